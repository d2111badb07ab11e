/* cosmo_pol_amd.h -- C ABI of the MI355X-native cosmo_pol hot path.
 *
 * Shared library: cosmo_pol_amd/csrc/libcosmo_pol_hip.so (hipcc, gfx950).
 * Plain C: opaque context, raw pointers and sizes, int status codes
 * (0 = ok, < 0 = error; text via cpol_last_error).  Never throws / aborts.
 * The caller owns every host buffer; device copies are owned by the context.
 * A context is NOT thread-safe: one per GPU per process (one process per GPU,
 * sweeps sharded by rays).
 *
 * What each entry point replaces in the reference (wolfidan/cosmo_pol):
 *
 *   cpol_stage_model    pycosmo variables handed to the workers through module
 *                       globals (radar_operator.py:199-215) and re-read for
 *                       every radial and variable by the SWIG call
 *                       (interpolation/interpolation.py:584-595)
 *   cpol_stage_hydro    Lookup_table objects of load_all_lut (lookup/lut.py:
 *                       27-76) + create_hydrometeor constants
 *                       (hydrometeors/hydrometeors.py:39, doppler_scatter.py:
 *                       111-122)
 *   cpol_interp_points  the native gate kernel itself, same inputs as
 *                       get_all_radar_pts (interpolation/interpolation_c.c:8)
 *                       but evaluated for every staged variable in one pass
 *   cpol_run_sweep      the body of the scan loop: for every radial
 *                       get_interpolated_radial (interpolation/interpolation.py:
 *                       91) -> melting (interpolation/melting.py:19) ->
 *                       get_radar_observables (scatter/doppler_scatter.py:49)
 *                       -> integrate_radials (interpolation.py:36), and
 *                       cut_at_sensitivity (doppler_scatter.py:804) for the
 *                       whole sweep: ONE call per sweep instead of one
 *                       pool.map task per radial (radar_operator.py:429-432)
 */
#ifndef COSMO_POL_AMD_H
#define COSMO_POL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the cpol_* functions below are its only exported symbols. */
#if defined(__GNUC__)
#define CPOL_API __attribute__((visibility("default")))
#else
#define CPOL_API
#endif

typedef struct cpol_ctx cpol_ctx;

enum {
    CPOL_OK = 0,
    CPOL_ERR_HIP = -1,          /* a HIP runtime call failed            */
    CPOL_ERR_ARG = -2,          /* invalid argument / not staged        */
    CPOL_ERR_DOMAIN = -3,       /* a gate lies outside the model domain
                                   (reference: IndexError,
                                   interpolation.py:572-580)            */
    CPOL_ERR_NOMEM = -4
};

#define CPOL_MAX_VARS   24
#define CPOL_MAX_HYDRO  8
#define CPOL_N_SZ       12      /* columns of a LUT row (compute_lut_sz.py:265-297) */
#define CPOL_TRAJ_STRIDE 4      /* doubles per (ray, vertical node) entry of the `traj` table */
#define CPOL_GEO_STRIDE  8      /* doubles per (ray, horizontal node) entry of the `geo` table */
#define CPOL_SITE_STRIDE 8      /* doubles per ray of the `site` table */

/* PSD families (how N(D) is evaluated on the device) */
enum {
    CPOL_PSD_GAMMA = 0,         /* N0 D^mu exp(-lambda D^nu)   hydrometeors.py:128-147 */
    CPOL_PSD_ICE_FIELD = 1,     /* 1-moment ice, Field (2005)  hydrometeors.py:1231-1373 */
    CPOL_PSD_MELTING = 2        /* melting snow / graupel      hydrometeors.py:303-478  */
};

/* how (lambda, N0) follow from the model moments */
enum {
    CPOL_RULE_RAIN_1MOM = 0,    /* hydrometeors.py:747-772   */
    CPOL_RULE_SNOW_1MOM = 1,    /* hydrometeors.py:879-907   */
    CPOL_RULE_GRAUPEL_1MOM = 2, /* hydrometeors.py:1025-1049 */
    CPOL_RULE_TWO_MOMENT = 3,   /* hydrometeors.py:212-256, 1341-1365 */
    CPOL_RULE_ICE_1MOM = 4,     /* hydrometeors.py:1302-1339 */
    CPOL_RULE_MELTING_SNOW = 5, /* hydrometeors.py:1398-1434 */
    CPOL_RULE_MELTING_GRAUPEL = 6 /* hydrometeors.py:1446-1481 */
};

/* mass source of a hydrometeor slot */
enum {
    CPOL_Q_MODEL = 0,           /* a staged model variable (var_q)             */
    CPOL_Q_MELT_SNOW = 1,       /* QmS_v diagnosed by the melting scheme       */
    CPOL_Q_MELT_GRAUPEL = 2     /* QmG_v                                       */
};

typedef struct {
    int32_t psd_family;         /* CPOL_PSD_*  */
    int32_t rule;               /* CPOL_RULE_* */
    int32_t q_source;           /* CPOL_Q_*    */
    int32_t var_q;              /* staged-variable index of the mass density   */
    int32_t var_qn;             /* ... of the number density (2-moment) or -1  */
    int32_t var_t;              /* ... of the temperature                      */
    int32_t n_e, n_t, n_d;      /* table shape [n_e, n_t, n_d, 12]             */
    int32_t second_axis_f64;    /* 1: second axis queried with a float64 value
                                   (wet fraction), 0: float32 (temperature)    */
    float   e_lo, e_step;       /* axes_limits[0][0], axes_step[0]  (float32)  */
    float   t_lo, t_step;       /* second axis ('t' or 'wc')        (float32)  */
    double  dD;                 /* d axis step D[1]-D[0] (as float32 value)    */
    /* generic power-law / PSD constants (python-float values of the reference) */
    double  a, b, alpha, beta, mu, nu;
    double  lambda_factor, ntot_factor, vel_factor;
    double  n0_fixed;           /* rain / graupel 1-mom intercept (0 if per gate) */
    double  x_min, x_max;       /* 2-moment mean-mass clip                     */
    double  c_n0, c_lam;        /* 2-moment unit factors 1000^-(1+mu), 1000^-nu */
    double  lam_exponent;       /* 1/(4+mu) | 1/(b+1) | -nu/b                  */
    double  n0_exponent;        /* (mu+1)/nu (2-moment)                        */
    /* melting species: rain and dry-solid partners                            */
    double  r_a, r_b, r_alpha, r_beta, r_n0, r_mu, r_lambda_factor, r_lam_exponent;
    double  r_dmin, r_dmax, s_dmin, s_dmax;
    int32_t solid_rule;         /* CPOL_RULE_SNOW_1MOM or _GRAUPEL_1MOM        */
    int32_t uniform_grid;       /* gamma family, nu == 1: aux[] holds the grid step
                                   and per-bin offsets for the exp recurrence   */
    int32_t numeric_intv;       /* gamma family whose fall-speed moments are summed
                                   numerically over ALL gates (2-moment ice,
                                   hydrometeors.py:1256-1275): aux[] = D^mu, D^nu,
                                   V(D) on the linspace grid, then its step       */
    int32_t tab_degree;         /* 0, or the degree of the polynomial tables appended to aux[]:
                                   melting family (CPOL_MELT_DEGREE): per wet-fraction bin of
                                   the table's second axis and diameter bin, polynomials in
                                   the wet fraction of the four fw-only factors of N(D);
                                   1-moment ice (CPOL_ICE_DEGREE): per panel in log2(lambda),
                                   polynomials of the three normalisation sums (which depend on
                                   lambda only).  See cpol_stage_hydro                     */
    int32_t pad_;
    uint64_t table_id;          /* 0, or a caller-chosen identity of EVERYTHING staged for this
                                   slot (descriptor, table, per-bin factors, aux, Doppler
                                   weights): the integral tables built for an id are kept (the
                                   CPOL_ITAB_CACHE most recent) and reused when a slot with the
                                   same id is staged again -- a caller that switches between a
                                   few table sets (Ku / Ka / ground radar, get_GPM_swath) pays
                                   for their construction once.  Equal ids must mean equal
                                   content                                                 */
} cpol_hydro_desc;
#define CPOL_ITAB_CACHE 24

#define CPOL_MELT_DEGREE 10     /* degree of those polynomials (11 coefficients)           */
#define CPOL_MELT_FUNCS  4      /* D_r, G, G*M, G*V                                        */
#define CPOL_ICE_DEGREE  10     /* degree of the ice normalisation-sum polynomials         */
#define CPOL_ICE_FUNCS   3      /* sum a D^b phi, sum V phi, sum phi over the norm. grid   */

typedef struct {
    int32_t n_rays, n_gates;
    int32_t n_sub;              /* kept antenna-quadrature points per radial    */
    int32_t n_hnodes, n_vnodes; /* distinct horizontal / vertical GH nodes      */
    int32_t with_melting, with_attenuation;
    int32_t integrate_model;    /* also return antenna-averaged model variables */
    int32_t apply_sensitivity;  /* 1: censor with tables->sens_thr (cut_at_sensitivity) */
    int32_t outputs_on_device;  /* 0: output pointers are host buffers, the call returns when
                                   they are filled; 1: device pointers, the kernels write
                                   them in place, the call returns at once; 2: PINNED host
                                   buffers (cpol_host_alloc), device-to-host copies are queued
                                   on the context's stream and the call returns at once --
                                   results (and a deferred CPOL_ERR_DOMAIN) after
                                   cpol_synchronize.  Carve the arrays from ONE allocation:
                                   when they span a window of at most 1.25 x their total
                                   size they are moved by a single copy (which also
                                   overwrites the padding bytes between them -- do not keep
                                   other data inside that window).  Every array must be
                                   aligned to its element size, as any C array is            */
    int32_t simulate_doppler;   /* 0 off, 1 / 2 / 3 = Doppler scheme of the reference (RVEL;
                                   3 = full Doppler spectrum, doppler_scatter.py:335-391) */
    int32_t geometry_mode;      /* CPOL_GEOM_*                                   */
    double  radar_lat, radar_lon, radar_alt;
    double  range0, range_step; /* RANGE_RADAR = range0 + k*range_step          */
    double  ke, re;             /* 4/3 and the earth radius (host evaluates quirk Q1) */
    double  sin_u1, cos_u1;     /* reduced latitude of the radar (Vincenty)     */
    double  wavelength;         /* mm                                           */
    double  k_squared;
    double  radial_res;         /* m                                            */
    double  c_zh;               /* wavelength^4 / (pi^5 K^2)                     */
    int32_t var_u, var_v, var_w; /* staged-variable indices of the wind (RVEL)    */
    int32_t var_rho;            /* ... of the air density (Doppler scheme 3) or -1 */
    /* Doppler scheme 3 */
    int32_t n_vbins;            /* len(VARRAY) = FFT_length + 1 (global_constants.py:171) */
    int32_t debug_flags;        /* 0 in production.  CPOL_DEBUG_EXACT_SUBBEAMS (bit 0): the non-central sub-beams take
                                   the long form of the geodesy the central one takes (5 Vincenty passes, atan2 ->
                                   degrees -> sincos, correctly rounded division / square root) instead of the short
                                   form (4 passes, reciprocal roots + Newton, short series): what
                                   tests/test_gpu_fullsize.py and tools/fast_sub_check.py compare the short form with */
    double  c_spectrum;         /* wavelength^4 / (pi^5 K^2 K^2)  (doppler_scatter.py:709) */
} cpol_sweep_params;

#define CPOL_DEBUG_EXACT_SUBBEAMS 1

/* ray-path models */
enum {
    CPOL_GEOM_GROUND_43 = 0,    /* 4/3-earth closed form (atm_refraction.py:181-220)    */
    CPOL_GEOM_SPACEBORNE = 1,   /* straight ray from orbit, KE = 1, gates below 35 km
                                   (atm_refraction.py:222-272, intended behaviour)      */
    CPOL_GEOM_HOST_PATHS = 2    /* (s, h, e) per gate supplied by the host, e.g. the
                                   Zeng & Blahak ODE (atm_refraction.py:79-148)          */
};

/* per-ray host-side tables (cpol_ray_tables fills them) */
typedef struct {
    const double *traj;         /* [n_rays][n_vnodes][CPOL_TRAJ_STRIDE = 4] : el_rad,
                                   sin el, cos el, el_deg                       */
    const double *geo;          /* [n_rays][n_hnodes][CPOL_GEO_STRIDE = 8] : sin a1,
                                   cos a1, sigma1, sin alpha, b*A, B, C, azimuth_rad */
    const int32_t *sub_h;       /* [n_sub] horizontal node of each kept sub-beam */
    const int32_t *sub_v;       /* [n_sub] vertical node                        */
    const double *sub_w;        /* [n_sub] quadrature weight                    */
    const double *sens_thr;     /* [n_gates] dBZ threshold per gate or NULL      */
    const double *site;         /* [n_rays][8] per-ray radar site or NULL (= the one
                                   of cpol_sweep_params): sin U1, cos U1, lon [deg],
                                   altitude [m], earth radius [m], first kept gate
                                   index, number of kept gates, unused             */
    const float *paths;         /* CPOL_GEOM_HOST_PATHS: [n_rays][n_vnodes][3][n_gates]
                                   float32 (s, h, e_deg), NaN = no gate            */
    const double *nyquist;      /* [n_rays] Nyquist velocity per ray [m/s] for the RVEL
                                   aliasing (utilities.py:142-156) or NULL (no folding) */
    uint64_t version;           /* 0: tables are uploaded on every call; otherwise the
                                   caller's tag of this table set -- a context keeps the
                                   device copies of the last 8 tags it has seen (same tag
                                   and same shapes: nothing is uploaded), so a scan that
                                   cycles through its elevations uploads each set once  */
    /* integration scheme 'ml' (interpolation.py:168-193, 423-436; per-gate weights as in
       doppler_scatter.py:124-129, 186-189, 259-264): NULL / 0 for scalar weights */
    const double *varray;       /* [n_vbins] velocity bins of the Doppler spectrum
                                   (Doppler scheme 3) or NULL                          */
    const int32_t *sub_smooth;  /* [n_sub] 1: the weight of this sub-beam is sub_w x the
                                   Gaussian-smoothed mask of the first / last melting-layer
                                   gate of the sub-beam; 0: sub_w at every gate          */
    const double *ml_filter;    /* [2 * ml_radius + 1] filter taps (scipy gaussian_filter) */
    int32_t ml_radius;
    int32_t pad_;
} cpol_ray_tables_t;

typedef struct {
    /* all [n_rays * n_gates]; NULL = not wanted */
    float  *ZH, *ZV, *ZDR, *KDP, *DELTA_HV, *PHIDP, *RHOHV, *ATT_H, *ATT_V;
    double *RVEL;
    double *mask;
    double *lats, *lons;        /* central sub-beam                             */
    float  *dist, *heights;
    double *model_vars;         /* [n_vars][n_rays*n_gates] (integrate_model)   */
    float  *sz_total;           /* [n_rays*n_gates][12]  (debug / parity)       */
    double *DSPECTRUM;          /* [n_rays*n_gates][n_vbins]  (Doppler scheme 3) */
    int8_t *mask_sum8;          /* [n_rays*n_gates] the radial mask as what it is made of: the SUM over the sub-beams of
                                   their mask codes (-1 below the topography, 0, +1 above the model top, 2 outside the
                                   domain; doppler_scatter.py:472-477), one byte per gate instead of the eight of `mask`.
                                   mask = mask_sum8 / n_sub, then values in (-1, 0] -> 0: the caller's two NumPy statements.
                                   Needs 2 * n_sub <= 127.  When it is asked for and `mask` is not, `mask` is not written. */
} cpol_outputs;

typedef struct {
    int64_t n_subbeam_gates;    /* N_sbg                                        */
    int64_t n_valid_items;      /* N_valid: (sub-beam gate, hydrometeor), QM>0  */
    int64_t n_gates;            /* output gates                                 */
    int64_t n_work_units;       /* work units of the integrating PSD kernels    */
    float   ms_traj, ms_interp, ms_classify, ms_bucket, ms_psd, ms_final, ms_total;
    int32_t n_table_items;      /* of n_valid_items: finished from the integral tables
                                   (the others were integrated bin by bin)      */
    int32_t pad_;
} cpol_counters_t;

CPOL_API int  cpol_create(int device, cpol_ctx **out);
CPOL_API void cpol_destroy(cpol_ctx *ctx);
/* A LANE of `parent`: a context that shares the parent's staged model cube and
 * scattering tables (read-only, no copy) and owns its own HIP stream, work buffers and
 * counters, so that independent sweeps (the elevations of a volume scan, consecutive
 * scans) are in flight together -- the reference runs them one after the other
 * (radar_operator.py:429-432).  Fork after staging; stage calls fail on a lane and on a
 * parent with live lanes; destroy lanes before their parent.  One host thread per lane. */
CPOL_API int  cpol_fork(cpol_ctx *parent, cpol_ctx **out);
CPOL_API const char *cpol_last_error(cpol_ctx *ctx);
/* use an externally created hipStream_t (e.g. torch's current stream); NULL = own stream */
CPOL_API int  cpol_set_stream(cpol_ctx *ctx, void *hip_stream);
/* waits for the context's stream.  Also the point where a DEFERRED domain error surfaces:
 * sweeps with outputs_on_device = 1 / 2 return before their kernels ran, so a gate outside
 * the model domain (reference: IndexError, interpolation.py:572-580) sets a sticky error word
 * on the device that stays set over later sweeps until cpol_synchronize or cpol_counters has
 * reported it ONCE as CPOL_ERR_DOMAIN (then it is cleared). */
CPOL_API int  cpol_synchronize(cpol_ctx *ctx);
/* page-locked host memory owned by the context (freed by cpol_host_free / cpol_destroy): the
 * target of outputs_on_device = 2, so that the device-to-host copy of one sweep overlaps the
 * kernels of the next (other lanes) instead of being staged through pageable memory.
 * ctx = NULL: a context-free block owned by the caller until cpol_host_free(NULL, p) -- for host-side
 * pools whose blocks (results handed to the user) must outlive the contexts that filled them; the
 * caller makes sure no copy into the block is in flight when it frees or re-uses it */
CPOL_API int  cpol_host_alloc(cpol_ctx *ctx, size_t bytes, void **out);
CPOL_API int  cpol_host_free(cpol_ctx *ctx, void *p);
/* a context-free block as cpol_host_alloc(NULL, ...) gives, taken from the NUMA node next to GPU
 * `device` whatever the calling thread's current device is (one process per GPU on a two-socket host:
 * helper threads have never called hipSetDevice); freed by cpol_host_free(NULL, p) */
CPOL_API int  cpol_host_alloc_near(int device, size_t bytes, void **out);
/* "0000:75:00.0" of GPU `device` (len >= 16): /sys/bus/pci/devices/<id>/numa_node and local_cpulist
 * tell a one-process-per-GPU launcher which cores to run the rank on (the reference's pool is not
 * placed at all, radar_operator.py:402) */
CPOL_API int  cpol_device_pci_bus_id(int device, char *buf, int len);
/* free and total device memory of the context's GPU in bytes (hipMemGetInfo; `free_bytes` also counts what the
 * context's own grow-only work buffers hold already: the room a launch sequence of THIS context has), and an estimate of
 * the work-buffer bytes ONE sub-beam gate of a launch sequence needs with the hydrometeors staged
 * now (`per_gate`: about 1.2 KB with six species): what a caller that packs many sweeps into one
 * cpol_run_sweep call sizes its batches by (the reference processes one radial at a time,
 * radar_operator.py:431) */
CPOL_API int  cpol_mem_info(cpol_ctx *ctx, size_t *free_bytes, size_t *total_bytes, size_t *per_gate);
/* the HIP stream (hipStream_t) the context launches on: to order foreign work (copies,
 * collectives) against a sweep with events */
CPOL_API int  cpol_get_stream(cpol_ctx *ctx, void **hip_stream);

/* data[v] and zlevels: [nz][ny][nx] float32, C order (x = rotated longitude
 * fastest), level 0 = model top; llc = (Lo1, La1), urc = (Lo2, La2), res =
 * (dlon, dlat) as float32; south_pole = (lat, lon) of the rotated south pole. */
CPOL_API int  cpol_stage_model(cpol_ctx *ctx, int n_vars, const float *const *data,
                      const float *zlevels, int nz, int ny, int nx,
                      const float llc[2], const float urc[2], const float res[2],
                      const double south_pole[2]);

/* table: float64 [n_e][n_t][n_d][12]; pre: [n_d] (or NULL) host-evaluated
 * N0*D^mu | D^mu; dnu: [n_d] D^nu; aux: family-specific per-bin tables.
 * Melting family with tab_degree = CPOL_MELT_DEGREE: aux = [n_t][2] (centre and 1 / half-width
 * of the wet-fraction interval of every bin of the second axis) followed by
 * [n_t][n_d][CPOL_MELT_FUNCS][CPOL_MELT_DEGREE + 1] monomial coefficients in
 * u = (fw - centre) / half-width of
 *   D_r(fw, k)   melted-equivalent diameter of bin k      (hydrometeors.py:382-383)
 *   G(fw, k)     N_r0 sqrt(D_r) V_r / V  dD_r/dD          (:372-390, 415-439)
 *   G M, G V     the same times the particle mass / fall speed (:393-412, :457-478)
 * so that N(D_k) = G exp(-lambda_r D_r): everything that depends on the wet fraction only
 * (two cube roots, a sixth / fourth root, two powers for graupel, a division per bin) is read
 * from the table; the rain slope lambda_r stays per item.
 * 1-moment ice with tab_degree = CPOL_ICE_DEGREE (and uniform_grid): the aux block of the
 * recurrence form (4 n_d + 8 + 8 n_d values) is followed by [log2(lambda_lo), panels per octave,
 * n_panels, 0] and [n_panels][CPOL_ICE_FUNCS][CPOL_ICE_DEGREE + 1] monomial coefficients in
 * u in [-1, 1] across a panel of log2(lambda): the renormalisation sums of IceParticle.set_psd /
 * integrate_V (hydrometeors.py:1331-1339, 1256-1275) are 1024-term sums that depend on the
 * item's lambda only, not on its LUT slice. */
CPOL_API int  cpol_stage_hydro(cpol_ctx *ctx, int slot, const cpol_hydro_desc *desc,
                      const double *table, const double *pre, const double *dnu,
                      const double *aux, int n_aux);
CPOL_API int  cpol_set_num_hydro(cpol_ctx *ctx, int n_hydro);
/* Finishes staging: builds the INTEGRAL TABLES now instead of at the first sweep / cpol_fork.
 * For every species whose N(D) has one per-item shape parameter lambda (all gamma-family species,
 * 1-moment ice) the 12 PSD-integrated entries of an item -- what get_N + lookup_line + einsum
 * (hydrometeors.py:128-147, lut.py:309-344, doppler_scatter.py:246-251) produce -- are a per-item
 * scale times a function of (LUT slice, lambda) only.  The integrating kernels evaluate that
 * function once per (slice, 1/8-octave panel of lambda, Chebyshev node) and store degree-10
 * polynomials ("1-D blocks": 11 rows of 16 float64 per (slice, panel)); a sweep then gathers
 * 12 x 11 coefficients per item instead of integrating 1024 diameter bins.
 * The melting species (wet fraction fw and rain-partner slope lambda_r) get "2-D blocks": per (slice,
 * 1/4-octave panel of lambda_r) the polynomial of total degree 10 in (fw inside the slice's wet-
 * fraction bin, position inside the panel), 66 rows of 16 float64; a sweep evaluates 14 functions
 * (12 columns + 2 Doppler sums) per item.
 * ACCURACY GATE: every block carries one more item at an off-node point, integrated by the same
 * kernel and compared with the polynomial function by function on the scale of the function over
 * the block.  1-D tables keep the longest run of lambda panels whose blocks all stay below 1e-10
 * (in practice all but the last panel, where exp(-lambda D^nu) of the bins behind the first goes
 * subnormal); a 1-D table with less than half of its panels left, and a 2-D table with any block
 * above 1e-10, is dropped.  Items outside the accepted ranges are integrated bin by bin as in round
 * 1.  cpol_debug_read("itab_check"): per slot the worst accepted deviation (negative: table dropped;
 * 0: none), then where it was found, then the number of (block, function) pairs above the limit;
 * "itab_detail<slot>": per lambda panel and per function; "itab_times": device ms of build / gate.
 * Environment (read at every build): CPOL_ITAB=0 no tables at all, CPOL_ITAB_MELT=0 none for the
 * melting species, CPOL_ITAB_MAX_DEV=<x> another limit than 1e-10.  (Per sweep: CPOL_SUBSUM=0,
 * CPOL_LOOKUP_TILE=0, see INTEGRATION.md.)
 * COST: a few ms of kernels per gamma / ice slot, ~90 ms per melting slot on full-size tables
 * (46 x 100 slices); TRANSIENT device memory of the build = n_e * n_t * n_panels * 12 (1-D) or
 * * 122 (2-D) synthetic items x ~164 B: 1 GB for a gamma slot, ~3.5 GB for a melting slot of that
 * size; RESIDENT tables 0.3 - 0.5 GB per 1-D slot, 1.5 GB per melting slot.
 * Depends on every staged table: call after the last cpol_stage_* (any later staging call
 * invalidates the tables; they are rebuilt on demand). */
CPOL_API int  cpol_prepare(cpol_ctx *ctx);

/* float32 functions of the gate temperature, TABULATED BY THE HOST over every float32 value
 * in [128 K, 512 K) (2^24 consecutive bit patterns from CPOL_TFUN_FIRST_BITS).  The reference
 * evaluates them with NumPy's float32 exp / power, which are not correctly rounded (1-2 ulp off
 * in 20-40 % of the arguments); a 1-ulp difference in the snow intercept becomes > 1e-5
 * relative in K_DP, a difference of near-equal float32 sums.  The host evaluates the
 * reference's own expressions with its own NumPy, the device looks the value up by the bit
 * pattern of T: the bits the reference would have used, on every host.  Optional: without a
 * table (or outside the range) the device computes the correctly rounded value.
 *   CPOL_TFUN_SNOW_N0     13.5*(5.65e5*exp(-0.107*(T-273.15)))/1000   hydrometeors.py:896
 *   CPOL_TFUN_ICE_MOM2_A  10**a(T - 273.15), Field et al. (2005)      hydrometeors.py:1287-1292 */
enum { CPOL_TFUN_SNOW_N0 = 0, CPOL_TFUN_ICE_MOM2_A = 1, CPOL_N_TFUN = 2 };
#define CPOL_TFUN_FIRST_BITS 0x43000000u      /* 128.0f */
#define CPOL_TFUN_COUNT      (1u << 24)       /* every float32 in [128, 512) */
CPOL_API int  cpol_stage_t_function(cpol_ctx *ctx, int which, const float *table /* [CPOL_TFUN_COUNT] */);

/* Doppler scheme 2 (doppler_scatter.py:283-296): per table slice and diameter bin the
 * trapezoid weight w_k (1/2 at both ends) times the horizontal radar cross-section
 * 2 pi (Z11 - Z12 - Z21 + Z22), and the same times the fall speed V(D_k):
 * weights [n_e][n_t][n_d][2] = (w rcs V, w rcs) float64.  Needed only when
 * cpol_sweep_params.simulate_doppler == 2. */
CPOL_API int  cpol_stage_doppler_weights(cpol_ctx *ctx, int slot, const double *weights);

/* Doppler scheme 3: per table slice and diameter bin the float32 horizontal radar cross
 * section 2 pi (Z11 - Z12 - Z21 + Z22) (doppler_scatter.py:686-695), and the float32
 * diameter grid np.linspace(d_min, d_max, n_d) of get_doppler_spectrum with D^mu, D^nu
 * evaluated in float32 (:676-681): dgrid = [3][n_d].  Power-law species only. */
CPOL_API int  cpol_stage_spectrum_tables(cpol_ctx *ctx, int slot, const float *rcs32, const float *dgrid);

/* gate kernel on explicit points: coords [n][2] (rotated lat, lon) float32,
 * heights [n] float32 -> out [n_vars][n] float32 with the reference's
 * sentinels (-9999 above the model top, NaN below topography). */
CPOL_API int  cpol_interp_points(cpol_ctx *ctx, int n, const float *coords, const float *heights,
                        float *out);

/* fills per-ray tables with libm (C callers); Python callers use numpy.
 * traj_out: [n_rays][n_vnodes][CPOL_TRAJ_STRIDE] doubles (el_rad, sin el, cos el, el_deg);
 * geo_out:  [n_rays][n_hnodes][CPOL_GEO_STRIDE] doubles. */
CPOL_API int  cpol_ray_tables(const cpol_sweep_params *p, const double *az_deg, const double *el_deg,
                     const double *pts_h_deg, const double *pts_v_deg,
                     double *traj_out, double *geo_out);

CPOL_API int  cpol_run_sweep(cpol_ctx *ctx, const cpol_sweep_params *p, const cpol_ray_tables_t *tables,
                    cpol_outputs *out);

/* CPOL_GEOM_SPACEBORNE helper: index of the first candidate gate below the
 * model-top ceiling for each (ray, vertical node): first_gate [n_rays*n_vnodes]
 * (host buffer).  `traj` [n_rays][n_vnodes][CPOL_TRAJ_STRIDE] and `site`
 * [n_rays][CPOL_SITE_STRIDE] as in cpol_ray_tables_t, n_cand[n_rays] candidate gates. */
CPOL_API int  cpol_spaceborne_first_gate(cpol_ctx *ctx, const cpol_sweep_params *p, const double *traj,
                                const double *site, const int32_t *n_cand, double ceiling_m,
                                int32_t *first_gate);

CPOL_API int  cpol_counters(cpol_ctx *ctx, cpol_counters_t *out);
/* on = 1: HIP events around every stage of every sweep (7 per sweep); on = 2: only around the
 * PSD x table stage (2 per sweep; ms_psd alone is filled in); on = 0: off */
CPOL_API int  cpol_enable_timing(cpol_ctx *ctx, int on);

/* debug / parity access to intermediate device buffers of the last sweep:
 * "sub_values" float32 [n_vars][n_sbg], "sub_mask" int8 [n_sbg], "sub_elev"
 * float32 [n_sbg], "sub_coords" float32 [n_sbg][2], "item_key" int32
 * [n_hydro][n_sbg], "sz_integ" float32 [n_rays*n_gates][n_hydro][12],
 * "traj" float32 [n_rays][n_vnodes][3][n_gates], "item_res" float64
 * [n_hydro][n_sbg][12], "sub_wgate" float64 [n_sbg] (integration scheme 'ml': the per-gate
 * sub-beam weights).  Without the debug reads enabled: "launch_forms" int32 [12] -- which launch
 * sequence the last cpol_run_sweep of this context took: [0] the CPOL_GATE1_RAY rule in force,
 * [1] k_gate1_ray ran, [2] the single-beam gate kernel, [3] k_interp_classify, [4] items off the
 * tables listed directly, [5] k_subbeam_sum, [6] table items evaluated in place, [7] the one
 * sub-beam on the coordinate polynomials, [8] n_sub, [9] lanes alive, [10] the build's CPOL_SCAN_FORM (1: wavefront range scans), [11] a HIP
 * graph was replayed; "poly_central", "host_times", "cache", "itab_check", "itab_times",
 * "itab_detail<slot>" (see cosmo_pol_amd/_native.py).  Returns bytes copied or < 0. */
CPOL_API int64_t cpol_debug_read(cpol_ctx *ctx, const char *name, void *dst, int64_t max_bytes);

/* test hook: evaluates one of the device math helpers of the melting-species kernel on
 * host arrays (op 0 exp, 1 log, 2 cbrt, 3 cbrt via x^(-1/6), 4 x^(1/6), 5 x^(1/4), 6 1/x) */
CPOL_API int  cpol_debug_math(cpol_ctx *ctx, int op, const double *x, double *y, int n);

#ifdef __cplusplus
}
#endif
#endif
