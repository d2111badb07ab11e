// cosmo_pol_hip.hip -- C ABI (include/cosmo_pol_amd.h) and launch sequence of
// the MI355X-native cosmo_pol hot path.  gfx950 only; no CPU fallback.
//
// One call of cpol_run_sweep = the work of one `pool.map(worker, azimuths)`
// of the reference (cosmo_pol/radar_operator.py:429-432) for the rays given:
//
//   k_interp_sweep    (ray, sub-beam, gate)           ray path (4/3 earth / orbit), geodesic,
//                                                     rotated pole, trilinear gather, all vars
//   k_classify        (sub-beam gate)                 melting, PSD parameters,
//                                                     LUT bins, bucket histogram
//   k_bucket_scan / k_bucket_scatter                  counting sort by LUT slice + unit list
//   k_psd_{uniform,gamma,ice,melting}                 PSD x table integration
//   k_final           (ray)                           sub-beam/hydrometeor sums, polarimetric
//                                                     variables, range scans, sensitivity
// = 6 launches for a 1-moment sweep (10 before the ray-path, unit-list and final merges).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <time.h>
#include <string.h>
#include <string>
#include <type_traits>
#include <vector>

#include "cpol_device.h"
#include "cpol_interp.inl"
#include "cpol_psd.inl"
#include "cpol_fused.inl"
#include "cpol_final.inl"
#include "cpol_gate.inl"
#include "cpol_spectrum.inl"

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

enum { EV_T0 = 0, EV_TRAJ, EV_INTERP, EV_CLASSIFY, EV_BUCKET, EV_PSD, EV_FINAL, EV_N };

// scattering table kept resident for a caller-supplied cpol_hydro_desc.table_id
struct TableCacheEntry {
    uint64_t id = 0;
    size_t bytes = 0;
    DevBuf buf;
    uint64_t used = 0;
};

// integral tables kept for a caller-supplied cpol_hydro_desc.table_id
struct ItabCacheEntry {
    uint64_t id = 0;
    int dop2 = 0;
    DevBuf tab, head;
    ItabDev t{};
    double check = 0.0, check_at = 0.0, check_edge = 0.0;
    uint64_t used = 0;          // build serial of the last use (LRU; entries of the current build are pinned)
};

}  // namespace

struct cpol_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // lanes (cpol_fork): a child shares the parent's staged model / tables read-only and
    // owns only its stream, work buffers and counters
    cpol_ctx *parent = nullptr;
    int n_children = 0;
    // HIP graph of the sweep's launch sequence (device outputs, unchanged arguments)
    bool use_graph = false;
    hipGraphExec_t graph_exec[2] = {nullptr, nullptr};     // one per counter set (the sets alternate sweep by sweep)
    uint64_t graph_key[2] = {0, 0};
    uint64_t stage_serial = 0;         // bumped by every staging call (kernel arguments change)
    std::string err;
    // model
    bool model_staged = false;
    ModelDev model{};
    DevBuf d_H, d_V;
    // hydrometeors
    HydroSet hs{};
    SpecSet ss{};
    DevBuf d_table[CPOL_MAX_HYDRO], d_pre[CPOL_MAX_HYDRO], d_dnu[CPOL_MAX_HYDRO],
        d_aux[CPOL_MAX_HYDRO], d_rcsw[CPOL_MAX_HYDRO], d_rcs32[CPOL_MAX_HYDRO], d_dgrid[CPOL_MAX_HYDRO];
    bool hydro_staged[CPOL_MAX_HYDRO] = {};
    // integral tables (built on the device by build_itabs after staging; lanes share the parent's)
    DevBuf d_itab[CPOL_MAX_HYDRO], d_itab_head[CPOL_MAX_HYDRO], d_itab_M, d_itab_M1;
    std::vector<ItabCacheEntry> itab_cache;
    std::vector<TableCacheEntry> table_cache;  // staged scattering tables with a table_id
    uint64_t table_clock = 0;
    uint64_t itab_builds = 0;
    double itab_check_at[CPOL_MAX_HYDRO] = {};
    double itab_check[CPOL_MAX_HYDRO] = {};     // worst deviation at the blocks' check points (negative: table rejected)
    std::vector<double> itab_detail[CPOL_MAX_HYDRO];   // 1-D tables: [log2_lo, ppo, d0, n_pan, worst per function (NF), worst per panel (n_pan)]
    double itab_ms[CPOL_MAX_HYDRO][2] = {};     // device time of the last build of the slot's table: all of it, the check alone
    double itab_check_edge[CPOL_MAX_HYDRO] = {};   // 1-D tables: worst deviation at the second check point (near the panel edge) alone
    double itab_bad[CPOL_MAX_HYDRO] = {};       // 1-D tables: (block, function) pairs at or above the accepted deviation
    ItabSet its{};
    uint64_t lut_serial = 0;           // bumped by the staging calls the integral tables depend on (not the model cube)
    uint64_t itab_serial = ~0ull;      // lut_serial the tables were built for
    DevBuf d_tfun[CPOL_N_TFUN];        // host-tabulated float32 functions of T (cpol_stage_t_function)
    const float *tfun[CPOL_N_TFUN] = {};
    // per-sweep work buffers (grow only)
    // per-sweep host tables: packed into ONE pinned staging buffer (ring of 4, an event each) and
    // moved by ONE host-to-device copy into the arena of a table set (tsets[]); v_* = views into the set in use
    // The per-ray tables of the last CPOL_TABLE_SETS scan geometries stay on the device, found again by the
    // caller's version tag: a scan that cycles through its elevations uploads each set once (a 35 KB copy in
    // front of a 0.09 ms sweep costs the stream 25-45 us: measured with tools/submit_cost.py, one lane,
    // 8 elevations in turn: 115 us per sweep against 70 with one resident set).
    struct TableSet {
        uint64_t version = 0, last_use = 0;
        long shape[6] = {0, 0, 0, 0, 0, 0};
        DevBuf buf;
        void *views[11] = {nullptr};
        // single-beam sweeps: the coordinate polynomials of this set's rays (k_trajectory), made once per (version, range grid)
        DevBuf poly;
        uint64_t poly_version = 0;
        double poly_scale = 0.0;
        // ... and per (model grid rotation, radar site): k_trajectory bakes the rotated-pole matrix of the staged model and
        // the site's constants into them (cpol_stage_model clears poly_version; the site is compared sweep by sweep)
        double poly_site[3] = {0.0, 0.0, 0.0};
    };
    static constexpr int N_TABLE_SETS = 8;
    TableSet tsets[N_TABLE_SETS];
    uint64_t tset_clock = 0;
    void *v_traj_in = nullptr, *v_geo = nullptr, *v_subh = nullptr, *v_subv = nullptr, *v_subw = nullptr,
         *v_sens = nullptr, *v_site = nullptr, *v_nyq = nullptr, *v_subsmooth = nullptr,
         *v_mlfilter = nullptr, *v_varray = nullptr;
    // sibling streams of the PSD stage: the kernel flavours of a sweep (recurrence / full-exp /
    // ice / melting) touch disjoint items, so they run side by side (fork after the bucket
    // sort, join before the final stage) instead of back to back
    hipStream_t aux[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    struct Staging { void *p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false; } stg[4];
    int stg_next = 0;
    DevBuf b_traj, b_wgate, b_clk, b_rayc, b_poly, d_geoM;
    DevBuf b_beam, b_spectrum, b_outwin;
    DevBuf b_vals, b_mask, b_elev, b_coords, b_qmelt, b_fwmelt, b_key, b_par, b_count, b_offset,
        b_units, b_totals, b_perm, b_res, b_pos, b_vn, b_icefirst, b_rvel, b_proj, b_blkranked, b_rec, b_vmask, b_gscan, b_defer;
    DevBuf b_out[16], b_szinteg, b_sztotal, b_model, b_ticket, b_mask8;
    DevBuf b_present;                  // k_gate1_ray's sweeps: one word per (ray, 64-gate tile), which hydrometeor slots may have an item there (k_interp_sweep writes, k_gate1_ray reads)
    // last sweep shapes (debug reads)
    long last_n_sbg = 0, last_n_rg = 0;
    int last_n_rays = 0, last_n_gates = 0, last_n_sub = 0, last_n_v = 0, last_n_keys = 0;
    // Two sets of sweep counters (b_count, b_totals), used in turn: the first kernel of a launch sequence clears
    // the set of the NEXT sweep, so that a kernel which counts can also be the first of its sequence (a kernel cannot
    // clear what its own workgroups add to), and no sequence needs a fill kernel.  Both sets are zero when (re)allocated.
    uint64_t sweep_serial = 0;
    long count_stride = 0;             // ints per set of b_count
    bool counters_dirty = false;       // a launch sequence began and did not end in sweep_serial advancing (an error return after
                                       // its first launch): both counter sets are cleared before the next sequence uses one
    int last_par = 0;                  // the set the last sweep used
    int subsum_coop_rounds = 6;        // CPOL_SUBSUM_COOP_ROUNDS: scalar-cache rounds per wavefront and sub-beam before the gather tail
    int rare_overlap = 0;              // CPOL_RARE_OVERLAP=1: k_psd_rare beside k_psd_lookup on a sibling stream instead of behind it (measured: the share of
                                       // one of 8 GPUs 1.42 -> 1.37 ms alone, 0.97 -> 0.99 with three lanes; the C3 sweep 151 -> 167 us: the fork and join cost
                                       // more than the idle launch -- off)
    int rare_direct = 1;               // CPOL_RARE_DIRECT=0: keep the counting sort of the items outside the tables (read when the context is created)
    int lookup_list = 1;               // CPOL_LOOKUP_LIST=0: k_psd_lookup starts one wavefront per tile instead of workgroups that list the tiles with work among their own; 2: the list for every launch size
    int lookup_split = 0;              // CPOL_LOOKUP_SPLIT=<n>: wavefronts per tile of k_psd_lookup (0: by launch size)
    int gate1_ray = -1;                // -1 (default): 1 in a context with lanes (sweeps in flight side by side: three launches instead of four per sweep are 6-8 % of
                                       // c2's throughput), 0 without (the isolated sweep: 88 against 91 us); CPOL_GATE1_RAY=0: never; 1: k_gate1_ray (items off the tables integrated in place, the range scans by k_scan_rays: a single-beam sweep
                                       // of three lean launches, no integrating launch); 2: also with tables that lost panels; 3: the scans inside the gate kernel (a ticket per ray)
    int gate1_species = 1;             // CPOL_GATE1_SPECIES=0 / 2: never / always k_gate1_species (one wavefront per species; default: small launches)
    int fuse_gate1 = 0;                // CPOL_FUSE_GATE1=1: k_interp_gate1 instead of k_interp_sweep + k_gate1 (measured slower where it matters)
    int fuse_classify = 1;             // CPOL_FUSE_CLASSIFY=0: k_interp_sweep + k_classify instead of k_interp_classify (read when the context is created)
    int gate1 = 1;                     // CPOL_GATE1=0 / 2: never / also with melting species: the single-beam fused kernel (read when the context is created)
    int subsum_scalar = 0;             // CPOL_SUBSUM_FORM=scalar: the cooperative form of k_subbeam_sum takes its rows through the scalar cache instead of LDS
    int upload_kernel = 0;             // CPOL_TABLE_UPLOAD=kernel: the per-ray tables by k_upload_tables instead of hipMemcpyAsync (a measurement knob)
    int last_forms[12] = {0};          // the launch forms of the last sweep (cpol_debug_read "launch_forms"): [0] g1r, [1] k_gate1_ray, [2] single-beam gate kernel,
                                       // [3] k_interp_classify, [4] items off the tables listed directly, [5] k_subbeam_sum, [6] table items evaluated in place,
                                       // [7] coordinate polynomials for the one sub-beam, [8] n_sub, [9] lanes alive, [10] CPOL_SCAN_FORM (1: the range scans by a whole wavefront), [11] HIP graph replayed
    int last_poly_central = 0;         // the last sweep's one sub-beam took the coordinate polynomials (cpol_debug_read "poly_central")
    int geo_poly_central = 1;          // CPOL_GEO_POLY_CENTRAL=0: a single-beam sweep keeps the long form of the geodesy for its (central) sub-beam even when
                                       // nobody asks for the float64 latitude / longitude; 2: the polynomials also with the debug reads enabled (tools/fast_sub_check.py)
    int geo_poly = 1;                  // CPOL_GEO_POLY=0: the non-central sub-beams take the short closed form of the geodesy instead of the per-ray polynomials
    int psd_rare = 1;                  // CPOL_PSD_RARE=0: one launch per integrating flavour also when the units are directly listed items (read when the context is created)
    int subsum_small = 0;              // CPOL_SUBSUM_SMALL=1: experiment: the gather form of k_subbeam_sum with three wavefronts per (tile, hydrometeor) and the whole block in flight (measured slower)
    int subsum_chain = 1;              // CPOL_SUBSUM_CHAIN=0: the team's terms pass through LDS and a barrier per round instead of its float32 sums waiting in LDS, handed
                                       // from sub-beam to sub-beam (share of one of 8 GPUs: 385 against 334 us)
    int subsum_team = -1;              // CPOL_SUBSUM_TEAM=W: k_subbeam_sum_team<W> (W = 2..8 wavefronts per (tile, species)) for every launch; 0: never (the one-wavefront
                                       // forms alone); -1 (default): W = 4 for the launches too small for the LDS form
    int subsum_coop = -1;              // CPOL_SUBSUM_COOP: k_subbeam_sum takes its coefficients through the scalar cache: 0 never, 1 always, -1 by launch size
    bool last_subsum = false;          // the 1-D table items of the last sweep never went through res[] (k_subbeam_sum)
    bool keep_debug = false;
    // host time of cpol_run_sweep by section (ns, summed; cpol_debug_read "host_times"): [0] calls, [1] per-ray tables
    // (staging memcpy + the H2D copy call), [2] work-buffer checks / allocations, [3] kernel launches, [4] the
    // device-to-host copy call(s), [5] everything
    // [6..9]: section [1] split: waiting for the staging slot's previous copy (hipEventSynchronize), filling the slot,
    // the hipMemcpyAsync call, hipEventRecord
    double host_ns[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool fail_next = false;            // test hook (cpol_debug_read "fail_next_sweep"): the next launch sequence returns an error
                                       // after its kernels are queued, as a failed copy or capture would
    // sticky domain-error word (device): OR-ed by the kernels of every sweep, cleared only
    // when it has been read AND reported (host-output sweeps, cpol_synchronize, cpol_counters)
    int *d_errword = nullptr;
    int *h_errword = nullptr;          // its page-locked host copy (report_domain_error reads it on the context's own stream: no copy through
                                       // pageable memory, nothing on the null stream)
    std::vector<void *> host_allocs;   // pinned host memory handed out by cpol_host_alloc
    // timing: one event set per sweep since cpol_enable_timing(ctx, 1); elapsed
    // times are collected (averaged) by cpol_counters after the stream drained,
    // so recording does not serialise the timed loop.
    int timing = 0;                    // 0 off, 1 every stage, 2 the PSD stage only
    std::vector<hipEvent_t *> ev_sets;
    size_t ev_used = 0;
    hipEvent_t *ev = nullptr;          // set of the sweep being recorded
    cpol_counters_t counters{};
};

namespace {

#define HIPCHK(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            char buf_[512];                                                               \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                 \
            ctx->err = buf_;                                                              \
            return CPOL_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

int ensure(cpol_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap && b.p) return CPOL_OK;
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;                      // (the buffer is empty and may be sized again by a later, smaller call)
        (void)hipGetLastError();            // reported through the return code: not left behind as the thread's last error
        ctx->err = std::string("hipMalloc failed: ") + hipGetErrorString(e);
        return CPOL_ERR_NOMEM;
    }
    b.cap = want;
    return CPOL_OK;
}

#define ENSURE(buf, bytes)                                   \
    do {                                                     \
        int rc_ = ensure(ctx, (buf), (size_t)(bytes));       \
        if (rc_ != CPOL_OK) return rc_;                      \
    } while (0)

int upload(cpol_ctx *ctx, DevBuf &b, const void *src, size_t bytes)
{
    ENSURE(b, bytes);
    HIPCHK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return CPOL_OK;
}

void free_buf(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

int build_itabs(cpol_ctx *ctx);

}  // namespace

namespace {

// Integral tables of every slot whose N(D) has one shape parameter (cpol_device.h: ItabDev).
// Runs after staging, before the first sweep / fork: the PSD kernels integrate one synthetic
// item (N0 = QM = 1) per (LUT slice, lambda panel, Chebyshev node); k_itab_fit turns the 11 node
// values of every (slice, panel, function) into polynomial coefficients.  CPOL_ITAB=0 disables
// (every item is then integrated bin by bin, as in round 1).
int build_itabs(cpol_ctx *ctx)
{
    if (ctx->parent) return CPOL_OK;                      // lanes copy the parent's tables
    if (ctx->itab_serial == ctx->lut_serial) return CPOL_OK;
    const bool enabled = !(getenv("CPOL_ITAB") && atoi(getenv("CPOL_ITAB")) == 0);   // read at every (re)build
    (void)hipGetLastError();            // a stale error of another user of the runtime in this thread is not ours
    ctx->its = ItabSet{};
    ctx->itab_serial = ctx->lut_serial;
    ctx->itab_builds++;
    for (int j = 0; j < CPOL_MAX_HYDRO; ++j) {
        ctx->itab_check[j] = ctx->itab_check_at[j] = ctx->itab_bad[j] = ctx->itab_check_edge[j] = 0.0;
        ctx->itab_ms[j][0] = ctx->itab_ms[j][1] = 0.0;
    }
    if (!enabled) return CPOL_OK;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int n_hyd = ctx->hs.n_hydro;
    constexpr int NC = CPOL_ITAB_NC, NC1 = CPOL_ITAB1_NC;
    // M[pw][q] = sum_n c2m[n][pw] w_n T_n(x_q): node values -> monomial coefficients, for n nodes:
    // out = [M | Tm (nodes -> Chebyshev) | C2M (T_n -> monomials)], each [n][n]
    auto basis_matrices = [](int n, std::vector<double> &out) {
        std::vector<long double> T((size_t)n * n, 0.0L);          // T[k][pw]: monomial coefficients of T_k
        T[0] = 1.0L;
        if (n > 1) T[(size_t)n + 1] = 1.0L;
        for (int k = 2; k < n; ++k)
            for (int pw = 0; pw < n; ++pw)
                T[(size_t)k * n + pw] = (pw > 0 ? 2.0L * T[(size_t)(k - 1) * n + pw - 1] : 0.0L) - T[(size_t)(k - 2) * n + pw];
        const long double pi = 3.141592653589793238462643383279502884L;
        out.assign((size_t)3 * n * n, 0.0);
        for (int pw = 0; pw < n; ++pw)
            for (int q = 0; q < n; ++q) {
                long double acc = 0.0L;
                for (int k = 0; k < n; ++k)
                    acc += T[(size_t)k * n + pw] * (k == 0 ? 1.0L : 2.0L) / n * cosl(pi * k * (q + 0.5L) / n);
                out[(size_t)pw * n + q] = (double)acc;
                out[(size_t)n * n + pw * n + q] = (double)((pw == 0 ? 1.0L : 2.0L) / n * cosl(pi * pw * (q + 0.5L) / n));
                out[(size_t)2 * n * n + pw * n + q] = (double)T[(size_t)pw * n + q];
            }
    };
    {
        std::vector<double> M2, M1;
        basis_matrices(NC, M2);                                    // 2-D blocks (k_itab_fit2: Tm, C2M)
        basis_matrices(NC1, M1);                                   // 1-D blocks (k_itab_fit: M)
        int rc = upload(ctx, ctx->d_itab_M, M2.data(), M2.size() * sizeof(double));
        if (rc != CPOL_OK) return rc;
        rc = upload(ctx, ctx->d_itab_M1, M1.data(), M1.size() * sizeof(double));
        if (rc != CPOL_OK) return rc;
        HIPCHK(hipStreamSynchronize(st));
    }
    const bool melt_enabled = !(getenv("CPOL_ITAB_MELT") && atoi(getenv("CPOL_ITAB_MELT")) == 0);
    // accepted deviation of a block's polynomial from the integrating kernel at the block's check point
    // (CPOL_ITAB_MAX_DEV: test knob -- a tiny value forces the fallback to the integrating kernels)
    const double max_dev = getenv("CPOL_ITAB_MAX_DEV") ? atof(getenv("CPOL_ITAB_MAX_DEV")) : CPOL_ITAB_MAX_DEVIATION;
    const bool dop2 = [&] { for (int j = 0; j < n_hyd; ++j) if (!ctx->hs.h[j].rcsw) return false; return n_hyd > 0; }();
    for (int j = 0; j < n_hyd; ++j) {
        const HydroDev &h = ctx->hs.h[j];
        const cpol_hydro_desc &d = h.d;
        const bool gamma = d.psd_family == CPOL_PSD_GAMMA && h.pre && h.dnu;
        const bool ice = d.psd_family == CPOL_PSD_ICE_FIELD && d.uniform_grid && d.tab_degree == CPOL_ICE_DEGREE;
        const bool melt = d.psd_family == CPOL_PSD_MELTING && melt_enabled;
        if (!gamma && !ice && !melt) continue;
        if (d.table_id) {
            bool hit = false;
            for (auto &e : ctx->itab_cache)
                if (e.id == d.table_id && e.dop2 == (int)dop2) {
                    ctx->its.t[j] = e.t;
                    ctx->itab_check[j] = e.check;
                    ctx->itab_check_edge[j] = e.check_edge;
                    ctx->itab_check_at[j] = e.check_at;
                    e.used = ctx->itab_builds;
                    hit = true;
                    break;
                }
            if (hit) continue;
        }
        // lambda range: 2^-10 .. the lambda at which exp(-lambda D_0^nu) leaves the double range
        // (gamma), resp. the end of the ice normalisation tables
        double d0 = 0.0, lo = -10.0, hi = 15.0;
        if (gamma) {
            HIPCHK(hipMemcpy(&d0, h.dnu, sizeof d0, hipMemcpyDeviceToHost));
            if (!(d0 > 0.0)) continue;
            hi = floor(log2(690.0 / d0) * CPOL_ITAB_PPO) / CPOL_ITAB_PPO;     // whole panels
            if (hi > 16.0) hi = 16.0;
            if (hi <= lo) continue;
        }
        if (melt) {
            // slope of the rain partner, lambda_r = (factor / QM)^(1/(4+mu)): 2^-1 (QM = 0.05 kg m-3) ..
            // 2^8.5 (3e-14: trilinear interpolation towards an empty model cell leaves such values);
            // items beyond are integrated
            lo = -1.0; hi = 8.5;
        }
        const int ppo = melt ? CPOL_ITAB2_PPO : CPOL_ITAB_PPO;
        const int n_pan = (int)floor((hi - lo) * ppo + 0.5);
        const int n_slices = d.n_e * d.n_t;
        const int per_block = melt ? CPOL_ITAB2_NODES : CPOL_ITAB1_NODES;      // the nodes + 1 check point
        const long n_items = (long)n_slices * n_pan * per_block;
        const bool melt_tab = melt && d.tab_degree == CPOL_MELT_DEGREE;
        const int unit_items = ((gamma && d.uniform_grid) || ice || melt_tab) ? 128 : 64;
        const int upers = (n_pan * per_block + unit_items - 1) / unit_items;
        const long n_units = (long)n_slices * upers;
        if (n_items >= (1L << 31)) continue;
        const size_t tab_bytes = (size_t)n_slices * n_pan * (melt ? CPOL_ITAB2_NB : NC1) * CPOL_ITAB_NFP * sizeof(double);
        DevBuf b_par, b_perm, b_units, b_tot, b_res, b_vn, b_det;
        std::vector<unsigned long long> det_bits;
        int rc;
        // destination: the slot's own buffers, or (table_id given) a cache entry of its own
        ItabCacheEntry *ce = nullptr;
        if (d.table_id) {
            if (ctx->itab_cache.size() >= CPOL_ITAB_CACHE) {          // evict the least recently used
                size_t victim = ctx->itab_cache.size();
                for (size_t k = 0; k < ctx->itab_cache.size(); ++k)
                    if (ctx->itab_cache[k].used != ctx->itab_builds &&
                        (victim == ctx->itab_cache.size() || ctx->itab_cache[k].used < ctx->itab_cache[victim].used))
                        victim = k;
                if (victim < ctx->itab_cache.size()) {
                    free_buf(ctx->itab_cache[victim].tab);
                    free_buf(ctx->itab_cache[victim].head);
                    ctx->itab_cache.erase(ctx->itab_cache.begin() + victim);
                }
            }
            if (ctx->itab_cache.size() < CPOL_ITAB_CACHE) {
                ctx->itab_cache.emplace_back();
                ce = &ctx->itab_cache.back();
                ce->id = d.table_id; ce->dop2 = (int)dop2; ce->used = ctx->itab_builds;
            }
        }
        DevBuf &dst_tab = ce ? ce->tab : ctx->d_itab[j];
        DevBuf &dst_head = ce ? ce->head : ctx->d_itab_head[j];
        auto drop_entry = [&] { if (ce) { free_buf(ce->tab); free_buf(ce->head); ctx->itab_cache.pop_back(); ce = nullptr; } };
        if (melt) {
            // centre and 1 / half-width of the wet-fraction bins of the table's second axis; bin 0
            // reaches down to fw = 0 and the last bin up to 1 (lut.py:336-341 clips the index)
            std::vector<double> head(2 * (size_t)d.n_t);
            for (int b = 0; b < d.n_t; ++b) {
                const double blo = b == 0 ? 0.0 : (double)d.t_lo + b * (double)d.t_step;
                const double bhi = b == d.n_t - 1 ? 1.0 : (double)d.t_lo + (b + 1) * (double)d.t_step;
                head[2 * b] = 0.5 * (blo + bhi);
                head[2 * b + 1] = 1.0 / (0.5 * (bhi - blo));
            }
            if ((rc = upload(ctx, dst_head, head.data(), head.size() * sizeof(double)))) { drop_entry(); return rc; }
            HIPCHK(hipStreamSynchronize(st));
        }
        if ((rc = ensure(ctx, b_par, (size_t)CPOL_MAX_PAR * n_items * sizeof(double))) ||
            (rc = ensure(ctx, b_perm, (size_t)n_items * sizeof(int))) ||
            (rc = ensure(ctx, b_units, (size_t)n_units * sizeof(WorkUnit))) ||
            (rc = ensure(ctx, b_tot, 8 * sizeof(long long))) ||
            (rc = ensure(ctx, b_res, (size_t)n_items * CPOL_N_SZ * sizeof(double))) ||
            (rc = ensure(ctx, b_vn, (size_t)n_items * 2 * sizeof(double))) ||
            (rc = ensure(ctx, b_det, (size_t)(2 * n_pan + CPOL_ITAB_NF) * sizeof(unsigned long long))) ||
            (rc = ensure(ctx, dst_tab, tab_bytes))) {
            free_buf(b_par); free_buf(b_perm); free_buf(b_units); free_buf(b_tot); free_buf(b_res); free_buf(b_vn); free_buf(b_det);
            drop_entry();
            return rc;
        }
        HIPCHK(hipMemsetAsync(b_vn.p, 0, (size_t)n_items * 2 * sizeof(double), st));
        ItabBuildArgs ba{};
        ba.par = (double *)b_par.p; ba.perm = (int *)b_perm.p; ba.units = (WorkUnit *)b_units.p;
        ba.totals = (long long *)b_tot.p; ba.n_items = n_items; ba.n_slices = n_slices; ba.n_pan = n_pan;
        ba.key_base = h.key_base; ba.unit_items = unit_items; ba.log2_lo = lo; ba.ppo = ppo;
        ba.two_d = melt ? 1 : 0; ba.n_t = d.n_t; ba.head = (const double *)dst_head.p;
        hipEvent_t evb[4] = {nullptr, nullptr, nullptr, nullptr};
        for (auto &e_ : evb) (void)hipEventCreate(&e_);
        (void)hipEventRecord(evb[0], st);
        hipLaunchKernelGGL(k_itab_nodes, dim3(cdiv(n_items > n_units ? n_items : n_units, 256)), dim3(256), 0, st, ba);
        // the slot's own kernels on the synthetic items (arrays of THIS slot only: the kernels
        // index [n_hydro][...][n] arrays with the slot number, hence the shifted bases)
        PsdArgs pa{};
        pa.units = (const WorkUnit *)b_units.p;
        pa.totals = (const long long *)b_tot.p;
        pa.perm = (const int *)b_perm.p;
        pa.par = (const double *)b_par.p - (long)j * CPOL_MAX_PAR * n_items;
        pa.res = (double *)b_res.p - (long)j * n_items * CPOL_N_SZ;
        pa.vn = (double *)b_vn.p - (long)j * n_items * 2;
        pa.par_w = (double *)b_par.p - (long)j * CPOL_MAX_PAR * n_items;
        pa.n_sbg = n_items;
        pa.clk = nullptr;
        pa.ice_force_sum = 0;
        const dim3 grd((unsigned)(n_units < 4096 ? n_units : 4096)), blk(CPOL_PSD_THREADS);
        if (melt) {
            if (melt_tab) {
                if (dop2) hipLaunchKernelGGL((k_psd_melting_tab<true>), grd, blk, 0, st, ctx->hs, pa);
                else hipLaunchKernelGGL((k_psd_melting_tab<false>), grd, blk, 0, st, ctx->hs, pa);
            } else {
                if (dop2) hipLaunchKernelGGL((k_psd<PSD_MODE_MELTING, true>), grd, blk, 0, st, ctx->hs, pa);
                else hipLaunchKernelGGL((k_psd<PSD_MODE_MELTING, false>), grd, blk, 0, st, ctx->hs, pa);
            }
        } else if (ice) {
            if (dop2) { hipLaunchKernelGGL((k_psd_ice2<true>), grd, blk, 0, st, ctx->hs, pa);
                        hipLaunchKernelGGL((k_psd<PSD_MODE_ICE, true>), grd, blk, 0, st, ctx->hs, pa); }
            else { hipLaunchKernelGGL((k_psd_ice2<false>), grd, blk, 0, st, ctx->hs, pa);
                   hipLaunchKernelGGL((k_psd<PSD_MODE_ICE, false>), grd, blk, 0, st, ctx->hs, pa); }
        } else if (d.uniform_grid) {
            if (dop2) hipLaunchKernelGGL((k_psd_uniform<true>), grd, dim3(CPOL_PSD_THREADS_U), 0, st, ctx->hs, pa);
            else hipLaunchKernelGGL((k_psd_uniform<false>), grd, dim3(CPOL_PSD_THREADS_U), 0, st, ctx->hs, pa);
        } else {
            if (dop2) hipLaunchKernelGGL((k_psd<PSD_MODE_GAMMA_EXP, true>), grd, blk, 0, st, ctx->hs, pa);
            else hipLaunchKernelGGL((k_psd<PSD_MODE_GAMMA_EXP, false>), grd, blk, 0, st, ctx->hs, pa);
        }
        double worst = 0.0;
        unsigned long long worst_bits = 0, edge_bits = 0;
        unsigned int n_bad = 0;
        if (melt) {
            ItabFit2Args fa{};
            fa.res = (const double *)b_res.p; fa.vn = (const double *)b_vn.p;
            fa.M = (const double *)ctx->d_itab_M.p; fa.tab = (double *)dst_tab.p;
            fa.n_blocks = (long)n_slices * n_pan;
            fa.worst = (unsigned long long *)b_tot.p + 3;
            HIPCHK(hipMemsetAsync(fa.worst, 0, sizeof(unsigned long long), st));
            hipLaunchKernelGGL(k_itab_fit2, dim3(cdiv(fa.n_blocks * CPOL_ITAB_NFP, 64)), dim3(64), 0, st, fa);
            (void)hipEventRecord(evb[1], st);
            hipLaunchKernelGGL(k_itab_check2, dim3(cdiv(fa.n_blocks * (CPOL_N_SZ + 2), 256)), dim3(256), 0, st, fa);
            (void)hipEventRecord(evb[2], st);
            HIPCHK(hipMemcpyAsync(&worst_bits, fa.worst, sizeof worst_bits, hipMemcpyDeviceToHost, st));
        } else {
            ItabFitArgs fa{};
            fa.res = (const double *)b_res.p; fa.vn = (const double *)b_vn.p; fa.par = (const double *)b_par.p;
            fa.M = (const double *)ctx->d_itab_M1.p; fa.tab = (double *)dst_tab.p;
            fa.n_items = n_items; fa.n_slices = n_slices; fa.n_pan = n_pan; fa.log2_lo = lo; fa.d0 = gamma ? d0 : 0.0;
            fa.worst = (unsigned long long *)b_tot.p + 3;
            fa.n_bad = (unsigned int *)((unsigned long long *)b_tot.p + 4);
            fa.max_dev = max_dev;
            HIPCHK(hipMemsetAsync(b_det.p, 0, (size_t)(2 * n_pan + CPOL_ITAB_NF) * sizeof(unsigned long long), st));
            fa.by_fn = (unsigned long long *)b_det.p;
            fa.by_pan = fa.by_fn + CPOL_ITAB_NF;            // [n_pan] both check points, then [n_pan] the edge point alone
            det_bits.resize((size_t)2 * n_pan + CPOL_ITAB_NF);
            HIPCHK(hipMemsetAsync(fa.worst, 0, 3 * sizeof(unsigned long long), st));      // worst, n_bad, worst at the edge point
            const dim3 fgrid(cdiv((long)n_slices * n_pan * CPOL_ITAB_NF, 256));
            // (the check is part of the fit kernel: its share of the build = the two check items of every block)
            (void)hipEventRecord(evb[1], st);
            (void)hipEventRecord(evb[2], st);
            hipLaunchKernelGGL(k_itab_fit, fgrid, dim3(256), 0, st, fa);
            HIPCHK(hipMemcpyAsync(&worst_bits, fa.worst, sizeof worst_bits, hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(&n_bad, fa.n_bad, sizeof n_bad, hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(&edge_bits, fa.worst + 2, sizeof edge_bits, hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(det_bits.data(), b_det.p, det_bits.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        }
        (void)hipEventRecord(evb[3], st);
        const hipError_t e = hipStreamSynchronize(st);
        {
            float ms_all = 0.f, ms_chk = 0.f;
            if (e == hipSuccess && hipEventElapsedTime(&ms_all, evb[0], evb[3]) == hipSuccess &&
                hipEventElapsedTime(&ms_chk, evb[1], evb[2]) == hipSuccess) {
                // the check = its kernel + the block's extra item in the integration (1 of per_block)
                ctx->itab_ms[j][0] = ms_all;
                ctx->itab_ms[j][1] = ms_chk + (ms_all - ms_chk) / per_block;
            }
            for (auto &e_ : evb) if (e_) (void)hipEventDestroy(e_);
        }
        free_buf(b_par); free_buf(b_perm); free_buf(b_units); free_buf(b_tot); free_buf(b_res); free_buf(b_vn); free_buf(b_det);
        ctx->itab_detail[j].clear();
        if (!det_bits.empty()) {
            ctx->itab_detail[j] = {lo, (double)ppo, gamma ? d0 : 0.0, (double)n_pan};     // (+ [pan_lo, pan_hi) at the end)
            for (unsigned long long b : det_bits) { double v; memcpy(&v, &b, sizeof v); ctx->itab_detail[j].push_back(v); }
        }
        if (e != hipSuccess || hipGetLastError() != hipSuccess) {
            ctx->err = "build_itabs: a kernel failed";
            drop_entry();
            return CPOL_ERR_HIP;
        }
        {
            const unsigned long long eb = worst_bits & ~0xFFFFFFull;
            memcpy(&worst, &eb, sizeof worst);
            ctx->itab_check_at[j] = (double)(worst_bits & 0xFFFFFFull);      // (block x functions + function) mod 2^24
        }
        ctx->itab_check[j] = worst;
        ctx->itab_bad[j] = (double)n_bad;
        { double ev = 0.0; memcpy(&ev, &edge_bits, sizeof ev); ctx->itab_check_edge[j] = ev; }   // (all panels; the accepted run: below)
        int pan_lo = 0, pan_hi = n_pan;
        // CPOL_ITAB_KEEP_PANELS=<lo>:<hi> (test knob, 1-D tables): only the panels [lo, hi) stay on the table -- the items
        // beyond go to the integrating kernels, as with a table that lost panels to the accuracy gate
        int keep_lo = 0, keep_hi = n_pan;
        if (!melt && getenv("CPOL_ITAB_KEEP_PANELS") && sscanf(getenv("CPOL_ITAB_KEEP_PANELS"), "%d:%d", &keep_lo, &keep_hi) == 2) {
            keep_lo = std::max(0, std::min(keep_lo, n_pan));
            keep_hi = std::max(keep_lo, std::min(keep_hi, n_pan));
        } else { keep_lo = 0; keep_hi = n_pan; }
        if (!melt && !(worst < max_dev)) {
            // 1-D table: keep the longest run of lambda panels whose blocks all pass (in practice everything
            // but the last panel, where exp(-lambda D^nu) of the bins behind the first one goes subnormal);
            // items with lambda outside the run are integrated bin by bin like items outside the table
            const std::vector<double> &dv = ctx->itab_detail[j];
            int best_lo = 0, best_n = 0, run_lo = 0;
            for (int p = 0; p <= n_pan; ++p) {
                const bool ok = p < n_pan && dv[4 + CPOL_ITAB_NF + p] < max_dev;
                if (ok) continue;
                if (p - run_lo > best_n) { best_n = p - run_lo; best_lo = run_lo; }
                run_lo = p + 1;
            }
            if (2 * best_n >= n_pan) {
                pan_lo = best_lo; pan_hi = best_lo + best_n;
                worst = 0.0;
                for (int p = pan_lo; p < pan_hi; ++p) worst = fmax(worst, dv[4 + CPOL_ITAB_NF + p]);
                ctx->itab_check[j] = worst;
            }
        }
        if (!(worst < max_dev)) {
            // the polynomial does not reproduce the integrating kernel between the nodes (melting: wet-
            // fraction bins too wide for the degree, coarse test tables; 1-D: coefficients that cancel):
            // this species stays on the integrating path
            ctx->itab_check[j] = -worst;
            if (ce) drop_entry(); else free_buf(ctx->d_itab[j]);
            continue;
        }
        ItabDev &t = ctx->its.t[j];
        t.tab = (const double *)dst_tab.p;
        t.head = melt ? (const double *)dst_head.p : nullptr;
        t.log2_lo = lo;
        t.d0 = gamma ? d0 : 0.0;
        t.n_pan = n_pan;
        if (!melt) { pan_lo = std::max(pan_lo, keep_lo); pan_hi = std::max(pan_lo, std::min(pan_hi, keep_hi)); }
        t.pan_lo = pan_lo; t.pan_hi = pan_hi;
        if (!ctx->itab_detail[j].empty()) {
            // (the edge point's worst over the accepted run of panels, like `check`)
            const std::vector<double> &dv = ctx->itab_detail[j];
            double we = 0.0;
            for (int p = pan_lo; p < pan_hi; ++p) we = fmax(we, dv[4 + CPOL_ITAB_NF + n_pan + p]);
            ctx->itab_check_edge[j] = we;
            ctx->itab_detail[j].push_back(pan_lo); ctx->itab_detail[j].push_back(pan_hi);
        }
        t.writes_vn = ice || dop2 || d.numeric_intv || melt;
        t.ppo = ppo;
        t.two_d = melt ? 1 : 0;
        t.par_slot = melt ? 2 : 0;
        t.n_t = d.n_t;
        if (ce) { ce->t = t; ce->check = ctx->itab_check[j]; ce->check_at = ctx->itab_check_at[j]; ce->check_edge = ctx->itab_check_edge[j]; }
    }
    return CPOL_OK;
}

}  // namespace

extern "C" {

int cpol_create(int device, cpol_ctx **out)
{
    if (!out) return CPOL_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return CPOL_ERR_HIP;
    cpol_ctx *ctx = new cpol_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return CPOL_ERR_HIP;
    }
    ctx->own_stream = true;
    if (getenv("CPOL_ERRWORD") && !strcmp(getenv("CPOL_ERRWORD"), "pageable")) ctx->h_errword = nullptr;      // (measurement knob: round 4's read)
    else if (hipHostMalloc((void **)&ctx->h_errword, 64, hipHostMallocDefault) != hipSuccess) { ctx->h_errword = nullptr; (void)hipGetLastError(); }
    if (hipMalloc((void **)&ctx->d_errword, sizeof(int)) != hipSuccess ||
        hipMemset(ctx->d_errword, 0, sizeof(int)) != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return CPOL_ERR_HIP;
    }
    // opt-in (CPOL_USE_GRAPH=1): replaying the captured sequence cuts the host time of a sweep
    // 4x (0.12 -> 0.03 ms) but is no faster on the device (0.232 vs 0.222 ms single lane) and
    // slows three-lane throughput by a quarter when graph launches and plain launches mix
    ctx->use_graph = getenv("CPOL_USE_GRAPH") && atoi(getenv("CPOL_USE_GRAPH")) != 0;
    // CPOL_SUBSUM_COOP=0 / 1: k_subbeam_sum never / always takes the coefficient rows through the scalar cache
    // (default: by launch size; the results are identical)
    if (getenv("CPOL_SUBSUM_COOP")) ctx->subsum_coop = atoi(getenv("CPOL_SUBSUM_COOP")) != 0 ? 1 : 0;
    if (getenv("CPOL_LOOKUP_LIST")) ctx->lookup_list = std::max(0, std::min(2, atoi(getenv("CPOL_LOOKUP_LIST"))));
    if (getenv("CPOL_LOOKUP_SPLIT")) ctx->lookup_split = std::max(0, std::min(16, atoi(getenv("CPOL_LOOKUP_SPLIT"))));
    if (getenv("CPOL_GATE1_SPECIES")) ctx->gate1_species = std::max(0, std::min(2, atoi(getenv("CPOL_GATE1_SPECIES"))));
    if (getenv("CPOL_GATE1_RAY")) ctx->gate1_ray = std::max(-1, std::min(3, atoi(getenv("CPOL_GATE1_RAY"))));   // (2: also with tables that lost panels; 3: the scans inside the gate kernel)
    if (getenv("CPOL_FUSE_GATE1")) ctx->fuse_gate1 = atoi(getenv("CPOL_FUSE_GATE1")) != 0 ? 1 : 0;
    if (getenv("CPOL_FUSE_CLASSIFY")) ctx->fuse_classify = atoi(getenv("CPOL_FUSE_CLASSIFY")) != 0 ? 1 : 0;
    if (getenv("CPOL_RARE_OVERLAP")) ctx->rare_overlap = atoi(getenv("CPOL_RARE_OVERLAP")) != 0 ? 1 : 0;
    if (getenv("CPOL_RARE_DIRECT")) ctx->rare_direct = atoi(getenv("CPOL_RARE_DIRECT")) != 0 ? 1 : 0;
    if (getenv("CPOL_GATE1")) ctx->gate1 = atoi(getenv("CPOL_GATE1"));
    if (getenv("CPOL_SUBSUM_FORM")) ctx->subsum_scalar = !strcmp(getenv("CPOL_SUBSUM_FORM"), "scalar") ? 1 : 0;
    if (getenv("CPOL_SUBSUM_CHAIN")) ctx->subsum_chain = atoi(getenv("CPOL_SUBSUM_CHAIN")) != 0 ? 1 : 0;
    if (getenv("CPOL_SUBSUM_TEAM")) ctx->subsum_team = atoi(getenv("CPOL_SUBSUM_TEAM"));
    if (getenv("CPOL_SUBSUM_SMALL")) ctx->subsum_small = atoi(getenv("CPOL_SUBSUM_SMALL")) != 0 ? 1 : 0;
    if (getenv("CPOL_TABLE_UPLOAD")) ctx->upload_kernel = strcmp(getenv("CPOL_TABLE_UPLOAD"), "kernel") == 0 ? 1 : 0;
    if (getenv("CPOL_GEO_POLY_CENTRAL")) ctx->geo_poly_central = std::max(0, std::min(2, atoi(getenv("CPOL_GEO_POLY_CENTRAL"))));
    if (getenv("CPOL_GEO_POLY")) ctx->geo_poly = atoi(getenv("CPOL_GEO_POLY")) != 0 ? 1 : 0;
    if (getenv("CPOL_PSD_RARE")) ctx->psd_rare = atoi(getenv("CPOL_PSD_RARE")) != 0 ? 1 : 0;
    if (getenv("CPOL_SUBSUM_COOP_ROUNDS")) ctx->subsum_coop_rounds = std::max(0, std::min(64, atoi(getenv("CPOL_SUBSUM_COOP_ROUNDS"))));
    *out = ctx;
    return CPOL_OK;
}

void cpol_destroy(cpol_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->parent) {
        // staged buffers belong to the parent
        ctx->d_H = DevBuf(); ctx->d_V = DevBuf();
        for (int j = 0; j < CPOL_MAX_HYDRO; ++j) {
            ctx->d_table[j] = DevBuf(); ctx->d_pre[j] = DevBuf(); ctx->d_dnu[j] = DevBuf();
            ctx->d_aux[j] = DevBuf(); ctx->d_rcsw[j] = DevBuf();
            ctx->d_rcs32[j] = DevBuf(); ctx->d_dgrid[j] = DevBuf();
        }
        for (auto &b : ctx->d_tfun) b = DevBuf();
        for (auto &b : ctx->d_itab) b = DevBuf();
        for (auto &b : ctx->d_itab_head) b = DevBuf();
        ctx->d_itab_M = DevBuf(); ctx->d_itab_M1 = DevBuf();
        ctx->parent->n_children -= 1;
    }
    for (auto &b : ctx->d_tfun) free_buf(b);
    for (auto &b : ctx->d_itab) free_buf(b);
    for (auto &b : ctx->d_itab_head) free_buf(b);
    for (auto &e : ctx->itab_cache) { free_buf(e.tab); free_buf(e.head); }
    ctx->itab_cache.clear();
    for (auto &e : ctx->table_cache) free_buf(e.buf);
    ctx->table_cache.clear();
    free_buf(ctx->d_itab_M);
    free_buf(ctx->d_itab_M1);
    for (int i = 0; i < 3; ++i) {
        if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
        if (ctx->aux[i]) { (void)hipStreamSynchronize(ctx->aux[i]); (void)hipStreamDestroy(ctx->aux[i]); }
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    for (auto &sg : ctx->stg) {
        if (sg.ev) (void)hipEventDestroy(sg.ev);
        if (sg.p) (void)hipHostFree(sg.p);
    }
    for (auto &ts : ctx->tsets) { free_buf(ts.buf); free_buf(ts.poly); }
    DevBuf *all[] = {&ctx->d_H, &ctx->d_V, &ctx->b_traj, &ctx->b_wgate, &ctx->b_clk, &ctx->b_rayc,
                     &ctx->b_beam, &ctx->b_spectrum, &ctx->b_outwin, &ctx->b_vals, &ctx->b_mask,
                     &ctx->b_elev, &ctx->b_coords, &ctx->b_qmelt, &ctx->b_fwmelt, &ctx->b_key,
                     &ctx->b_par, &ctx->b_count, &ctx->b_offset, &ctx->b_units,
                     &ctx->b_totals, &ctx->b_perm, &ctx->b_res, &ctx->b_pos,
                     &ctx->b_vn, &ctx->b_icefirst, &ctx->b_rvel, &ctx->b_proj, &ctx->b_blkranked, &ctx->b_rec, &ctx->b_vmask, &ctx->b_gscan, &ctx->b_defer,
                     &ctx->b_szinteg, &ctx->b_sztotal, &ctx->b_model, &ctx->b_ticket, &ctx->b_poly, &ctx->d_geoM, &ctx->b_present, &ctx->b_mask8};
    for (DevBuf *b : all) free_buf(*b);
    for (auto &b : ctx->b_out) free_buf(b);
    for (int j = 0; j < CPOL_MAX_HYDRO; ++j) {
        free_buf(ctx->d_table[j]);
        free_buf(ctx->d_pre[j]);
        free_buf(ctx->d_dnu[j]);
        free_buf(ctx->d_aux[j]);
        free_buf(ctx->d_rcsw[j]);
        free_buf(ctx->d_rcs32[j]);
        free_buf(ctx->d_dgrid[j]);
    }
    for (hipEvent_t *set : ctx->ev_sets) {
        for (int k = 0; k < EV_N; ++k) (void)hipEventDestroy(set[k]);
        delete[] set;
    }
    for (hipGraphExec_t g : ctx->graph_exec) if (g) (void)hipGraphExecDestroy(g);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->d_errword) (void)hipFree(ctx->d_errword);
    if (ctx->h_errword) (void)hipHostFree(ctx->h_errword);
    for (void *h : ctx->host_allocs) (void)hipHostFree(h);
    delete ctx;
}

int cpol_fork(cpol_ctx *parent, cpol_ctx **out)
{
    if (!out) return CPOL_ERR_ARG;
    *out = nullptr;
    if (!parent || parent->parent) {
        if (parent) parent->err = "cpol_fork: fork the root context, not a lane";
        return CPOL_ERR_ARG;
    }
    cpol_ctx *ctx = parent;            // for HIPCHK
    HIPCHK(hipSetDevice(parent->device));
    HIPCHK(hipStreamSynchronize(parent->stream));      // staging has landed
    cpol_ctx *c = new cpol_ctx();
    c->device = parent->device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        parent->err = "cpol_fork: hipStreamCreate failed";
        return CPOL_ERR_HIP;
    }
    c->own_stream = true;
    c->h_errword = nullptr;
    if (parent->h_errword && hipHostMalloc((void **)&c->h_errword, 64, hipHostMallocDefault) != hipSuccess) { c->h_errword = nullptr; (void)hipGetLastError(); }
    if (hipMalloc((void **)&c->d_errword, sizeof(int)) != hipSuccess ||
        hipMemset(c->d_errword, 0, sizeof(int)) != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        parent->err = "cpol_fork: hipMalloc failed";
        return CPOL_ERR_NOMEM;
    }
    c->use_graph = parent->use_graph;
    c->subsum_coop = parent->subsum_coop;
    c->subsum_small = parent->subsum_small;
    c->subsum_team = parent->subsum_team;
    c->subsum_chain = parent->subsum_chain;
    c->psd_rare = parent->psd_rare;
    c->geo_poly = parent->geo_poly;
    c->geo_poly_central = parent->geo_poly_central;
    c->upload_kernel = parent->upload_kernel;
    c->subsum_scalar = parent->subsum_scalar;
    c->rare_direct = parent->rare_direct;
    c->rare_overlap = parent->rare_overlap;
    c->fuse_classify = parent->fuse_classify;
    c->fuse_gate1 = parent->fuse_gate1;
    c->gate1_species = parent->gate1_species;
    c->gate1_ray = parent->gate1_ray;
    c->lookup_split = parent->lookup_split;
    c->lookup_list = parent->lookup_list;
    c->gate1 = parent->gate1;
    c->subsum_coop_rounds = parent->subsum_coop_rounds;
    c->parent = parent;
    c->model_staged = parent->model_staged;
    c->model = parent->model;
    c->hs = parent->hs;
    c->ss = parent->ss;
    for (int j = 0; j < CPOL_MAX_HYDRO; ++j) c->hydro_staged[j] = parent->hydro_staged[j];
    for (int k = 0; k < CPOL_N_TFUN; ++k) c->tfun[k] = parent->tfun[k];
    {   // the parent's integral tables are complete before any lane exists
        const int rc_it = build_itabs(parent);
        if (rc_it != CPOL_OK) { (void)hipStreamDestroy(c->stream); (void)hipFree(c->d_errword); delete c; return rc_it; }
    }
    c->its = parent->its;
    c->itab_serial = parent->lut_serial;
    parent->n_children += 1;
    *out = c;
    return CPOL_OK;
}

const char *cpol_last_error(cpol_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int cpol_set_stream(cpol_ctx *ctx, void *hip_stream)
{
    if (!ctx) return CPOL_ERR_ARG;
    if (ctx->own_stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
        ctx->own_stream = false;
    }
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
    } else {
        HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    return CPOL_OK;
}

int cpol_get_stream(cpol_ctx *ctx, void **hip_stream)
{
    if (!ctx || !hip_stream) return CPOL_ERR_ARG;
    *hip_stream = (void *)ctx->stream;
    return CPOL_OK;
}

// reads the sticky domain-error word (the stream has drained); a set word is cleared and
// reported ONCE as CPOL_ERR_DOMAIN -- so a sweep that left the model domain is never lost
// behind later sweeps of the same context (the reference raises IndexError at that radial)
static int report_domain_error(cpol_ctx *ctx)
{
    // (called behind a hipStreamSynchronize of ctx->stream.  The word travels into page-locked memory on that stream: a
    // blocking hipMemcpy into a stack variable goes through the null stream and the runtime's staging of pageable
    // memory -- after the first of those, the first asynchronous copy of every following sweep took ~340 us with ~200
    // page faults for the next two synchronisation periods: round 4's "slow mode", profiles/r5_host_mode_probe.txt)
    int flag = 0;
    if (ctx->h_errword) {
        HIPCHK(hipMemcpyAsync(ctx->h_errword, ctx->d_errword, sizeof flag, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        flag = *(volatile int *)ctx->h_errword;
    } else {
        HIPCHK(hipMemcpy(&flag, ctx->d_errword, sizeof flag, hipMemcpyDeviceToHost));
    }
    if (!flag) return CPOL_OK;
    HIPCHK(hipMemsetAsync(ctx->d_errword, 0, sizeof(int), ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->err = "RADAR DOMAIN IS NOT ENTIRELY CONTAINED IN COSMO SIMULATION DOMAIN";
    return CPOL_ERR_DOMAIN;
}

int cpol_synchronize(cpol_ctx *ctx)
{
    if (!ctx) return CPOL_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return report_domain_error(ctx);
}

int cpol_host_alloc(cpol_ctx *ctx, size_t bytes, void **out)
{
    if (!out || bytes == 0) return CPOL_ERR_ARG;
    *out = nullptr;
    if (!ctx) {
        // context-free block: owned by the caller until cpol_host_free(NULL, p) (a host-side pool whose
        // blocks outlive the contexts that copy into them)
        void *h = nullptr;
        if (hipHostMalloc(&h, bytes, hipHostMallocDefault) != hipSuccess || !h) return CPOL_ERR_NOMEM;
        *out = h;
        return CPOL_OK;
    }
    HIPCHK(hipSetDevice(ctx->device));
    void *h = nullptr;
    if (hipHostMalloc(&h, bytes, hipHostMallocDefault) != hipSuccess || !h) {
        ctx->err = "cpol_host_alloc: hipHostMalloc failed";
        return CPOL_ERR_NOMEM;
    }
    ctx->host_allocs.push_back(h);
    *out = h;
    return CPOL_OK;
}

int cpol_host_alloc_near(int device, size_t bytes, void **out)
{
    // a context-free block like cpol_host_alloc(NULL, ...), but placed for `device`: the runtime takes
    // page-locked memory from the NUMA node next to the CURRENT device of the calling thread, which a
    // helper thread of a rank on GPU 5 has never set
    if (!out || bytes == 0 || device < 0) return CPOL_ERR_ARG;
    *out = nullptr;
    int prev = -1, n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); return CPOL_ERR_HIP; }
    if (device >= n_dev) return CPOL_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return CPOL_ERR_ARG; }
    void *h = nullptr;
    hipError_t e = hipHostMalloc(&h, bytes, hipHostMallocDefault);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess || !h) { (void)hipGetLastError(); return CPOL_ERR_NOMEM; }
    *out = h;
    return CPOL_OK;
}

int cpol_device_pci_bus_id(int device, char *buf, int len)
{
    if (!buf || len < 16 || device < 0) return CPOL_ERR_ARG;
    buf[0] = 0;
    if (hipDeviceGetPCIBusId(buf, len, device) == hipSuccess) return CPOL_OK;
    (void)hipGetLastError();            // (reported through the return code: not left behind as the thread's last error)
    return CPOL_ERR_HIP;
}

int cpol_mem_info(cpol_ctx *ctx, size_t *free_bytes, size_t *total_bytes, size_t *per_gate)
{
    if (!ctx) return CPOL_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    // "free" for a launch sequence of THIS context = what the device has free + what the context's grow-only work
    // buffers hold already (a scan that fitted as one sequence the first time fits again: a caller that sizes its
    // batches by this figure groups the same scan the same way on every call)
    {
        const DevBuf *work[] = {&ctx->b_traj, &ctx->b_wgate, &ctx->b_rayc, &ctx->b_beam, &ctx->b_vals, &ctx->b_mask, &ctx->b_elev,
                                &ctx->b_coords, &ctx->b_qmelt, &ctx->b_fwmelt, &ctx->b_key, &ctx->b_par, &ctx->b_units,
                                &ctx->b_perm, &ctx->b_res, &ctx->b_pos, &ctx->b_vn, &ctx->b_proj, &ctx->b_rec, &ctx->b_vmask,
                                &ctx->b_gscan, &ctx->b_defer, &ctx->b_szinteg};
        for (const DevBuf *b : work) if (b->p) f += b->cap;
        if (f > t) f = t;
    }
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    if (per_gate) {
        // the arenas cpol_run_sweep sizes by n_sbg (vals, masks, elevation, melting scratch, per-item key / pos /
        // par / rec / perm / res / vn, velocity terms, per-gate weights)
        const size_t n_hyd = (size_t)(ctx->hs.n_hydro > 0 ? ctx->hs.n_hydro : 1), n_vars = (size_t)(ctx->model.n_vars > 0 ? ctx->model.n_vars : 9);
        *per_gate = n_vars * 4 + 1 + 4 + 8 + 16 + 1 + 8 + 8
                  + n_hyd * (4 + 4 + CPOL_MAX_PAR * 8 + 16 + 4 + CPOL_N_SZ * 8 + 16);
    }
    return CPOL_OK;
}

int cpol_host_free(cpol_ctx *ctx, void *p)
{
    if (!p) return CPOL_ERR_ARG;
    if (!ctx) return hipHostFree(p) == hipSuccess ? CPOL_OK : CPOL_ERR_HIP;   // a context-free block (see above)
    for (size_t i = 0; i < ctx->host_allocs.size(); ++i)
        if (ctx->host_allocs[i] == p) {
            HIPCHK(hipStreamSynchronize(ctx->stream));      // no copy into it may be in flight
            (void)hipHostFree(p);
            ctx->host_allocs.erase(ctx->host_allocs.begin() + (long)i);
            return CPOL_OK;
        }
    ctx->err = "cpol_host_free: not a cpol_host_alloc pointer of this context";
    return CPOL_ERR_ARG;
}

int cpol_enable_timing(cpol_ctx *ctx, int on)
{
    if (!ctx) return CPOL_ERR_ARG;
    ctx->timing = (on == 2) ? 2 : (on != 0);
    ctx->ev_used = 0;                  // restart the averaging window
    return CPOL_OK;
}

int cpol_stage_model(cpol_ctx *ctx, int n_vars, const float *const *data, const float *zlevels,
                     int nz, int ny, int nx, const float llc[2], const float urc[2],
                     const float res[2], const double south_pole[2])
{
    if (!ctx || !data || !zlevels || n_vars < 1 || n_vars > CPOL_MAX_VARS || nz < 3 || ny < 2 ||
        nx < 2) {
        if (ctx) ctx->err = "cpol_stage_model: bad arguments (need nz >= 3, ny, nx >= 2)";
        return CPOL_ERR_ARG;
    }
    (void)hipGetLastError();            // a stale error of another user of the runtime in this thread is not ours
    if (ctx->parent || ctx->n_children) {
        ctx->err = "cpol_stage_model: not on a lane, and not while lanes of this context exist (cpol_fork)";
        return CPOL_ERR_ARG;
    }
    HIPCHK(hipSetDevice(ctx->device));
    const long ncell = (long)ny * nx;
    const size_t plane_bytes = (size_t)nz * ncell * sizeof(float);
    const size_t h_bytes = (plane_bytes + 7) & ~(size_t)7;        // + (top, lowest level) per column behind the levels
    ENSURE(ctx->d_H, h_bytes + (size_t)ncell * sizeof(float2));
    ENSURE(ctx->d_V, plane_bytes * n_vars);
    DevBuf tmp;
    int rc = ensure(ctx, tmp, plane_bytes);
    if (rc != CPOL_OK) return rc;
    const int blk = 256, grd = cdiv(ncell, blk);
    HIPCHK(hipMemcpyAsync(tmp.p, zlevels, plane_bytes, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_stage_heights, dim3(grd), dim3(blk), 0, ctx->stream, (const float *)tmp.p,
                       (float *)ctx->d_H.p, (float2 *)((char *)ctx->d_H.p + h_bytes), nz, ncell);
    for (int v = 0; v < n_vars; ++v) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipMemcpyAsync(tmp.p, data[v], plane_bytes, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_stage_variable, dim3(grd), dim3(blk), 0, ctx->stream,
                           (const float *)tmp.p, (float *)ctx->d_V.p, nz, ncell, n_vars, v);
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    free_buf(tmp);
    ModelDev &m = ctx->model;
    m.H = (const float *)ctx->d_H.p;
    m.HT = (const float2 *)((const char *)ctx->d_H.p + h_bytes);
    m.V = (const float *)ctx->d_V.p;
    m.n_vars = n_vars; m.nz = nz; m.ny = ny; m.nx = nx;
    m.llc0 = llc[0]; m.llc1 = llc[1];
    m.urc0 = urc[0]; m.urc1 = urc[1];
    m.res0 = res[0]; m.res1 = res[1];
    m.rres0 = 1.0 / (double)m.res0; m.rres1 = 1.0 / (double)m.res1;      // (IEEE division on the host: correctly rounded)
    // rotation constants (oracle/cosmo_pol_oracle/geodesy.py: rotation_constants)
    const double theta = (90.0 + south_pole[0]) * CPOL_DEG, phi = south_pole[1] * CPOL_DEG;
    const double ct = cos(theta), st = sin(theta), cp = cos(phi), sp = sin(phi);
    m.ctcp = ct * cp; m.ctsp = ct * sp; m.st = st; m.nsp = -sp; m.cp = cp;
    m.nstcp = -st * cp; m.stsp = st * sp; m.ct = ct;
    ctx->model_staged = true;
    ctx->stage_serial++;
    // the coordinate polynomials of the resident table sets hold the PREVIOUS model's rotated-pole matrix (round-5 advisor
    // finding: a second cube with another south pole, the same rays again -> wrong grid cells without an error)
    for (auto &ts : ctx->tsets) ts.poly_version = 0;
    return CPOL_OK;
}

static int n_par_of(int rule)
{
    switch (rule) {
    case CPOL_RULE_ICE_1MOM: return 3;
    case CPOL_RULE_MELTING_SNOW:
    case CPOL_RULE_MELTING_GRAUPEL: return 3;
    default: return 2;
    }
}

int cpol_set_num_hydro(cpol_ctx *ctx, int n_hydro)
{
    if (!ctx || n_hydro < 0 || n_hydro > CPOL_MAX_HYDRO) return CPOL_ERR_ARG;
    if (ctx->parent || ctx->n_children) {
        ctx->err = "cpol_set_num_hydro: not on a lane, and not while lanes of this context exist";
        return CPOL_ERR_ARG;
    }
    long total = 0;
    for (int j = 0; j < n_hydro; ++j) total += (long)ctx->hs.h[j].d.n_e * ctx->hs.h[j].d.n_t;
    if (total > 1024L * CPOL_SCAN_MAX_PER) {           // checked BEFORE any state changes
        ctx->err = "too many LUT slices (elevation x temperature bins) for the bucket scan";
        return CPOL_ERR_ARG;
    }
    ctx->hs.n_hydro = n_hydro;
    ctx->stage_serial++;
    ctx->lut_serial++;
    int base = 0;
    for (int j = 0; j < n_hydro; ++j) {
        ctx->hs.h[j].key_base = base;
        base += ctx->hs.h[j].d.n_e * ctx->hs.h[j].d.n_t;
    }
    ctx->hs.n_keys = base;
    return CPOL_OK;
}

int cpol_stage_hydro(cpol_ctx *ctx, int slot, const cpol_hydro_desc *desc, const double *table,
                     const double *pre, const double *dnu, const double *aux, int n_aux)
{
    if (!ctx || !desc || !table || slot < 0 || slot >= CPOL_MAX_HYDRO || desc->n_e < 1 ||
        desc->n_t < 1 || desc->n_d < 2) {
        if (ctx) ctx->err = "cpol_stage_hydro: bad arguments";
        return CPOL_ERR_ARG;
    }
    if (ctx->parent || ctx->n_children) {
        ctx->err = "cpol_stage_hydro: not on a lane, and not while lanes of this context exist (cpol_fork)";
        return CPOL_ERR_ARG;
    }
    HIPCHK(hipSetDevice(ctx->device));
    const size_t tb = (size_t)desc->n_e * desc->n_t * desc->n_d * CPOL_N_SZ * sizeof(double);
    const size_t db = (size_t)desc->n_d * sizeof(double);
    int rc;
    HydroDev &h = ctx->hs.h[slot];
    const double *dev_table = nullptr;
    if (desc->table_id) {
        // the scattering table of a known identity stays resident (CPOL_ITAB_CACHE most recent):
        // a switch back to a table set seen before uploads only the small per-bin arrays
        for (auto &e : ctx->table_cache)
            if (e.id == desc->table_id && e.bytes == tb) { dev_table = (const double *)e.buf.p; e.used = ++ctx->table_clock; break; }
        if (!dev_table) {
            if (ctx->table_cache.size() >= CPOL_ITAB_CACHE) {
                // evict the least recently used entry that no staged slot points at
                size_t victim = ctx->table_cache.size();
                for (size_t k = 0; k < ctx->table_cache.size(); ++k) {
                    bool live = false;
                    for (int q = 0; q < CPOL_MAX_HYDRO; ++q)
                        live = live || (q != slot && ctx->hydro_staged[q] && ctx->hs.h[q].table == ctx->table_cache[k].buf.p);
                    if (!live && (victim == ctx->table_cache.size() || ctx->table_cache[k].used < ctx->table_cache[victim].used))
                        victim = k;
                }
                if (victim < ctx->table_cache.size()) {
                    free_buf(ctx->table_cache[victim].buf);
                    ctx->table_cache.erase(ctx->table_cache.begin() + victim);
                }
            }
            if (ctx->table_cache.size() < CPOL_ITAB_CACHE) {
                ctx->table_cache.emplace_back();
                TableCacheEntry &e = ctx->table_cache.back();
                if ((rc = upload(ctx, e.buf, table, tb)) != CPOL_OK) { ctx->table_cache.pop_back(); return rc; }
                e.id = desc->table_id; e.bytes = tb; e.used = ++ctx->table_clock;
                dev_table = (const double *)e.buf.p;
            }
        }
    }
    if (!dev_table) {
        if ((rc = upload(ctx, ctx->d_table[slot], table, tb)) != CPOL_OK) return rc;
        dev_table = (const double *)ctx->d_table[slot].p;
    }
    h.d = *desc;
    h.table = dev_table;
    h.pre = h.dnu = h.aux = nullptr;
    h.rcsw = nullptr;
    ctx->ss.s[slot] = SpecDev{};
    if (pre) {
        if ((rc = upload(ctx, ctx->d_pre[slot], pre, db)) != CPOL_OK) return rc;
        h.pre = (const double *)ctx->d_pre[slot].p;
    }
    if (dnu) {
        if ((rc = upload(ctx, ctx->d_dnu[slot], dnu, db)) != CPOL_OK) return rc;
        h.dnu = (const double *)ctx->d_dnu[slot].p;
    }
    if (aux && n_aux > 0) {
        if ((rc = upload(ctx, ctx->d_aux[slot], aux, (size_t)n_aux * sizeof(double))) != CPOL_OK)
            return rc;
        h.aux = (const double *)ctx->d_aux[slot].p;
    }
    if (desc->psd_family == CPOL_PSD_GAMMA && (!pre || !dnu)) {
        ctx->err = "cpol_stage_hydro: gamma family needs pre[] and dnu[]";
        return CPOL_ERR_ARG;
    }
    if (desc->psd_family == CPOL_PSD_GAMMA && desc->numeric_intv &&
        (desc->uniform_grid || !aux || n_aux < 3 * desc->n_d + 1)) {
        ctx->err = "cpol_stage_hydro: numeric_intv needs aux[3*n_d+1] and excludes uniform_grid";
        return CPOL_ERR_ARG;
    }
    if (desc->psd_family == CPOL_PSD_GAMMA && desc->uniform_grid &&
        (!aux || n_aux < 5 * desc->n_d + 1)) {
        ctx->err = "cpol_stage_hydro: uniform_grid needs aux[1 + 5 n_d]";
        return CPOL_ERR_ARG;
    }
    if (desc->psd_family == CPOL_PSD_MELTING && desc->tab_degree != 0 &&
        (desc->tab_degree != CPOL_MELT_DEGREE || !aux ||
         (long)n_aux < 2L * desc->n_t + (long)desc->n_t * desc->n_d * CPOL_MELT_FUNCS * (CPOL_MELT_DEGREE + 1))) {
        ctx->err = "cpol_stage_hydro: tab_degree must be 0 or CPOL_MELT_DEGREE with aux[2 n_t + n_t n_d 4 (degree+1)]";
        return CPOL_ERR_ARG;
    }
    if (desc->psd_family == CPOL_PSD_ICE_FIELD && desc->tab_degree != 0 &&
        (desc->tab_degree != CPOL_ICE_DEGREE || !desc->uniform_grid || !aux || n_aux < 12 * desc->n_d + 12 ||
         (long)n_aux < 12L * desc->n_d + 12 + (long)aux[12 * desc->n_d + 10] * CPOL_ICE_FUNCS * (CPOL_ICE_DEGREE + 1))) {
        ctx->err = "cpol_stage_hydro: ice tab_degree must be 0 or CPOL_ICE_DEGREE with the lambda-panel tables behind the recurrence block of aux[]";
        return CPOL_ERR_ARG;
    }
    if (desc->psd_family == CPOL_PSD_ICE_FIELD && (!aux || n_aux < 4 * desc->n_d + 1)) {
        ctx->err = "cpol_stage_hydro: ice family needs aux[4*n_d+1]";
        return CPOL_ERR_ARG;
    }
    h.n_par = n_par_of(desc->rule);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->hydro_staged[slot] = true;
    ctx->stage_serial++;
    ctx->lut_serial++;
    const int rc_n = cpol_set_num_hydro(ctx, slot >= ctx->hs.n_hydro ? slot + 1 : ctx->hs.n_hydro);
    if (rc_n != CPOL_OK) {
        ctx->hydro_staged[slot] = false;               // the slot does not count as staged
        return rc_n;
    }
    return CPOL_OK;
}

int cpol_stage_t_function(cpol_ctx *ctx, int which, const float *table)
{
    if (!ctx || !table || which < 0 || which >= CPOL_N_TFUN) {
        if (ctx) ctx->err = "cpol_stage_t_function: bad arguments";
        return CPOL_ERR_ARG;
    }
    if (ctx->parent || ctx->n_children) {
        ctx->err = "cpol_stage_t_function: not on a lane, and not while lanes of this context exist (cpol_fork)";
        return CPOL_ERR_ARG;
    }
    HIPCHK(hipSetDevice(ctx->device));
    int rc;
    if ((rc = upload(ctx, ctx->d_tfun[which], table, (size_t)CPOL_TFUN_COUNT * sizeof(float))) != CPOL_OK)
        return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->tfun[which] = (const float *)ctx->d_tfun[which].p;
    ctx->stage_serial++;
    return CPOL_OK;
}

int cpol_prepare(cpol_ctx *ctx)
{
    if (!ctx) return CPOL_ERR_ARG;
    if (ctx->parent) return CPOL_OK;
    for (int j = 0; j < ctx->hs.n_hydro; ++j)
        if (!ctx->hydro_staged[j]) { ctx->err = "cpol_prepare: hydrometeor slot not staged"; return CPOL_ERR_ARG; }
    return build_itabs(ctx);
}

int cpol_stage_doppler_weights(cpol_ctx *ctx, int slot, const double *weights)
{
    if (!ctx || !weights || slot < 0 || slot >= CPOL_MAX_HYDRO || !ctx->hydro_staged[slot]) {
        if (ctx) ctx->err = "cpol_stage_doppler_weights: stage the hydrometeor first";
        return CPOL_ERR_ARG;
    }
    if (ctx->parent || ctx->n_children) {
        ctx->err = "cpol_stage_doppler_weights: not on a lane, and not while lanes of this context exist (cpol_fork)";
        return CPOL_ERR_ARG;
    }
    HIPCHK(hipSetDevice(ctx->device));
    const cpol_hydro_desc &d = ctx->hs.h[slot].d;
    const size_t bytes = (size_t)d.n_e * d.n_t * d.n_d * 2 * sizeof(double);
    int rc;
    if ((rc = upload(ctx, ctx->d_rcsw[slot], weights, bytes)) != CPOL_OK) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->hs.h[slot].rcsw = (const double *)ctx->d_rcsw[slot].p;
    ctx->stage_serial++;
    ctx->lut_serial++;
    return CPOL_OK;
}

int cpol_stage_spectrum_tables(cpol_ctx *ctx, int slot, const float *rcs32, const float *dgrid)
{
    if (!ctx || !rcs32 || !dgrid || slot < 0 || slot >= CPOL_MAX_HYDRO || !ctx->hydro_staged[slot]) {
        if (ctx) ctx->err = "cpol_stage_spectrum_tables: stage the hydrometeor first";
        return CPOL_ERR_ARG;
    }
    if (ctx->parent || ctx->n_children) {
        ctx->err = "cpol_stage_spectrum_tables: not on a lane, and not while lanes of this context exist (cpol_fork)";
        return CPOL_ERR_ARG;
    }
    HIPCHK(hipSetDevice(ctx->device));
    HydroDev &h = ctx->hs.h[slot];
    // (melting species: rcs32 alone is read -- their diameter grid belongs to the gate's wet fraction, cpol_spectrum.inl)
    const size_t nr = (size_t)h.d.n_e * h.d.n_t * h.d.n_d;
    int rc;
    if ((rc = upload(ctx, ctx->d_rcs32[slot], rcs32, nr * sizeof(float))) != CPOL_OK) return rc;
    if ((rc = upload(ctx, ctx->d_dgrid[slot], dgrid, (size_t)3 * h.d.n_d * sizeof(float))) != CPOL_OK) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->ss.s[slot].rcs32 = (const float *)ctx->d_rcs32[slot].p;
    ctx->ss.s[slot].dgrid = (const float *)ctx->d_dgrid[slot].p;
    ctx->ss.s[slot].step32 = dgrid[1] - dgrid[0];
    return CPOL_OK;
}

int cpol_interp_points(cpol_ctx *ctx, int n, const float *coords, const float *heights, float *out)
{
    if (!ctx || !ctx->model_staged || n < 1 || !coords || !heights || !out) {
        if (ctx) ctx->err = "cpol_interp_points: model not staged or bad arguments";
        return CPOL_ERR_ARG;
    }
    (void)hipGetLastError();            // a stale error of another user of the runtime in this thread is not ours
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf c, h, o;
    int rc;
    if ((rc = upload(ctx, c, coords, (size_t)n * 2 * sizeof(float))) != CPOL_OK) return rc;
    if ((rc = upload(ctx, h, heights, (size_t)n * sizeof(float))) != CPOL_OK) return rc;
    const size_t ob = (size_t)n * ctx->model.n_vars * sizeof(float);
    if ((rc = ensure(ctx, o, ob)) != CPOL_OK) return rc;
    hipLaunchKernelGGL(k_interp_points, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, ctx->model,
                       (const float *)c.p, (const float *)h.p, (float *)o.p, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, o.p, ob, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    free_buf(c); free_buf(h); free_buf(o);
    return CPOL_OK;
}

int cpol_ray_tables(const cpol_sweep_params *p, const double *az_deg, const double *el_deg,
                    const double *pts_h_deg, const double *pts_v_deg, double *traj_out,
                    double *geo_out)
{
    if (!p || !az_deg || !el_deg || !pts_h_deg || !pts_v_deg || !traj_out || !geo_out)
        return CPOL_ERR_ARG;
    const double a = 6378137.0, f = CPOL_WGS84_F, b = (1.0 - f) * a;
    for (int r = 0; r < p->n_rays; ++r) {
        for (int j = 0; j < p->n_vnodes; ++j) {
            double el = (pts_v_deg[j] + el_deg[r]) * CPOL_DEG;
            double *o = traj_out + ((long)r * p->n_vnodes + j) * 4;
            o[0] = el; o[1] = sin(el); o[2] = cos(el); o[3] = pts_v_deg[j] + el_deg[r];
        }
        for (int i = 0; i < p->n_hnodes; ++i) {
            double alpha1 = (pts_h_deg[i] + az_deg[r]) * CPOL_DEG;
            double sin_a1 = sin(alpha1), cos_a1 = cos(alpha1);
            double tan_u1 = p->sin_u1 / p->cos_u1;
            double sigma1 = atan2(tan_u1, cos_a1);
            double sin_alpha = p->cos_u1 * sin_a1;
            double cos2_alpha = 1.0 - sin_alpha * sin_alpha;
            double u2 = cos2_alpha * (a * a - b * b) / (b * b);
            double A = 1.0 + u2 / 16384.0 * (4096.0 + u2 * (-768.0 + u2 * (320.0 - 175.0 * u2)));
            double B = u2 / 1024.0 * (256.0 + u2 * (-128.0 + u2 * (74.0 - 47.0 * u2)));
            double C = f / 16.0 * cos2_alpha * (4.0 + f * (4.0 - 3.0 * cos2_alpha));
            double *o = geo_out + ((long)r * p->n_hnodes + i) * 8;
            o[0] = sin_a1; o[1] = cos_a1; o[2] = sigma1; o[3] = sin_alpha;
            o[4] = b * A; o[5] = B; o[6] = C; o[7] = alpha1;
        }
    }
    return CPOL_OK;
}

// Measurement knob (CPOL_TABLE_UPLOAD=kernel): the per-ray tables of a sweep from the page-locked staging slot (host memory
// the device can address) into the table set's device buffer by a kernel instead of hipMemcpyAsync.  It showed that the
// ~340 us the first asynchronous copy of a sweep cost in round 4's "slow mode" belong to whichever copy call comes first
// (host-to-device with the memcpy, the device-to-host copy of the results with this kernel): see report_domain_error.
__global__ __launch_bounds__(256) void k_upload_tables(uint4 *__restrict__ dst, const uint4 *__restrict__ src, long n16)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

static inline double now_ns()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e9 + (double)ts.tv_nsec;
}

static int copy_out(cpol_ctx *ctx, void *dst, const void *src, size_t bytes, bool dst_on_device)
{
    if (!dst) return CPOL_OK;
    HIPCHK(hipMemcpyAsync(dst, src, bytes,
                          dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                          ctx->stream));
    return CPOL_OK;
}

int cpol_run_sweep(cpol_ctx *ctx, const cpol_sweep_params *p, const cpol_ray_tables_t *t,
                   cpol_outputs *out)
{
    if (!ctx) return CPOL_ERR_ARG;
    if (!p || !t || !out || !ctx->model_staged || ctx->hs.n_hydro < 1 || p->n_rays < 1 ||
        p->n_gates < 1 || p->n_sub < 1 || p->n_hnodes < 1 || p->n_vnodes < 1 || !t->traj ||
        !t->geo || !t->sub_h || !t->sub_v || !t->sub_w) {
        ctx->err = "cpol_run_sweep: model / hydrometeors not staged or bad arguments";
        return CPOL_ERR_ARG;
    }
    const double t_enter = now_ns();
    (void)hipGetLastError();            // a stale error of another user of the runtime in this thread is not ours
    for (int j = 0; j < ctx->hs.n_hydro; ++j)
        if (!ctx->hydro_staged[j]) { ctx->err = "cpol_run_sweep: hydrometeor slot not staged"; return CPOL_ERR_ARG; }
    HIPCHK(hipSetDevice(ctx->device));
    if (!ctx->parent && ctx->itab_serial != ctx->lut_serial) {
        const int rc_it = build_itabs(ctx);                // once per staged table set
        if (rc_it != CPOL_OK) return rc_it;
    }
    hipStream_t st = ctx->stream;
    const int n_rays = p->n_rays, ng = p->n_gates, n_sub = p->n_sub;
    const int n_h = p->n_hnodes, n_v = p->n_vnodes;
    const int n_vars = ctx->model.n_vars, n_hyd = ctx->hs.n_hydro, n_keys = ctx->hs.n_keys;
    const long n_rg = (long)n_rays * ng;
    const long n_sbg = n_rg * n_sub;
    if (n_sbg >= (1L << 31)) { ctx->err = "cpol_run_sweep: too many sub-beam gates in one call"; return CPOL_ERR_ARG; }
    {
        // re-validated here: a C caller may have reached this state through cpol_stage_hydro alone
        long total = 0;
        for (int j = 0; j < n_hyd; ++j) total += (long)ctx->hs.h[j].d.n_e * ctx->hs.h[j].d.n_t;
        if (total != n_keys || n_keys > 1024 * CPOL_SCAN_MAX_PER) {
            ctx->err = "cpol_run_sweep: inconsistent / too many LUT slices for the bucket scan (restage the hydrometeors)";
            return CPOL_ERR_ARG;
        }
        if ((long)n_rays * n_sub >= (1L << 31) || cdiv(ng, 256) > 65535 || p->outputs_on_device < 0 ||
            p->outputs_on_device > 2) {
            ctx->err = "cpol_run_sweep: launch-grid limits exceeded (n_rays * n_sub < 2^31, n_gates <= 65535 * 256) or bad outputs_on_device";
            return CPOL_ERR_ARG;
        }
    }
    int rc;

    // ---- per-sweep host tables -> device (skipped when the caller's tag is unchanged) ----
    const int mode = p->geometry_mode;
    if ((mode == CPOL_GEOM_SPACEBORNE && !t->site) || (mode == CPOL_GEOM_HOST_PATHS && !t->paths) ||
        mode < 0 || mode > 2) {
        ctx->err = "cpol_run_sweep: geometry_mode needs tables->site (spaceborne) / tables->paths (host paths)";
        return CPOL_ERR_ARG;
    }
    const bool cut = p->apply_sensitivity && t->sens_thr;
    const bool ml = t->sub_smooth != nullptr;
    if (ml && (!t->ml_filter || t->ml_radius < 0 || t->ml_radius > 64)) {
        ctx->err = "cpol_run_sweep: sub_smooth needs ml_filter / ml_radius";
        return CPOL_ERR_ARG;
    }
    const long shape[6] = {n_rays, ng, n_sub, n_h, n_v, (ml ? 64 + t->ml_radius * 128L : 0) + (long)mode * 8 + (t->nyquist ? 4 : 0) + (t->site ? 2 : 0) + (cut ? 1 : 0)
                           + (t->varray ? 16 + p->n_vbins * 65536L : 0)};
    void **const views[11] = {&ctx->v_traj_in, &ctx->v_site, &ctx->v_geo, &ctx->v_subh, &ctx->v_subv, &ctx->v_subw,
                              &ctx->v_sens, &ctx->v_nyq, &ctx->v_subsmooth, &ctx->v_mlfilter, &ctx->v_varray};
    cpol_ctx::TableSet *set = nullptr, *lru = &ctx->tsets[0];
    for (auto &ts : ctx->tsets) {
        if (t->version != 0 && ts.version == t->version && memcmp(shape, ts.shape, sizeof shape) == 0) set = &ts;
        if (ts.last_use < lru->last_use) lru = &ts;
    }
    const bool reuse = set != nullptr;
    if (!set) set = lru;
    set->last_use = ++ctx->tset_clock;
    if (reuse) {
        for (int k = 0; k < 11; ++k) *views[k] = set->views[k];
    } else {
        // pack every table of this sweep into one pinned staging slot, one H2D copy
        struct Item { const void *src; size_t bytes; };
        const Item items[11] = {
            {t->traj, (size_t)n_rays * n_v * CPOL_TRAJ_STRIDE * sizeof(double)},
            {t->site, t->site ? (size_t)n_rays * CPOL_SITE_STRIDE * sizeof(double) : 0},
            {t->geo, (size_t)n_rays * n_h * CPOL_GEO_STRIDE * sizeof(double)},
            {t->sub_h, (size_t)n_sub * sizeof(int)},
            {t->sub_v, (size_t)n_sub * sizeof(int)},
            {t->sub_w, (size_t)n_sub * sizeof(double)},
            {cut ? t->sens_thr : nullptr, cut ? (size_t)ng * sizeof(double) : 0},
            {t->nyquist, t->nyquist ? (size_t)n_rays * sizeof(double) : 0},
            {ml ? t->sub_smooth : nullptr, ml ? (size_t)n_sub * sizeof(int) : 0},
            {ml ? t->ml_filter : nullptr, ml ? (size_t)(2 * t->ml_radius + 1) * sizeof(double) : 0},
            {t->varray, t->varray ? (size_t)p->n_vbins * sizeof(double) : 0},
        };
        size_t total = 0;
        for (const Item &it : items) total += (it.bytes + 63) & ~(size_t)63;
        set->version = 0;                                        // (not valid until the copy is queued)
        set->poly_version = 0;                                   // (other rays: their polynomials are made again)
        ENSURE(set->buf, total);
        cpol_ctx::Staging &sg = ctx->stg[ctx->stg_next];
        ctx->stg_next = (ctx->stg_next + 1) % 4;
        const double t_s0 = now_ns();
        if (sg.used) HIPCHK(hipEventSynchronize(sg.ev));        // its last copy has left the buffer
        const double t_s1 = now_ns();
        if (sg.cap < total) {
            if (sg.p) (void)hipHostFree(sg.p);
            sg.p = nullptr; sg.cap = 0;
            HIPCHK(hipHostMalloc(&sg.p, total + total / 4 + 4096, hipHostMallocDefault));
            sg.cap = total + total / 4 + 4096;
        }
        if (!sg.ev) HIPCHK(hipEventCreateWithFlags(&sg.ev, hipEventDisableTiming));
        size_t off = 0;
        for (int k = 0; k < 11; ++k) {
            const Item &it = items[k];
            set->views[k] = it.bytes ? (void *)((char *)set->buf.p + off) : nullptr;
            *views[k] = set->views[k];
            if (it.bytes) memcpy((char *)sg.p + off, it.src, it.bytes);
            off += (it.bytes + 63) & ~(size_t)63;
        }
        const double t_s2 = now_ns();
        if (ctx->upload_kernel) {
            const long n16 = (long)(total / 16);          // (every table is padded to 64 bytes)
            hipLaunchKernelGGL(k_upload_tables, dim3((unsigned)std::min<long>(cdiv(n16, 256), 256)), dim3(256), 0, ctx->stream,
                               (uint4 *)set->buf.p, (const uint4 *)sg.p, n16);
        } else {
            HIPCHK(hipMemcpyAsync(set->buf.p, sg.p, total, hipMemcpyHostToDevice, ctx->stream));
        }
        const double t_s3 = now_ns();
        HIPCHK(hipEventRecord(sg.ev, ctx->stream));
        const double t_s4 = now_ns();
        ctx->host_ns[6] += t_s1 - t_s0; ctx->host_ns[7] += t_s2 - t_s1; ctx->host_ns[8] += t_s3 - t_s2; ctx->host_ns[9] += t_s4 - t_s3;
        sg.used = true;
        set->version = t->version;
        memcpy(set->shape, shape, sizeof shape);
    }
    const double t_tables = now_ns();
    // ---- work buffers ----
    // several sub-beams: the ray paths (shared by the horizontal nodes of a vertical node) and the per-ray
    // constants of the geodesic come from k_trajectory instead of once per sub-beam gate
    const bool ray_prep = n_sub >= CPOL_RAY_PREP_MIN_SUB;
    const bool prep_paths = ray_prep && n_h > 1 && mode != CPOL_GEOM_HOST_PATHS;
    if (mode == CPOL_GEOM_HOST_PATHS || ctx->keep_debug || prep_paths)
        ENSURE(ctx->b_traj, (size_t)n_rays * n_v * 3 * ng * sizeof(float));
    if (ray_prep) ENSURE(ctx->b_rayc, ((size_t)n_rays * n_h + n_rays) * 2 * sizeof(double));
    // the rotated coordinates of the non-central sub-beams as polynomials of the arc distance (cpol_interp.inl: k_trajectory):
    // ground radars on the 4/3-earth ray paths, one site (CPOL_GEO_POLY=0: the short closed form of round 4)
    const bool geo_poly = ray_prep && ctx->geo_poly && mode == CPOL_GEOM_GROUND_43 && !t->site;
    // single-beam sweeps (round 5): the one sub-beam takes the polynomials too when its float64 latitude / longitude are not
    // outputs; they belong to the resident table set of the rays and are made once per (version, range grid)
    const bool poly_single = !ray_prep && ctx->geo_poly && ctx->geo_poly_central && mode == CPOL_GEOM_GROUND_43 && !t->site &&
                             (!ctx->keep_debug || ctx->geo_poly_central == 2) && t->version != 0 && !(p->debug_flags & CPOL_DEBUG_EXACT_SUBBEAMS);
    // (round 6: also when the float64 latitude / longitude are outputs -- the long form then runs for those two arrays alone and
    // the float32 grid coordinates still come from the guarded polynomials: identical calls give identical bits whether or not the
    // caller fetches the gate coordinates, round-5 advisor finding)
    // arc distance <= slant range; a margin of 1e-3 for the asin of the 4/3-earth formula
    const double geo_poly_scale = 2.0 / ((p->range0 + (double)(ng - 1) * p->range_step) * 1.001);
    if (geo_poly || poly_single) {
        if (geo_poly) ENSURE(ctx->b_poly, (size_t)n_rays * n_h * 2 * CPOL_GEO_NP * sizeof(double));
        if (!ctx->d_geoM.p) {
            // Chebyshev-node values -> monomial coefficients (as build_itabs' M), extended precision on the host
            constexpr int NP = CPOL_GEO_NP;
            long double T[NP][NP] = {};
            T[0][0] = 1.0L;
            if (NP > 1) T[1][1] = 1.0L;
            for (int k = 2; k < NP; ++k)
                for (int pw = 0; pw < NP; ++pw) T[k][pw] = (pw > 0 ? 2.0L * T[k - 1][pw - 1] : 0.0L) - T[k - 2][pw];
            const long double pi = 3.141592653589793238462643383279502884L;
            double M[NP * NP];
            for (int pw = 0; pw < NP; ++pw)
                for (int q = 0; q < NP; ++q) {
                    long double acc = 0.0L;
                    for (int k = 0; k < NP; ++k) acc += T[k][pw] * (k == 0 ? 1.0L : 2.0L) / NP * cosl(pi * k * (q + 0.5L) / NP);
                    M[pw * NP + q] = (double)acc;
                }
            rc = upload(ctx, ctx->d_geoM, M, sizeof M);
            if (rc != CPOL_OK) return rc;
            HIPCHK(hipStreamSynchronize(ctx->stream));           // (M is a stack array)
        }
        if (poly_single && (!set->poly.p || set->poly_version != set->version || set->poly_scale != geo_poly_scale ||
                            set->poly_site[0] != p->radar_lon || set->poly_site[1] != p->sin_u1 || set->poly_site[2] != p->cos_u1)) {
            ENSURE(set->poly, (size_t)n_rays * n_h * 2 * CPOL_GEO_NP * sizeof(double));
            TrajArgs tp{};
            tp.geo = (const double *)ctx->v_geo;
            tp.n_rays = n_rays; tp.n_h = n_h; tp.n_v = n_v; tp.n_gates = ng; tp.mode = mode;
            tp.lon1 = p->radar_lon; tp.sin_u1 = p->sin_u1; tp.cos_u1 = p->cos_u1;
            tp.poly = (double *)set->poly.p;
            tp.poly_M = (const double *)ctx->d_geoM.p;
            tp.poly_scale = geo_poly_scale;
            hipLaunchKernelGGL(k_trajectory, dim3((unsigned)cdiv((long)n_rays * n_h, 256 / CPOL_GEO_NP)), dim3(256), 0, ctx->stream, ctx->model, tp);
            set->poly_version = set->version;
            set->poly_scale = geo_poly_scale;
            set->poly_site[0] = p->radar_lon; set->poly_site[1] = p->sin_u1; set->poly_site[2] = p->cos_u1;
        }
    }
    ENSURE(ctx->b_vals, (size_t)n_vars * n_sbg * sizeof(float));
    ENSURE(ctx->b_mask, (size_t)n_sbg);
    ENSURE(ctx->b_elev, (size_t)n_sbg * sizeof(float));
    if (ctx->keep_debug) ENSURE(ctx->b_coords, (size_t)n_sbg * 2 * sizeof(float));
    if (ml) ENSURE(ctx->b_wgate, (size_t)n_sbg * sizeof(double));
    ENSURE(ctx->b_qmelt, (size_t)2 * n_sbg * sizeof(float));
    ENSURE(ctx->b_fwmelt, (size_t)2 * n_sbg * sizeof(double));
    ENSURE(ctx->b_key, (size_t)n_hyd * n_sbg * sizeof(int));
    ENSURE(ctx->b_pos, (size_t)n_hyd * n_sbg * sizeof(int));
    ENSURE(ctx->b_par, (size_t)n_hyd * CPOL_MAX_PAR * n_sbg * sizeof(double));
    // per set: [n_keys] bucket counts, [n_keys + 2]: items ranked, [n_keys + 3 ...): items on integral tables (count_table_items)
    const long cnt_stride = n_keys + 3 + CPOL_COUNT_SLOTS;
    const bool was_dirty = ctx->counters_dirty;          // (the previous sequence was cut short: its rays' tickets may be half taken too)
    {
        void *const was = ctx->b_count.p, *const was_t = ctx->b_totals.p;
        ENSURE(ctx->b_count, (size_t)2 * cnt_stride * sizeof(int));
        ENSURE(ctx->b_totals, 2 * 4 * sizeof(long long));
        // (counters_dirty: the previous call returned an error after its counting kernel was queued -- an ENSURE that
        // found no memory, a failed copy or capture: its set holds counts and the serial did not advance, so the set
        // would be used again as it is; the memsets are queued behind whatever that call left on the stream)
        if (ctx->b_count.p != was || ctx->b_totals.p != was_t || ctx->count_stride != cnt_stride || ctx->counters_dirty) {
            HIPCHK(hipMemsetAsync(ctx->b_count.p, 0, (size_t)2 * cnt_stride * sizeof(int), ctx->stream));
            HIPCHK(hipMemsetAsync(ctx->b_totals.p, 0, 2 * 4 * sizeof(long long), ctx->stream));
            ctx->count_stride = cnt_stride;
            ctx->counters_dirty = false;
        }
    }
    const int par = (int)(ctx->sweep_serial & 1);        // (the serial advances once the sequence is queued)
    int *const cnt_p = (int *)ctx->b_count.p + par * cnt_stride, *const cnt_next = (int *)ctx->b_count.p + (par ^ 1) * cnt_stride;
    long long *const tot_p = (long long *)ctx->b_totals.p + par * 4, *const tot_next = (long long *)ctx->b_totals.p + (par ^ 1) * 4;
    ENSURE(ctx->b_blkranked, (size_t)cdiv(n_sbg, CPOL_CLASSIFY_THREADS) * sizeof(int));
    ENSURE(ctx->b_rec, (size_t)n_hyd * n_sbg * sizeof(double2));
    ENSURE(ctx->b_vmask, (size_t)n_sbg);
    ENSURE(ctx->b_offset, (size_t)2 * n_keys * sizeof(int));        // item and unit offsets
    // (work units: up to 64 / 128 sorted items each; declared here, sized below once the launch mode is known)
    long unit_cap = (long)n_hyd * n_sbg / 64 + n_keys + 64;
    ENSURE(ctx->b_perm, (size_t)n_hyd * n_sbg * sizeof(int));
    ENSURE(ctx->b_res, (size_t)n_hyd * n_sbg * CPOL_N_SZ * sizeof(double));
    const bool doppler = p->simulate_doppler != 0;
    const bool dop2 = p->simulate_doppler == 2;
    const bool dop3 = p->simulate_doppler == 3;
    const int n_vb = p->n_vbins;
    bool spec_melt = false;                // Doppler scheme 3: a melting species is staged (its fall-speed tables take LDS of k_spec_gate)
    if (dop3) {
        if (n_vb < 2 || n_vb > 4097 || !t->varray || p->var_rho < 0 || p->var_rho >= n_vars) {
            ctx->err = "cpol_run_sweep: Doppler scheme 3 needs n_vbins in [2, 4097], tables->varray and var_rho";
            return CPOL_ERR_ARG;
        }
        for (int j = 0; j < n_hyd; ++j) {
            if (!ctx->ss.s[j].rcs32) { ctx->err = "cpol_run_sweep: Doppler scheme 3 needs cpol_stage_spectrum_tables"; return CPOL_ERR_ARG; }
            if (ctx->hs.h[j].d.n_d != ctx->hs.h[0].d.n_d) {     // the LDS image is sized from slot 0
                ctx->err = "cpol_run_sweep: Doppler scheme 3 needs the same number of diameter bins in every table";
                return CPOL_ERR_ARG;
            }
        }
        for (int j = 0; j < n_hyd; ++j) spec_melt = spec_melt || ctx->hs.h[j].d.psd_family == CPOL_PSD_MELTING;
        if ((spec_melt ? ((size_t)2 * ctx->hs.h[0].d.n_d + 256) * sizeof(double) : 0) + ((size_t)n_hyd * (ctx->hs.h[0].d.n_d + n_vb) + n_vb) * sizeof(float) > 160 * 1024 - 256) {
            // (gfx950: 160 KB of LDS per CU, all of it available to ONE workgroup of k_spec_gate when the launch asks for it --
            // six species with FFT_length = 2048, the upper end of the reference's valid range (cfg.py:91), need 100 KB)
            ctx->err = "cpol_run_sweep: Doppler scheme 3: n_hydro x (n_d + n_vbins) exceeds the 160 KB of LDS of a gfx950 CU";
            return CPOL_ERR_ARG;
        }
    }
    if (dop2)
        for (int j = 0; j < n_hyd; ++j)
            if (!ctx->hs.h[j].rcsw) { ctx->err = "cpol_run_sweep: Doppler scheme 2 needs cpol_stage_doppler_weights"; return CPOL_ERR_ARG; }
    if (doppler) {
        ENSURE(ctx->b_vn, (size_t)n_hyd * n_sbg * 2 * sizeof(double));
        ENSURE(ctx->b_icefirst, (size_t)n_rays * n_sub * sizeof(IceFirst));
        if (!dop3 && n_sub >= 4) ENSURE(ctx->b_proj, (size_t)n_sbg * sizeof(double));     // k_rvel_terms
    }
    // (every argument check and every allocation of the sequence happens before its first launch: an error return
    // further down would leave the sweep's counter set half used; see counters_dirty)
    if ((size_t)3 * ng * sizeof(float) > 64 * 1024) { ctx->err = "cpol_run_sweep: n_gates too large for the range scans (3 * n_gates floats of LDS)"; return CPOL_ERR_ARG; }
    if (out->mask_sum8 && 2 * n_sub > 127) { ctx->err = "cpol_run_sweep: outputs->mask_sum8 needs 2 * n_sub <= 127 (one byte per gate)"; return CPOL_ERR_ARG; }
    if (dop3) {
        ENSURE(ctx->b_beam, (size_t)n_sbg * n_vb * sizeof(float));
    }
    // ---- outputs: where the kernels write each array, and how it reaches the caller ----
    enum { O_ZH, O_ZV, O_ZDR, O_KDP, O_DHV, O_PHIDP, O_RHOHV, O_ATTH, O_ATTV, O_MASK, O_LAT, O_LON,
           O_DIST, O_HGT, O_RVEL, O_MODEL, O_SZT, O_SPEC, O_MASK8, O_N };
    const bool dev = p->outputs_on_device == 1;
    const bool async_host = p->outputs_on_device == 2;    // pinned host buffers, no wait
    // the sub-beam sums by one thread per (gate, hydrometeor) with the 1-D table items evaluated in
    // place (k_subbeam_sum); CPOL_SUBSUM=0: k_psd_lookup stores them and k_final walks the rows
    static const bool subsum_enabled = !(getenv("CPOL_SUBSUM") && atoi(getenv("CPOL_SUBSUM")) == 0);
    bool subsum = subsum_enabled;
    {
        bool any1d = false;
        for (int j = 0; j < n_hyd; ++j) any1d = any1d || (ctx->its.t[j].tab && !ctx->its.t[j].two_d);
        // (not with Doppler scheme 3: k_spec_atten reads every item's columns from res[]; with one
        // sub-beam there is nothing to accumulate and the extra launch costs more than it saves)
        subsum = subsum && any1d && !dop3;
    }
    // ... with fewer than 4 sub-beams k_final itself evaluates them in place (no extra launch, and
    // k_psd_lookup no longer writes 96 B per item for k_final to read back)
    // -- where that saves the k_psd_lookup launch altogether (no melting species, no Doppler sums from the tables: the
    // C2 sweep 122 -> 119 us and 46 MB less traffic).  Where k_psd_lookup runs anyway it keeps storing the columns: the
    // gathers inside the per-ray workgroups of k_final cost more than the stored rows (C3 sweep at 3 deg: 199 against
    // 188 us; 512-thread workgroups held to 128 / 168 VGPRs: 138 / 159 us on the C2 sweep; round 4)
    bool final_inplace = subsum && n_sub < 4;
    for (int j = 0; j < n_hyd && final_inplace; ++j) {
        const ItabDev &tj = ctx->its.t[j];
        if (tj.tab && (tj.two_d || (doppler && tj.writes_vn))) final_inplace = false;
    }
    subsum = subsum && n_sub >= 4;
    // The single-beam fast path (cpol_gate.inl): one sub-beam per radial and every slot on an integral table --
    // k_gate1 takes a gate from its interpolated model values to its polarimetric variables in one kernel
    // (k_classify + k_psd_lookup + the per-gate half of k_final), items outside the tables go to the integrating
    // kernels as one-item work units without the counting sort.  CPOL_GATE1=0 (or CPOL_SUBSUM=0, debug reads, a
    // slot without a table, Doppler scheme 3) keeps the general launch sequence; the results are bit-identical.
    // Not with melting species: their 2-D blocks are walked by whole wavefronts, which inside this one fat kernel
    // (six species in sequence per thread) is no faster than k_classify + k_psd_lookup, and the four idle
    // launches of the integrating flavours cost more than the two bucket launches they replace (C3 sweep at
    // 3 deg: 203 us this way against 188; the kernel handles them -- CPOL_GATE1=2 forces it, tests do).
    const int gate1_env = ctx->gate1;
    bool gate1 = gate1_env != 0 && subsum_enabled && n_sub == 1 && !ctx->keep_debug && !dop3 && !ml;
    for (int j = 0; j < n_hyd && gate1; ++j)
        gate1 = ctx->its.t[j].tab != nullptr && (gate1_env == 2 || !ctx->its.t[j].two_d);
    if (gate1) final_inplace = true;       // (k_final's recomputed gates take the table items from their records)
    if (gate1) {
        ENSURE(ctx->b_gscan, (size_t)3 * n_rg * sizeof(float));
        ENSURE(ctx->b_defer, (size_t)n_rg);
    }
    // k_gate1_ray (cpol_gate.inl): the single-beam kernel with one wavefront per species, the ray's range scans by the
    // workgroup that finishes the ray last and the items outside the tables integrated in place -- the sweep is
    // k_interp_sweep + that kernel, no integrating launch, no k_final.  Only where the in-place integration mirrors the
    // integrating kernels: gamma-family species without Doppler-scheme-2 sums and without per-ray fall-speed totals
    // (numeric_intv), whose tables kept all panels but the tail (an item off the table costs a wavefront ~40 us:
    // fine for the handful a volume has, not for a table that lost half of its panels to the accuracy gate).
    const int g1r = ctx->gate1_ray >= 0 ? ctx->gate1_ray
                  : ((ctx->parent ? ctx->parent->n_children : ctx->n_children) >= 2 ? 1 : 0);
    bool gate1_ray = gate1 && g1r && !ctx->fuse_gate1 && !p->with_melting && !dop2 && n_rays <= 65535;
    for (int j = 0; j < n_hyd && gate1_ray; ++j) {
        const cpol_hydro_desc &d = ctx->hs.h[j].d;
        const ItabDev &tj = ctx->its.t[j];
        gate1_ray = d.psd_family == CPOL_PSD_GAMMA && !d.numeric_intv && d.q_source == CPOL_Q_MODEL && tj.tab && !tj.two_d &&
                    ((tj.pan_lo == 0 && tj.pan_hi >= tj.n_pan - 2) || g1r >= 2) && ctx->hs.h[j].pre && ctx->hs.h[j].dnu;
    }
    {
        const long g1_waves = n_rg * n_hyd / 64 / 1024;
        gate1_ray = gate1_ray && (ctx->gate1_species == 2 || (ctx->gate1_species == 1 && g1_waves < 48));
    }
    if (gate1_ray && CPOL_GATE1_PRESENT) {
        // (the presence words: every one "anything may be here" until k_interp_sweep has written it -- a wavefront none of whose gates
        // exists leaves its word alone)
        void *const was = ctx->b_present.p;
        ENSURE(ctx->b_present, (size_t)n_rays * cdiv(ng, 64) * sizeof(unsigned));
        if (ctx->b_present.p != was) HIPCHK(hipMemsetAsync(ctx->b_present.p, 0xFF, ctx->b_present.cap, ctx->stream));
    }
    if (gate1_ray && g1r == 3) {
        void *const was = ctx->b_ticket.p;
        ENSURE(ctx->b_ticket, (size_t)n_rays * sizeof(int));
        if (ctx->b_ticket.p != was || was_dirty)                 // (the kernel leaves every ticket at 0 behind it)
            HIPCHK(hipMemsetAsync(ctx->b_ticket.p, 0, ctx->b_ticket.cap, ctx->stream));
    }
    // Every slot on an integral table: the items outside the tables (a handful per volume) are listed directly as
    // one-item work units by k_classify / k_gate1 -- key in b_pos, gate in b_perm, count in b_totals -- and the
    // counting sort (LDS ranking in k_classify, k_bucket_scan, k_bucket_scatter) is not run at all.  CPOL_RARE_DIRECT=0,
    // debug reads (bucket counts) or a slot without a table keep the sort: with EVERY item integrated one item per
    // work unit would waste 63 of 64 lanes.
    bool rare_direct = gate1 || (ctx->rare_direct != 0 && !ctx->keep_debug);
    for (int j = 0; j < n_hyd && rare_direct; ++j) rare_direct = ctx->its.t[j].tab != nullptr;
    if (rare_direct) unit_cap = (long)n_hyd * n_sbg;           // (virtual units: no b_units entries are written)
    else ENSURE(ctx->b_units, (size_t)unit_cap * sizeof(WorkUnit));
    const bool want_szi = ctx->keep_debug || subsum;
    if (want_szi) ENSURE(ctx->b_szinteg, (size_t)n_rg * n_hyd * CPOL_N_SZ * sizeof(float));
    const bool want_szt = out->sz_total != nullptr || ctx->keep_debug;
    const bool want_model = p->integrate_model && out->model_vars;
    // every slot on a table, no debug reads: the gate kernel classifies its gates itself (k_interp_classify,
    // cpol_fused.inl)
    const bool fused = ctx->fuse_classify != 0 && rare_direct && !gate1 && !ml && !dop3 && !ctx->keep_debug;
    // k_interp_gate1 (CPOL_FUSE_GATE1=1, not the default): the single-beam kernel interpolates its gates too.  Measured: the
    // isolated C2 sweep 95.4 -> 88.6 us (one lane back to back: 70 -> 62 us per sweep), but with three lanes in flight 42.2 ->
    // 44.8 us per sweep, and the Ku swath of config 5 (9 800 rays) 0.93 -> 1.18 ms: at the 3 wavefronts per SIMD k_gate1 needs,
    // the interpolation -- VALU-bound at 5 -- loses more than the saved launch and the 14 MB of vals[] give back.
    const bool fused_gate1 = ctx->fuse_gate1 != 0 && gate1;
    void *const user_out[O_N] = {out->ZH, out->ZV, out->ZDR, out->KDP, out->DELTA_HV, out->PHIDP,
                                 out->RHOHV, out->ATT_H, out->ATT_V, out->mask, out->lats, out->lons,
                                 out->dist, out->heights, out->RVEL, out->model_vars, out->sz_total,
                                 out->DSPECTRUM, out->mask_sum8};
    size_t obytes[O_N];
    bool produced[O_N];
    DevBuf *own[O_N];
    for (int k = 0; k < 14; ++k) {
        obytes[k] = (size_t)n_rg * ((k == O_MASK || k == O_LAT || k == O_LON) ? sizeof(double) : sizeof(float));
        produced[k] = true;
        own[k] = &ctx->b_out[k];
    }
    obytes[O_RVEL] = (size_t)n_rg * sizeof(double);            produced[O_RVEL] = doppler;    own[O_RVEL] = &ctx->b_rvel;
    obytes[O_MODEL] = (size_t)n_vars * n_rg * sizeof(double);  produced[O_MODEL] = want_model; own[O_MODEL] = &ctx->b_model;
    obytes[O_SZT] = (size_t)n_rg * CPOL_N_SZ * sizeof(float);  produced[O_SZT] = want_szt;    own[O_SZT] = &ctx->b_sztotal;
    obytes[O_SPEC] = (size_t)n_rg * n_vb * sizeof(double);     produced[O_SPEC] = dop3;       own[O_SPEC] = &ctx->b_spectrum;
    // the radial mask as one byte per gate (the sum of the sub-beams' codes): only when asked for; the float64 form is then
    // written only if it is asked for too
    obytes[O_MASK8] = (size_t)n_rg;                            produced[O_MASK8] = out->mask_sum8 != nullptr; own[O_MASK8] = &ctx->b_mask8;
    if (out->mask_sum8 && !out->mask && !ctx->keep_debug) produced[O_MASK] = false;
    void *T[O_N];
    // pinned-host mode: when the requested arrays lie in one window of the caller's slab (only
    // alignment padding between them) the kernels write into a device image of that window and
    // ONE device-to-host copy moves it (instead of up to 18 copies of a few hundred KB each)
    char *win_lo = nullptr, *win_hi = nullptr;
    size_t win_sum = 0;
    if (async_host && !ctx->keep_debug)
        for (int k = 0; k < O_N; ++k) {
            if (!produced[k] || !user_out[k]) continue;
            char *a = (char *)user_out[k];
            if (!win_lo || a < win_lo) win_lo = a;
            if (!win_hi || a + obytes[k] > win_hi) win_hi = a + obytes[k];
            win_sum += obytes[k];
        }
    const bool window = win_lo && (size_t)(win_hi - win_lo) <= win_sum + win_sum / 4 + 4096;
    // (the device image keeps the window's alignment modulo 64 B, so that every array of the image is
    // aligned exactly like its host counterpart: a float32 array at an address 4 mod 8 followed by a
    // float64 array must not shift the latter to a misaligned device address)
    const size_t win_skew = window ? ((size_t)(uintptr_t)win_lo & 63) : 0;
    if (window) ENSURE(ctx->b_outwin, (size_t)(win_hi - win_lo) + 64);
    for (int k = 0; k < O_N; ++k) {
        T[k] = nullptr;
        if (!produced[k]) continue;
        if (dev && user_out[k] && !(ctx->keep_debug && k == O_SZT)) { T[k] = user_out[k]; continue; }   // in place
        if (window && user_out[k]) { T[k] = (char *)ctx->b_outwin.p + win_skew + ((char *)user_out[k] - win_lo); continue; }
        ENSURE(*own[k], obytes[k]);
        T[k] = own[k]->p;
    }

    // the bucket counters start at zero: cleared by k_interp_sweep (no fill kernel); the domain
    // error word is sticky (cleared when reported)
    int *const d_errflag = ctx->d_errword;

    const bool tm = ctx->timing == 1;          // events around every stage
    const bool tm_psd = ctx->timing != 0;      // ... or only around the PSD stage
    if (tm_psd) {
        if (ctx->ev_used == ctx->ev_sets.size()) {
            hipEvent_t *set = new hipEvent_t[EV_N];
            for (int k = 0; k < EV_N; ++k) HIPCHK(hipEventCreate(&set[k]));
            ctx->ev_sets.push_back(set);
        }
        ctx->ev = ctx->ev_sets[ctx->ev_used++];
        if (tm) HIPCHK(hipEventRecord(ctx->ev[EV_T0], st));
    }

    // The launch sequence of a sweep.  Opt-in (CPOL_USE_GRAPH=1): with device outputs and
    // nothing to upload it is captured into a HIP graph and replayed while the arguments stay
    // the same (one graph launch instead of ten kernel launches).
    auto launch_all = [&]() -> int {
    // ---- 1. ray paths: evaluated inside k_interp_sweep; host-supplied paths are uploaded ----
    if (mode == CPOL_GEOM_HOST_PATHS) {
        HIPCHK(hipMemcpyAsync(ctx->b_traj.p, t->paths, (size_t)n_rays * n_v * 3 * ng * sizeof(float),
                              hipMemcpyHostToDevice, st));
    }
    if (ray_prep || (ctx->keep_debug && mode != CPOL_GEOM_HOST_PATHS)) {
        // ray paths + per-ray constants ahead of the sweep kernel (and the parity access to the paths,
        // cpol_debug_read "traj"): same device functions as the in-place evaluation
        const bool paths = mode != CPOL_GEOM_HOST_PATHS && (prep_paths || ctx->keep_debug);
        TrajArgs ta{};
        ta.ray_traj = (const double *)ctx->v_traj_in;
        ta.site = t->site ? (const double *)ctx->v_site : nullptr;
        ta.traj_out = paths ? (float *)ctx->b_traj.p : nullptr;
        ta.n_rays = n_rays; ta.n_v = n_v; ta.n_gates = ng; ta.mode = mode;
        ta.range0 = p->range0; ta.range_step = p->range_step;
        ta.ke = p->ke; ta.re = p->re; ta.alt = p->radar_alt;
        ta.geo = (const double *)ctx->v_geo;
        ta.ray_const = ray_prep ? (double *)ctx->b_rayc.p : nullptr;
        ta.n_h = n_h; ta.lon1 = p->radar_lon;
        if (geo_poly) {
            ta.poly = (double *)ctx->b_poly.p;
            ta.poly_M = (const double *)ctx->d_geoM.p;
            ta.poly_scale = geo_poly_scale;
            ta.sin_u1 = p->sin_u1; ta.cos_u1 = p->cos_u1;
        }
        hipLaunchKernelGGL(k_trajectory, dim3((unsigned)(n_rays * n_v), paths ? cdiv(ng, 256) : 1), dim3(256), 0, st, ctx->model, ta);
    }
    if (tm) HIPCHK(hipEventRecord(ctx->ev[EV_TRAJ], st));

    // ---- 2. gate interpolation ----
    InterpArgs ia{};
    ia.traj = (mode == CPOL_GEOM_HOST_PATHS || prep_paths) ? (const float *)ctx->b_traj.p : nullptr;
    ia.ray_const = ray_prep ? (const double *)ctx->b_rayc.p : nullptr;
    ia.rp.ray_traj = (const double *)ctx->v_traj_in;
    ia.rp.site = t->site ? (const double *)ctx->v_site : nullptr;
    ia.rp.n_v = n_v; ia.rp.mode = mode;
    ia.rp.range0 = p->range0; ia.rp.range_step = p->range_step;
    ia.rp.ke = p->ke; ia.rp.re = p->re; ia.rp.alt = p->radar_alt;
    ia.zero_buf = cnt_next;
    ia.zero_n = (int)cnt_stride;
    ia.zero_buf2 = (int *)tot_next;     // (4 long long: k_classify / k_gate1 count the items outside the tables into them)
    ia.zero_n2 = 8;
    ia.geo = (const double *)ctx->v_geo;
    ia.sub_h = (const int *)ctx->v_subh;
    ia.sub_v = (const int *)ctx->v_subv;
    ia.vals = (float *)ctx->b_vals.p;
    ia.mask = (signed char *)ctx->b_mask.p;
    ia.elev = (float *)ctx->b_elev.p;
    ia.coords = ctx->keep_debug ? (float *)ctx->b_coords.p : nullptr;
    ia.lats = (double *)T[O_LAT];
    ia.lons = (double *)T[O_LON];
    ia.dist = (float *)T[O_DIST];
    ia.heights = (float *)T[O_HGT];
    ia.error_flag = d_errflag;
    ia.n_rays = n_rays; ia.n_gates = ng; ia.n_sub = n_sub; ia.n_h = n_h; ia.n_v = n_v;
    ia.central_sub = n_sub / 2;
    ia.sin_u1 = p->sin_u1; ia.cos_u1 = p->cos_u1; ia.lon1 = p->radar_lon;
    ia.site = t->site ? (const double *)ctx->v_site : nullptr;
    ia.exact_sub = (p->debug_flags & CPOL_DEBUG_EXACT_SUBBEAMS) ? 1 : 0;
    // (the one sub-beam of a single-beam sweep: the polynomials of its table set, unless its float64 coordinates are outputs)
    const bool poly_central = poly_single;
    if (poly_central && !ctx->keep_debug && !user_out[O_LAT] && !user_out[O_LON]) ia.lats = ia.lons = nullptr;   // (nobody reads the library's own copies)
    ia.poly = geo_poly ? (const double *)ctx->b_poly.p : poly_central ? (const double *)set->poly.p : nullptr;
    ia.poly_scale = (geo_poly || poly_central) ? geo_poly_scale : 0.0;
    ia.poly_central = poly_central ? 1 : 0;
    static const int use_present = getenv("CPOL_GATE1_PRESENT") ? atoi(getenv("CPOL_GATE1_PRESENT")) : 1;
    if (gate1_ray && use_present && CPOL_GATE1_PRESENT) {
        ia.present = (unsigned *)ctx->b_present.p;
        ia.n_pres = n_hyd;
        for (int j = 0; j < n_hyd; ++j) ia.pres_var[j] = ctx->hs.h[j].d.var_q;
    }
    ctx->last_poly_central = ia.poly_central;
    static const int exp_skip_interp = getenv("CPOL_EXP_SKIP") ? (atoi(getenv("CPOL_EXP_SKIP")) & 1) : 0;
    if (!fused && !fused_gate1 && !exp_skip_interp)
    hipLaunchKernelGGL(k_interp_sweep, dim3((unsigned)(n_rays * n_sub), cdiv(ng, 256)), dim3(256), 0, st,
                       ctx->model, ia);
    if (tm && !fused && !fused_gate1) HIPCHK(hipEventRecord(ctx->ev[EV_INTERP], st));

    // ---- 2b. the arguments of the final stage (k_gate1, the single-beam fast path, needs them already) ----
    FinalArgs fa{};
    fa.res = (const double *)ctx->b_res.p;
    fa.key = (const int *)ctx->b_key.p;
    fa.vmask = (const unsigned char *)ctx->b_vmask.p;
    fa.sub_mask = (const signed char *)ctx->b_mask.p;
    fa.vals = (const float *)ctx->b_vals.p;
    fa.sub_w = (const double *)ctx->v_subw;
    fa.sz_integ = want_szi ? (float *)ctx->b_szinteg.p : nullptr;
    fa.sz_total = want_szt ? (float *)T[O_SZT] : nullptr;
    fa.ZH = (float *)T[O_ZH]; fa.ZV = (float *)T[O_ZV];
    fa.ZDR = (float *)T[O_ZDR]; fa.KDP = (float *)T[O_KDP];
    fa.DELTA_HV = (float *)T[O_DHV]; fa.RHOHV = (float *)T[O_RHOHV];
    fa.ATT_H = (float *)T[O_ATTH]; fa.ATT_V = (float *)T[O_ATTV];
    fa.mask = (double *)T[O_MASK];
    fa.mask8 = (signed char *)T[O_MASK8];
    fa.model_vars = want_model ? (double *)T[O_MODEL] : nullptr;
    fa.n_rays = n_rays; fa.n_gates = ng; fa.n_sub = n_sub; fa.n_hydro = n_hyd; fa.n_vars = n_vars;
    fa.c_zh = (float)p->c_zh;
    fa.c_kdp = (float)(1e-3 * (180.0 / 3.14159265358979323846) * p->wavelength);
    fa.c_2w = (float)(2 * p->wavelength);
    double sum_w = 0;
    for (int s = 0; s < n_sub; ++s) sum_w += t->sub_w[s];
    fa.sum_w = sum_w;
    fa.with_attenuation = p->with_attenuation;
    fa.res_km = (float)(p->radial_res / 1000.);
    fa.wgate = ml ? (const double *)ctx->b_wgate.p : nullptr;
    fa.RVEL = nullptr;
    if (doppler) {
        fa.RVEL = (double *)T[O_RVEL];
        fa.vn = (const double *)ctx->b_vn.p;
        fa.ice_first = (const IceFirst *)ctx->b_icefirst.p;
        fa.geo = (const double *)ctx->v_geo;
        fa.sub_h = (const int *)ctx->v_subh;
        fa.elev = (const float *)ctx->b_elev.p;
        fa.n_h = n_h;
        fa.var_u = p->var_u; fa.var_v = p->var_v; fa.var_w = p->var_w;
        fa.nyquist = t->nyquist ? (const double *)ctx->v_nyq : nullptr;
        if (fa.var_u < 0 || fa.var_v < 0 || fa.var_w < 0 || fa.var_u >= n_vars ||
            fa.var_v >= n_vars || fa.var_w >= n_vars) {
            ctx->err = "cpol_run_sweep: simulate_doppler needs var_u / var_v / var_w";
            return CPOL_ERR_ARG;
        }
        for (int j = 0; j < n_hyd && !dop3; ++j) {
            const cpol_hydro_desc &d = ctx->hs.h[j].d;
            // 1: vn[] per gate -- written by the PSD stage (scheme 2, melting species) or, for the analytic
            // moments of the gamma species under scheme 1, by k_classify; 2: summed over the ray (1-moment
            // ice, numeric integrate_V) and credited to the first valid gate
            fa.vsrc[j] = (dop2 || d.psd_family == CPOL_PSD_MELTING) ? 1
                       : (d.psd_family == CPOL_PSD_ICE_FIELD || d.numeric_intv) ? 2 : 1;
        }
    }

    fa.eval_1d = final_inplace ? 1 : 0;
    fa.rec = (const double2 *)ctx->b_rec.p;
    for (int j = 0; j < n_hyd; ++j) fa.key_base[j] = ctx->hs.h[j].key_base;
    bool any_vsrc2 = false;
    for (int j = 0; j < n_hyd; ++j) any_vsrc2 = any_vsrc2 || (doppler && !dop3 && fa.vsrc[j] == 2);
    if (gate1) {
        fa.pre_gate = 1;
        fa.ice_redo = any_vsrc2 ? 1 : 0;
        fa.defer = (const unsigned char *)ctx->b_defer.p;
        fa.sk = (const float *)ctx->b_gscan.p;
        fa.sh = fa.sk + n_rg;
        fa.sv = fa.sk + 2 * n_rg;
    }

    // ---- 3. melting + PSD parameters + bucket histogram ----
    ClassifyArgs ca{};
    ca.vals = (float *)ctx->b_vals.p;
    ca.mask = (const signed char *)ctx->b_mask.p;
    ca.elev = (const float *)ctx->b_elev.p;
    ca.q_melt = ctx->keep_debug ? (float *)ctx->b_qmelt.p : nullptr;
    ca.fw_melt = ctx->keep_debug ? (double *)ctx->b_fwmelt.p : nullptr;
    ca.key = (int *)ctx->b_key.p;
    ca.pos = (int *)ctx->b_pos.p;
    ca.par = (double *)ctx->b_par.p;
    ca.count = cnt_p;
    ca.n_sbg = n_sbg;
    ca.present = ia.present;              // (k_gate1_ray: the presence words k_interp_sweep has just written)
    ca.with_melting = p->with_melting;
    ca.var_qr = ca.var_qs = ca.var_qg = -1;
    ca.doppler = doppler ? 1 : 0;
    ca.tfun_snow = ctx->tfun[CPOL_TFUN_SNOW_N0];
    ca.tfun_ice = ctx->tfun[CPOL_TFUN_ICE_MOM2_A];
    // integral tables: not with Doppler scheme 3 + ice (k_spec_gate needs every item's parameters as
    // the integrating kernels leave them; the lookup writes the same slots, so it is fine) -- always on
    ca.n_lookup = cnt_p + n_keys + 1;
    ca.blk_ranked = (int *)ctx->b_blkranked.p;
    ca.rec = (double2 *)ctx->b_rec.p;
    ca.vmask = (unsigned char *)ctx->b_vmask.p;
    ca.vn = (doppler && !dop2 && !dop3) ? (double *)ctx->b_vn.p : nullptr;     // analytic moments (Doppler scheme 1)
    ca.keep_par = (ctx->keep_debug || dop3) ? 1 : 0;
    if (rare_direct) {
        ca.rare_key = (int *)ctx->b_pos.p;
        ca.rare_perm = (int *)ctx->b_perm.p;
        ca.rare_totals = (unsigned long long *)tot_p;
    }
    // k_psd_lookup's workgroups first list the tiles among their own that hold a species with a 2-D table (LookupArgs::tile_scan):
    // when the lookup walks tiles and has nothing but the 2-D tables to evaluate, from 262 144 tiles on (C4 volume, 735 000 tiles:
    // 1.30 -> 1.20 ms; its share of 1/8, 92 000 tiles: 247 -> 239-257 us, nothing gained.  Measured and dropped on the way: the
    // list in global memory, made by k_interp_classify with one bit per tile and an atomicOr per item -- lookup 1.30 -> 1.09 ms
    // and 239 -> 205 us, but the classification 3.24 -> 4.31 ms: device-scope atomics; made by a kernel of its own with one
    // atomicAdd per wavefront and pass -- the same lookup times, and 244 / 51 us for that kernel: one address, ~11 ns per atomic)
    static const int tile_env = getenv("CPOL_LOOKUP_TILE") ? atoi(getenv("CPOL_LOOKUP_TILE")) : 1;
    constexpr int TILE_GATES = 1 << CPOL_TILE_GATES_LOG2, TILE_RAYS = 64 >> CPOL_TILE_GATES_LOG2;
    bool use_tile_list = false;
    const long n_tiles = (long)cdiv(n_rays, TILE_RAYS) * n_sub * cdiv(ng, TILE_GATES);
    if (ctx->lookup_list && rare_direct && !gate1 && tile_env && n_rays >= TILE_RAYS && (subsum || final_inplace) && (n_tiles >= 262144 || ctx->lookup_list == 2) && n_tiles < (1L << 31)) {
        bool any_2d = false, other = false;
        for (int j = 0; j < n_hyd; ++j) {
            const ItabDev &tj = ctx->its.t[j];
            if (!tj.tab) continue;
            if (tj.two_d) any_2d = true;
            else if ((dop3 && ctx->hs.h[j].d.psd_family == CPOL_PSD_ICE_FIELD) || (final_inplace && doppler && tj.writes_vn)) other = true;
        }
        use_tile_list = any_2d && !other;
    }
    for (int j = 0; j < n_hyd; ++j) {
        const cpol_hydro_desc &d = ctx->hs.h[j].d;
        if (d.q_source != CPOL_Q_MODEL) continue;
        if (d.rule == CPOL_RULE_RAIN_1MOM) ca.var_qr = d.var_q;
        if (d.rule == CPOL_RULE_SNOW_1MOM) ca.var_qs = d.var_q;
        if (d.rule == CPOL_RULE_GRAUPEL_1MOM) ca.var_qg = d.var_q;
    }
    if (p->with_melting && (ca.var_qr < 0 || ca.var_qs < 0 || ca.var_qg < 0)) {
        ctx->err = "cpol_run_sweep: melting needs 1-moment rain, snow and graupel slots";
        return CPOL_ERR_ARG;
    }
    if (ml) {
        MlArgs ma{};
        ma.vals = (const float *)ctx->b_vals.p;
        ma.sub_w = (const double *)ctx->v_subw;
        ma.sub_smooth = (const int *)ctx->v_subsmooth;
        ma.taps = (const double *)ctx->v_mlfilter;
        ma.wgate = (double *)ctx->b_wgate.p;
        ma.n_sbg = n_sbg; ma.n_sub = n_sub; ma.n_gates = ng; ma.radius = t->ml_radius;
        ma.with_melting = p->with_melting;
        ma.var_qr = ca.var_qr; ma.var_qs = ca.var_qs; ma.var_qg = ca.var_qg;
        hipLaunchKernelGGL(k_ml_weights, dim3(n_rays * n_sub), dim3(64), 0, st, ma);
        ca.wgate = (const double *)ctx->b_wgate.p;
    }
    // the variables later kernels read: U, V, W (the Doppler terms); all of them for the integrated model variables
    ia.store_mask = want_model ? 0xffffffffu : 0u;
    if (doppler) ia.store_mask |= (1u << p->var_u) | (1u << p->var_v) | (1u << p->var_w);
    if (fused) {
        hipLaunchKernelGGL(k_interp_classify, dim3((unsigned)(n_rays * n_sub), cdiv(ng, 256)), dim3(256),
                           (size_t)n_vars * 256 * sizeof(float), st, ctx->model, ia, ctx->hs, ctx->its, ca);
        if (tm) HIPCHK(hipEventRecord(ctx->ev[EV_INTERP], st));
    } else if (gate1) {
        GateArgs ga{};
        ga.sk = (float *)ctx->b_gscan.p;
        ga.sh = ga.sk + n_rg;
        ga.sv = ga.sk + 2 * n_rg;
        ga.defer = (unsigned char *)ctx->b_defer.p;
        ga.unit_key = (int *)ctx->b_pos.p;
        ga.perm = (int *)ctx->b_perm.p;
        ga.totals = (unsigned long long *)tot_p;
        ga.res = (double *)ctx->b_res.p;
        ga.store_items = any_vsrc2 ? 1 : 0;
        ga.analytic_vn = ca.vn ? 1 : 0;
        if (doppler) ca.vn = (double *)ctx->b_vn.p;        // (also the table-borne sums of a species summed over the ray)
        bool melt_tab = false;
        for (int j = 0; j < n_hyd; ++j) melt_tab = melt_tab || ctx->its.t[j].two_d;
        const dim3 ggrid((unsigned)n_rays, cdiv(ng, CPOL_GATE1_THREADS));
        const size_t glds = (size_t)n_vars * CPOL_GATE1_THREADS * sizeof(float);
        // one wavefront per species (k_gate1_species) where no melting species and no per-ray fall-speed sums are involved
        // -- for small launches: the C2 sweep (8 wavefronts per SIMD) 33.6 -> 30.7 us and 42.9 -> 39.2 us per sweep with three
        // lanes in flight; the C5 Ku swath (235 per SIMD, five species) 429 -> 674 us: every species' wavefront repeats the
        // gate's loads and wavefronts 1.. idle while wavefront 0 finishes the gates; five C2 sweeps as one sequence (44 per
        // SIMD) 112 -> 100 us (CPOL_GATE1_SPECIES=0 / 2: never / always)
        const long g1_waves_per_simd = n_rg * n_hyd / 64 / 1024;
        const bool by_species = (ctx->gate1_species == 2 || (ctx->gate1_species == 1 && g1_waves_per_simd < 48)) &&
                                !fused_gate1 && !melt_tab && !ga.store_items && !p->with_melting;
        if (gate1_ray && by_species) {
            // the whole rest of the sweep in this launch: no integrating kernels, no k_final
            ga.ticket = (int *)ctx->b_ticket.p;
            ScanRayArgs rr{};
            rr.PHIDP = (float *)T[O_PHIDP];
            rr.RVEL = nullptr;
            rr.sens_thr = cut ? (const double *)ctx->v_sens : nullptr;
            rr.radial_res = (float)p->radial_res;
            const size_t lds_terms = (size_t)n_hyd * (64 * GATE1S_BYTES + GATE1S_BLK_BYTES), lds_scan = (size_t)3 * ng * sizeof(float);
            const dim3 rgrid((unsigned)cdiv(ng, 64), (unsigned)n_rays);
            if (g1r == 3) {
                hipLaunchKernelGGL(k_gate1_ray_scan, rgrid, dim3(64 * n_hyd), lds_terms > lds_scan ? lds_terms : lds_scan, st,
                                   ctx->hs, ctx->its, ca, fa, ga, rr);
            } else {
                // (CPOL_EXP_SKIP, timing experiments only -- wrong results: bit 0 the gate interpolation, bit 1 this kernel, bit 2 the scans)
                static const int exp_skip = getenv("CPOL_EXP_SKIP") ? atoi(getenv("CPOL_EXP_SKIP")) : 0;
                if (!(exp_skip & 2))
                hipLaunchKernelGGL(k_gate1_ray, rgrid, dim3(64 * n_hyd), lds_terms, st, ctx->hs, ctx->its, ca, fa, ga, rr);
                if (tm) HIPCHK(hipEventRecord(ctx->ev[EV_CLASSIFY], st));
                if (tm_psd) { HIPCHK(hipEventRecord(ctx->ev[EV_BUCKET], st)); HIPCHK(hipEventRecord(ctx->ev[EV_PSD], st)); }
                if (!(exp_skip & 4))
                hipLaunchKernelGGL(k_scan_rays, dim3((unsigned)n_rays), dim3(256), lds_scan, st, fa, ga, rr);
                if (tm) HIPCHK(hipEventRecord(ctx->ev[EV_FINAL], st));
                HIPCHK(hipGetLastError());
                return CPOL_OK;
            }
            if (tm) HIPCHK(hipEventRecord(ctx->ev[EV_CLASSIFY], st));
            if (tm_psd) { HIPCHK(hipEventRecord(ctx->ev[EV_BUCKET], st)); HIPCHK(hipEventRecord(ctx->ev[EV_PSD], st)); }
            if (tm) HIPCHK(hipEventRecord(ctx->ev[EV_FINAL], st));
            HIPCHK(hipGetLastError());
            return CPOL_OK;
        }
        if (by_species) hipLaunchKernelGGL(k_gate1_species, dim3((unsigned)cdiv(n_rg, 64)), dim3(64 * n_hyd),
                                           (size_t)n_hyd * (64 * GATE1S_BYTES + GATE1S_BLK_BYTES), st, ctx->hs, ctx->its, ca, fa, ga);
        else if (fused_gate1 && melt_tab) hipLaunchKernelGGL((k_interp_gate1<true>), ggrid, dim3(CPOL_GATE1_THREADS), glds, st,
                                                        ctx->model, ia, ctx->hs, ctx->its, ca, fa, ga);
        else if (fused_gate1) hipLaunchKernelGGL((k_interp_gate1<false>), ggrid, dim3(CPOL_GATE1_THREADS), glds, st,
                                                 ctx->model, ia, ctx->hs, ctx->its, ca, fa, ga);
        else if (melt_tab) hipLaunchKernelGGL((k_gate1<true>), dim3(cdiv(n_rg, CPOL_GATE1_THREADS)), dim3(CPOL_GATE1_THREADS), 0, st,
                                         ctx->hs, ctx->its, ca, fa, ga);
        else hipLaunchKernelGGL((k_gate1<false>), dim3(cdiv(n_rg, CPOL_GATE1_THREADS)), dim3(CPOL_GATE1_THREADS), 0, st,
                                ctx->hs, ctx->its, ca, fa, ga);
        if (tm && fused_gate1) HIPCHK(hipEventRecord(ctx->ev[EV_INTERP], st));
    } else
    hipLaunchKernelGGL(k_classify, dim3(cdiv(n_sbg, CPOL_CLASSIFY_THREADS)),
                       dim3(CPOL_CLASSIFY_THREADS), 0, st, ctx->hs, ctx->its, ca);
    if (tm) HIPCHK(hipEventRecord(ctx->ev[EV_CLASSIFY], st));

    // ---- 4. counting sort by LUT slice ----
    ScanArgs sa{};
    sa.count = cnt_p;
    sa.offset = (int *)ctx->b_offset.p;
    sa.uoffset = (int *)ctx->b_offset.p + n_keys;
    sa.units = (WorkUnit *)ctx->b_units.p;
    sa.totals = tot_p;
    sa.n_keys = n_keys;
    sa.n_hydro = n_hyd;
    for (int j = 0; j < n_hyd; ++j) {
        const cpol_hydro_desc &d = ctx->hs.h[j].d;
        sa.key_base[j] = ctx->hs.h[j].key_base;
        sa.unit_shift[j] = ((d.psd_family == CPOL_PSD_GAMMA && d.uniform_grid) ||
                            (d.psd_family == CPOL_PSD_MELTING && d.tab_degree == CPOL_MELT_DEGREE) ||
                            (d.psd_family == CPOL_PSD_ICE_FIELD && d.uniform_grid && d.tab_degree == CPOL_ICE_DEGREE)) ? 7 : 6;
    }
    // (Measured and dropped: this chain -- scan, scatter, the integrating kernels, all idle when every item lies
    // on a table -- on a sibling stream beside k_psd_lookup, forked and joined with events: the isolated C2 sweep
    // 122 -> 135 us, the C3 volume 468 -> 476 us, the 225-ray C4 share 1.568 -> 1.553 ms: a cross-stream event
    // costs the device about as much as the three idle launches it would hide.)
    if (!rare_direct)
    hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(1024), 0, st, sa);
    // (a fixed grid: the workgroups stride over the k_classify gate ranges and skip the empty ones)
    const long n_cblk = cdiv(n_sbg, CPOL_CLASSIFY_THREADS);
    if (!rare_direct)
    hipLaunchKernelGGL(k_bucket_scatter, dim3((unsigned)(n_cblk < 2048 ? n_cblk : 2048)), dim3(256), 0, st,
                       (const int *)ctx->b_key.p, (const int *)ctx->b_pos.p,
                       (int *)ctx->b_perm.p, (const int *)ctx->b_blkranked.p,
                       (const unsigned char *)ctx->b_vmask.p, n_sbg, n_hyd, sa);      // + the work-unit list
    if (tm_psd) HIPCHK(hipEventRecord(ctx->ev[EV_BUCKET], st));

    // ---- 5a. items on an integral table: 15 x 11 coefficients gathered, no diameter-bin loop ----
    bool rare_forked = false;           // k_psd_rare goes to a sibling stream that forked in front of k_psd_lookup
    {
        bool any = false;
        for (int j = 0; j < n_hyd; ++j) any = any || ctx->its.t[j].tab != nullptr;
        if (any) {
            LookupArgs la{};
            la.key = (const int *)ctx->b_key.p;
            la.vmask = (const unsigned char *)ctx->b_vmask.p;
            la.rec = (const double2 *)ctx->b_rec.p;
            la.par = (const double *)ctx->b_par.p;
            la.par_w = dop3 ? (double *)ctx->b_par.p : nullptr;
            la.res = (double *)ctx->b_res.p;
            la.vn = doppler ? (double *)ctx->b_vn.p : nullptr;
            la.n_sbg = n_sbg;
            la.skip_res_1d = (subsum || final_inplace) ? 1 : 0;
            la.vn_1d = final_inplace ? 1 : 0;
            bool launch = !(subsum || final_inplace);   // with k_subbeam_sum / in-place evaluation in k_final: only for 2-D tables, Doppler sums and the ice intercept
            for (int j = 0; j < n_hyd && !launch; ++j) {
                const ItabDev &tj = ctx->its.t[j];
                if (!tj.tab) continue;
                launch = tj.two_d || (la.par_w && ctx->hs.h[j].d.psd_family == CPOL_PSD_ICE_FIELD) ||
                         (final_inplace && la.vn && tj.writes_vn);      // (k_subbeam_sum evaluates the Doppler sums itself)
            }
            bool two_d = false;
            for (int j = 0; j < n_hyd; ++j) two_d = two_d || (ctx->its.t[j].tab && ctx->its.t[j].two_d);
            la.tile = (two_d && tile_env && n_rays >= TILE_RAYS) ? 1 : 0;
            la.n_rays = n_rays; la.n_sub = n_sub; la.n_gates = ng;
            const long n_thr = la.tile ? (long)cdiv(n_rays, TILE_RAYS) * n_sub * cdiv(ng, TILE_GATES) * 64 : n_sbg;
            // few wavefronts (a single sweep with one sub-beam): the walk of the busiest tile is the kernel's duration;
            // its distinct blocks and species are dealt to `split` wavefronts (CPOL_LOOKUP_SPLIT=<n>; default by launch size)
            const long lookup_waves_per_simd = n_thr / 64 / 1024;
            la.split = ctx->lookup_split > 0 ? ctx->lookup_split : (la.tile && lookup_waves_per_simd < 16) ? 4 : 1;
            long grid_x = cdiv(n_thr, CPOL_LOOKUP_THREADS);
            if (use_tile_list && la.tile) {
                // a fixed grid walks the list: enough wavefronts to fill the chip three times over (5 per SIMD resident), never more than tiles
                // workgroups that own many tiles each: the chip filled `fill` times over (5 wavefronts per SIMD resident), a workgroup's
                // list never longer than its LDS array
                la.tile_scan = 1;
                la.n_tiles = n_tiles;
                for (int j = 0; j < n_hyd; ++j) if (ctx->its.t[j].tab && ctx->its.t[j].two_d) la.species2d |= 1u << j;
                static const long fill = getenv("CPOL_LOOKUP_FILL") ? atol(getenv("CPOL_LOOKUP_FILL")) : 12;      // (C4 volume, 1 / 3 / 6 / 12 / 24 / 48: 1.28 / 1.25 / 1.24 / 1.20 / 1.23 / 1.26 ms)
                const long cap_wg = std::max<long>(fill * 1024 * 5 * CPOL_WAVE / CPOL_LOOKUP_THREADS, cdiv(n_tiles, CPOL_LOOKUP_LIST_CAP));
                if (grid_x > cap_wg) grid_x = cap_wg;
            }
            if (launch && !gate1) {
                // the items outside the tables (k_psd_rare, below) and the 2-D table items (here) are disjoint and both wait for
                // the classification alone: k_psd_rare goes to a sibling stream that forks HERE and joins before the sub-beam
                // sums -- its one busy workgroup (69-80 us for a single item of the C4 volume) runs beside the lookup
                if (ctx->rare_overlap && rare_direct && ctx->psd_rare && !ctx->use_graph && !ctx->keep_debug) {
                    if (!ctx->ev_fork) HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
                    HIPCHK(hipEventRecord(ctx->ev_fork, st));
                    rare_forked = true;
                }
                hipLaunchKernelGGL(k_psd_lookup, dim3((unsigned)grid_x, la.split), dim3(CPOL_LOOKUP_THREADS), 0, st, ctx->hs, ctx->its, la);
            }
        }
    }


    // ---- 5. PSD x scattering table: one launch per kernel flavour present ----
    {
        PsdArgs pa{};
        pa.unit_key = rare_direct ? (const int *)ctx->b_pos.p : nullptr;
        pa.units = (const WorkUnit *)ctx->b_units.p;
        pa.totals = tot_p;
        pa.perm = (const int *)ctx->b_perm.p;
        pa.par = (const double *)ctx->b_par.p;
        pa.res = (double *)ctx->b_res.p;
        pa.vn = doppler ? (double *)ctx->b_vn.p : nullptr;
        pa.n_sbg = n_sbg;
        pa.par_w = dop3 ? (double *)ctx->b_par.p : nullptr;
        pa.clk = nullptr;
        static const int ice_force_sum = getenv("CPOL_ICE_FORCE_SUM") ? atoi(getenv("CPOL_ICE_FORCE_SUM")) : 0;
        pa.ice_force_sum = ice_force_sum;
        if (ctx->keep_debug) {
            ENSURE(ctx->b_clk, 2048 * 4 * sizeof(long long));
            HIPCHK(hipMemsetAsync(ctx->b_clk.p, 0, 2048 * 4 * sizeof(long long), st));
            pa.clk = (long long *)ctx->b_clk.p;
        }
        bool need[4] = {false, false, false, false};
        // CPOL_PSD_ONLY: experiment knob (tools/psd_flavours.py): bit mask of the flavours to launch
        // (1 gamma-exp, 2 recurrence, 4 ice, 8 melting); results are then incomplete
        static const int only = getenv("CPOL_PSD_ONLY") ? atoi(getenv("CPOL_PSD_ONLY")) : 15;
        for (int j = 0; j < n_hyd; ++j) {
            const cpol_hydro_desc &d = ctx->hs.h[j].d;
            int mode = d.psd_family == CPOL_PSD_ICE_FIELD ? PSD_MODE_ICE
                     : d.psd_family == CPOL_PSD_MELTING ? PSD_MODE_MELTING
                     : d.uniform_grid ? PSD_MODE_GAMMA_UNIFORM : PSD_MODE_GAMMA_EXP;
            if (only & (1 << mode)) need[mode] = true;
        }
        // persistent grids: 1024 workgroups walk the unit list with a static stride (smaller
        // grids, 512 / 768, measured equal or slower; CPOL_PSD_GRID* are experiment knobs)
        static const long grid_u = getenv("CPOL_PSD_GRID") ? atol(getenv("CPOL_PSD_GRID")) : 1024;
        static const long grid_g = getenv("CPOL_PSD_GRID_GENERIC") ? atol(getenv("CPOL_PSD_GRID_GENERIC")) : 1024;
        // (items listed directly: the units are single items outside the tables, a handful per volume -- a small
        // grid costs an idle launch less; a flood of them is still processed, by 128 workgroups)
        const long cap_u = rare_direct ? 128 : grid_u, cap_g = rare_direct ? 128 : grid_g;
        const dim3 grd_u((unsigned)(unit_cap < cap_u ? unit_cap : cap_u));
        const dim3 grd((unsigned)(unit_cap < cap_g ? unit_cap : cap_g)), blk(CPOL_PSD_THREADS);
        // The flavours touch disjoint items and could run side by side.  Measured (MI355X, one
        // sweep): on sibling streams (fork after the bucket sort, join before the final stage) the
        // PSD stage took 762 vs 734 us on C3 and 29.2 vs 27.1 ms on C4 -- every flavour is a
        // persistent grid that fills the chip and is VALU-bound, so overlap only adds the event
        // traffic.  Back to back on the sweep's stream is the default; CPOL_PSD_SIBLINGS=1 forks.
        static const bool siblings = getenv("CPOL_PSD_SIBLINGS") && atoi(getenv("CPOL_PSD_SIBLINGS")) != 0;
        const int order[4] = {PSD_MODE_MELTING, PSD_MODE_ICE, PSD_MODE_GAMMA_UNIFORM, PSD_MODE_GAMMA_EXP};
        // items listed directly (every slot on a table): ONE launch runs every flavour (k_psd_rare); CPOL_PSD_RARE=0: a launch
        // per flavour as before (same bits: tests/test_gpu_edges.py)
        if (rare_direct && ctx->psd_rare && only == 15 && !siblings && !ctx->keep_debug) {
            int modes = 0;
            for (int m = 0; m < 4; ++m) if (need[m]) modes |= 1 << m;
            for (int jj = 0; jj < n_hyd; ++jj) {
                const cpol_hydro_desc &dd = ctx->hs.h[jj].d;
                if (dd.psd_family == CPOL_PSD_MELTING && dd.tab_degree != CPOL_MELT_DEGREE) modes |= 16;
            }
            pa.ice_same_launch = 1;
            hipStream_t sr = st;
            if (rare_forked) {
                if (!ctx->aux[0]) HIPCHK(hipStreamCreateWithFlags(&ctx->aux[0], hipStreamNonBlocking));
                if (!ctx->ev_join[0]) HIPCHK(hipEventCreateWithFlags(&ctx->ev_join[0], hipEventDisableTiming));
                sr = ctx->aux[0];
                HIPCHK(hipStreamWaitEvent(sr, ctx->ev_fork, 0));
            }
            if (dop2) hipLaunchKernelGGL((k_psd_rare<true>), grd, blk, 0, sr, ctx->hs, pa, modes);
            else hipLaunchKernelGGL((k_psd_rare<false>), grd, blk, 0, sr, ctx->hs, pa, modes);
            if (rare_forked) {
                HIPCHK(hipEventRecord(ctx->ev_join[0], sr));
                HIPCHK(hipStreamWaitEvent(st, ctx->ev_join[0], 0));
                rare_forked = false;
            }
            for (int m = 0; m < 4; ++m) need[m] = false;
        }
        int n_need = 0;
        for (int m = 0; m < 4; ++m) n_need += need[m] ? 1 : 0;
        const bool fork = siblings && n_need > 1;
        if (fork) {
            if (!ctx->ev_fork) HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
            HIPCHK(hipEventRecord(ctx->ev_fork, st));
        }
        int n_aux = 0;
        for (int q = 0, launched = 0; q < 4; ++q) {
            const int M = order[q];
            if (!need[M]) continue;
            hipStream_t s_ = st;
            if (fork && launched > 0) {
                const int i = n_aux++;
                if (!ctx->aux[i]) HIPCHK(hipStreamCreateWithFlags(&ctx->aux[i], hipStreamNonBlocking));
                if (!ctx->ev_join[i]) HIPCHK(hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming));
                s_ = ctx->aux[i];
                HIPCHK(hipStreamWaitEvent(s_, ctx->ev_fork, 0));
            }
            ++launched;
            switch (M) {
            case PSD_MODE_GAMMA_UNIFORM: {
                const dim3 blk_u(CPOL_PSD_THREADS_U);
                // CPOL_PSD_LDS_PAD: experiment knob (extra dynamic LDS limits the workgroups per CU)
                static const size_t lds_pad = getenv("CPOL_PSD_LDS_PAD") ? atol(getenv("CPOL_PSD_LDS_PAD")) : 0;
                if (dop2) hipLaunchKernelGGL((k_psd_uniform<true>), grd_u, blk_u, lds_pad, s_, ctx->hs, pa);
                else hipLaunchKernelGGL((k_psd_uniform<false>), grd_u, blk_u, lds_pad, s_, ctx->hs, pa);
                break; }
            case PSD_MODE_GAMMA_EXP:
                if (dop2) hipLaunchKernelGGL((k_psd<PSD_MODE_GAMMA_EXP, true>), grd, blk, 0, s_, ctx->hs, pa);
                else hipLaunchKernelGGL((k_psd<PSD_MODE_GAMMA_EXP, false>), grd, blk, 0, s_, ctx->hs, pa);
                break;
            case PSD_MODE_ICE: {
                // slots with lambda tables: k_psd_ice2 takes the units inside the tabulated range
                // (all of them, in practice), k_psd<ICE> sums the others
                bool tab = false;
                for (int jj = 0; jj < n_hyd; ++jj) {
                    const cpol_hydro_desc &dd = ctx->hs.h[jj].d;
                    if (dd.psd_family == CPOL_PSD_ICE_FIELD && dd.uniform_grid && dd.tab_degree == CPOL_ICE_DEGREE) tab = true;
                }
                if (tab) {
                    if (dop2) hipLaunchKernelGGL((k_psd_ice2<true>), grd, blk, 0, s_, ctx->hs, pa);
                    else hipLaunchKernelGGL((k_psd_ice2<false>), grd, blk, 0, s_, ctx->hs, pa);
                }
                if (dop2) hipLaunchKernelGGL((k_psd<PSD_MODE_ICE, true>), grd, blk, 0, s_, ctx->hs, pa);
                else hipLaunchKernelGGL((k_psd<PSD_MODE_ICE, false>), grd, blk, 0, s_, ctx->hs, pa);
                break; }
            default: {
                // melting slots with fw tables go to the table-driven kernel, the others (none
                // in the product's own staging) to the direct one; each skips foreign units
                bool tab = false, direct = false;
                for (int jj = 0; jj < n_hyd; ++jj) {
                    const cpol_hydro_desc &dd = ctx->hs.h[jj].d;
                    if (dd.psd_family != CPOL_PSD_MELTING) continue;
                    (dd.tab_degree == CPOL_MELT_DEGREE ? tab : direct) = true;
                }
                if (tab) {
                    if (dop2) hipLaunchKernelGGL((k_psd_melting_tab<true>), grd, blk, 0, s_, ctx->hs, pa);
                    else hipLaunchKernelGGL((k_psd_melting_tab<false>), grd, blk, 0, s_, ctx->hs, pa);
                }
                if (direct) {
                    if (dop2) hipLaunchKernelGGL((k_psd_melting<true>), grd, blk, 0, s_, ctx->hs, pa);
                    else hipLaunchKernelGGL((k_psd_melting<false>), grd, blk, 0, s_, ctx->hs, pa);
                }
                break; }
            }
            if (s_ != st) HIPCHK(hipEventRecord(ctx->ev_join[n_aux - 1], s_));
        }
        for (int i = 0; i < n_aux; ++i) HIPCHK(hipStreamWaitEvent(st, ctx->ev_join[i], 0));
    }
    // ---- 5d. sub-beam sums per (gate, hydrometeor); the items on 1-D tables are evaluated here ----
    if (subsum) {
        SubsumArgs sa2{};
        sa2.key = (const int *)ctx->b_key.p;
        sa2.vmask = (const unsigned char *)ctx->b_vmask.p;
        sa2.rec = (const double2 *)ctx->b_rec.p;
        sa2.res = (const double *)ctx->b_res.p;
        sa2.vn = doppler ? (double *)ctx->b_vn.p : nullptr;
        sa2.sub_w = (const double *)ctx->v_subw;
        sa2.wgate = ml ? (const double *)ctx->b_wgate.p : nullptr;
        sa2.sz_integ = (float *)ctx->b_szinteg.p;
        sa2.n_rays = n_rays; sa2.n_gates = ng; sa2.n_sub = n_sub; sa2.n_hydro = n_hyd;
        // lanes of a wavefront = a tile of neighbouring rays x consecutive gates (16 x 4 from 16 rays on)
        int tl = CPOL_TILE_GATES_LOG2;
        while (tl < 6 && (CPOL_WAVE >> tl) > n_rays) ++tl;
        sa2.tile_log2 = tl;
        sa2.coop_rounds = ctx->subsum_coop_rounds;
        const long tiles = (long)cdiv(n_rays, CPOL_WAVE >> tl) * cdiv(ng, 1 << tl);
        // the scalar-cache form needs many wavefronts per SIMD to hide its waits (C4 volume, rays per sweep:
        // 45 / 90 / 180 / 360 -> PSD stage 1.08 / 1.56 / 2.08 / 3.57 ms against 0.85 / 1.48 / 2.35 / 4.71 ms with
        // the gather): from ~32 wavefronts per SIMD on (the scalar-cache form; see below for the LDS form).  CPOL_SUBSUM_COOP=0 / 1: never / always.
        // With lanes (cpol_fork) other sweeps share the GPU and hide the waits: measured with three lanes in
        // flight, the share of one of 8 / 4 GPUs (11 / 21 wavefronts per SIMD): 1.30 / 2.24 ms per volume share
        // against 1.31 / ~2.5 ms with the gather -- from ~12 there.
        const long waves_per_simd = tiles * n_hyd / 1024;
        const int lanes_alive = ctx->parent ? ctx->parent->n_children : ctx->n_children;
        // (round 4, with the validity bits read up front: the share of one of 8 GPUs -- 11 wavefronts per SIMD --
        // with three lanes in flight 1.19 ms per volume share this way against 1.27 with the gather: from 8 there)
        // (the LDS form, round 4: the share of one of 8 / 4 / 2 GPUs alone -- 11 / 21 / 43 wavefronts per SIMD -- PSD stage 905 / 1281 /
        // 1748 us against 846 / 1413 / 2312 with the gather: from 16 there)
        const bool coop = ctx->subsum_coop == 1 || (ctx->subsum_coop < 0 && waves_per_simd >= (lanes_alive >= 2 ? 8 : 16));
        // CPOL_SUBSUM_SMALL=1 (experiment, never the default): the gather form with three wavefronts per (tile,
        // hydrometeor), 4 columns each, and all rows of the block requested at once -- see the note on SPLIT in
        // cpol_final.inl: slower than the plain gather on the share (571 vs 533 us) and with lanes (1.41 vs 1.27 ms)
        const bool small = !coop && ctx->subsum_small == 1;
        // the cooperative form: coefficient rows through LDS (default since round 4) or through the scalar cache
        // (CPOL_SUBSUM_FORM=scalar, read when the context is created)
        // the team form (round 5: W wavefronts per (tile, species) share the sub-beams, the float32 sums stay ordered): what bounds a small
        // launch is the length of its longest wavefront's chain (cpol_final.inl).  Share of one of 8 / 4 / 2 GPUs alone (11 / 21 / 43
        // wavefronts per SIMD), ms per volume share: 1.47 / 2.18 / 3.57 with the rule above, 1.23 / 2.00 / 3.53 with W = 2 (W = 4: 1.23 /
        // 2.05 / 3.66; W = 4 with the sums handed on in LDS instead of a barrier per round -- the default -- 1.16 / 1.98 / 3.44); with three lanes in flight 0.97 / 1.71 either way (a context WITH lanes that runs one share at
        // a time: 1.42 with the LDS form its rule picked, 1.23 with the team); the whole volume 1.71 (LDS form) against 1.94 ms.
        const int team = ctx->subsum_team >= 0 ? ctx->subsum_team       // (a form forced through CPOL_SUBSUM_COOP stays what was asked for)
                       : (ctx->subsum_coop < 0 && waves_per_simd < 50 ? 4 : 0);      // (whatever the lanes: a context with lanes may still run one sweep at a time)
#define CPOL_TEAM_CASE(W) case W: if (ctx->subsum_chain) hipLaunchKernelGGL((k_subbeam_sum_team<W, true>), dim3((unsigned)tiles, n_hyd), dim3(CPOL_WAVE * W), 0, st, ctx->hs, ctx->its, sa2); \
                          else hipLaunchKernelGGL((k_subbeam_sum_team<W, false>), dim3((unsigned)tiles, n_hyd), dim3(CPOL_WAVE * W), 0, st, ctx->hs, ctx->its, sa2); break;
        if (team >= 2 && team <= 8)
            switch (team) { CPOL_TEAM_CASE(2) CPOL_TEAM_CASE(3) CPOL_TEAM_CASE(4) CPOL_TEAM_CASE(5) CPOL_TEAM_CASE(6) CPOL_TEAM_CASE(7)
                            case 8: hipLaunchKernelGGL((k_subbeam_sum_team<8, true>), dim3((unsigned)tiles, n_hyd), dim3(CPOL_WAVE * 8), 0, st, ctx->hs, ctx->its, sa2); break; }
#undef CPOL_TEAM_CASE
        else if (coop && ctx->subsum_scalar) hipLaunchKernelGGL(k_subbeam_sum_scalar, dim3((unsigned)tiles, n_hyd), dim3(CPOL_SUBSUM_THREADS), 0, st, ctx->hs, ctx->its, sa2);
        else if (coop) hipLaunchKernelGGL(k_subbeam_sum_lds, dim3((unsigned)tiles, n_hyd), dim3(CPOL_SUBSUM_THREADS), 0, st, ctx->hs, ctx->its, sa2);
        else if (small) hipLaunchKernelGGL((k_subbeam_sum_gather<3, 10>), dim3((unsigned)tiles, n_hyd * 3), dim3(CPOL_SUBSUM_THREADS), 0, st, ctx->hs, ctx->its, sa2);
        else hipLaunchKernelGGL((k_subbeam_sum_gather<1, 2>), dim3((unsigned)tiles, n_hyd), dim3(CPOL_SUBSUM_THREADS), 0, st, ctx->hs, ctx->its, sa2);
    }
    if (tm_psd) HIPCHK(hipEventRecord(ctx->ev[EV_PSD], st));

    // ---- 6. accumulation + polarimetric variables + scans (the arguments: see 2b above) ----
    if (doppler)
        for (int j = 0; j < n_hyd && !dop3; ++j)
            if (fa.vsrc[j] == 2)
                hipLaunchKernelGGL(k_ice_first, dim3(n_rays * n_sub), dim3(64), 0, st,
                                   (const unsigned char *)ctx->b_vmask.p, j,
                                   (const double *)ctx->b_vn.p + (long)j * n_sbg * 2,
                                   (IceFirst *)ctx->b_icefirst.p, ng);
    ScanRayArgs ra{};
    ra.PHIDP = (float *)T[O_PHIDP];
    ra.RVEL = nullptr;
    ra.sens_thr = cut ? (const double *)ctx->v_sens : nullptr;
    ra.radial_res = (float)p->radial_res;
    if (dop3) {
        // ---- 6b. Doppler spectrum (scheme 3): RVEL comes from the spectrum ----
        fa.RVEL = nullptr;
        SpecArgs sp{};
        sp.vals = (const float *)ctx->b_vals.p;
        sp.mask = (const signed char *)ctx->b_mask.p;
        sp.elev = (const float *)ctx->b_elev.p;
        sp.key = (const int *)ctx->b_key.p;
        sp.par = (const double *)ctx->b_par.p;
        sp.wgate = ml ? (const double *)ctx->b_wgate.p : nullptr;
        sp.geo = (const double *)ctx->v_geo;
        sp.sub_h = (const int *)ctx->v_subh;
        sp.varray = (const double *)ctx->v_varray;
        sp.beam = (float *)ctx->b_beam.p;
        sp.n_sbg = n_sbg; sp.n_gates = ng; sp.n_sub = n_sub; sp.n_h = n_h; sp.n_v = n_vb;
        sp.var_u = p->var_u; sp.var_v = p->var_v; sp.var_w = p->var_w; sp.var_rho = p->var_rho;
        sp.c_spec = (float)p->c_spectrum;
        // [2][n_d] + [threads] float64 (only with melting species), [n_hyd][n_d] + [n_hyd + 1][n_v] float32 (cpol_spectrum.inl)
        sp.n_melt_rows = spec_melt ? 2 : 0;
        const size_t lds = (spec_melt ? ((size_t)2 * ctx->hs.h[0].d.n_d + CPOL_SPEC_THREADS) * sizeof(double) : 0)
                           + ((size_t)n_hyd * (ctx->hs.h[0].d.n_d + n_vb) + n_vb) * sizeof(float);
        if (lds > 64 * 1024) {
            // beyond the default 64 KB per workgroup: ask for it (an attribute of the kernel; the largest request so far is kept)
            static size_t lds_allowed = 64 * 1024;
            if (lds > lds_allowed) {
                HIPCHK(hipFuncSetAttribute((const void *)k_spec_gate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                lds_allowed = lds;
            }
        }
        hipLaunchKernelGGL(k_spec_gate, dim3((unsigned)n_sbg), dim3(CPOL_SPEC_THREADS), lds, st, ctx->hs, ctx->ss, sp);
        if (p->with_attenuation) {
            SpecAttenArgs sa2{};
            sa2.key = (const int *)ctx->b_key.p;
            sa2.res = (const double *)ctx->b_res.p;
            sa2.beam = (float *)ctx->b_beam.p;
            sa2.n_sbg = n_sbg; sa2.n_gates = ng; sa2.n_v = n_vb; sa2.n_hydro = n_hyd;
            sa2.c_att = 4.343e-3 * 2 * p->wavelength;
            sa2.res_km = p->radial_res / 1000.;
            hipLaunchKernelGGL(k_spec_atten, dim3(n_rays * n_sub), dim3(64), (size_t)ng * sizeof(double), st, sa2);
        }
        SpecFinalArgs sf{};
        sf.beam = (const float *)ctx->b_beam.p;
        sf.sub_w = (const double *)ctx->v_subw;
        sf.wgate = ml ? (const double *)ctx->b_wgate.p : nullptr;
        sf.varray = (const double *)ctx->v_varray;
        sf.nyquist = t->nyquist ? (const double *)ctx->v_nyq : nullptr;
        sf.spectrum = (double *)T[O_SPEC];
        sf.RVEL = (double *)T[O_RVEL];
        sf.n_rays = n_rays; sf.n_gates = ng; sf.n_sub = n_sub; sf.n_v = n_vb;
        // cut_at_sensitivity censors the spectrum bin by bin (doppler_scatter.py:839-850)
        sf.sens_thr = cut ? (const double *)ctx->v_sens : nullptr;
        hipLaunchKernelGGL(k_spec_final, dim3((unsigned)n_rg), dim3(64), 0, st, sf);
        ra.RVEL = (double *)T[O_RVEL];       // censored with the other observables in k_final
    }
    fa.pre_integ = subsum ? 1 : 0;
    if (subsum) fa.sz_integ = (float *)ctx->b_szinteg.p;
    fa.proj = nullptr;
    if (fa.RVEL && n_sub >= 4) {
        // the per-sub-beam velocity terms by one thread per sub-beam gate (k_final adds them in order)
        hipLaunchKernelGGL(k_rvel_terms, dim3((unsigned)(n_rays * n_sub), cdiv(ng, 256)), dim3(256), 0, st,
                           fa, (double *)ctx->b_proj.p);
        fa.proj = (const double *)ctx->b_proj.p;
    }
    // (one workgroup per ray: with no more rays than CUs the kernel lasts as long as ONE workgroup -- 512
    // threads halve its gate loop; the share of one of 8 GPUs of a 5 x 360-ray volume is 225 rays)
    static const int final512 = getenv("CPOL_FINAL_512") ? atoi(getenv("CPOL_FINAL_512")) : -1;   // experiment knob
    if ((final512 == 1 || (final512 < 0 && n_rays <= 256)) && ng > CPOL_FINAL_THREADS)
        hipLaunchKernelGGL((k_final<2 * CPOL_FINAL_THREADS>), dim3(n_rays), dim3(2 * CPOL_FINAL_THREADS), (size_t)3 * ng * sizeof(float), st, fa, ra, ctx->its);
    else
        hipLaunchKernelGGL((k_final<CPOL_FINAL_THREADS>), dim3(n_rays), dim3(CPOL_FINAL_THREADS), (size_t)3 * ng * sizeof(float), st, fa, ra, ctx->its);
    if (tm) HIPCHK(hipEventRecord(ctx->ev[EV_FINAL], st));
    HIPCHK(hipGetLastError());

    return CPOL_OK;
    };

    const int forms[12] = {g1r, (int)gate1_ray, (int)gate1, (int)fused, (int)rare_direct, (int)subsum, (int)final_inplace, (int)poly_single,
                           n_sub, ctx->parent ? ctx->parent->n_children : ctx->n_children, CPOL_SCAN_FORM, 0};
    memcpy(ctx->last_forms, forms, sizeof forms);
    const double t_buffers = now_ns();
    ctx->counters_dirty = true;         // until the sequence is queued completely (cleared where sweep_serial advances)
    // graph key: every value that ends up in a kernel argument
    const bool graphable = ctx->use_graph && dev && !tm_psd && !ctx->keep_debug && mode != CPOL_GEOM_HOST_PATHS &&
                           !dop3 && reuse && !want_szt && !want_model;
    if (graphable) {
        uint64_t key = 1469598103934665603ull;
        auto mix = [&](const void *ptr, size_t n) {
            const unsigned char *c = (const unsigned char *)ptr;
            for (size_t i = 0; i < n; ++i) { key ^= c[i]; key *= 1099511628211ull; }
        };
        mix(p, sizeof *p);
        mix(&t->version, sizeof t->version);
        mix(T, sizeof T);
        mix(&ctx->stage_serial, sizeof ctx->stage_serial);
        void *arena[] = {ctx->b_traj.p, ctx->b_vals.p, ctx->b_mask.p, ctx->b_elev.p, ctx->b_qmelt.p,
                         ctx->b_fwmelt.p, ctx->b_key.p, ctx->b_pos.p, ctx->b_par.p, (void *)cnt_p,
                         ctx->b_offset.p, ctx->b_units.p, (void *)tot_p, ctx->b_perm.p,
                         ctx->b_res.p, ctx->b_vn.p, ctx->b_icefirst.p, ctx->b_wgate.p, ctx->b_blkranked.p, ctx->b_rec.p,
                         ctx->b_vmask.p, ctx->b_rayc.p, ctx->v_traj_in, ctx->v_geo, ctx->v_subh,
                         ctx->v_subv, ctx->v_subw, ctx->v_sens, ctx->v_site, ctx->v_nyq,
                         ctx->v_subsmooth, ctx->v_mlfilter, (void *)st, poly_single ? set->poly.p : nullptr, gate1_ray ? ctx->b_present.p : nullptr};
        mix(arena, sizeof arena);
        // (the launch forms chosen above from state outside *p: a graph captured before the lanes were forked must not keep
        // replaying the four-launch sequence once k_gate1_ray is the default, nor the reverse)
        mix(forms, sizeof forms);
        hipGraphExec_t &gexec = ctx->graph_exec[par];
        if (!gexec || ctx->graph_key[par] != key) {
            if (gexec) { (void)hipGraphExecDestroy(gexec); gexec = nullptr; }
            hipGraph_t graph = nullptr;
            HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            const int lrc = launch_all();
            const hipError_t ce = hipStreamEndCapture(st, &graph);
            if (lrc != CPOL_OK || ce != hipSuccess || !graph) {
                if (graph) (void)hipGraphDestroy(graph);
                if (lrc != CPOL_OK) return lrc;
                ctx->err = "cpol_run_sweep: stream capture failed";
                return CPOL_ERR_HIP;
            }
            const hipError_t ie = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ie != hipSuccess) { gexec = nullptr; ctx->err = "cpol_run_sweep: hipGraphInstantiate failed"; return CPOL_ERR_HIP; }
            ctx->graph_key[par] = key;
        }
        HIPCHK(hipGraphLaunch(gexec, st));
        ctx->last_forms[11] = 1;
    } else {
        if ((rc = launch_all()) != CPOL_OK) return rc;
    }

    const double t_launched = now_ns();
    if (ctx->fail_next) {
        ctx->fail_next = false;
        ctx->err = "cpol_run_sweep: failure requested by the test hook (fail_next_sweep)";
        return CPOL_ERR_HIP;
    }
    // ---- outputs that the kernels did not write in place ----
    if (window) {
        HIPCHK(hipMemcpyAsync(win_lo, (const char *)ctx->b_outwin.p + win_skew, (size_t)(win_hi - win_lo), hipMemcpyDeviceToHost, st));
    } else {
        for (int k = 0; k < O_N; ++k) {
            if (!produced[k] || !user_out[k] || T[k] == user_out[k]) continue;
            if ((rc = copy_out(ctx, user_out[k], T[k], obytes[k], dev))) return rc;
        }
    }

    {
        const double t_done = now_ns();
        ctx->host_ns[0] += 1.0;
        ctx->host_ns[1] += t_tables - t_enter;
        ctx->host_ns[2] += t_buffers - t_tables;
        ctx->host_ns[3] += t_launched - t_buffers;
        ctx->host_ns[4] += t_done - t_launched;
        ctx->host_ns[5] += t_done - t_enter;
    }
    ctx->last_n_sbg = n_sbg; ctx->last_n_rg = n_rg; ctx->last_n_rays = n_rays;
    ctx->last_n_gates = ng; ctx->last_n_sub = n_sub; ctx->last_n_v = n_v;
    ctx->last_par = par;
    ++ctx->sweep_serial;
    ctx->counters_dirty = false;
    ctx->last_n_keys = n_keys; ctx->last_subsum = subsum || final_inplace;
    ctx->counters.n_subbeam_gates = n_sbg;
    ctx->counters.n_gates = n_rg;

    if ((!dev && !async_host) || ctx->keep_debug) {
        // blocking host outputs: wait, then surface the domain error.  Device outputs and
        // non-blocking pinned-host outputs (outputs_on_device = 2) return at once: results
        // are valid, and a domain error is reported, after cpol_synchronize / cpol_counters.
        HIPCHK(hipStreamSynchronize(st));
        return report_domain_error(ctx);
    }
    return CPOL_OK;
}

int cpol_spaceborne_first_gate(cpol_ctx *ctx, const cpol_sweep_params *p, const double *traj,
                               const double *site, const int32_t *n_cand, double ceiling_m,
                               int32_t *first_gate)
{
    if (!ctx || !p || !traj || !site || !n_cand || !first_gate || p->n_rays < 1 || p->n_vnodes < 1) {
        if (ctx) ctx->err = "cpol_spaceborne_first_gate: bad arguments";
        return CPOL_ERR_ARG;
    }
    (void)hipGetLastError();            // a stale error of another user of the runtime in this thread is not ours
    HIPCHK(hipSetDevice(ctx->device));
    const int n = p->n_rays * p->n_vnodes;
    DevBuf d_tr, d_site, d_nc, d_out;
    int rc;
    if ((rc = upload(ctx, d_tr, traj, (size_t)n * 4 * sizeof(double)))) return rc;
    if ((rc = upload(ctx, d_site, site, (size_t)p->n_rays * 8 * sizeof(double)))) return rc;
    if ((rc = upload(ctx, d_nc, n_cand, (size_t)p->n_rays * sizeof(int)))) return rc;
    if ((rc = ensure(ctx, d_out, (size_t)n * sizeof(int)))) return rc;
    hipLaunchKernelGGL(k_spaceborne_first_gate, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream,
                       (const double *)d_tr.p, (const double *)d_site.p, (const int *)d_nc.p,
                       (int *)d_out.p, p->n_rays, p->n_vnodes, p->range0, p->range_step, ceiling_m);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(first_gate, d_out.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost,
                          ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    free_buf(d_tr); free_buf(d_site); free_buf(d_nc); free_buf(d_out);
    return CPOL_OK;
}

int cpol_counters(cpol_ctx *ctx, cpol_counters_t *out)
{
    if (!ctx || !out) return CPOL_ERR_ARG;
    if (ctx->last_n_sbg > 0) {
        // device-side totals of the LAST sweep (valid once the stream drained)
        HIPCHK(hipStreamSynchronize(ctx->stream));
        long long totals[2] = {0, 0};
        HIPCHK(hipMemcpy(totals, (const long long *)ctx->b_totals.p + ctx->last_par * 4, sizeof totals, hipMemcpyDeviceToHost));
        long long n_lookup = 0;
        {
            std::vector<int> slots(CPOL_COUNT_SLOTS);       // the kernels count into one of many words (count_table_items)
            HIPCHK(hipMemcpy(slots.data(), (const int *)ctx->b_count.p + ctx->last_par * ctx->count_stride + ctx->last_n_keys + 3, slots.size() * sizeof(int), hipMemcpyDeviceToHost));
            for (int v : slots) n_lookup += v;
        }
        ctx->counters.n_table_items = n_lookup;
        totals[0] += n_lookup;                              // valid items = integrated + looked up
        ctx->counters.n_valid_items = totals[0];
        ctx->counters.n_work_units = totals[1];
        if (ctx->ev_used > 0) {
            // average stage durations over the sweeps recorded since enable_timing
            double acc[EV_N] = {}, tot = 0;
            for (size_t i = 0; i < ctx->ev_used; ++i) {
                hipEvent_t *e = ctx->ev_sets[i];
                float ms = 0;
                if (ctx->timing == 2) {                     // PSD stage only
                    HIPCHK(hipEventElapsedTime(&ms, e[EV_BUCKET], e[EV_PSD]));
                    acc[EV_PSD] += ms;
                    continue;
                }
                for (int k = 1; k < EV_N; ++k) {
                    HIPCHK(hipEventElapsedTime(&ms, e[k - 1], e[k]));
                    acc[k] += ms;
                }
                HIPCHK(hipEventElapsedTime(&ms, e[EV_T0], e[EV_FINAL]));
                tot += ms;
            }
            const double inv = 1.0 / (double)ctx->ev_used;
            ctx->counters.ms_traj = (float)(acc[EV_TRAJ] * inv);
            ctx->counters.ms_interp = (float)(acc[EV_INTERP] * inv);
            ctx->counters.ms_classify = (float)(acc[EV_CLASSIFY] * inv);
            ctx->counters.ms_bucket = (float)(acc[EV_BUCKET] * inv);
            ctx->counters.ms_psd = (float)(acc[EV_PSD] * inv);
            ctx->counters.ms_final = (float)(acc[EV_FINAL] * inv);
            ctx->counters.ms_total = (float)(tot * inv);
        }
        *out = ctx->counters;
        return report_domain_error(ctx);
    }
    *out = ctx->counters;
    return CPOL_OK;
}

int cpol_debug_math(cpol_ctx *ctx, int op, const double *x, double *y, int n)
{
    if (!ctx || !x || !y || n < 1) return CPOL_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf dx, dy;
    int rc;
    if ((rc = upload(ctx, dx, x, (size_t)n * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, dy, (size_t)n * sizeof(double)))) { free_buf(dx); return rc; }
    hipLaunchKernelGGL(k_debug_math, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, op,
                       (const double *)dx.p, (double *)dy.p, n);
    hipError_t e = hipMemcpyAsync(y, dy.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    free_buf(dx);
    free_buf(dy);
    if (e != hipSuccess) { ctx->err = std::string("cpol_debug_math: ") + hipGetErrorString(e); return CPOL_ERR_HIP; }
    return CPOL_OK;
}

int64_t cpol_debug_read(cpol_ctx *ctx, const char *name, void *dst, int64_t max_bytes)
{
    if (!ctx || !name) return CPOL_ERR_ARG;
    if (!strcmp(name, "enable")) { ctx->keep_debug = true; return 0; }
    if (!strcmp(name, "disable")) { ctx->keep_debug = false; return 0; }
    if (!strcmp(name, "fail_next_sweep")) { ctx->fail_next = true; return 0; }
    if (!strcmp(name, "poly_central")) {
        if (!dst || max_bytes < (int64_t)sizeof(int)) return CPOL_ERR_ARG;
        memcpy(dst, &ctx->last_poly_central, sizeof(int));
        return (int64_t)sizeof(int);
    }
    if (!strcmp(name, "launch_forms")) {
        if (!dst || max_bytes < (int64_t)sizeof ctx->last_forms) return CPOL_ERR_ARG;
        memcpy(dst, ctx->last_forms, sizeof ctx->last_forms);
        return (int64_t)sizeof ctx->last_forms;
    }
    if (!strcmp(name, "host_times")) {
        // host time of cpol_run_sweep by section since the last read (see cpol_ctx::host_ns); reading resets
        if (!dst || max_bytes < (int64_t)sizeof ctx->host_ns) return CPOL_ERR_ARG;
        memcpy(dst, ctx->host_ns, sizeof ctx->host_ns);
        for (double &v : ctx->host_ns) v = 0.0;
        return (int64_t)sizeof ctx->host_ns;
    }
    if (!strcmp(name, "cache")) {
        // [integral-table cache entries, scattering-table cache entries, integral-table builds] (table_id)
        const cpol_ctx *own = ctx->parent ? ctx->parent : ctx;
        const double v[3] = {(double)own->itab_cache.size(), (double)own->table_cache.size(), (double)own->itab_builds};
        if (!dst || max_bytes < (int64_t)sizeof v) return CPOL_ERR_ARG;
        memcpy(dst, v, sizeof v);
        return (int64_t)sizeof v;
    }
    if (!strncmp(name, "itab_detail", 11) && name[11] >= '0' && name[11] < '0' + CPOL_MAX_HYDRO && !name[12]) {
        // slot j ("itab_detail<j>"): log2_lo, panels per octave, d0, n_pan, then the worst deviation at the
        // check points per function (CPOL_ITAB_NF values), per lambda panel (n_pan values: both check points;
        // n_pan values: the point near the panel edge alone), the accepted run of panels [lo, hi); 1-D tables only
        const cpol_ctx *own = ctx->parent ? ctx->parent : ctx;
        const std::vector<double> &v = own->itab_detail[name[11] - '0'];
        const int64_t nb = (int64_t)(v.size() * sizeof(double));
        if (!dst || max_bytes < nb) return nb ? -nb - 1000 : 0;      // (size query: -(bytes) - 1000)
        if (nb) memcpy(dst, v.data(), (size_t)nb);
        return nb;
    }
#ifdef CPOL_SUBSUM_STATS
    if (!strcmp(name, "subsum_stats")) {
        if (!dst || max_bytes < 32) return CPOL_ERR_ARG;
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_subsum_stats), 32));
        unsigned long long z[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_subsum_stats), z, 32));
        return 32;
    }
#endif
#ifdef CPOL_RARE_TRACE
    if (!strcmp(name, "rare_trace")) {
        if (!dst || max_bytes < 64 * 8 * 8) return CPOL_ERR_ARG;
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rare_trace), 64 * 8 * 8));
        return 64 * 8 * 8;
    }
#endif
#if defined(CPOL_SUBSUM_TRACE) || defined(CPOL_LOOKUP_TRACE) || defined(CPOL_INTERP_TRACE)
    if (!strcmp(name, "subsum_trace")) {
        const int64_t nb = (int64_t)sizeof(unsigned long long) * CPOL_SUBSUM_TRACE_W * CPOL_SUBSUM_TRACE_N;
        if (!dst || max_bytes < nb) return CPOL_ERR_ARG;
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_subsum_trace), (size_t)nb));
        return nb;
    }
#endif
    if (!strcmp(name, "itab_times")) {
        // per hydrometeor slot: device ms of the last integral-table build, and of its accuracy check
        const cpol_ctx *own = ctx->parent ? ctx->parent : ctx;
        if (!dst || max_bytes < (int64_t)sizeof own->itab_ms) return CPOL_ERR_ARG;
        memcpy(dst, own->itab_ms, sizeof own->itab_ms);
        return (int64_t)sizeof own->itab_ms;
    }
    if (!strcmp(name, "itab_check")) {
        // per hydrometeor slot: worst |polynomial - integrating kernel| / scale over the check points of its
        // integral table (1-D and 2-D); negative: table rejected (the slot is integrated bin by bin);
        // 0: no table.  Then (optional) where, and the number of (block, function) pairs above the limit
        const cpol_ctx *own = ctx->parent ? ctx->parent : ctx;
        if (!dst || max_bytes < (int64_t)sizeof own->itab_check) return CPOL_ERR_ARG;
        memcpy(dst, own->itab_check, sizeof own->itab_check);
        if (max_bytes >= 4 * (int64_t)sizeof own->itab_check) {
            memcpy((char *)dst + sizeof own->itab_check, own->itab_check_at, sizeof own->itab_check_at);
            memcpy((char *)dst + 2 * sizeof own->itab_check, own->itab_bad, sizeof own->itab_bad);
            memcpy((char *)dst + 3 * sizeof own->itab_check, own->itab_check_edge, sizeof own->itab_check_edge);
            return 4 * (int64_t)sizeof own->itab_check;
        }
        if (max_bytes >= 3 * (int64_t)sizeof own->itab_check) {
            memcpy((char *)dst + sizeof own->itab_check, own->itab_check_at, sizeof own->itab_check_at);
            memcpy((char *)dst + 2 * sizeof own->itab_check, own->itab_bad, sizeof own->itab_bad);
            return 3 * (int64_t)sizeof own->itab_check;
        }
        if (max_bytes >= 2 * (int64_t)sizeof own->itab_check) {
            memcpy((char *)dst + sizeof own->itab_check, own->itab_check_at, sizeof own->itab_check_at);
            return 2 * (int64_t)sizeof own->itab_check;
        }
        return (int64_t)sizeof own->itab_check;
    }
    if (!dst || ctx->last_n_sbg <= 0) return CPOL_ERR_ARG;
    const long n_sbg = ctx->last_n_sbg, n_rg = ctx->last_n_rg;
    const int n_hyd = ctx->hs.n_hydro, n_vars = ctx->model.n_vars;
    const void *src = nullptr;
    int64_t bytes = 0;
    if (!strcmp(name, "sub_values")) { src = ctx->b_vals.p; bytes = (int64_t)n_vars * n_sbg * 4; }
    else if (!strcmp(name, "sub_mask")) { src = ctx->b_mask.p; bytes = n_sbg; }
    else if (!strcmp(name, "sub_elev")) { src = ctx->b_elev.p; bytes = n_sbg * 4; }
    else if (!strcmp(name, "sub_coords")) { src = ctx->b_coords.p; bytes = n_sbg * 8; }
    else if (!strcmp(name, "sub_wgate")) { src = ctx->b_wgate.p; bytes = n_sbg * 8; }      // integration scheme 'ml': per-gate weights of every sub-beam
    else if (!strcmp(name, "item_key")) { src = ctx->b_key.p; bytes = (int64_t)n_hyd * n_sbg * 4; }
    else if (!strcmp(name, "item_res")) {
        if (ctx->last_subsum) {
            ctx->err = "cpol_debug_read: item_res is incomplete (the items on 1-D integral tables are evaluated inside "
                       "k_subbeam_sum / k_final and never stored; CPOL_SUBSUM=0 keeps them)";
            return CPOL_ERR_ARG;
        }
        src = ctx->b_res.p; bytes = (int64_t)n_hyd * n_sbg * CPOL_N_SZ * 8;
    }
    else if (!strcmp(name, "item_rec")) { src = ctx->b_rec.p; bytes = (int64_t)n_hyd * n_sbg * 16; }
    else if (!strcmp(name, "item_par")) { src = ctx->b_par.p; bytes = (int64_t)n_hyd * CPOL_MAX_PAR * n_sbg * 8; }
    else if (!strcmp(name, "q_melt")) { src = ctx->b_qmelt.p; bytes = 2 * n_sbg * 4; }
    else if (!strcmp(name, "fw_melt")) { src = ctx->b_fwmelt.p; bytes = 2 * n_sbg * 8; }
    else if (!strcmp(name, "sz_integ")) { src = ctx->b_szinteg.p; bytes = (int64_t)n_rg * n_hyd * CPOL_N_SZ * 4; }
    else if (!strcmp(name, "sz_total")) { src = ctx->b_sztotal.p; bytes = (int64_t)n_rg * CPOL_N_SZ * 4; }
    else if (!strcmp(name, "traj")) { src = ctx->b_traj.p; bytes = (int64_t)ctx->last_n_rays * ctx->last_n_v * 3 * ctx->last_n_gates * 4; }
    else if (!strcmp(name, "psd_clock")) { src = ctx->b_clk.p; bytes = 2048 * 4 * 8; }
    else if (!strcmp(name, "bucket_count")) { src = (const int *)ctx->b_count.p + ctx->last_par * ctx->count_stride; bytes = (int64_t)ctx->last_n_keys * 4; }
    else { ctx->err = std::string("cpol_debug_read: unknown buffer ") + name; return CPOL_ERR_ARG; }
    if (!src) { ctx->err = "cpol_debug_read: buffer not kept (call with name \"enable\" first)"; return CPOL_ERR_ARG; }
    if (bytes > max_bytes) { ctx->err = "cpol_debug_read: destination too small"; return CPOL_ERR_ARG; }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost) != hipSuccess) {
        ctx->err = "cpol_debug_read: copy failed";
        return CPOL_ERR_HIP;
    }
    return bytes;
}

}  // extern "C"
