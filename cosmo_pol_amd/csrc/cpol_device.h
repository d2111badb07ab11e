// cpol_device.h -- device-side data structures and small helpers shared by the
// HIP kernels of libcosmo_pol_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cosmo_pol_amd.h"

#define CPOL_WAVE 64

// WGS84 (Vincenty direct; mirrors oracle/cosmo_pol_oracle/geodesy.py)
#define CPOL_WGS84_F (1.0 / 298.257223563)
#define CPOL_VINCENTY_ITERS 5
#define CPOL_DEG (3.14159265358979323846 / 180.0)

// Model cube resident in HBM, re-laid-out at staging time so that ONE grid
// column is contiguous:  H[ny][nx][nz]  and  V[ny][nx][nz][n_vars]
// (the reference layout [nz][ny][nx] strides a column over planes of
//  ny*nx*4 B = 3.6 MB; see DESIGN.md "data layout in HBM").
struct ModelDev {
    const float *H;        // z-levels, column-major per cell, descending with k
    const float2 *HT;      // per cell (top = H[cell][0], lowest level = H[cell][nz - 1]): one 8-B load for both
    const float *V;        // variables interleaved per (cell, level)
    int n_vars, nz, ny, nx;
    float llc0, llc1;      // Lo1 (lon), La1 (lat)
    float urc0, urc1;
    float res0, res1;      // dlon, dlat
    double rres0, rres1;   // RN64(1 / res): the float32 quotients by the grid resolution as float64 products (div32_by, cpol_interp.inl)
    // rotated-pole rotation: products of sin/cos evaluated on the host
    double ctcp, ctsp, st, nsp, cp, nstcp, stsp, ct;
};

// Per-hydrometeor staged data
struct HydroDev {
    cpol_hydro_desc d;
    const double *table;   // [n_e][n_t][n_d][12]
    const double *pre;     // [n_d]
    const double *dnu;     // [n_d]
    const double *aux;     // family specific
    const double *rcsw;    // [n_e][n_t][n_d][2] Doppler scheme 2 weights or NULL
    int key_base;          // first bucket id of this hydrometeor
    int n_par;             // per-item parameter count
};

struct HydroSet {
    int n_hydro;
    int n_keys;
    HydroDev h[CPOL_MAX_HYDRO];
};

// Doppler scheme 3 (spectrum): float32 radar cross sections per table bin and the float32
// diameter grid of get_doppler_spectrum with its float32 powers.  Kept OUT of HydroDev:
// growing that struct (a by-value kernel argument of the PSD kernels) by 24 bytes per
// slot made the dominant PSD kernel 25 % slower in isolation.
struct SpecDev {
    const float *rcs32;    // [n_e][n_t][n_d] or NULL
    const float *dgrid;    // [3][n_d]: D, D^mu, D^nu (float32) or NULL
    float step32;          // D[1] - D[0] in float32
    int pad_spec;
};

struct SpecSet {
    SpecDev s[CPOL_MAX_HYDRO];
};

// Integral tables ("itab"), built on the device at staging time (cosmo_pol_hip.hip::build_itabs):
// for a species whose N(D) has ONE per-item shape parameter lambda (every gamma-family species,
// 1-moment ice) the 12 PSD-integrated scattering entries of an item are  scale_item x F_c(slice,
// lambda): F_c is evaluated ONCE per (LUT slice, 1/8-octave panel of lambda, Chebyshev node) by
// the PSD kernels themselves and stored as degree-10 polynomials in the position inside the panel;
// a sweep then gathers 15 x 11 coefficients per item instead of integrating 1024 diameter bins.
#define CPOL_ITAB_DEGREE 10      // 2-D blocks (melting species): total degree of the polynomial in (u, w)
#define CPOL_ITAB_NC     (CPOL_ITAB_DEGREE + 1)
// 1-D blocks: degree of the polynomial in the panel position and panels per octave of lambda.  What a sweep pays per item
// is (degree + 1) rows of 128 B gathered and `degree` Horner steps of 12 (14) chains; what the table costs is
// panels x (degree + 1) rows.  Measured pairs that pass the 1e-10 gate on every table of the bench, the goldens and
// the Mie sets: see DESIGN.md 3.4 / profiles/r5_itab1_degree.txt (make EXTRA="-DCPOL_ITAB1_DEGREE=.. -DCPOL_ITAB_PPO=..").
#ifndef CPOL_ITAB1_DEGREE
#define CPOL_ITAB1_DEGREE 10
#endif
#define CPOL_ITAB1_NC    (CPOL_ITAB1_DEGREE + 1)
#define CPOL_ITAB_NF     15      // 12 columns, 2 Doppler sums (v, n), ice: normalised N0 (Doppler spectrum)
#define CPOL_ITAB_NFP    16      // functions per coefficient row (padded: one row = 128 B)
#ifndef CPOL_ITAB_PPO
#define CPOL_ITAB_PPO    8       // panels per octave of lambda (1-D blocks)
#endif
// Every 1-D block is verified when it is built: two more items per (slice, panel), at the off-node
// positions CPOL_ITAB1_CHECK_U (mid-panel: T_11(0.37) = 0.86 of the nodal polynomial's maximum) and
// CPOL_ITAB1_CHECK_U2 (between the last two Chebyshev nodes, where the interpolation error of a function
// with a nearby singularity peaks -- and where the panel borders on the next one; round 4), are
// integrated by the same kernel and compared with the polynomial, function by function, on the scale of
// the function over the block (end of k_itab_fit).  A slot whose worst deviation reaches
// CPOL_ITAB_MAX_DEVIATION keeps its items on the integrating kernels.
#define CPOL_ITAB1_NODES (CPOL_ITAB1_NC + 2)
#define CPOL_ITAB1_CHECK_U 0.37
#define CPOL_ITAB1_CHECK_U2 0.96      // ~cos(pi / 11): the extremum of T_11 between the last two nodes
#define CPOL_ITAB_MAX_DEVIATION 1e-10
// Melting species: N(D) has TWO per-item parameters, the wet fraction fw (which also selects the
// LUT slice, floor bin of the table's second axis) and the slope lambda_r of the rain partner, and
// every integrated entry is  QM x F_c(slice, fw, lambda_r)  (the normalisation by the mass
// integral makes F_c a ratio of two sums over the bins, a very smooth function).  F_c is stored
// per (slice, 1/4-octave panel of lambda_r) as a polynomial of total degree 10 in (u = position of
// fw inside the slice's wet-fraction bin, w = position inside the panel): the 11 x 11 tensor
// Chebyshev interpolant with the terms T_a(u) T_b(w), a + b > 10, dropped (their coefficients are of
// the size of the 1-D tails), converted to monomials u^a w^b, a + b <= 10: 66 rows of 128 B, the
// rows of w^b (a = 0 .. 10 - b) at row CPOL_ITAB2_ROW(b).
#ifndef CPOL_ITAB2_PPO
#define CPOL_ITAB2_PPO   4
#endif
#define CPOL_ITAB2_NB    (CPOL_ITAB_NC * (CPOL_ITAB_NC + 1) / 2)       // coefficient rows per block
#define CPOL_ITAB2_ROW(b) ((b) * CPOL_ITAB_NC - (b) * ((b) - 1) / 2)
#define CPOL_ITAB2_NODES (CPOL_ITAB_NC * CPOL_ITAB_NC + 1)              // build items per block: the nodes + 1 check point
#define CPOL_ITAB2_CHECK_U 0.37
#define CPOL_ITAB2_CHECK_W (-0.61)
#define CPOL_ITAB2_MAX_DEVIATION CPOL_ITAB_MAX_DEVIATION   // accepted |polynomial - integrating kernel| / |value| at the check points (measured on
                                         // the full-size tables: 1.7e-12, in a column that nearly cancels at 88 deg elevation)
struct ItabDev {
    const double *tab;     // 1-D: [n_slices][n_pan][CPOL_ITAB1_NC][CPOL_ITAB_NFP] monomial coefficients (power-major:
                           //   one 128-B row holds the coefficient of u^q of all functions), or NULL
                           // 2-D: [n_slices][n_pan][CPOL_ITAB2_NB rows (w^b u^a, a + b <= 10)][CPOL_ITAB_NFP]
    const double *head;    // 2-D: [n_t][2] centre and 1 / half-width of the wet-fraction bins
    double log2_lo;        // lambda of panel 0, node u = -1:  2^log2_lo
    double d0;             // gamma family: the tabulated function is exp(+lambda d0) x integral (d0 = D_0^nu)
    int n_pan;             // panels per slice in the table (its stride)
    int pan_lo, pan_hi;    // panels [pan_lo, pan_hi) passed the accuracy gate: items with lambda outside are integrated
    int writes_vn;         // the direct kernels of this slot write vn (Doppler scheme 2, numeric integrate_V, ice)
    int ppo;               // panels per octave
    int two_d;             // melting species (2-D blocks)
    int par_slot;          // parameter slot of the item that holds lambda (0; melting: 2)
    int n_t;               // 2-D: slices per elevation (= wet-fraction bins)
};

struct ItabSet {
    ItabDev t[CPOL_MAX_HYDRO];
};

struct WorkUnit {          // one wave of the PSD kernel
    int key;               // bucket id (hydrometeor, e bin, t bin)
    int start;             // first position in perm[]
    int count;             // 1..64 items
    int pad;
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & (CPOL_WAVE - 1); }

__device__ __forceinline__ double shfl_f64(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src);
    hi = __shfl(hi, src);
    return __hiloint2double(hi, lo);
}

#ifndef CPOL_GATE1_PRESENT
#define CPOL_GATE1_PRESENT 0      // 1 (build knob, measured and rejected in round 6): k_interp_sweep leaves one word per (ray, 64-gate tile) saying which
                                  // hydrometeor slots have a positive mass density anywhere in the tile, and the wavefront of k_gate1_ray whose species is
                                  // absent from its tile (60 % of them on the C2 sweep, 2.4 us of life each) loads nothing, writes nothing, takes its
                                  // ticket and leaves.  Same bits (edges, headline, parity tests), and no gain: pipelined sweep 33.2 against 32.7 us,
                                  // isolated 79.6 against 77.8 (k_interp_sweep pays for the words) -- the wave slots those wavefronts held were not
                                  // what the kernel waits for (profiles/r6_variants.txt, item 13)
#endif
#if defined(CPOL_SUBSUM_TRACE) || defined(CPOL_LOOKUP_TRACE) || defined(CPOL_INTERP_TRACE)
// (CPOL_INTERP_TRACE, tools/interp_trace.py: the phases of every wavefront of k_interp_sweep / k_interp_classify)
// measurement builds (tools/subsum_trace.py): per wavefront of k_subbeam_sum* (index blockIdx.y * gridDim.x + blockIdx.x, times W
// + wave for the team form) or of k_psd_lookup (CPOL_LOOKUP_TRACE: global wavefront index), the first CPOL_SUBSUM_TRACE_N of them:
// start and end on the 100-MHz clock, iterations with work, HW_ID | XCC_ID << 32, then (team form) seven phase times in 10-ns
// units, two per word
#define CPOL_SUBSUM_TRACE_N 131072
#define CPOL_SUBSUM_TRACE_W 8
__device__ unsigned long long g_subsum_trace[CPOL_SUBSUM_TRACE_W * CPOL_SUBSUM_TRACE_N];
#endif
