// cpol_final.inl -- sub-beam / hydrometeor accumulation, polarimetric variables,
// range scans, radial mask, antenna-averaged model variables, sensitivity cut.
//
// Reference functions replaced (wolfidan/cosmo_pol):
//   accumulation over sub-beams (float32 store, quirks Q3-Q5)
//                                scatter/doppler_scatter.py:133-134, 259-268, 400-401
//   get_pol_from_sz              scatter/doppler_scatter.py:491-544
//   PHIDP / attenuated ZDR       scatter/doppler_scatter.py:406-415,
//                                utilities/utilities.py:175-199 (nan_cumsum/-prod)
//   radial mask                  scatter/doppler_scatter.py:472-477
//   integrate_radials            interpolation/interpolation.py:36-89
//   cut_at_sensitivity           scatter/doppler_scatter.py:804-862
// float32 statements keep NumPy's operand order (python scalars are cast to
// float32 first); libm-type float32 functions are evaluated in float64 and
// rounded once, which reproduces a correctly rounded float32 libm.

// IceParticle.integrate_V sums over ALL valid gates of a sub-beam and the result is
// zero-padded into the first valid gate (hydrometeors.py:1256-1275 +
// utilities.py:231-261): per (ray, sub-beam) totals and the gate that receives them
struct IceFirst {
    int first_gate;
    int pad;
    double v, n;
};

// one wavefront per (ray, sub-beam): ordered (deterministic) reduction over the gates
__global__ __launch_bounds__(64) void k_ice_first(const unsigned char *__restrict__ vmask, int j,
                                                   const double *__restrict__ vn_j,
                                                   IceFirst *__restrict__ out, int n_gates)
{
    const long rs = blockIdx.x;
    const int lane = threadIdx.x;
    double v = 0.0, nn = 0.0;
    int first = 0x7fffffff;
    for (int g = lane; g < n_gates; g += 64) {
        const long sbg = rs * n_gates + g;
        if ((vmask[sbg] >> j) & 1) {
            v += vn_j[sbg * 2];
            nn += vn_j[sbg * 2 + 1];
            first = min(first, g);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        v += shfl_f64(v, (lane + off) & 63) * ((lane < off) ? 1.0 : 0.0);
        nn += shfl_f64(nn, (lane + off) & 63) * ((lane < off) ? 1.0 : 0.0);
        first = min(first, __shfl(first, (lane + off) & 63));
    }
    if (lane == 0) {
        IceFirst r;
        r.first_gate = first; r.pad = 0; r.v = v; r.n = nn;
        out[rs] = r;
    }
}

// Integration scheme 'ml': per-gate weights of every sub-beam.  One wavefront per
// (ray, sub-beam): first / last melting-layer gate (melting.py:41, values BEFORE the
// melting scheme zeroes them), then w x 1 (high-weight nodes) or w x the two edge
// deltas smoothed by scipy.ndimage.gaussian_filter(sigma = 2, mode 'reflect')
// (interpolation.py:423-436).  Summation order of correlate1d's symmetric branch:
// centre tap, then pairs from the outermost tap inwards.
struct MlArgs {
    const float *vals;          // [n_vars][n_sbg], before k_classify
    const double *sub_w;
    const int *sub_smooth;
    const double *taps;         // [2r+1]
    double *wgate;              // [n_sbg]
    long n_sbg;
    int n_sub, n_gates, radius, with_melting;
    int var_qr, var_qs, var_qg;
};

__device__ __forceinline__ int reflect_index(int i, int n)
{
    // scipy 'reflect': (d c b a | a b c d | d c b a)
    while (i < 0 || i >= n) i = (i < 0) ? -i - 1 : 2 * n - 1 - i;
    return i;
}

__global__ __launch_bounds__(64) void k_ml_weights(MlArgs a)
{
    const long rs = blockIdx.x;
    const int s = (int)(rs % a.n_sub);
    const int lane = threadIdx.x;
    const long base = rs * a.n_gates;
    const double w = a.sub_w[s];
    if (!a.sub_smooth[s]) {
        for (int g = lane; g < a.n_gates; g += 64) a.wgate[base + g] = w * 1.0;
        return;
    }
    int first = 0x7fffffff, last = -1;
    if (a.with_melting) {
        for (int g = lane; g < a.n_gates; g += 64) {
            const float qr = a.vals[a.var_qr * a.n_sbg + base + g];
            const float qsg = a.vals[a.var_qs * a.n_sbg + base + g] + a.vals[a.var_qg * a.n_sbg + base + g];
            if (qr > 0.f && qsg > 0.f) { first = min(first, g); last = max(last, g); }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            first = min(first, __shfl_xor(first, off));
            last = max(last, __shfl_xor(last, off));
        }
    }
    const int r = a.radius;
    for (int g = lane; g < a.n_gates; g += 64) {
        double m = 0.0;
        if (last >= 0) {
            m = ((g == first || g == last) ? 1.0 : 0.0) * a.taps[r];
            for (int k = r; k >= 1; --k) {
                const int lo = reflect_index(g - k, a.n_gates), hi = reflect_index(g + k, a.n_gates);
                const double pair = ((lo == first || lo == last) ? 1.0 : 0.0)
                                  + ((hi == first || hi == last) ? 1.0 : 0.0);
                m += pair * a.taps[r - k];
            }
        }
        a.wgate[base + g] = w * m;
    }
}

struct FinalArgs {
    const double *res;          // [n_hydro][n_sbg][12]
    const int *key;             // [n_hydro][n_sbg]
    const unsigned char *vmask; // [n_sbg] bit j: hydrometeor j present
    const signed char *sub_mask;
    const float *vals;          // [n_vars][n_sbg]
    const double *sub_w;        // [n_sub]
    float *sz_integ;            // [n_rg][n_hydro][12] or NULL
    int pre_integ;              // sz_integ already holds the sub-beam sums (k_subbeam_sum)
    int eval_1d;                // fewer than 4 sub-beams: the items on 1-D integral tables are evaluated HERE from their
                                // record (key, panel position, scale) -- k_psd_lookup neither stores their 12 columns
                                // (96 B per item) nor does this kernel read them back; res[] holds only the melting
                                // items and the items integrated bin by bin
    const double2 *rec;         // [n_hydro][n_sbg] {panel position (-1: not on the table), scale} (k_classify)
    int key_base[CPOL_MAX_HYDRO];
    // single-beam fast path: k_gate1 has finished the gates and left the operands of the range scans; k_final
    // recomputes (final_gate) only the deferred gates -- an item outside the integral tables -- and, when a
    // species' fall-speed sums are totals over the ray (`ice_redo`), the gate that receives the total
    int pre_gate, ice_redo;
    const unsigned char *defer; // [n_rg]
    const float *sk, *sh, *sv;  // [n_rg]
    float *sz_total;            // [n_rg][12] or NULL
    float *ZH, *ZV, *ZDR, *KDP, *DELTA_HV, *RHOHV, *ATT_H, *ATT_V;   // work / outputs
    double *mask;               // [n_rg] or NULL
    signed char *mask8;         // [n_rg] or NULL: the sum of the sub-beams' mask codes (cpol_outputs.mask_sum8)
    double *model_vars;         // [n_vars][n_rg] or NULL
    int with_attenuation;
    float res_km;               // (float)(radial_res / 1000.)
    int n_rays, n_gates, n_sub, n_hydro, n_vars;
    float c_zh, c_kdp, c_2w;    // wavelength^4/(pi^5 K^2), 1e-3*(180/pi)*wavelength, 2*wavelength
    double sum_w;
    // Doppler scheme 1 (radial velocity)
    double *RVEL;               // [n_rg] or NULL
    const double *vn;           // [n_hydro][n_sbg][2] fall-speed moments of every item
    const IceFirst *ice_first;  // [n_rays*n_sub]
    const double *geo;          // [n_rays][n_h][8]: sin / cos of the sub-beam azimuth
    const int *sub_h;
    const float *elev;          // folded elevation per sub-beam gate (quirk Q8)
    int n_h, var_u, var_v, var_w;
    int vsrc[CPOL_MAX_HYDRO];   // 1: vn per gate, 2: ice (first valid gate)
    const double *proj;         // [n_sbg] subbeam_proj of every sub-beam gate (k_rvel_terms) or NULL: evaluated in place
    const double *nyquist;      // [n_rays] or NULL
    const double *wgate;        // [n_sbg] per-gate sub-beam weights (scheme 'ml') or NULL
};

// one output gate; returns the operands of the three range scans (2 KDP with NaN -> 0, and
// the two-way attenuation factors of the gate, NaN -> 1)
// Radial velocity seen by one sub-beam at one gate (doppler_scatter.py:276-281, 313-333;
// proj_vel :46-47): mean fall speed of the hydrometeors present, wind projected on the beam.
// ... the wind projected on the beam minus the mean fall speed v / nn of the species present
__device__ __forceinline__ double proj_from_moments(const FinalArgs &a, int ray, int s, long sbg, long n_sbg, double v, double nn)
{
    const double vh = v / nn;
    const double *gc = a.geo + ((long)ray * a.n_h + a.sub_h[s]) * 8;
    const float th = a.elev[sbg] * 0.017453292f;          // np.deg2rad on float32
    const double ct = (double)(float)cos((double)th), st = (double)(float)sin((double)th);
    const double U = (double)a.vals[(long)a.var_u * n_sbg + sbg];
    const double V = (double)a.vals[(long)a.var_v * n_sbg + sbg];
    const double W = (double)a.vals[(long)a.var_w * n_sbg + sbg];
    return (U * gc[0] + V * gc[1]) * ct + (W - vh) * st;
}

__device__ __forceinline__ double subbeam_proj(const FinalArgs &a, int ray, int s, int gate, long sbg, long n_sbg)
{
    double v = 0.0, nn = 0.0;
    const unsigned present = a.vmask[sbg];
    for (int j = 0; j < a.n_hydro; ++j) {
        if (!((present >> j) & 1u)) continue;
        double vj, nj;
        if (a.vsrc[j] == 1) {
            // (gamma species, Doppler scheme 1: the analytic moments, written by k_classify)
            const double2 w = *reinterpret_cast<const double2 *>(a.vn + ((long)j * n_sbg + sbg) * 2);
            vj = w.x; nj = w.y;
        } else {
            const IceFirst f = a.ice_first[(long)ray * a.n_sub + s];
            const bool here = (f.first_gate == gate);
            vj = here ? f.v : 0.0;
            nj = here ? f.n : 0.0;
        }
        if (vj == vj) v += vj;                  // nansum_arr
        if (nj == nj) nn += nj;
    }
    return proj_from_moments(a, ray, s, sbg, n_sbg, v, nn);
}

// With many sub-beams the terms above are evaluated by one thread per sub-beam gate first (a
// thread of k_final would walk n_sub x n_hydro dependent loads: 0.53 of its 1.13 ms on the C4
// sweep); k_final then adds them in the reference's order.
__global__ __launch_bounds__(256) void k_rvel_terms(FinalArgs a, double *__restrict__ out)
{
    const int rs = blockIdx.x;                            // ray * n_sub + sub
    const int gate = blockIdx.y * blockDim.x + threadIdx.x;
    if (gate >= a.n_gates) return;
    const int ray = rs / a.n_sub, s = rs % a.n_sub;
    const long n_sbg = (long)a.n_rays * a.n_sub * a.n_gates;
    const long sbg = (long)rs * a.n_gates + gate;
    out[sbg] = subbeam_proj(a, ray, s, gate, sbg, n_sbg);
}

// Sub-beam accumulation of one (output gate, hydrometeor) per thread (doppler_scatter.py:133-134,
// 259-268: nansum of [float32 acc, float64 term] stored back as float32, sub-beams in order).  The
// items on a 1-D integral table are EVALUATED here (their 12 columns never go through memory:
// k_psd_lookup wrote 96 B per item and k_final read them back -- 2.9 GB of the 9.8 GB a C4 sweep
// moved); items of the melting species and items integrated bin by bin are read from res[].
// Bound: the texture-address path of the CU (66 16-B gathers per evaluation, ~16 cycles each):
// 1.07 ms on the C4 sweep, 0.20 ms on the 45-ray share of one of 8 GPUs; requesting key and
// parameters of the next sub-beam ahead of the gather changes neither (1.03 / 0.196 ms); two
// sub-beams per step with both gathers in flight is slower (1.38 / 0.36 ms, 128 VGPRs).
struct SubsumArgs {
    const int *key;             // [n_hydro][n_sbg] (valid where vmask says so)
    const unsigned char *vmask; // [n_sbg] bit j: hydrometeor j present
    const double2 *rec;         // [n_hydro][n_sbg] {panel position (-1: not on the table), scale}
    const double *res;          // [n_hydro][n_sbg][12]
    double *vn;                 // [n_hydro][n_sbg][2] or NULL: the Doppler sums of the 1-D table items of a slot whose
                                // table carries them (1-moment ice, Doppler scheme 2, numeric integrate_V) are
                                // evaluated here too (functions 12, 13 of the block: one more 16-B load per row)
    const double *sub_w;        // [n_sub]
    const double *wgate;        // [n_sbg] or NULL (scheme 'ml')
    float *sz_integ;            // [n_rg][n_hydro][12]
    int n_rays, n_gates, n_sub, n_hydro;
    int tile_log2;              // lanes of a wavefront = 2^(6 - tile_log2) rays x 2^tile_log2 gates
    int coop_rounds;            // COOP: distinct blocks per wavefront and sub-beam through the scalar cache before the
                                // per-lane gather takes the remaining lanes (6; measured on the C4 volume: every lane
                                // through that one-row-at-a-time gather 3.2 ms, the normal case 2.04 ms)
};

#ifndef CPOL_SUBSUM_THREADS
#define CPOL_SUBSUM_THREADS 64       // (measured on the C4 sweep / its 45-ray share: 64 -> 0.97 / 0.190 ms, 128 -> 1.01 / 0.192, 256 -> 1.05 / 0.205)
#endif
#ifndef CPOL_SUBSUM_WPE
#define CPOL_SUBSUM_WPE 0            // experiment knob: minimum waves per SIMD asked of the register allocator (0: its own choice, 4)
#endif
#ifndef CPOL_SUBSUM_WPE_MAX
#define CPOL_SUBSUM_WPE_MAX 8
#endif
#if CPOL_SUBSUM_WPE
#define CPOL_SUBSUM_ATTR __attribute__((amdgpu_waves_per_eu(CPOL_SUBSUM_WPE, CPOL_SUBSUM_WPE_MAX)))
#else
#define CPOL_SUBSUM_ATTR
#endif
// Column split (template parameter SPLIT, blockIdx.y = j * SPLIT + part): two or three threads per (gate,
// hydrometeor), 6 or 4 columns each (the validity bits, key and record are read once per part).  Measured in
// round 3 on the C4 volume / the 225-ray share of one of 8 GPUs with the loop as it was then (one dependent
// validity load per sub-beam, rows requested two at a time): SPLIT 1: 3.58 / 0.55 ms; 2: 4.07 / 0.53; 3: 4.64 /
// 0.56 -- no gain, because what bounded the share was the NUMBER OF MEMORY ROUND TRIPS per wavefront and
// sub-beam (validity byte, then key + record, then six pairs of rows: 8), not registers or occupancy.
// Round 4: the validity bits of all sub-beams are read up front and key + record one sub-beam ahead (C4 volume,
// scalar-cache form: 2.06 -> 1.86 ms; the gather form on the share: 531 -> 533 us, unchanged), and SPLIT 3 with
// ALL rows of the block requested at once (GUNROLL = 10: 22 loads in flight per lane, one round trip per
// sub-beam) was tried for the small launches: 571 us -- so the round trips are not what bounds the gather form
// either; what does is the number of 16-byte gather instructions a CU's texture-address path takes (16 cycles
// each whatever the lanes in use: the tile's lanes are a third full on average).  With every item of a
// wavefront forced onto ONE block (wrong results, timing only) the scalar-cache form takes 1.37 ms: the
// per-sub-beam overhead of its partly filled wavefronts, not the number of distinct blocks, is most of it.
#ifndef CPOL_SUBSUM_LDS_UNROLL
#define CPOL_SUBSUM_LDS_UNROLL 2
#endif
#ifndef CPOL_SUBSUM_ROW_UNROLL
#define CPOL_SUBSUM_ROW_UNROLL 2     // coefficient rows per scalar-memory wait (3 rows = 72 SGPRs no longer fit: 3.57 -> 4.34 ms)
#endif
typedef double __attribute__((address_space(4))) sconst_f64;       // read-only data behind a wave-uniform address

// a * b + c with c in a scalar register pair (a wave-uniform coefficient): written out because the
// compiler forms v_fmac (addend = destination) and first copies every coefficient into vector registers
__device__ __forceinline__ double fma_sgpr(double a, double b, double c_uniform)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_uniform));
    return r;
}

// Thread -> (ray, gate): the 64 lanes of a wavefront are a TILE of 2^(6 - tile_log2) neighbouring rays x
// 2^tile_log2 consecutive gates (16 x 4 when the sweep has 16 rays or more), not 64 gates of one ray:
// neighbouring rays at one range see almost the same PSD slope, so the items of a tile sit on very few
// distinct (LUT slice, lambda panel) blocks -- measured on the C4 sweep at 3 degrees: 1.7 (snow), 1.8 (ice),
// 3.2 (graupel) distinct blocks per tile and sub-beam, against 7.5 / 9.2 / 6.2 along 64 gates of a ray.
// Forms, bit-identical (same coefficients, same Horner order), chosen per launch by the host:
//   COOP = false  every lane gathers the rows of its own block: 66 16-byte loads per evaluation through
//                 the vector L1, which bounds a large launch (~2 500 cycles per wavefront and sub-beam on a
//                 CU); the tile alone helps (lanes on one block share its lines): C4 volume 3.59 -> 3.2 ms.
//                 <false, 3, 10>: the small launches (the share of one of 8 GPUs: 11 wavefronts per SIMD):
//                 latency-bound, see the note on SPLIT above.
//   COOP = true   the wavefront walks over its distinct blocks: the block address is made wave-uniform
//                 (readlane), the coefficient rows arrive through the SCALAR cache into SGPRs and the lanes
//                 on that block run their Horner chains with a scalar operand -- no per-lane gather for them;
//                 after `coop_rounds` (6) blocks the lanes still without a value take a one-row-at-a-time
//                 gather, which bounds a tile of scattered items (91 VGPRs with that tail, 5 wavefronts per
//                 SIMD; 57 / 7 without it and the same 2.04 ms).  2.0 ms on the C4 volume; its wavefronts wait for the
//                 scalar cache row pair by row pair (s_load returns out of order: nothing to pipeline
//                 inside a wavefront; three waves per tile with 4 columns each and the whole block in
//                 one wait: 5.5-6.0 ms of PSD stage, the scalar cache does not keep up), so it needs many
//                 wavefronts per SIMD to pay.  The host picks it from ~32 wavefronts per SIMD on.
//                 (Workgroup ids reordered so that the 8 ids of an XCD in every 64 take 8 consecutive ray
//                 tiles of one range -- blocks shared through that XCD's L2: 2.05 -> 2.23 ms, dropped.)
#ifdef CPOL_SUBSUM_STATS
// measurement build (tools/variants.sh "-DCPOL_SUBSUM_STATS"): [0] wavefront iterations that did work, [1] lanes
// with an item in them, [2] scalar-cache rounds, [3] wavefront iterations skipped (species absent in the whole tile)
__device__ unsigned long long g_subsum_stats[4];
#endif
#ifdef CPOL_SUBSUM_TRACE
// (its own build flag: the counters above are atomics and stretch the launch sevenfold; the buffer: cpol_device.h)
#endif
enum { SUBSUM_GATHER = 0, SUBSUM_SCALAR = 1, SUBSUM_LDS = 2 };
// (Measured and dropped, round 5: the gather form with every lane walking ITS OWN present sub-beams -- the set bits of its
// validity word -- instead of the wavefront walking the union of its lanes' sub-beams: the 225-ray share 849 -> 885 us of
// PSD stage; the longest lane of a tile is as long as the union.  And the table rows themselves are not what bounds it
// either: degree-6 blocks on panels a quarter as wide, 7 rows gathered instead of 11, 842 -> 806 us, with k_final
// paying for it elsewhere: profiles/r5_itab1_degree.txt.)
template <int FORM, int SPLIT, int GUNROLL>
__device__ __forceinline__ void subbeam_sum_body(const HydroSet &hs, const ItabSet &its, const SubsumArgs &a)
{
    constexpr bool COOP = FORM != SUBSUM_GATHER;
    static_assert(CPOL_SUBSUM_THREADS == CPOL_WAVE, "one wavefront per workgroup: the tile walk uses wave-wide ballots");
    constexpr int NC = CPOL_ITAB1_NC, NFP = CPOL_ITAB_NFP, NB = NC * NFP;
    constexpr int NP = CPOL_N_SZ / 2 / SPLIT;                            // double2 column pairs of this thread
    static_assert(NP * SPLIT * 2 == CPOL_N_SZ, "SPLIT must divide 6");
    const long n_rg = (long)a.n_rays * a.n_gates;
    const int j = blockIdx.y / SPLIT, part = blockIdx.y % SPLIT;
    const int f0 = part * NP;                                            // first column pair
    const int tg = a.tile_log2, lane = threadIdx.x;
    const int gate_tiles = (a.n_gates + (1 << tg) - 1) >> tg;
    const int ray = (int)(blockIdx.x / gate_tiles) * (CPOL_WAVE >> tg) + (lane >> tg);
    const int gate = (int)(blockIdx.x % gate_tiles) * (1 << tg) + (lane & ((1 << tg) - 1));
    const bool in = ray < a.n_rays && gate < a.n_gates;
    const long rg = in ? (long)ray * a.n_gates + gate : 0;
    const long n_sbg = n_rg * a.n_sub;
    const long sbg0 = in ? (long)ray * a.n_sub * a.n_gates + gate : 0;
    const ItabDev &t = its.t[j];
    const bool tab1 = t.tab && !t.two_d;                                 // uniform
    const int key_base = hs.h[j].key_base;
    const double2 *R = a.rec + (long)j * n_sbg;
    const int *K = a.key + (long)j * n_sbg;
    const bool want_vn = tab1 && a.vn && t.writes_vn && part == 0;       // uniform
    double wtot = 0.0;
    if (a.wgate && in)
        for (int s = 0; s < a.n_sub; ++s) wtot += a.wgate[sbg0 + (long)s * a.n_gates];
    float acc[2 * NP];
#pragma unroll
    for (int c = 0; c < 2 * NP; ++c) acc[c] = __builtin_nanf("");
#ifdef CPOL_SUBSUM_TRACE
    const unsigned long long trace_t0 = wall_clock64();
    unsigned trace_work = 0;
#endif
    int pre_b = -1;                                     // SUBSUM_LDS: block whose pieces are already on their way into s_blk[cur]
    int cur = 0;
    // sub-beams in chunks of 64: the validity bits of a whole chunk are read FIRST (independent byte
    // loads, 16 in flight), so that neither the skip of an absent sub-beam nor the loads of a present one
    // wait for a validity byte inside the loop
    for (int s_lo = 0; s_lo < a.n_sub; s_lo += 64) {
        const int n_here = min(64, a.n_sub - s_lo);
        unsigned long long pm = 0;                                       // bit q: species j present at sub-beam s_lo + q (this lane)
#pragma unroll 16
        for (int q = 0; q < n_here; ++q) {
            const unsigned vb = in ? a.vmask[sbg0 + (long)(s_lo + q) * a.n_gates] : 0u;
            pm |= (unsigned long long)((vb >> j) & 1u) << q;
        }
        unsigned long long wp = 0;                                       // ... at any lane of the wavefront (uniform)
        for (int q = 0; q < n_here; ++q)
            if (__builtin_amdgcn_ballot_w64((pm >> q) & 1ull)) wp |= 1ull << q;
#ifdef CPOL_SUBSUM_STATS
        if (lane == 0 && part == 0) atomicAdd(&g_subsum_stats[3], (unsigned long long)(n_here - __popcll(wp)));
#endif
        if (!wp) continue;
        // the chunk's quadrature weights: lane q keeps sub_w[s_lo + q], an iteration reads its own by readlane (no load there)
        const double wq = (!a.wgate && lane < n_here) ? a.sub_w[s_lo + lane] : 0.0;
        // key + record of the NEXT present sub-beam are requested before the rows of the current one
        int q_next = __ffsll((long long)wp) - 1;
        int key_n = 0;
        double2 rc_n = make_double2(-1.0, 0.0);
        {
            const long sbg = sbg0 + (long)(s_lo + q_next) * a.n_gates;
            const bool pr = (pm >> q_next) & 1ull;
            key_n = pr ? K[sbg] : 0;
            rc_n = (pr && tab1) ? R[sbg] : make_double2(-1.0, 0.0);
        }
        while (wp) {
            const int q_cur = q_next;
            wp &= wp - 1;
            const int s = s_lo + q_cur;
            const long sbg = sbg0 + (long)s * a.n_gates;
            const bool present = (pm >> q_cur) & 1ull;
            const double w_sub = readlane_f64(wq, q_cur);                // (read here, where every lane is active)
            const int key = key_n;
            const double2 rc = rc_n;
            if (wp) {
                q_next = __ffsll((long long)wp) - 1;
                const long sbg_n = sbg0 + (long)(s_lo + q_next) * a.n_gates;
                const bool pr = (pm >> q_next) & 1ull;
                key_n = pr ? K[sbg_n] : 0;
                rc_n = (pr && tab1) ? R[sbg_n] : make_double2(-1.0, 0.0);
            }
            const bool on_tab = rc.x >= 0.0;
            const int pn = on_tab ? min((int)rc.x, t.n_pan - 1) : 0;
            const double u = 2.0 * (rc.x - (double)pn) - 1.0;
            const int blk_id = on_tab ? (key - key_base) * t.n_pan + pn : -1;
            double2 v[NP];
            double2 wv = make_double2(0.0, 0.0);
            // ---- the tile's distinct blocks, one after the other: coefficients through the scalar cache ----
            unsigned long long todo = __builtin_amdgcn_ballot_w64(on_tab);
#ifdef CPOL_SUBSUM_STATS
            if (lane == 0 && part == 0) { atomicAdd(&g_subsum_stats[0], 1ull); atomicAdd(&g_subsum_stats[1], (unsigned long long)__popcll(todo)); }
#endif
#ifdef CPOL_SUBSUM_TRACE
            ++trace_work;
#endif
            // ---- SUBSUM_LDS: the block's 88 16-byte pieces go straight from global memory into LDS (two
            // global_load_lds_dwordx4 of the wavefront: no vector registers in between), every coefficient pair is
            // then read back by ALL lanes from ONE address (ds_read_b128 broadcast) and used as a vector operand.  Two
            // buffers: the pieces of the NEXT block -- of this sub-beam, or the first one of the next sub-beam with
            // work -- are requested before the Horner chains / the accumulation of the current one start. ----
            if constexpr (FORM == SUBSUM_LDS) {
            constexpr int PIECES = NC * NFP / 2;          // 16-byte pieces of a block (88 at degree 10, 56 at degree 6)
            __shared__ double2 s_blk[2][PIECES > CPOL_WAVE ? PIECES : CPOL_WAVE];
            constexpr int REST = PIECES - CPOL_WAVE;
            // the 88 pieces of block `blk` straight from global memory into s_blk[bufi] (global_load_lds_dwordx4:
            // no vector registers in between); their arrival is awaited with s_waitcnt vmcnt(0) before the first read
            auto request = [&](int blk, int bufi) {
                const double2 *src = reinterpret_cast<const double2 *>(t.tab + (long)blk * NB);
                // (a block shorter than a wavefront: the lanes beyond it fetch its last piece again, into the buffer's
                // tail, which nobody reads -- every lane of the instruction writes its own 16-byte slot of LDS)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + (PIECES >= CPOL_WAVE ? lane : min(lane, PIECES - 1))),
                                                 (void __attribute__((address_space(3))) *)s_blk[bufi], 16, 0, 0);
                if (lane < REST)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + CPOL_WAVE + lane),
                                                     (void __attribute__((address_space(3))) *)(s_blk[bufi] + CPOL_WAVE), 16, 0, 0);
            };
            if (todo) {
                int b = __builtin_amdgcn_readlane(blk_id, __ffsll((long long)todo) - 1);
                if (b != pre_b) request(b, cur);
                pre_b = -1;
                for (int round = 0; todo && round < a.coop_rounds; ++round) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    const bool mine = blk_id == b;
                    todo &= ~__builtin_amdgcn_ballot_w64(mine);
                    const bool more = todo && round + 1 < a.coop_rounds;
                    if (more) {
                        b = __builtin_amdgcn_readlane(blk_id, __ffsll((long long)todo) - 1);
                        request(b, cur ^ 1);               // in flight during the Horner chains below
                    }
                    const double2 *sb = s_blk[cur];
                    if (mine) {
#pragma unroll
                        for (int f = 0; f < NP; ++f) v[f] = sb[(NC - 1) * (NFP / 2) + f0 + f];
#pragma unroll CPOL_SUBSUM_LDS_UNROLL
                        for (int q = NC - 2; q >= 0; --q) {
#pragma unroll
                            for (int f = 0; f < NP; ++f) {
                                const double2 cq = sb[q * (NFP / 2) + f0 + f];
                                v[f].x = fma(v[f].x, u, cq.x);
                                v[f].y = fma(v[f].y, u, cq.y);
                            }
                        }
                        if (want_vn) {
                            wv = sb[(NC - 1) * (NFP / 2) + CPOL_N_SZ / 2];
                            for (int q = NC - 2; q >= 0; --q) {
                                const double2 cq = sb[q * (NFP / 2) + CPOL_N_SZ / 2];
                                wv.x = fma(wv.x, u, cq.x);
                                wv.y = fma(wv.y, u, cq.y);
                            }
                        }
                    }
                    if (more) cur ^= 1;
                }
            }
            if (wp) {
                // the first block of the NEXT sub-beam with work (its key / record were requested at the top of this one)
                const bool on_n = rc_n.x >= 0.0;
                const unsigned long long mn = __builtin_amdgcn_ballot_w64(on_n);
                if (mn) {
                    const int pn_n = on_n ? min((int)rc_n.x, t.n_pan - 1) : 0;
                    const int blk_n = on_n ? (key_n - key_base) * t.n_pan + pn_n : -1;
                    pre_b = __builtin_amdgcn_readlane(blk_n, __ffsll((long long)mn) - 1);
                    cur ^= 1;                              // (the buffer the last Horner chains did not read)
                    request(pre_b, cur);
                }
            }
            }
            for (int round = 0; FORM == SUBSUM_SCALAR && todo && round < a.coop_rounds; ++round) {
#ifdef CPOL_SUBSUM_STATS
                if (lane == 0) atomicAdd(&g_subsum_stats[2], 1ull);
#endif
                const int leader = __ffsll((long long)todo) - 1;
                const int b = __builtin_amdgcn_readlane(blk_id, leader);
#ifdef CPOL_SUBSUM_FAKE_ONE_BLOCK
                const bool mine = blk_id >= 0;                           // TIMING EXPERIMENT ONLY (wrong results): every item on the leader's block
#else
                const bool mine = blk_id == b;                           // (blk_id = -1 never matches: b >= 0)
#endif
                // (constant address space: a wave-uniform address there is what the backend turns into s_load)
                const sconst_f64 *B = (const sconst_f64 *)(unsigned long long)(t.tab + (long)b * NB + 2 * f0);
                if (mine) {
#pragma unroll
                    for (int f = 0; f < NP; ++f) v[f] = make_double2(B[(NC - 1) * NFP + 2 * f], B[(NC - 1) * NFP + 2 * f + 1]);
                    // (one row per trip: unrolled, the scheduler requests all 11 rows at once and the 264 SGPRs they
                    // would need spill into vector registers)
#pragma unroll CPOL_SUBSUM_ROW_UNROLL
                    for (int q = NC - 2; q >= 0; --q) {
#pragma unroll
                        for (int f = 0; f < NP; ++f) {
                            v[f].x = fma_sgpr(v[f].x, u, B[q * NFP + 2 * f]);
                            v[f].y = fma_sgpr(v[f].y, u, B[q * NFP + 2 * f + 1]);
                        }
                    }
                    if (want_vn) {
                        const sconst_f64 *W = (const sconst_f64 *)(unsigned long long)(t.tab + (long)b * NB + CPOL_N_SZ);
                        wv = make_double2(W[(NC - 1) * NFP], W[(NC - 1) * NFP + 1]);
#pragma unroll CPOL_SUBSUM_ROW_UNROLL
                        for (int q = NC - 2; q >= 0; --q) {
                            wv.x = fma_sgpr(wv.x, u, W[q * NFP]);
                            wv.y = fma_sgpr(wv.y, u, W[q * NFP + 1]);
                        }
                    }
                }
                todo &= ~__builtin_amdgcn_ballot_w64(mine);
            }
            // ---- per-lane gather: COOP = false; COOP = true: the lanes still without a value after
            // `coop_rounds` blocks (a tile whose items are scattered over many blocks: noise-like
            // fields) -- one row at a time there, so that the rare tail costs the common path no registers; it
            // bounds a wavefront's sub-beam at 6 scalar rounds + one gather instead of up to 64 rounds ----
            if (on_tab && (!COOP || ((todo >> lane) & 1ull))) {
                const double2 *blk = reinterpret_cast<const double2 *>(t.tab + (long)blk_id * NB);
                // the Doppler sums of the block (functions 12, 13): with the whole block in flight (GUNROLL = NC - 1)
                // their rows are requested together with the columns, not one dependent load after the other
                double2 wq[(!COOP && GUNROLL >= NC - 1) ? NC : 1];
                if (!COOP && GUNROLL >= NC - 1 && want_vn) {
#pragma unroll
                    for (int q = 0; q < NC; ++q) wq[q] = blk[q * (NFP / 2) + CPOL_N_SZ / 2];
                }
                // this thread's columns of the block (itab1_columns on a part of the row)
#pragma unroll
                for (int f = 0; f < NP; ++f) v[f] = blk[(NC - 1) * (NFP / 2) + f0 + f];
#pragma unroll (COOP ? 1 : GUNROLL)
                for (int q = NC - 2; q >= 0; --q) {
#pragma unroll
                    for (int f = 0; f < NP; ++f) {
                        const double2 cq = blk[q * (NFP / 2) + f0 + f];
                        v[f].x = fma(v[f].x, u, cq.x);
                        v[f].y = fma(v[f].y, u, cq.y);
                    }
                }
                if (want_vn) {
                    if (!COOP && GUNROLL >= NC - 1) {
                        wv = wq[NC - 1];
#pragma unroll
                        for (int q = NC - 2; q >= 0; --q) {
                            wv.x = fma(wv.x, u, wq[q].x);
                            wv.y = fma(wv.y, u, wq[q].y);
                        }
                    } else {
                        wv = blk[(NC - 1) * (NFP / 2) + CPOL_N_SZ / 2];
#pragma unroll (COOP ? 1 : NC - 1)
                        for (int q = NC - 2; q >= 0; --q) {
                            const double2 cq = blk[q * (NFP / 2) + CPOL_N_SZ / 2];
                            wv.x = fma(wv.x, u, cq.x);
                            wv.y = fma(wv.y, u, cq.y);
                        }
                    }
                }
            }
            if (!present) continue;
            if (on_tab) {
#pragma unroll
                for (int f = 0; f < NP; ++f) { v[f].x *= rc.y; v[f].y *= rc.y; }
                if (want_vn)
                    *reinterpret_cast<double2 *>(a.vn + ((long)j * n_sbg + sbg) * 2) = make_double2(wv.x * rc.y, wv.y * rc.y);
            } else {
                const double2 *r = reinterpret_cast<const double2 *>(a.res + ((long)j * n_sbg + sbg) * CPOL_N_SZ) + f0;
#pragma unroll
                for (int c = 0; c < NP; ++c) v[c] = r[c];
            }
            const double w = a.wgate ? a.wgate[sbg] / wtot : w_sub;
#pragma unroll
            for (int c = 0; c < 2 * NP; ++c) {
                // nansum([float32 acc, float64 term]) stored back as float32
                double y = ((c & 1) ? v[c / 2].y : v[c / 2].x) * w;
                const float xa = (acc[c] == acc[c]) ? acc[c] : 0.f;      // (NaN -> 0 before the conversion: one select instead of two)
                const double x = (double)xa;
                if (!(y == y)) y = 0.0;
                acc[c] = (float)(x + y);
            }
        }
    }
#ifdef CPOL_SUBSUM_TRACE
    {
        const unsigned long w = (unsigned long)blockIdx.y * gridDim.x + blockIdx.x;
        if (lane == 0 && w < CPOL_SUBSUM_TRACE_N) {
            g_subsum_trace[CPOL_SUBSUM_TRACE_W * w] = trace_t0;
            g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + 1] = wall_clock64();
            g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + 2] = trace_work;
            g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                        (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32;
        }
    }
#endif
    if (!in) return;
    float2 *o = reinterpret_cast<float2 *>(a.sz_integ + (rg * a.n_hydro + j) * CPOL_N_SZ) + f0;
#pragma unroll
    for (int c2 = 0; c2 < NP; ++c2) o[c2] = make_float2(acc[2 * c2], acc[2 * c2 + 1]);
}

// the three forms as kernels: the LDS form asks the register allocator for 5 wavefronts per SIMD (96 VGPRs; its own choice
// of 98 gives 4: C4 volume 1.77 against 1.68 ms), the others keep the allocator's choice (scalar: 96 / 5, gather: 104 / 4)
__global__ __launch_bounds__(CPOL_SUBSUM_THREADS) __attribute__((amdgpu_waves_per_eu(5, 5)))
void k_subbeam_sum_lds(HydroSet hs, ItabSet its, SubsumArgs a) { subbeam_sum_body<SUBSUM_LDS, 1, 2>(hs, its, a); }
__global__ __launch_bounds__(CPOL_SUBSUM_THREADS) CPOL_SUBSUM_ATTR
void k_subbeam_sum_scalar(HydroSet hs, ItabSet its, SubsumArgs a) { subbeam_sum_body<SUBSUM_SCALAR, 1, 2>(hs, its, a); }
template <int SPLIT, int GUNROLL>
__global__ __launch_bounds__(CPOL_SUBSUM_THREADS) CPOL_SUBSUM_ATTR
void k_subbeam_sum_gather(HydroSet hs, ItabSet its, SubsumArgs a) { subbeam_sum_body<SUBSUM_GATHER, SPLIT, GUNROLL>(hs, its, a); }

// ---- the TEAM form (round 5): W wavefronts of one workgroup share a (tile, species) ----
// What bounds the forms above on a small launch is the LENGTH OF ONE WAVEFRONT'S CHAIN, not throughput
// (tools/subsum_trace.py on the 225-ray share of one of 8 GPUs: all 11 250 wavefronts have started after 68 us, the launch
// ends after 617 us with ONE wavefront that walked its 49 sub-beams at 12.6 us each; median with work 166 us).  The order of
// the float32 accumulation is what makes a chain of it -- but only the accumulation: the terms themselves (block fetch,
// Horner chains, scale, weight) do not depend on the running sum.  So the r-th present sub-beam goes to wavefront r mod W,
// the W terms of a round pass through LDS as float64 (12 columns x 64 lanes x 8 B per wavefront), and after a barrier
// wavefront k adds column pairs [k PPW, (k + 1) PPW) of the round's sub-beams IN ORDER to its float32 registers: same
// terms, same order, same bits, and a chain of ceil(49 / W) rounds.
// CHAIN (the default, W = 4): no barrier per round and no exchange of terms -- the tile's float32 sums wait in LDS (3 KB per
// workgroup), a counter says whose turn it is; the wavefront whose sub-beam comes next waits for the counter, adds its 12
// terms per lane to the sums in LDS and moves the counter on.  The wavefronts of a team then run apart (each is held up only
// by the sub-beam right before its own), and a wavefront's LDS shrinks from 9.4 to 4 KB: share 385 -> 352 us, 334 at 5
// wavefronts per SIMD (96 VGPRs).  Every turn belongs to a wavefront that takes it before it reaches the next barrier
// (the exchange of validity bits of the next chunk of 64 sub-beams, the end): nobody waits for a wavefront that waits.
#ifndef CPOL_TEAM_NBUF
#define CPOL_TEAM_NBUF 2          // block buffers per wavefront (1.4 KB each): the blocks of the next round requested ahead (3: 380 -> 402 us, fewer workgroups per CU)
#endif
static_assert(CPOL_TEAM_NBUF >= 2 && CPOL_TEAM_NBUF <= 3, "k_subbeam_sum_team keeps the ids of at most three blocks requested ahead");
#ifndef CPOL_TEAM_WPE
#define CPOL_TEAM_WPE 5              // wavefronts per SIMD asked of the register allocator (100 VGPRs by its own choice: 4; chain form, W = 4, share: 352 -> 334 us; 6: 391, spills)
#endif
#if CPOL_TEAM_WPE
#define CPOL_TEAM_ATTR __attribute__((amdgpu_waves_per_eu(CPOL_TEAM_WPE, CPOL_TEAM_WPE)))
#else
#define CPOL_TEAM_ATTR
#endif
template <int W, bool CHAIN>
__global__ __launch_bounds__(CPOL_WAVE * W) CPOL_TEAM_ATTR
void k_subbeam_sum_team(HydroSet hs, ItabSet its, SubsumArgs a)
{
    constexpr int NC = CPOL_ITAB1_NC, NFP = CPOL_ITAB_NFP, NB = NC * NFP;
    constexpr int NP = CPOL_N_SZ / 2;                                    // double2 column pairs of an item
    constexpr int PPW = (NP + W - 1) / W;                                // pairs a wavefront accumulates
    constexpr int PIECES = NC * NFP / 2;
    constexpr int REST = PIECES - CPOL_WAVE;
    __shared__ double2 s_y[CHAIN ? 1 : W][CHAIN ? 1 : NP][CPOL_WAVE];
    __shared__ float2 s_acc[CHAIN ? NP : 1][CPOL_WAVE];     // CHAIN: the running float32 sums of the tile, handed from sub-beam to sub-beam
    __shared__ int s_turn;                                   // CHAIN: sub-beams (in the order of the sums) added so far
    __shared__ unsigned long long s_pm[W][CPOL_WAVE];
    constexpr int NBUF = CPOL_TEAM_NBUF;
    __shared__ double2 s_blk[W][NBUF][PIECES > CPOL_WAVE ? PIECES : CPOL_WAVE];
    const long n_rg = (long)a.n_rays * a.n_gates;
    const int j = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tg = a.tile_log2, lane = threadIdx.x & (CPOL_WAVE - 1);
    const int gate_tiles = (a.n_gates + (1 << tg) - 1) >> tg;
    const int ray = (int)(blockIdx.x / gate_tiles) * (CPOL_WAVE >> tg) + (lane >> tg);
    const int gate = (int)(blockIdx.x % gate_tiles) * (1 << tg) + (lane & ((1 << tg) - 1));
    const bool in = ray < a.n_rays && gate < a.n_gates;
    const long rg = in ? (long)ray * a.n_gates + gate : 0;
    const long n_sbg = n_rg * a.n_sub;
    const long sbg0 = in ? (long)ray * a.n_sub * a.n_gates + gate : 0;
    const ItabDev &t = its.t[j];
    const bool tab1 = t.tab && !t.two_d;                                 // uniform
    const int key_base = hs.h[j].key_base;
    const double2 *R = a.rec + (long)j * n_sbg;
    const int *K = a.key + (long)j * n_sbg;
    const bool want_vn = tab1 && a.vn && t.writes_vn;                    // uniform
    double wtot = 0.0;
    if (a.wgate && in)
        for (int s = 0; s < a.n_sub; ++s) wtot += a.wgate[sbg0 + (long)s * a.n_gates];
    float acc[2 * PPW];
#pragma unroll
    for (int c = 0; c < 2 * PPW; ++c) acc[c] = __builtin_nanf("");
    if constexpr (CHAIN) {
        for (int f = wave; f < NP; f += W) s_acc[f][lane] = make_float2(__builtin_nanf(""), __builtin_nanf(""));
        if (threadIdx.x == 0) s_turn = 0;                                // (the first barrier below comes before the first sum)
    }
    int turn_base = 0;                                                   // CHAIN: sub-beams with work in the chunks before this one
    double2 (*my_blk)[PIECES > CPOL_WAVE ? PIECES : CPOL_WAVE] = s_blk[wave];
    auto request = [&](int blk, int bufi) {
        const double2 *src = reinterpret_cast<const double2 *>(t.tab + (long)blk * NB);
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + (PIECES >= CPOL_WAVE ? lane : min(lane, PIECES - 1))),
                                         (void __attribute__((address_space(3))) *)my_blk[bufi], 16, 0, 0);
        if (lane < REST)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + CPOL_WAVE + lane),
                                             (void __attribute__((address_space(3))) *)(my_blk[bufi] + CPOL_WAVE), 16, 0, 0);
    };
    // the w-th lowest set bit of m (-1: fewer than w + 1 bits), and m without its W lowest bits
    auto pick = [&](unsigned long long m, int w) {
        for (int i = 0; i < w && m; ++i) m &= m - 1;
        return m ? __ffsll((long long)m) - 1 : -1;
    };
    auto drop = [&](unsigned long long m) {
        for (int i = 0; i < W && m; ++i) m &= m - 1;
        return m;
    };
    int pb0 = -1, pb1 = -1, pb2 = -1, npre = 0;          // blocks of the coming round already on their way into buffers 0 .. npre - 1 (uniform)
#ifdef CPOL_SUBSUM_TRACE
    const unsigned long long trace_t0 = wall_clock64();
    unsigned trace_work = 0;
    unsigned trace_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long trace_last = trace_t0;
#define TP(i) { const unsigned long long now_ = wall_clock64(); trace_ph[i] += (unsigned)(now_ - trace_last); trace_last = now_; }
#else
#define TP(i)
#endif
    for (int s_lo = 0; s_lo < a.n_sub; s_lo += 64) {
        const int n_here = min(64, a.n_sub - s_lo);
        // bit q: species j present at sub-beam s_lo + q (this lane) -- wavefront k reads the validity bytes of sub-beams
        // k, k + W, ... and the W parts meet in LDS (a tile without the species costs the workgroup what it costs one wavefront)
        unsigned long long pm = 0;
#pragma unroll 4
        for (int q = wave; q < n_here; q += W) {
            const unsigned vb = in ? a.vmask[sbg0 + (long)(s_lo + q) * a.n_gates] : 0u;
            pm |= (unsigned long long)((vb >> j) & 1u) << q;
        }
        if (s_lo) __syncthreads();                                       // (the parts of the chunk before are no longer read)
        s_pm[wave][lane] = pm;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < W; ++k) pm |= s_pm[k][lane];
        unsigned long long wp = 0;                                       // ... at any lane of the tile (the same in all W wavefronts)
        for (int q = 0; q < n_here; ++q)
            if (__builtin_amdgcn_ballot_w64((pm >> q) & 1ull)) wp |= 1ull << q;
        if (!wp) continue;
        const unsigned long long wp_all = wp;
        // key + record of this wavefront's sub-beams TWO rounds ahead, the first block ONE round ahead: what a round waits
        // for (s_waitcnt vmcnt(0) below: every load in flight) was requested a round ago or more
        auto load_kr = [&](int q, int &key_o, double2 &rc_o) {
            key_o = 0;
            rc_o = make_double2(-1.0, 0.0);
            if (q >= 0) {
                const long sbg = sbg0 + (long)(s_lo + q) * a.n_gates;
                const bool pr = (pm >> q) & 1ull;
                key_o = pr ? K[sbg] : 0;
                rc_o = (pr && tab1) ? R[sbg] : make_double2(-1.0, 0.0);
            }
        };
        int q1 = pick(wp, wave), q2 = pick(drop(wp), wave);
        int key1, key2;
        double2 rc1, rc2;
        load_kr(q1, key1, rc1);
        load_kr(q2, key2, rc2);
        // the chunk's quadrature weights: lane q keeps sub_w[s_lo + q], a round reads its own by readlane (no load in the round)
        const double wq = (!a.wgate && lane < n_here) ? a.sub_w[s_lo + lane] : 0.0;
        npre = 0;
        while (wp) {
            const unsigned long long wp_after = drop(wp);
            const unsigned long long round_bits = wp ^ wp_after;
            const int q_cur = q1;
            const int key = key1;
            const double2 rc = rc1;
            q1 = q2; key1 = key2; rc1 = rc2;                              // the round after this one
            q2 = pick(drop(wp_after), wave);                             // ... and the one after that (loads: below)
            TP(0)
            if (q_cur >= 0) {                                            // uniform in the wavefront
#ifdef CPOL_SUBSUM_TRACE
                ++trace_work;
#endif
                const int s = s_lo + q_cur;
                const long sbg = sbg0 + (long)s * a.n_gates;
                const bool present = (pm >> q_cur) & 1ull;
                const double w_sub = readlane_f64(wq, q_cur);            // (read here, where every lane is active)
                const bool on_tab = rc.x >= 0.0;
                const int pn = on_tab ? min((int)rc.x, t.n_pan - 1) : 0;
                const double u = 2.0 * (rc.x - (double)pn) - 1.0;
                const int blk_id = on_tab ? (key - key_base) * t.n_pan + pn : -1;
                double2 v[NP];
                double2 wv = make_double2(0.0, 0.0);
                unsigned long long todo = __builtin_amdgcn_ballot_w64(on_tab);
                const bool todo0 = todo != 0;
                if (todo) {
                    // the tile's distinct blocks in the order of their first lanes: block k sits in buffer k mod NBUF -- the first
                    // `npre` were requested during the round before (below), the others are requested one step ahead
                    int b = __builtin_amdgcn_readlane(blk_id, __ffsll((long long)todo) - 1);
                    if (!(npre > 0 && pb0 == b)) { npre = 0; request(b, 0); }
                    for (int k = 0; ; ++k) {
                        TP(2)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                        TP(1)
                        // key + record two rounds ahead: requested after the round's first wait, so that they travel during the
                        // chains and nothing waits for them (the next wait of this kind is a round away)
                        if (k == 0) load_kr(q2, key2, rc2);
                        const bool mine = blk_id == b;
                        todo &= ~__builtin_amdgcn_ballot_w64(mine);
                        const bool more = todo && k + 1 < a.coop_rounds;
                        int b_next = 0;
                        if (more) {
                            b_next = __builtin_amdgcn_readlane(blk_id, __ffsll((long long)todo) - 1);
                            const int pbn = k == 0 ? pb1 : pb2;
                            if (!(k + 1 < npre && k + 1 < NBUF && pbn == b_next)) request(b_next, (k + 1) % NBUF);
                        }
                        const double2 *sb = my_blk[k % NBUF];
                        if (mine) {
#pragma unroll
                            for (int f = 0; f < NP; ++f) v[f] = sb[(NC - 1) * (NFP / 2) + f];
#pragma unroll CPOL_SUBSUM_LDS_UNROLL
                            for (int q = NC - 2; q >= 0; --q) {
#pragma unroll
                                for (int f = 0; f < NP; ++f) {
                                    const double2 cq = sb[q * (NFP / 2) + f];
                                    v[f].x = fma(v[f].x, u, cq.x);
                                    v[f].y = fma(v[f].y, u, cq.y);
                                }
                            }
                            if (want_vn) {
                                wv = sb[(NC - 1) * (NFP / 2) + CPOL_N_SZ / 2];
                                for (int q = NC - 2; q >= 0; --q) {
                                    const double2 cq = sb[q * (NFP / 2) + CPOL_N_SZ / 2];
                                    wv.x = fma(wv.x, u, cq.x);
                                    wv.y = fma(wv.y, u, cq.y);
                                }
                            }
                        }
                        if (!more) break;
                        b = b_next;
                    }
                }
                TP(2)
                npre = 0;
                if (q1 >= 0) {
                    // the first NBUF distinct blocks of this wavefront's sub-beam of the next round: on their way during the
                    // barrier and the sums (every buffer is free: this round's chains are done)
                    const bool on_n = rc1.x >= 0.0;
                    unsigned long long mn = __builtin_amdgcn_ballot_w64(on_n);
                    const int pn_n = on_n ? min((int)rc1.x, t.n_pan - 1) : 0;
                    const int blk_n = on_n ? (key1 - key_base) * t.n_pan + pn_n : -1;
#pragma unroll
                    for (int k = 0; k < NBUF; ++k) {
                        if (!mn) break;
                        const int bn = __builtin_amdgcn_readlane(blk_n, __ffsll((long long)mn) - 1);
                        mn &= ~__builtin_amdgcn_ballot_w64(blk_n == bn);
                        if (k == 0) pb0 = bn; else if (k == 1) pb1 = bn; else pb2 = bn;
                        request(bn, k);
                        npre = k + 1;
                    }
                }
                if (!todo0) load_kr(q2, key2, rc2);                        // (a round without an item on the table)
                // the lanes still without a value after `coop_rounds` blocks: one row at a time (see subbeam_sum_body)
                if (on_tab && ((todo >> lane) & 1ull)) {
                    const double2 *blk = reinterpret_cast<const double2 *>(t.tab + (long)blk_id * NB);
#pragma unroll
                    for (int f = 0; f < NP; ++f) v[f] = blk[(NC - 1) * (NFP / 2) + f];
#pragma unroll 1
                    for (int q = NC - 2; q >= 0; --q) {
#pragma unroll
                        for (int f = 0; f < NP; ++f) {
                            const double2 cq = blk[q * (NFP / 2) + f];
                            v[f].x = fma(v[f].x, u, cq.x);
                            v[f].y = fma(v[f].y, u, cq.y);
                        }
                    }
                    if (want_vn) {
                        wv = blk[(NC - 1) * (NFP / 2) + CPOL_N_SZ / 2];
#pragma unroll 1
                        for (int q = NC - 2; q >= 0; --q) {
                            const double2 cq = blk[q * (NFP / 2) + CPOL_N_SZ / 2];
                            wv.x = fma(wv.x, u, cq.x);
                            wv.y = fma(wv.y, u, cq.y);
                        }
                    }
                }
                if (present) {
                    if (on_tab) {
#pragma unroll
                        for (int f = 0; f < NP; ++f) { v[f].x *= rc.y; v[f].y *= rc.y; }
                        if (want_vn)
                            *reinterpret_cast<double2 *>(a.vn + ((long)j * n_sbg + sbg) * 2) = make_double2(wv.x * rc.y, wv.y * rc.y);
                    } else {
                        const double2 *r = reinterpret_cast<const double2 *>(a.res + ((long)j * n_sbg + sbg) * CPOL_N_SZ);
#pragma unroll
                        for (int c = 0; c < NP; ++c) v[c] = r[c];
                    }
                    const double w = a.wgate ? a.wgate[sbg] / wtot : w_sub;
#pragma unroll
                    for (int f = 0; f < NP; ++f) {
                        double yx = v[f].x * w, yy = v[f].y * w;
                        if (!(yx == yx)) yx = 0.0;
                        if (!(yy == yy)) yy = 0.0;
                        if constexpr (CHAIN) v[f] = make_double2(yx, yy);
                        else s_y[wave][f][lane] = make_double2(yx, yy);
                    }
                }
                if constexpr (CHAIN) {
                    // this sub-beam's turn in the order of the sums: the tile's float32 sums wait in LDS; the wavefront whose
                    // sub-beam comes next adds its terms and passes them on (nansum([float32 sum, float64 term]) as float32)
                    const int turn = turn_base + __popcll(wp_all & ((1ull << q_cur) - 1ull));
                    TP(3)
                    while (__hip_atomic_load(&s_turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != turn) __builtin_amdgcn_s_sleep(1);
                    TP(4)
                    if (present) {
#pragma unroll
                        for (int f = 0; f < NP; ++f) {
                            float2 sa = s_acc[f][lane];
                            const float xa = (sa.x == sa.x) ? sa.x : 0.f, xb = (sa.y == sa.y) ? sa.y : 0.f;
                            sa.x = (float)((double)xa + v[f].x);
                            sa.y = (float)((double)xb + v[f].y);
                            s_acc[f][lane] = sa;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) __hip_atomic_store(&s_turn, turn + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    TP(5)
                }
            }
            if constexpr (!CHAIN) {
            TP(3)
            __syncthreads();
            TP(4)
            // nansum([float32 acc, float64 term]) stored back as float32, sub-beam after sub-beam
            if (wave * PPW < NP) {
                unsigned long long rb = round_bits;
                for (int i = 0; rb; ++i, rb &= rb - 1) {
                    const int q = __ffsll((long long)rb) - 1;
                    if (!((pm >> q) & 1ull)) continue;
#pragma unroll
                    for (int f = 0; f < PPW; ++f) {
                        if (wave * PPW + f >= NP) break;
                        const double2 y = s_y[i][wave * PPW + f][lane];
                        const float xa = (acc[2 * f] == acc[2 * f]) ? acc[2 * f] : 0.f;
                        const float xb = (acc[2 * f + 1] == acc[2 * f + 1]) ? acc[2 * f + 1] : 0.f;
                        acc[2 * f] = (float)((double)xa + y.x);
                        acc[2 * f + 1] = (float)((double)xb + y.y);
                    }
                }
            }
            TP(5)
            __syncthreads();
            TP(6)
            }
            wp = wp_after;
        }
        turn_base += __popcll(wp_all);
    }
#ifdef CPOL_SUBSUM_TRACE
    {
        const unsigned long w = ((unsigned long)blockIdx.y * gridDim.x + blockIdx.x) * W + wave;
        if (lane == 0 && w < CPOL_SUBSUM_TRACE_N) {
            for (int i = 0; i < 4; ++i) g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + 4 + i] = trace_ph[2 * i] | (unsigned long long)trace_ph[2 * i + 1] << 32;
            g_subsum_trace[CPOL_SUBSUM_TRACE_W * w] = trace_t0;
            g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + 1] = wall_clock64();
            g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + 2] = trace_work | (unsigned long long)trace_ph[4] << 32;
            g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                        (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32;
        }
    }
#endif
#undef TP
    if constexpr (CHAIN) {
        __syncthreads();                                                 // (every sub-beam has had its turn)
        if (!in) return;
        float2 *o = reinterpret_cast<float2 *>(a.sz_integ + (rg * a.n_hydro + j) * CPOL_N_SZ);
        for (int f = wave; f < NP; f += W) o[f] = s_acc[f][lane];
    } else {
    if (!in || wave * PPW >= NP) return;
    float2 *o = reinterpret_cast<float2 *>(a.sz_integ + (rg * a.n_hydro + j) * CPOL_N_SZ) + wave * PPW;
#pragma unroll
    for (int f = 0; f < PPW; ++f)
        if (wave * PPW + f < NP) o[f] = make_float2(acc[2 * f], acc[2 * f + 1]);
    }
}


#ifndef CPOL_SKIP_RVEL
#define CPOL_SKIP_RVEL 0      // experiment knob (tools/variants.sh): time of the RVEL loop
#endif
#ifndef CPOL_EXP_FINISH_CHEAP
#define CPOL_EXP_FINISH_CHEAP 0   // experiment knob (timing only, wrong values): the two exp10 and the atan2 of gate_finish replaced by one operation each
#endif
#ifndef CPOL_FINAL_BATCH
#define CPOL_FINAL_BATCH 2
#endif
#ifndef CPOL_FINAL_ROW_UNROLL
#define CPOL_FINAL_ROW_UNROLL 5      // coefficient rows of an in-place table evaluation requested together (k_final)
#endif
#ifndef CPOL_FINAL_SBATCH
#define CPOL_FINAL_SBATCH 7          // sub-beams whose RVEL terms / masks are requested together (experiment knob)
#endif
// Everything of an output gate behind the hydrometeor sums tot[12]: exact zeros -> NaN, get_pol_from_sz, the
// operands of the range scans, RVEL, the radial mask, the antenna-averaged model variables.  `have_moments`: the
// fall-speed sums (mom_v, mom_n) of the ONE sub-beam of a single-beam sweep come from the caller's registers
// (k_gate1); else they are read from vn[] / ice_first[] (subbeam_proj).
__device__ __forceinline__ void gate_finish(const FinalArgs &a, int ray, int gate, float (&tot)[CPOL_N_SZ], bool have_moments,
                                            double mom_v, double mom_n, double wtot, float &k2_out, float &fh_out, float &fv_out)
{
    const long n_rg = (long)a.n_rays * a.n_gates;
    const long rg = (long)ray * a.n_gates + gate;
    const long n_sbg = n_rg * a.n_sub;
    const long sbg0 = (long)ray * a.n_sub * a.n_gates + gate;      // + sub * n_gates
    const float qnan = __builtin_nanf("");
#pragma unroll
    for (int c = 0; c < CPOL_N_SZ; ++c) if (tot[c] == 0.0f) tot[c] = qnan;     // Q5
    if (a.sz_total) {
#pragma unroll
        for (int c = 0; c < CPOL_N_SZ; ++c) a.sz_total[rg * CPOL_N_SZ + c] = tot[c];
    }

    // ---- get_pol_from_sz (float32) ----
    const float two_pi = (float)(2 * 3.14159265358979323846);
    const float b = tot[0] - tot[1] - tot[2] + tot[3];
    const float cc = tot[0] + tot[1] + tot[2] + tot[3];
    const float xs_h = two_pi * b;
    const float xs_v = two_pi * cc;
    a.ZH[rg] = a.c_zh * xs_h;
    a.ZV[rg] = a.c_zh * xs_v;
    a.ZDR[rg] = xs_h / xs_v;
    const float kdp = a.c_kdp * (tot[10] - tot[8]);
    a.KDP[rg] = kdp;
    const float k2 = 2.0f * kdp;
    k2_out = (k2 == k2) ? k2 : 0.0f;                           // nan_cumsum
    const float att_h = 4.343e-3f * (a.c_2w * tot[11]);
    const float att_v = 4.343e-3f * (a.c_2w * tot[9]);
    a.ATT_H[rg] = att_h;
    a.ATT_V[rg] = att_v;
    fh_out = fv_out = 1.0f;
    if (a.with_attenuation) {
        // 10**(-0.1*A*(radial_res/1000.)) in float32 (doppler_scatter.py:413-414); NaN -> 1
#if CPOL_EXP_FINISH_CHEAP          // (timing experiment only: wrong values)
        float fh = 1.0f - 0.1f * att_h * a.res_km;
        float fv = 1.0f - 0.1f * att_v * a.res_km;
#else
        float fh = (float)exp10((double)(-0.1f * att_h * a.res_km));
        float fv = (float)exp10((double)(-0.1f * att_v * a.res_km));
#endif
        fh_out = (fh == fh) ? fh : 1.0f;
        fv_out = (fv == fv) ? fv : 1.0f;
    }
    const float t47 = tot[4] + tot[7], t65 = tot[6] - tot[5];
    const float aa = t47 * t47 + t65 * t65;
    a.RHOHV[rg] = sqrtf(aa / (b * cc));
#if CPOL_EXP_FINISH_CHEAP
    a.DELTA_HV[rg] = (tot[5] - tot[6]) / (-tot[4] - tot[7]);
#else
    a.DELTA_HV[rg] = (float)atan2((double)(tot[5] - tot[6]), (double)(-tot[4] - tot[7]));
#endif

    // ---- radial velocity, Doppler scheme 1 (doppler_scatter.py:276-281, 313-333, 418-420) ----
    if (a.RVEL && !CPOL_SKIP_RVEL) {
        double rv = __builtin_nan(""), tw = 0.0;
        // with many sub-beams (the terms come from k_rvel_terms): in groups of CPOL_FINAL_SBATCH, the
        // terms (and per-gate weights) of a group requested together; the float64 accumulation keeps
        // the reference's order.  (C4 volume: k_final 467 -> 258 us; its 225-ray share 100 -> 70 us.)
        int s_done = 0;
        if (a.proj)
            for (; s_done + CPOL_FINAL_SBATCH <= a.n_sub; s_done += CPOL_FINAL_SBATCH) {
                double pj[CPOL_FINAL_SBATCH], wj[CPOL_FINAL_SBATCH];
#pragma unroll
                for (int q = 0; q < CPOL_FINAL_SBATCH; ++q) {
                    const long sbg = sbg0 + (long)(s_done + q) * a.n_gates;
                    pj[q] = a.proj[sbg];
                    wj[q] = a.wgate ? a.wgate[sbg] : a.sub_w[s_done + q];
                }
#pragma unroll
                for (int q = 0; q < CPOL_FINAL_SBATCH; ++q) {
                    const double proj = pj[q], w = wj[q];
                    if (proj == proj) tw += w;
                    double x = (rv == rv) ? rv : 0.0;
                    double y = proj * w;
                    if (!(y == y)) y = 0.0;
                    rv = x + y;
                }
            }
        for (int s = s_done; s < a.n_sub; ++s) {
            const long sbg = sbg0 + (long)s * a.n_gates;
            const double proj = a.proj ? a.proj[sbg] : have_moments ? proj_from_moments(a, ray, s, sbg, n_sbg, mom_v, mom_n)
                                                       : subbeam_proj(a, ray, s, gate, sbg, n_sbg);
            const double w = a.wgate ? a.wgate[sbg] : a.sub_w[s];
            if (proj == proj) tw += w;
            double x = (rv == rv) ? rv : 0.0;
            double y = proj * w;
            if (!(y == y)) y = 0.0;
            rv = x + y;
        }
        rv = rv / tw;
        if (a.nyquist) {
            // aliasing (utilities.py:142-156)
            const double nyq = a.nyquist[ray], pi = 3.14159265358979323846;
            const double theta = (rv + nyq) / (2 * nyq) * pi - pi / 2.;
            const double fold = atan(tan(theta));
            rv = (fold + pi / 2) * (2 * nyq) / pi - nyq;
        }
        a.RVEL[rg] = rv;
    }

    // ---- radial mask (doppler_scatter.py:472-477) ----
    double msum = 0.0;
    {
        int s = 0;
        for (; s + CPOL_FINAL_SBATCH <= a.n_sub; s += CPOL_FINAL_SBATCH) {
            signed char mj[CPOL_FINAL_SBATCH];
#pragma unroll
            for (int q = 0; q < CPOL_FINAL_SBATCH; ++q) mj[q] = a.sub_mask[sbg0 + (long)(s + q) * a.n_gates];
#pragma unroll
            for (int q = 0; q < CPOL_FINAL_SBATCH; ++q) msum += (double)mj[q];
        }
        for (; s < a.n_sub; ++s) msum += (double)a.sub_mask[sbg0 + (long)s * a.n_gates];
    }
    if (a.mask8) a.mask8[rg] = (signed char)(int)msum;        // (a sum of at most 63 codes in [-1, 2]: exact)
    msum /= (double)a.n_sub;
    if (msum > -1.0 && msum <= 0.0) msum = 0.0;
    if (a.mask) a.mask[rg] = msum;

    // ---- integrate_radials: NaN-skipping weighted sum (float64) ----
    if (a.model_vars) {
        for (int v = 0; v < a.n_vars; ++v) {
            double acc = 0.0;
            for (int s = 0; s < a.n_sub; ++s) {
                const long sbg = sbg0 + (long)s * a.n_gates;
                double y = a.wgate ? (double)a.vals[(long)v * n_sbg + sbg] * a.wgate[sbg] / wtot
                                   : (double)a.vals[(long)v * n_sbg + sbg] * a.sub_w[s] / a.sum_w;
                if (y == y) acc += y;
            }
            a.model_vars[(long)v * n_rg + rg] = acc;
        }
    }
}

__device__ __forceinline__ void final_gate(const FinalArgs &a, const ItabSet &its, int ray, int gate, float &k2_out,
                                           float &fh_out, float &fv_out)
{
    const long n_rg = (long)a.n_rays * a.n_gates;
    const long rg = (long)ray * a.n_gates + gate;
    const long n_sbg = n_rg * a.n_sub;
    const long sbg0 = (long)ray * a.n_sub * a.n_gates + gate;      // + sub * n_gates
    const float qnan = __builtin_nanf("");

    // scheme 'ml': weights renormalised gate by gate (doppler_scatter.py:124-129)
    double wtot = 0.0;
    if (a.wgate)
        for (int s = 0; s < a.n_sub; ++s) wtot += a.wgate[sbg0 + (long)s * a.n_gates];

    float tot[CPOL_N_SZ];
    for (int j = 0; j < a.n_hydro; ++j) {
        float acc[CPOL_N_SZ];
#pragma unroll
        for (int c = 0; c < CPOL_N_SZ; ++c) acc[c] = qnan;
        if (a.pre_integ) {
            const float4 *pi = reinterpret_cast<const float4 *>(a.sz_integ + (rg * a.n_hydro + j) * CPOL_N_SZ);
#pragma unroll
            for (int c4 = 0; c4 < CPOL_N_SZ / 4; ++c4) {
                const float4 q4 = pi[c4];
                acc[4 * c4] = q4.x; acc[4 * c4 + 1] = q4.y; acc[4 * c4 + 2] = q4.z; acc[4 * c4 + 3] = q4.w;
            }
        } else if (a.eval_1d && its.t[j].tab && !its.t[j].two_d) {
            // (uniform per hydrometeor) one to three sub-beams, the table items evaluated in place: the lane
            // gathers the 11 rows of ITS (slice, panel) block, 12 Horner chains advance together
            // (itab1_columns: the same coefficients in the same order as k_psd_lookup / k_subbeam_sum)
            const ItabDev &t = its.t[j];
            for (int sq = 0; sq < a.n_sub; ++sq) {
                const long sbg = sbg0 + (long)sq * a.n_gates;
                if (!((a.vmask[sbg] >> j) & 1)) continue;
                const double2 rc = a.rec[(long)j * n_sbg + sbg];
                double2 v[CPOL_N_SZ / 2];
                if (rc.x >= 0.0) {
                    const int key = a.key[(long)j * n_sbg + sbg];
                    const int pn = min((int)rc.x, t.n_pan - 1);
                    const double u = 2.0 * (rc.x - (double)pn) - 1.0;
                    // (itab1_columns with the rows requested CPOL_FINAL_ROW_UNROLL at a time: all 66 loads in
                    // flight at once need 264 VGPRs)
                    constexpr int NC = CPOL_ITAB1_NC, NFP = CPOL_ITAB_NFP;
                    const double2 *blk = reinterpret_cast<const double2 *>(
                        t.tab + ((long)(key - a.key_base[j]) * t.n_pan + pn) * (NC * NFP));
#pragma unroll
                    for (int f = 0; f < CPOL_N_SZ / 2; ++f) v[f] = blk[(NC - 1) * (NFP / 2) + f];
#pragma unroll CPOL_FINAL_ROW_UNROLL
                    for (int q = NC - 2; q >= 0; --q) {
#pragma unroll
                        for (int f = 0; f < CPOL_N_SZ / 2; ++f) {
                            const double2 cq = blk[q * (NFP / 2) + f];
                            v[f].x = fma(v[f].x, u, cq.x);
                            v[f].y = fma(v[f].y, u, cq.y);
                        }
                    }
#pragma unroll
                    for (int f = 0; f < CPOL_N_SZ / 2; ++f) { v[f].x *= rc.y; v[f].y *= rc.y; }
                } else {                                       // (an item outside the table: integrated bin by bin)
                    const double2 *r = reinterpret_cast<const double2 *>(a.res + ((long)j * n_sbg + sbg) * CPOL_N_SZ);
#pragma unroll
                    for (int c = 0; c < CPOL_N_SZ / 2; ++c) v[c] = r[c];
                }
                const double w = a.wgate ? a.wgate[sbg] / wtot : a.sub_w[sq];
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ; ++c) {
                    double y = ((c & 1) ? v[c / 2].y : v[c / 2].x) * w;
                    double x = (double)acc[c];
                    if (!(x == x)) x = 0.0;
                    if (!(y == y)) y = 0.0;
                    acc[c] = (float)(x + y);
                }
            }
        } else
        // sub-beams in groups of CPOL_FINAL_BATCH: the keys and the result rows of a group are read
        // before the first of them is used (independent loads in flight: a thread walks
        // n_hydro x n_sub rows, 294 with 7x7 sub-beams), the float32 accumulation keeps its order.
        // C4 sweep at 3 degrees: one row at a time 1.34 ms, groups of 2 / 4 / 7: 1.13 / 1.15 / 1.96 ms
        // (512 threads per ray: no change); reading the rows of invalid items too: 1.63 ms.
        for (int s0 = 0; s0 < a.n_sub; s0 += CPOL_FINAL_BATCH) {
            int kk[CPOL_FINAL_BATCH];
            double2 rr[CPOL_FINAL_BATCH][CPOL_N_SZ / 2];
#pragma unroll
            for (int q = 0; q < CPOL_FINAL_BATCH; ++q) {
                const int sq = min(s0 + q, a.n_sub - 1);
                kk[q] = (s0 + q < a.n_sub && ((a.vmask[sbg0 + (long)sq * a.n_gates] >> j) & 1)) ? 0 : -1;
            }
#pragma unroll
            for (int q = 0; q < CPOL_FINAL_BATCH; ++q) {
                if (kk[q] < 0) continue;
                const long sbg = sbg0 + (long)(s0 + q) * a.n_gates;
                const double2 *r = reinterpret_cast<const double2 *>(a.res + ((long)j * n_sbg + sbg) * CPOL_N_SZ);
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ / 2; ++c) rr[q][c] = r[c];
            }
#pragma unroll
            for (int q = 0; q < CPOL_FINAL_BATCH; ++q) {
                if (kk[q] < 0) continue;
                const int sq = s0 + q;
                const long sbg = sbg0 + (long)sq * a.n_gates;
                const double w = a.wgate ? a.wgate[sbg] / wtot : a.sub_w[sq];
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ; ++c) {
                    // nansum([float32 acc, float64 term]) stored back as float32
                    double y = ((c & 1) ? rr[q][c / 2].y : rr[q][c / 2].x) * w;
                    double x = (double)acc[c];
                    if (!(x == x)) x = 0.0;
                    if (!(y == y)) y = 0.0;
                    acc[c] = (float)(x + y);
                }
            }
        }
        if (a.sz_integ && !a.pre_integ) {
#pragma unroll
            for (int c = 0; c < CPOL_N_SZ; ++c)
                a.sz_integ[(rg * a.n_hydro + j) * CPOL_N_SZ + c] = acc[c];
        }
        // np.nansum(sz_integ, axis=1): float32, hydrometeors in order
#pragma unroll
        for (int c = 0; c < CPOL_N_SZ; ++c) {
            float v = (acc[c] == acc[c]) ? acc[c] : 0.0f;
            tot[c] = (j == 0) ? v : tot[c] + v;
        }
    }
    gate_finish(a, ray, gate, tot, false, 0.0, 0.0, wtot, k2_out, fh_out, fv_out);
}

// one strictly sequential float32 scan over sv[0 .. ng) in LDS by ONE lane (MUL: running product, else running sum), the
// operands of the next 8 steps requested before the 8 dependent operations of the current ones
template <bool MUL>
__device__ __forceinline__ void scan_lds_sequential(float *sv, int ng)
{
    float c = 0.0f;
    int gg = 0;
    float nx[8];
    if (ng >= 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) nx[q] = sv[q];
    }
    for (; gg + 8 <= ng; gg += 8) {
        float q8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) q8[q] = nx[q];
        if (gg + 16 <= ng) {
#pragma unroll
            for (int q = 0; q < 8; ++q) nx[q] = sv[gg + 8 + q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) { c = (gg + q == 0) ? q8[0] : (MUL ? c * q8[q] : c + q8[q]); q8[q] = c; }
#pragma unroll
        for (int q = 0; q < 8; ++q) sv[gg + q] = q8[q];
    }
    for (; gg < ng; ++gg) { c = (gg == 0) ? sv[gg] : (MUL ? c * sv[gg] : c + sv[gg]); sv[gg] = c; }
}

// The same scan by a whole wavefront, SAME BITS: gate 64 r + l sits in lane l of row r; inside a row the running value moves
// one lane to the right per step (v_mov_b32_dpp wave_shr:1) and every lane applies its own operand -- after step t the lanes
// 0..t hold their final values, because lane l's value was formed from the FINAL value of lane l - 1: the additions happen in
// gate order with the same operands as the one-lane loop, 63 steps of two instructions per row instead of 64 dependent
// LDS round trips.  Lane 0 is not written by the shifted operation and keeps its value; the first gate of the ray is taken
// as it is (np.cumsum's first element), the first gate of a later row continues from the last lane of the row before; lanes
// behind the last gate carry the identity (-0.0 for the sum: x + -0.0 == x for every x, signed zeros included; 1.0 for the product).
template <bool MUL>
__device__ __forceinline__ void scan_lds_wave_exact(float *sv, int ng, int lane)
{
    const float ident = MUL ? 1.0f : -0.0f;
    float carry = ident;
    for (int row0 = 0; row0 < ng; row0 += 64) {
        const int gg = row0 + lane;
        const float x = gg < ng ? sv[gg] : ident;
        float s = x;
        if (lane == 0 && row0 > 0) s = MUL ? carry * x : carry + x;
        // (always 63 steps, unrolled: the lanes behind the ray's last gate carry the identity.  The operation itself takes its
        // first operand through DPP: s[l] = s[l - 1] op x[l]; lane 0 has no source lane and, with bound_ctrl off, is not
        // written -- it keeps its value.  Two wait states between a VALU write and a DPP read of the same register; five after
        // the EXEC write that ends the `lane == 0` branch above -- the compiler cannot see into the block.)
        if (MUL) asm volatile("s_nop 4\n\t.rept 63\n\ts_nop 1\n\tv_mul_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t.endr" : "+&v"(s) : "v"(x));
        else     asm volatile("s_nop 4\n\t.rept 63\n\ts_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t.endr" : "+&v"(s) : "v"(x));
        if (gg < ng) sv[gg] = s;
        carry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 63));
    }
}

#ifndef CPOL_SCAN_FORM
#define CPOL_SCAN_FORM 1          // 0: one lane walks the ray through LDS (rounds 1-5); 1: the wavefront form above (same bits); 2: no scan at all (measurement of the kernel's floor: wrong results)
#endif

// One workgroup per ray: every thread finishes gates (final_gate), the operands of the three
// strictly sequential float32 range scans (np.cumsum / np.cumprod order is part of the
// numerical contract) go through LDS, lane 0 of waves 0..2 runs one scan each, then all
// threads finish PHIDP, the attenuated ZDR and the sensitivity cut.  (One launch instead of
// the former k_final_gate + k_final_ray pair; no global round trip of the scan operands.)
struct ScanRayArgs {
    float *PHIDP;
    double *RVEL;               // RVEL of the spectrum kernels (Doppler scheme 3) or NULL
    const double *sens_thr;     // [n_gates] or NULL
    float radial_res;           // float32 cast of the python scalar
};

#ifndef CPOL_FINAL_THREADS
#define CPOL_FINAL_THREADS 256
#endif
// (Measured and dropped, round 4: a second instantiation for sweeps whose gates k_gate1 has finished -- scans, cut and the
// rare recomputed gates -- held to 64 VGPRs so that a CU holds eight workgroups instead of two (the allocator's 243
// registers are for the per-gate function, which almost never runs there): 840 B of scratch per lane, and the C2 sweep's
// k_final 18.5 -> 27.9 us, the C5 Ku swath's 183 -> 319 us; at 128 / 168 VGPRs the swath takes 948 / 921 us against 945.)
// THREADS: 256 (a gate or two per thread); 512 when the sweep has fewer rays than the GPU has CUs
// (the share of one of N GPUs: the kernel's duration is that of ONE workgroup then)
#ifndef CPOL_FINAL_WPE
#define CPOL_FINAL_WPE 0             // experiment knob: wavefronts per SIMD asked of the register allocator (0: its own choice)
#endif
#if CPOL_FINAL_WPE
#define CPOL_FINAL_ATTR __attribute__((amdgpu_waves_per_eu(CPOL_FINAL_WPE, CPOL_FINAL_WPE)))
#else
#define CPOL_FINAL_ATTR
#endif
template <int THREADS>
__global__ __launch_bounds__(THREADS) CPOL_FINAL_ATTR void k_final(FinalArgs a, ScanRayArgs r, ItabSet its)
{
    extern __shared__ float lds[];          // [3][n_gates]
    const int ray = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0: KDP sum, 1: FH product, 2: FV product
    const int ng = a.n_gates;
    const long base = (long)ray * ng;
    float *s_k = lds, *s_h = lds + ng, *s_v = lds + 2 * ng;
    for (int g = tid; g < ng; g += THREADS) {
        float k2, fh, fv;
        bool redo = true;
        if (a.pre_gate) {
            redo = a.defer[base + g] != 0;
            if (a.ice_redo) redo = redo || a.ice_first[ray].first_gate == g;      // (one sub-beam: [ray * n_sub + 0])
        }
        if (redo) final_gate(a, its, ray, g, k2, fh, fv);
        else { k2 = a.sk[base + g]; fh = a.sh[base + g]; fv = a.sv[base + g]; }
        s_k[g] = k2;
        s_h[g] = fh;
        s_v[g] = fv;
    }
    __syncthreads();
    // strictly sequential float32 scans (np.cumsum / np.cumprod order is part of the numerical contract), one wavefront per scan.
    // Round 6: the whole wavefront scans, same bits (scan_lds_wave_exact below: the running value moves one lane to the right per
    // step through DPP, 63 steps of one shifted add / multiply per row of 64 gates) -- k_scan_rays 17.5 -> 10.2 us on the C2
    // sweep, of which 6.8 are the kernel without any scan (profiles/r6_scan_variants.txt).  CPOL_SCAN_FORM=0: one lane walks
    // the ray through LDS in chunks of 8 (rounds 1-5).  (Measured and dropped, round 4: every step taking its operand with
    // v_readlane while all lanes carry the running value: k_final 18.4 -> 35.7 us.)
    if (wave == 0 || (a.with_attenuation && wave < 3)) {          // (wave-uniform)
        float *sv = (wave == 0) ? s_k : (wave == 1 ? s_h : s_v);
#if CPOL_SCAN_FORM == 1
        if (wave == 0) scan_lds_wave_exact<false>(sv, ng, lane);
        else scan_lds_wave_exact<true>(sv, ng, lane);
#elif CPOL_SCAN_FORM == 0
        if (lane == 0) {
            if (wave == 0) scan_lds_sequential<false>(sv, ng);
            else scan_lds_sequential<true>(sv, ng);
        }
#endif
    }
    __syncthreads();
    double *rvel = a.RVEL ? a.RVEL : r.RVEL;
    for (int g = tid; g < ng; g += THREADS) {
        const long i = base + g;
        // (this thread wrote these four values itself in the first loop)
        float zh = a.ZH[i], zv = a.ZV[i];
        float phidp = s_k[g] * r.radial_res / 1000.0f + a.DELTA_HV[i];
        float zdr = a.ZDR[i];
        if (a.with_attenuation) zdr = (zh * s_h[g]) / (zv * s_v[g]);
        bool cut = false;
        if (r.sens_thr) {
            // 10*np.log10(ZH) (float32) < threshold(r) (float64)
            float dbz = 10.0f * (float)log10((double)zh);
            cut = (double)dbz < r.sens_thr[g];
        }
        if (cut) {
            const float qnan = __builtin_nanf("");
            a.ZH[i] = qnan; a.ZV[i] = qnan; a.KDP[i] = qnan; a.RHOHV[i] = qnan;
            zdr = qnan; phidp = qnan;
            if (rvel) rvel[i] = __builtin_nan("");
        }
        r.PHIDP[i] = phidp;
        a.ZDR[i] = zdr;
    }
}
