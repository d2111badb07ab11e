// cpol_psd.inl -- melting diagnosis, PSD parameters, LUT-slice bucketing and the
// PSD x scattering-table integration (the dominant kernel).
//
// Reference functions replaced (wolfidan/cosmo_pol):
//   melting                         interpolation/melting.py:19-90
//   Rain/Snow/Graupel.set_psd       hydrometeors/hydrometeors.py:747-772, 879-907, 1025-1049
//   _Hydrometeor.set_psd (2-moment) hydrometeors/hydrometeors.py:212-256
//   IceParticle.set_psd / get_N     hydrometeors/hydrometeors.py:1302-1373, 1231-1250
//   Melting*.set_psd / get_N / ...  hydrometeors/hydrometeors.py:333-478, 1398-1481
//   _Hydrometeor.get_N              hydrometeors/hydrometeors.py:128-147
//   Lookup_table.lookup_line        lookup/lut.py:309-344  (floor-bin, no interpolation)
//   PSD integral (rectangle rule)   scatter/doppler_scatter.py:246-251
//
// Design: the reference gathers one [1024 x 12] float64 LUT slice per valid
// (gate, hydrometeor) item and contracts it with N(D).  Items with the same
// (hydrometeor, elevation bin, temperature bin) share the slice, so items are
// counting-sorted by slice ("bucket") and cut into work units of up to 64 (128 in the
// recurrence flavour) items of ONE bucket.  A workgroup of 8 wavefronts integrates a
// unit with the item on the lane and the 1024 diameter bins split over the waves: the
// slice row, D^mu and D^nu are wave-uniform and arrive through the scalar data path
// (s_load -> SGPR operand of v_fma_f64), the 12 float64 accumulators per item stay in
// VGPRs, and the only cross-wave step is the fixed-order combine of the 8 partial sums.

// ---------------------------------------------------------------- helpers
__device__ __forceinline__ int clip_bin(float q, int n)
{
    // np.array(np.floor(x), dtype=int) then clip to [0, n-1]   (lut.py:336-341)
    float fl = floorf(q);
    int i;
    if (!(fl == fl)) i = 0;                       // NaN -> INT_MIN -> clipped to 0
    else if (fl >= 2147483648.0f) i = n - 1;
    else if (fl <= -2147483648.0f) i = 0;
    else i = (int)fl;
    return min(max(i, 0), n - 1);
}

__device__ __forceinline__ int clip_bin64(double q, int n)
{
    double fl = floor(q);
    int i;
    if (!(fl == fl)) i = 0;
    else if (fl >= 2147483648.0) i = n - 1;
    else if (fl <= -2147483648.0) i = 0;
    else i = (int)fl;
    return min(max(i, 0), n - 1);
}

// Block-aggregated bucket ranking.  Global atomics are the scarce resource here
// (measured: ~17k same-sweep atomics cost 54 us, 5x the rest of the kernel), so
// the wavefronts of a workgroup (CPOL_RANK_WAVES) first merge their per-key counts
// in a small LDS hash table; ONE returning global atomic per distinct key per
// workgroup then reserves the key's range and every item derives its final
// position inside its bucket:  pos = base(block,key) + items of earlier waves
// + rank inside the wave.  (Order inside a bucket is irrelevant: every item
// writes its own result slot.)
#define CPOL_RANK_SLOTS 128
#ifndef CPOL_RANK_WAVES
#define CPOL_RANK_WAVES 4            // wavefronts per classify workgroup.  16 merged the most atomics (round 1) but left ONE
#endif                               // workgroup per CU at the kernel's 123 VGPRs: its 16 wavefronts end at different times
                                     // (gates without hydrometeors) and the CU idles until the last one.  C4 volume, all
                                     // at 4 wavefronts per SIMD: 16 / 8 / 4 / 2 / 1 -> 2.45 / 2.36 / 2.14 / 3.61 / 6.44 ms
                                     // (below 4 the per-workgroup reset of the LDS tables dominates); with every item
                                     // ranked (CPOL_ITAB=0, C4 sweep) 1.26 -> 0.92 ms, on the small C2 sweep 22 -> 30 us.
struct RankShared {
    int key[CPOL_RANK_SLOTS];
    int base[CPOL_RANK_SLOTS];
    int cnt[CPOL_RANK_SLOTS][CPOL_RANK_WAVES];
};

__device__ __forceinline__ void rank_reset(RankShared &sh)
{
    for (int i = threadIdx.x; i < CPOL_RANK_SLOTS; i += blockDim.x) sh.key[i] = -1;
    for (int i = threadIdx.x; i < CPOL_RANK_SLOTS * CPOL_RANK_WAVES; i += blockDim.x)
        (&sh.cnt[0][0])[i] = 0;
}

// phase A (per hydrometeor, no barrier): wave-level grouping + LDS table insert.
// returns a packed ticket: bit 31 = final position already known (table overflow ->
// direct global claim), else (slot << 6) | rank-in-wave; -1 for invalid lanes
__device__ __forceinline__ int rank_insert(RankShared &sh, int *__restrict__ count, int key, bool valid)
{
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    unsigned long long todo = __ballot(valid);
    int ticket = -1;
    while (todo) {                                  // wave-uniform loop
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __shfl(key, leader);
        const unsigned long long m = __ballot(valid && key == k);
        int sl = -1, db = -1;
        if (lane == leader) {
            int h = (int)(((unsigned)k * 2654435761u) >> 25) & (CPOL_RANK_SLOTS - 1);
            for (int probe = 0; probe < CPOL_RANK_SLOTS; ++probe) {
                int prev = atomicCAS(&sh.key[h], -1, k);
                if (prev == -1 || prev == k) { sl = h; break; }
                h = (h + 1) & (CPOL_RANK_SLOTS - 1);
            }
            if (sl >= 0) sh.cnt[sl][wave] = __popcll(m);
            else db = atomicAdd(&count[k], __popcll(m));     // table full: direct claim
        }
        sl = __shfl(sl, leader);
        db = __shfl(db, leader);
        if (valid && key == k) {
            const int rank = __popcll(m & ((1ull << lane) - 1ull));
            ticket = (sl >= 0) ? ((sl << 6) | rank) : (int)(0x80000000u | (unsigned)(db + rank));
        }
        todo &= ~m;
    }
    return ticket;
}

// phase B (one thread per (hydrometeor, slot), after a barrier): exclusive prefix over
// the waves and ONE returning global atomic per distinct key of the workgroup
__device__ __forceinline__ void rank_reserve(RankShared &sh, int *__restrict__ count, int t)
{
    const int k = sh.key[t];
    if (k < 0) return;
    int run = 0;
#pragma unroll
    for (int w = 0; w < CPOL_RANK_WAVES; ++w) {
        int c = sh.cnt[t][w];
        sh.cnt[t][w] = run;
        run += c;
    }
    sh.base[t] = atomicAdd(&count[k], run);
}

// phase C (after a barrier): ticket -> position inside the bucket
__device__ __forceinline__ int rank_position(const RankShared &sh, int ticket)
{
    if (ticket < 0) return (ticket == -1) ? -1 : (int)((unsigned)ticket & 0x7fffffffu);
    const int slot = ticket >> 6, rank = ticket & 63;
    return sh.base[slot] + sh.cnt[slot][threadIdx.x >> 6] + rank;
}

// ---------------------------------------------------------------- per-bin math of the melting flavour
// The melting species integrate over a per-ITEM diameter grid, so nothing can be tabulated
// on the host: every (item, bin) needs two cube roots, a square root, an exponential, a
// fourth root (snow) or two powers (graupel) and a division.  The OCML versions cost 32 /
// 22 / 42 / 44 / 215 / 12 VALU instructions; the versions below reach 1e-15 relative
// accuracy (tests/test_gpu_math.py) in 15-32, from float32 hardware seeds refined by
// Newton steps in float64 and short polynomials.  Inputs are positive and normal here.
// a * b + c with a wave-uniform addend c taken from an SGPR pair: `v_fma_f64 dst, a, b, s[c]`.
// Left to itself the compiler prefers the two-address form v_fmac_f64 (addend = destination)
// and first copies a uniform / literal addend into VGPRs with one v_mov_b64 or two v_mov_b32:
// 2-3 VALU issues per Horner step instead of one (tools/isa_loops.py: 97 v_mov_b64 per two
// bins of the melting loop, 160 v_mov_b32 in its table-driven form).  The kernels below are
// VALU-issue bound, so the addend goes through the scalar operand explicitly.
__device__ __forceinline__ double fma_sc(double a, double b, double c_uniform)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_uniform));
    return r;
}

__device__ __forceinline__ double cp_exp(double x)
{
    // x = n ln2 + r, |r| <= ln2/2; Taylor to r^12/12! (next term < 2e-16)
    const double n = rint(x * 1.4426950408889634);
    double r = fma(n, -0.6931471803691238, x);
    r = fma(n, -1.9082149292705877e-10, r);
    double p = 1.0 / 479001600.0;
    p = fma_sc(p, r, 1.0 / 39916800.0);
    p = fma_sc(p, r, 1.0 / 3628800.0);
    p = fma_sc(p, r, 1.0 / 362880.0);
    p = fma_sc(p, r, 1.0 / 40320.0);
    p = fma_sc(p, r, 1.0 / 5040.0);
    p = fma_sc(p, r, 1.0 / 720.0);
    p = fma_sc(p, r, 1.0 / 120.0);
    p = fma_sc(p, r, 1.0 / 24.0);
    p = fma_sc(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

__device__ __forceinline__ double cp_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = r * fma(-x, r, 2.0);
    r = r * fma(-x, r, 2.0);
    return r;
}

__device__ __forceinline__ double cp_log(double x)
{
    // x = m 2^e, m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(s), s = (m-1)/(m+1), |s| < 0.1716
    int e;
    double m = frexp(x, &e);
    if (m < 0.7071067811865476) { m *= 2.0; e -= 1; }
    const double s = (m - 1.0) * cp_rcp(m + 1.0);
    const double z = s * s;
    // (addends through the scalar operand, as in cp_exp: as VGPR literals the compiler hoists them out
    // of k_classify's hydrometeor loop and holds 20 registers across it -- that kernel has 128)
    double p = 1.0 / 21.0;
    p = fma_sc(p, z, 1.0 / 19.0);
    p = fma_sc(p, z, 1.0 / 17.0);
    p = fma_sc(p, z, 1.0 / 15.0);
    p = fma_sc(p, z, 1.0 / 13.0);
    p = fma_sc(p, z, 1.0 / 11.0);
    p = fma_sc(p, z, 1.0 / 9.0);
    p = fma_sc(p, z, 1.0 / 7.0);
    p = fma_sc(p, z, 1.0 / 5.0);
    p = fma_sc(p, z, 1.0 / 3.0);
    p = fma_sc(p, z, 1.0);
    const double en = (double)e;
    return fma(en, 0.6931471803691238, fma(en, 1.9082149292705877e-10, 2.0 * s * p));
}

// x^(-1/6) -> cbrt(x) = x y^4 and sqrt(cbrt(x)) = x y^5 from ONE Newton sequence
__device__ __forceinline__ void cp_cbrt_and_sixth(double x, double &cb, double &sx)
{
    double y = (double)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)x) * (-1.0f / 6.0f));
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double y2 = y * y, y4 = y2 * y2;
        y = y * fma(-x * y2, y4, 7.0) * (1.0 / 6.0);       // y (7 - x y^6) / 6
    }
    const double y2 = y * y, y4 = y2 * y2;
    cb = x * y4;
    sx = cb * y;
}

__device__ __forceinline__ double cp_cbrt(double x)
{
    double y = (double)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)x) * (-1.0f / 3.0f));
#pragma unroll
    for (int it = 0; it < 2; ++it)
        y = y * fma(-x * y, y * y, 4.0) * (1.0 / 3.0);      // y (4 - x y^3) / 3
    return x * (y * y);
}

__device__ __forceinline__ double cp_fourth_root(double x)
{
    double y = (double)__builtin_amdgcn_rsqf(__builtin_amdgcn_sqrtf((float)x));   // x^(-1/4)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double y2 = y * y;
        y = y * fma(-x * y2, y2, 5.0) * 0.25;               // y (5 - x y^4) / 4
    }
    return x * (y * y * y);
}

// test hook (cpol_debug_math): evaluates one of the functions above on the device
__global__ void k_debug_math(int op, const double *__restrict__ x, double *__restrict__ y, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a, b;
    switch (op) {
    case 0: y[i] = cp_exp(x[i]); break;
    case 1: y[i] = cp_log(x[i]); break;
    case 2: y[i] = cp_cbrt(x[i]); break;
    case 3: cp_cbrt_and_sixth(x[i], a, b); y[i] = a; break;
    case 4: cp_cbrt_and_sixth(x[i], a, b); y[i] = b; break;
    case 5: y[i] = cp_fourth_root(x[i]); break;
    case 6: y[i] = cp_rcp(x[i]); break;
    case 7: {       // x[i] holds two float32 (n, d): 0 if div32_by(n, rcp_for_div32(d)) has the bits of n / d (two NaNs count as equal), else 1
        const float n = __int_as_float(__double2loint(x[i])), d = __int_as_float(__double2hiint(x[i]));
        const float q = n / d, p = div32_by(n, rcp_for_div32(d));
        y[i] = (__float_as_int(q) == __float_as_int(p) || (q != q && p != p)) ? 0.0 : 1.0;
        break; }
    default: y[i] = x[i];
    }
}

// x^y for x > 0 from the short log / exp above: ~55 VALU instructions instead of the 215 of
// the OCML pow, relative error < 1e-14 for |y log x| < 10 (PSD slopes, fall-speed moments)
#ifdef CPOL_POW_NOINLINE
__device__ __attribute__((noinline)) double cp_pow(double x, double y) { return cp_exp(y * cp_log(x)); }
#else
__device__ __forceinline__ double cp_pow(double x, double y) { return cp_exp(y * cp_log(x)); }
#endif

// float32 power via float64 (rounds to the correctly rounded float32 result in
// all but ~1e-8 of the cases; numpy's float32 power calls libm powf)
__device__ __forceinline__ float pow10_f32(float x) { return (float)pow(10.0, (double)x); }
__device__ __forceinline__ float exp_f32(float x) { return (float)exp((double)x); }

// ---------------------------------------------------------------- classify
struct ClassifyArgs {
    float *vals;                 // [n_vars][n_sbg] (QR/QS/QG zeroed in place by melting)
    const signed char *mask;
    const float *elev;           // folded
    float *q_melt;               // [2][n_sbg] QmS_v, QmG_v (float32 values); NULL unless debug reads are enabled
    double *fw_melt;             // [2][n_sbg] fwet_mS, fwet_mG
    int *key;                    // [n_hydro][n_sbg]
    int *pos;                    // [n_hydro][n_sbg] position inside the bucket
    double *par;                 // [n_hydro][CPOL_MAX_PAR][n_sbg]
    int *count;                  // [n_keys]
    long n_sbg;
    int with_melting;
    int var_qr, var_qs, var_qg;
    int doppler;                 // also store the analytic fall-speed moments (integrate_V)
    const double *wgate;         // [n_sbg] per-gate sub-beam weights (scheme 'ml') or NULL
    // float32 functions of T tabulated by the host's NumPy over every float32 in [128, 512) K
    // (cpol_stage_t_function): the reference's float32 exp / power are 1-2 ulp off the
    // correctly rounded value in 20-40 % of the arguments, and K_DP shows it
    const float *tfun_snow;      // snow intercept N0(T)  (hydrometeors.py:896) or NULL
    const float *tfun_ice;       // 10**a(T) of the Field (2005) moment relation (:1287) or NULL
    int *n_lookup;               // [1] items ranked for the integrating kernels; [2 .. 2 + CPOL_COUNT_SLOTS): items taken by the
                                 // integral tables, counted into one of many words (count_table_items); [0] unused
    double2 *rec;                // [n_hydro][n_sbg] items of a slot with an integral table: {panel position (-1: not
                                 // on the table), scale}
    double *vn;                  // [n_hydro][n_sbg][2] analytic fall-speed moments of the gamma species (Doppler
                                 // scheme 1) or NULL
    unsigned char *vmask;        // [n_sbg] bit j: hydrometeor j present (what the kernels that only ask
                                 // "key >= 0 ?" read instead of n_hydro keys)
    int keep_par;                // also write par[] of the table items (debug reads, Doppler scheme 3)
    // every slot has an integral table: the items outside the tables are a handful per volume, and the kernel lists
    // them directly as one-item work units -- no ranking tables, no counting sort, no scan / scatter launches
    int *rare_key;               // [n_hydro * n_sbg] LUT slice of listed item u (non-NULL: this mode)
    int *rare_perm;              // [n_hydro * n_sbg] its sub-beam gate
    unsigned long long *rare_totals;   // [0] items, [1] units (= items) listed; cleared by k_interp_sweep
    int *blk_ranked;             // [gridDim.x] items of this workgroup's gates ranked for the integrating kernels
                                 // (k_bucket_scatter skips the gates of a workgroup without any; pos[] is only
                                 // written where there are some)
    const unsigned *present;     // k_gate1_ray: [n_rays][ceil(n_gates / 64)] bit s = slot s may have an item in the tile (k_interp_sweep's
                                 // InterpArgs.present) or NULL
};

// The count of the items on integral tables (reported by cpol_counters, nothing on the device reads it): ONE word
// took an atomic per workgroup of k_classify -- 172 000 on the C4 volume, each waiting for the one before it
// at the same L2 address, ~11 ns apiece: 0.12 ms of the kernel's 2.0, and 3.2 ms of k_interp_classify's 6.6 when that
// kernel added per wavefront.  The words are spread over CPOL_COUNT_SLOTS; the host adds them up.
#define CPOL_COUNT_SLOTS 1024
__device__ __forceinline__ void count_table_items(int *n_lookup, int items)
{
    const unsigned slot = (blockIdx.x * 4u + (threadIdx.x >> 6)) & (CPOL_COUNT_SLOTS - 1);
    if (items) atomicAdd(n_lookup + 2 + slot, items);
}

#ifndef CPOL_EXP_NO_TFUN
#define CPOL_EXP_NO_TFUN 0
#endif
__device__ __forceinline__ bool tfun_lookup(const float *tab, float T, float &out)
{
    const unsigned idx = __float_as_uint(T) - CPOL_TFUN_FIRST_BITS;
    if (!tab || idx >= CPOL_TFUN_COUNT) return false;
#if CPOL_EXP_NO_TFUN               // (timing experiment only, wrong values: no gather into the 64-MB table)
    out = T * 1.0e-3f;
#else
    out = tab[idx];
#endif
    return true;
}

#define CPOL_MAX_PAR 6

// One (gate, hydrometeor): validity, LUT slice, PSD parameters, position on the integral table's panel axis,
// scale of a table item, analytic fall-speed moments -- everything k_classify hands on per item, computed
// in registers (k_classify stores it; k_gate1, the fused single-beam kernel, uses it in place).  The model variables
// of the gate are vals[var * n + i] (the sweep's array, or the workgroup's LDS copy in k_interp_classify); gi = the
// gate's index in the sweep.
struct ClassItem {
    int key;                 // LUT slice of a present species (-1: absent)
    double p0, p1, p2;       // parameter slots 0..2 of the integrating kernels
    double pf;               // position on the panel axis of the slot's integral table (where `lookup`)
    double scale;            // scale of a table item: gamma N0 exp(-lambda d0), ice and melting QM
    double dv, dn;           // analytic fall-speed moments (where `has_vn`)
    bool valid, lookup, has_vn;
};

__device__ __forceinline__ void classify_item(const HydroDev &h, const ItabDev &tj, const ClassifyArgs &a, const float *vals, long n, long i, long gi,
                                              bool in, float qm, double fw, float T0, int var_t0, float e, ClassItem &o)
{
    const cpol_hydro_desc &d = h.d;
    // NaN -> false (doppler_scatter.py:185); scheme 'ml': only gates where the
    // sub-beam has a non-zero weight (:186-189)
    const bool valid = in && (qm > 0.f) && (!a.wgate || a.wgate[gi] > 0.0);
    int key = -1;
    double p0 = 0.0, p1 = 0.0, p2 = 0.0;         // parameter slots 0..2
    double loglam = 0.0;                          // log of a slope parameter when the rule formed it as
    int ll_slot = -1;                             // (slot whose logarithm it is) exp(y log x): the table position and the fall-speed moments
                                                  // take it from there instead of a logarithm of the result
    if (valid) {
        const float T = d.var_t == var_t0 ? T0 : vals[d.var_t * n + i];
        // lut.py:336-341: float32 arithmetic for float32 queries
        int eb = clip_bin((e - d.e_lo) / d.e_step, d.n_e);
        int tb = d.second_axis_f64 ? clip_bin64((fw - (double)d.t_lo) / (double)d.t_step, d.n_t)
                                   : clip_bin((T - d.t_lo) / d.t_step, d.n_t);
        key = h.key_base + eb * d.n_t + tb;
        const double q = (double)qm;
        double lamf = 0.0, n0v = 0.0, qnv = 0.0;   // final lambda / N0 (get_N units)
        double ll = 0.0;                           // log(lambda) where the rule has it from its power (has_ll)
        // the item's parameters (slots 0..2 of par[], read by the integrating kernels; slot 1 by
        // the 2-D lookup of the melting species)
        switch (d.rule) {
        case CPOL_RULE_RAIN_1MOM:
        case CPOL_RULE_GRAUPEL_1MOM:
            ll = d.lam_exponent * cp_log(d.lambda_factor / q);
            lamf = cp_exp(ll);                            // = cp_pow(lambda_factor / q, lam_exponent)
            loglam = ll; ll_slot = 0;
            n0v = d.n0_fixed;
            p0 = lamf;
            p1 = 1.0;                                     // N0 folded into pre[]
            break;
        case CPOL_RULE_SNOW_1MOM: {
            // hydrometeors.py:896-899: float32 chain, then float64 from lambda_factor on
            float n0;
            if (!tfun_lookup(a.tfun_snow, T, n0))
                n0 = 13.5f * (565000.0f * exp_f32(-0.107f * (T - 273.15f))) / 1000.0f;
            float an0 = (float)d.a * n0;
            ll = d.lam_exponent * cp_log((double)an0 * d.lambda_factor / q);
            lamf = cp_exp(ll);
            loglam = ll; ll_slot = 0;
            n0v = (double)n0;
            p0 = lamf;
            p1 = n0v;
            break; }
        case CPOL_RULE_TWO_MOMENT: {
            // hydrometeors.py:231-246
            qnv = (double)vals[d.var_qn * n + i];
            double xm = q / (qnv + 2.220446049250313e-16);
            xm = fmin(fmax(xm, d.x_min), d.x_max);
            double lam = cp_pow(d.lambda_factor * xm, d.lam_exponent);
            double n0 = (d.nu / d.ntot_factor) * qnv * cp_pow(lam, d.n0_exponent);
            lamf = lam * d.c_lam;
            n0v = n0 * d.c_n0;
            p0 = lamf;
            p1 = n0v;
            break; }
        case CPOL_RULE_ICE_1MOM: {
            // hydrometeors.py:1277-1299 (float32 polynomials), :1320-1328
            const float Tc = T - 273.15f;
            const float n3 = 3.0f;
            float pa;
            if (!tfun_lookup(a.tfun_ice, T, pa)) {
                pa = 5.065339f - 0.062659f * Tc - (float)(3.032362 * 3) + 0.029469f * Tc * n3
                    - 0.000285f * (Tc * Tc) + (float)(0.312550 * 9) + 0.000204f * (Tc * Tc) * n3
                    + 0.003199f * Tc * 9.0f - (float)(0.015952 * 27);
                pa = pow10_f32(pa);
            }
            float pb = 0.476221f - 0.015896f * Tc + (float)(0.165977 * 3) + 0.007468f * Tc * n3
                - 0.000141f * (Tc * Tc) + (float)(0.060366 * 9) + 0.000079f * (Tc * Tc) * n3
                + 0.000594f * Tc * 9.0f - (float)(0.003577 * 27);
            const double qb = q / 3.0;                        // QM / BM_I
            const double Q2 = cp_pow(qb / (double)pa, (double)(1.0f / pb));
            const double Q22 = Q2 * Q2;
            double N0 = (Q22 * Q22) / (q * q * q);            // Q2^4 QM^-3 (b = 3)
            N0 /= 100000.0;
            p0 = Q2 / q;                                      // lambda (exponent 1/(b-2) = 1)
            p1 = N0;
            p2 = q;
            break; }
        case CPOL_RULE_MELTING_SNOW:
        case CPOL_RULE_MELTING_GRAUPEL: {
            p0 = q;
            p1 = fw;
            ll = d.r_lam_exponent * cp_log(d.r_lambda_factor / q);
            p2 = cp_exp(ll);                                         // rain partner
            loglam = ll; ll_slot = 2;                                // (the melting tables sit on slot 2)
            // (the dry partner's PSD does not enter get_N, hydrometeors.py:372-390)
            break; }
        default: break;
        }
    }
    // items whose lambda lies on the slot's integral table are finished from ONE 16-byte record {position on
    // the panel axis, scale of the item}; only the others are sorted by LUT slice for the integrating kernels
    bool lookup = false;
    double pf = -1.0, scale = 0.0;
    if (valid && tj.tab) {
        const double lg = ll_slot == tj.par_slot ? loglam : cp_log(tj.par_slot == 2 ? p2 : p0);
        pf = (lg * 1.4426950408889634 - tj.log2_lo) * (double)tj.ppo;
        lookup = pf >= (double)tj.pan_lo && pf < (double)tj.pan_hi;   // NaN -> false
        // scale: gamma N0 exp(-lambda d0) (the table holds exp(+lambda d0) x integral); ice and melting: QM
        scale = tj.two_d ? p0 : d.psd_family == CPOL_PSD_ICE_FIELD ? p2 : p1 * cp_exp(-(p0 * tj.d0));
    }
    double dv = 0.0, dn = 0.0;
    const bool has_vn = valid && a.doppler && d.psd_family == CPOL_PSD_GAMMA && !d.numeric_intv;
    if (has_vn) {
        // _Hydrometeor.integrate_V (hydrometeors.py:178-199): analytic moments (a species with
        // numeric_intv gets its sums from the PSD stage instead); lambda = p0, N0 = the fixed
        // intercept (rain, graupel) or p1 (few values stay live across the table decision above:
        // k_classify is held to 128 VGPRs, CPOL_CLASSIFY_WPE)
        const double n0v = (d.rule == CPOL_RULE_RAIN_1MOM || d.rule == CPOL_RULE_GRAUPEL_1MOM) ? d.n0_fixed : p1;
        const double lp = ll_slot == 0 ? loglam : cp_log(p0);  // one logarithm of lambda for both powers
        dv = d.vel_factor * n0v * d.alpha / d.nu * cp_exp(-(d.beta + d.mu + 1) / d.nu * lp);
        dn = (d.rule == CPOL_RULE_TWO_MOMENT)
            ? (double)vals[d.var_qn * n + i] : d.ntot_factor * n0v / d.nu * cp_exp(-(d.mu + 1) / d.nu * lp);
    }
    o.key = key; o.p0 = p0; o.p1 = p1; o.p2 = p2; o.pf = lookup ? pf : -1.0; o.scale = scale;
    o.dv = dv; o.dn = dn; o.valid = valid; o.lookup = lookup; o.has_vn = has_vn;
}


// (Measured with the stores of key / rec / vn dropped one by one: 2.55 ms each time against 2.57 -- the kernel
// waits for its own dependent float64 chains at 4 wavefronts per SIMD, not for memory; the Doppler moments
// are 0.27 ms of it.)
#define CPOL_CLASSIFY_THREADS (CPOL_RANK_WAVES * CPOL_WAVE)
// (Measured and dropped: finishing the table items inside this kernel -- the lookup fused in, the
// parameters never leaving the registers, 2 GB less HBM traffic on the C4 sweep -- took 3.35 ms
// against 0.69 ms + 2.21 ms for k_classify + k_psd_lookup: the lookup is bound by the vector-L1
// gather, not by HBM, and inside the 16-wave ranking workgroups it spills; 4- and 8-wave
// workgroups 3.6 / 4.5 ms.)
// (Measured and dropped, round 4: both float32 functions of T requested right after T is loaded, ahead of the
// hydrometeor loop and whether or not snow / ice are present at the gate -- they sit behind `valid` in the
// dependent chain of those species: C4 volume 2.11 -> 2.20 ms, C2 sweep 19.7 -> 21.5 us.)
#ifndef CPOL_CLASSIFY_WPE
#define CPOL_CLASSIFY_WPE 4          // wavefronts per SIMD asked of the register allocator (128 VGPRs: what 16-wave workgroups
                                     // imposed; without it the allocator takes more registers and occupancy drops)
#endif
#if CPOL_CLASSIFY_WPE
#define CPOL_CLASSIFY_ATTR __attribute__((amdgpu_waves_per_eu(CPOL_CLASSIFY_WPE, CPOL_CLASSIFY_WPE)))
#else
#define CPOL_CLASSIFY_ATTR
#endif
__global__ __launch_bounds__(CPOL_CLASSIFY_THREADS) CPOL_CLASSIFY_ATTR void k_classify(HydroSet hs, ItabSet its, ClassifyArgs a)
{
    // one LDS ranking table per hydrometeor: all global atomics of the workgroup are
    // issued in ONE round (phase B) instead of one dependent round per hydrometeor
    __shared__ RankShared sh[CPOL_MAX_HYDRO];
    __shared__ int s_lookup, s_ranked;
    const bool direct = a.rare_key != nullptr;                 // (uniform)
    if (!direct)
        for (int j = 0; j < hs.n_hydro; ++j) rank_reset(sh[j]);
    if (threadIdx.x == 0) { s_lookup = 0; s_ranked = 0; }
    __syncthreads();
    int my_lookup = 0, my_ranked = 0;
    unsigned ranked_mask = 0;                         // bit j: item j of this gate goes to the integrating kernels
    unsigned vbits = 0;                               // bit j: hydrometeor j is present at this gate (key >= 0)
    const long sbg = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = sbg < a.n_sbg;
    const long n = a.n_sbg;
    const long i = in ? sbg : 0;
    float qms = 0.f, qmg = 0.f;
    double fws = 0.0, fwg = 0.0;
    if (a.with_melting && in) {
        // melting.py:34-83, float32 arithmetic as NumPy evaluates it
        float qr = a.vals[a.var_qr * n + i], qs = a.vals[a.var_qs * n + i],
              qg = a.vals[a.var_qg * n + i];
        float qsg = qs + qg;
        if (qr > 0.f && qsg > 0.f) {
            qms = qs + qr * (qs / qsg);
            qmg = qg + qr * (qg / qsg);
            if (qms > 0.f || qmg > 0.f) {
                a.vals[a.var_qr * n + i] = 0.f;
                a.vals[a.var_qs * n + i] = 0.f;
                a.vals[a.var_qg * n + i] = 0.f;
            }
            fws = (double)(qr * qs / qsg) / (double)qms;
            fwg = (double)(qr * qg / qsg) / (double)qmg;
        }
        if (a.q_melt) {                               // parity access for the tests (debug mode only)
            a.q_melt[i] = qms;
            a.q_melt[n + i] = qmg;
            a.fw_melt[i] = fws;
            a.fw_melt[n + i] = fwg;
        }
    }
    const float e = in ? a.elev[i] : 0.f;
    // the loads of the loop below are issued one hydrometeor ahead (the mass density of species j + 1 while
    // species j is computed) and the temperature once: the kernel waits for memory, not for arithmetic
    const int var_t0 = hs.h[0].d.var_t;
    const float T0 = in ? a.vals[var_t0 * n + i] : 0.f;
    float q_ahead = (in && hs.h[0].d.q_source == CPOL_Q_MODEL) ? a.vals[hs.h[0].d.var_q * n + i] : 0.f;
    for (int j = 0; j < hs.n_hydro; ++j) {
        const HydroDev &h = hs.h[j];
        const cpol_hydro_desc &d = h.d;
        float qm = 0.f;
        double fw = 0.0;
        if (in) {
            if (d.q_source == CPOL_Q_MODEL) qm = q_ahead;
            else if (d.q_source == CPOL_Q_MELT_SNOW) { qm = qms; fw = fws; }
            else { qm = qmg; fw = fwg; }
        }
        if (in && j + 1 < hs.n_hydro && hs.h[j + 1].d.q_source == CPOL_Q_MODEL)
            q_ahead = a.vals[hs.h[j + 1].d.var_q * n + i];
        const ItabDev &tj = its.t[j];
        ClassItem it;
        classify_item(h, tj, a, a.vals, n, i, i, in, qm, fw, T0, var_t0, e, it);
        const bool valid = it.valid, lookup = it.lookup;
        const int key = it.key;
        if (valid) {
            double *P = a.par + ((long)j * CPOL_MAX_PAR) * n + i;
            my_lookup += lookup ? 1 : 0;
            if (!lookup || a.keep_par) {                     // read by the integrating kernels / the spectrum kernels
                P[0] = it.p0;
                P[n] = it.p1;
                if (h.n_par >= 3) P[2 * n] = it.p2;
                if (tj.tab) P[4 * n] = it.pf;                 // (debug reads; -1: not on the table)
            } else if (tj.two_d) {
                P[n] = it.p1;                                 // wet fraction: position inside the 2-D block
            }
            if (tj.tab) a.rec[(long)j * n + i] = make_double2(it.pf, it.scale);
            if (it.has_vn) {
                if (a.vn) *reinterpret_cast<double2 *>(a.vn + ((long)j * n + i) * 2) = make_double2(it.dv, it.dn);
                if (a.keep_par) { P[2 * n] = it.dv; P[3 * n] = it.dn; }
            }
            vbits |= 1u << j;
        }
        my_ranked += (valid && !lookup) ? 1 : 0;
        if (direct) {
            if (valid && !lookup) {
                const unsigned long long idx = atomicAdd(a.rare_totals + 1, 1ull);
                atomicAdd(a.rare_totals, 1ull);
                a.rare_key[idx] = key;
                a.rare_perm[idx] = (int)i;
            }
            if (valid || (in && a.keep_par)) a.key[(long)j * n + i] = key;
            continue;
        }
        const int ticket = rank_insert(sh[j], a.count, key, valid && !lookup);
        // (the LUT slice of a PRESENT hydrometeor; who is present is in vmask[] -- the -1 of the others is
        // only written for the debug reads and the spectrum kernels)
        if (valid || (in && a.keep_par)) a.key[(long)j * n + i] = key;
        if (valid && !lookup) {                       // (rare: pos[] is written only where an item is ranked)
            a.pos[(long)j * n + i] = ticket;
            ranked_mask |= 1u << j;
        }
    }
    if (in) a.vmask[i] = (unsigned char)vbits;
    // a.n_lookup[0]: items on integral tables; [1]: items ranked for the integrating kernels (0 in
    // the normal case: k_bucket_scan / k_bucket_scatter then return at once)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        my_lookup += __shfl_xor(my_lookup, off);
        my_ranked += __shfl_xor(my_ranked, off);
    }
    if (lane_id() == 0 && my_lookup) atomicAdd(&s_lookup, my_lookup);
    if (lane_id() == 0 && my_ranked) atomicAdd(&s_ranked, my_ranked);
    __syncthreads();
    if (threadIdx.x == 0) count_table_items(a.n_lookup, s_lookup);
    if (threadIdx.x == 0 && s_ranked) atomicAdd(a.n_lookup + 1, s_ranked);
    if (direct) return;
    if (threadIdx.x == 0) a.blk_ranked[blockIdx.x] = s_ranked;
    if (s_ranked == 0) return;                        // workgroup-uniform: every ticket is -1; pos[] stays unwritten
    for (int idx = threadIdx.x; idx < hs.n_hydro * CPOL_RANK_SLOTS; idx += blockDim.x)
        rank_reserve(sh[idx / CPOL_RANK_SLOTS], a.count, idx % CPOL_RANK_SLOTS);
    __syncthreads();
    if (in) {
        // this workgroup ranked something: k_bucket_scatter reads pos[] of all its gates
        for (int j = 0; j < hs.n_hydro; ++j) {
            const long o = (long)j * n + i;
            a.pos[o] = ((ranked_mask >> j) & 1u) ? rank_position(sh[j], a.pos[o]) : -1;
        }
    }
}

// ---------------------------------------------------------------- bucket scan
// single block: exclusive scans of the bucket counts (items and 64-item units)
struct ScanArgs {
    const int *count;
    int *offset;                // [n_keys] first item slot of the bucket
    int *uoffset;               // [n_keys] first work unit of the bucket
    WorkUnit *units;            // capacity >= n_items/64 + n_keys
    long long *totals;          // [0] = n_valid items, [1] = n_units
    int n_keys;
    int n_hydro;
    int key_base[CPOL_MAX_HYDRO];
    int unit_shift[CPOL_MAX_HYDRO];   // log2(items per work unit): 6, or 7 for the 2-items-per-lane flavour
};

// log2(items per work unit) of bucket k.  (Constant indices into the by-value argument arrays: a
// run-time index makes every access a dependent scalar load from the kernarg segment -- with the
// 32 x 2 unrolled call sites of k_bucket_scan that was 35 us of serialised s_load / s_waitcnt for a
// 15 k-bucket scan that otherwise takes 4.)
__device__ __forceinline__ int unit_shift_of(const ScanArgs &a, int k)
{
    int sh = a.unit_shift[0];
#pragma unroll
    for (int q = 1; q < CPOL_MAX_HYDRO; ++q)
        if (q < a.n_hydro && k >= a.key_base[q]) sh = a.unit_shift[q];
    return sh;
}

__device__ __forceinline__ int2 wave_inclusive_scan2(int2 v)
{
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < CPOL_WAVE; off <<= 1) {
        int a = __shfl_up(v.x, off), b = __shfl_up(v.y, off);
        if (lane >= off) { v.x += a; v.y += b; }
    }
    return v;
}

#define CPOL_SCAN_MAX_PER 32
// Single workgroup: exclusive scans of the bucket counts (items and work units).
__global__ __launch_bounds__(1024) void k_bucket_scan(ScanArgs a)
{
    __shared__ int2 s_wave[16];
    const int t = threadIdx.x, lane = lane_id(), wave = t >> 6;
    if (a.count[a.n_keys + 2] == 0) {                       // no item left to the integrating kernels
        if (t == 0) { a.totals[0] = 0; a.totals[1] = 0; a.totals[2] = 0; }
        return;
    }
    const int per = (a.n_keys + 1023) / 1024;               // <= CPOL_SCAN_MAX_PER (host checks)
    const int k0 = t * per;
    // the thread's counts in registers: the (up to 32) loads are independent and issue together
    int cnt[CPOL_SCAN_MAX_PER];
#pragma unroll
    for (int i = 0; i < CPOL_SCAN_MAX_PER; ++i) {
        const int k = k0 + i;
        cnt[i] = (i < per && k < a.n_keys) ? a.count[k] : 0;
    }
    int2 mine = make_int2(0, 0);
#pragma unroll
    for (int i = 0; i < CPOL_SCAN_MAX_PER; ++i) {
        const int k = k0 + i;
        if (i < per && k < a.n_keys) {
            const int sh = unit_shift_of(a, k);
            mine.x += cnt[i];
            mine.y += (cnt[i] + (1 << sh) - 1) >> sh;
        }
    }
    int2 inc = wave_inclusive_scan2(mine);
    if (lane == CPOL_WAVE - 1) s_wave[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        int2 w = (lane < 16) ? s_wave[lane] : make_int2(0, 0);
        int2 wi = wave_inclusive_scan2(w);
        if (lane < 16) s_wave[lane] = make_int2(wi.x - w.x, wi.y - w.y);   // exclusive
    }
    __syncthreads();
    int ibase = s_wave[wave].x + inc.x - mine.x;
    int ubase = s_wave[wave].y + inc.y - mine.y;
#pragma unroll
    for (int i = 0; i < CPOL_SCAN_MAX_PER; ++i) {
        const int k = k0 + i;
        if (i < per && k < a.n_keys) {
            const int sh = unit_shift_of(a, k);
            a.offset[k] = ibase;
            a.uoffset[k] = ubase;
            ibase += cnt[i];
            ubase += (cnt[i] + (1 << sh) - 1) >> sh;
        }
    }
    if (t == 1023) { a.totals[0] = ibase; a.totals[1] = ubase; a.totals[2] = 0; }   // [2]: ice fallback flag
}

// perm[offset[key] + pos] = sub-beam gate (no atomics: pos comes from k_classify), and -- the
// former k_make_units, folded in to save a launch -- the work-unit list: the wavefronts of the
// grid take the buckets round-robin, one wavefront writes the units of one bucket.
__global__ __launch_bounds__(256) void k_bucket_scatter(const int *__restrict__ key,
                                                         const int *__restrict__ pos,
                                                         int *__restrict__ perm,
                                                         const int *__restrict__ blk_ranked,
                                                         const unsigned char *__restrict__ vmask,
                                                         long n_sbg, int n_hydro, ScanArgs a)
{
    const int n_waves = gridDim.x * (blockDim.x >> 6);
    if (a.count[a.n_keys + 2] == 0) return;                 // grid-uniform: nothing was ranked
    for (int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); k < a.n_keys; k += n_waves) {
        const int c = a.count[k];
        if (c == 0) continue;                                   // wave-uniform
        const int ib = a.offset[k], ub = a.uoffset[k];
        const int sh = unit_shift_of(a, k), pu = 1 << sh;
        for (int u = lane_id(); u * pu < c; u += CPOL_WAVE) {
            WorkUnit w;
            w.key = k;
            w.start = ib + u * pu;
            w.count = min(pu, c - u * pu);
            w.pad = 0;
            a.units[ub + u] = w;
        }
    }
    // The workgroups walk the gate ranges of the k_classify workgroups (CPOL_CLASSIFY_THREADS gates each)
    // and skip those that ranked nothing -- the rule: a handful of items per volume lie outside the
    // integral tables, and neither key[] nor pos[] (unwritten there) of the other gates is read.
    const long n_cblk = (n_sbg + CPOL_CLASSIFY_THREADS - 1) / CPOL_CLASSIFY_THREADS;
    for (long cb = blockIdx.x; cb < n_cblk; cb += gridDim.x) {
        if (blk_ranked[cb] == 0) continue;                      // workgroup-uniform
        for (int q = threadIdx.x; q < CPOL_CLASSIFY_THREADS; q += blockDim.x) {
            const long sbg = cb * CPOL_CLASSIFY_THREADS + q;
            if (sbg >= n_sbg) break;
            const unsigned present = vmask[sbg];
            for (int j = 0; j < n_hydro; ++j) {
                if (!((present >> j) & 1u)) continue;
                const int k = key[(long)j * n_sbg + sbg];
                const int ps = pos[(long)j * n_sbg + sbg];      // < 0: the item went to an integral table
                if (ps >= 0) perm[a.offset[k] + ps] = (int)sbg;
            }
        }
    }
}

// ---------------------------------------------------------------- PSD x LUT
// Work decomposition: persistent 512-thread workgroups (8 wavefronts) walk the list of
// work units (up to 64 items of one LUT slice; 128 in the recurrence flavour, two per
// lane).  The item sits on the lane; wave w integrates diameter bins [w*n_d/8,
// (w+1)*n_d/8); the 8 partial sums per (item, column) are combined through LDS in a
// fixed order (deterministic).  All units of all hydrometeors are covered by ONE launch
// per kernel flavour: a workgroup skips the units of the other flavours.
#define CPOL_PSD_WAVES 8
#define CPOL_PSD_THREADS (CPOL_PSD_WAVES * CPOL_WAVE)
// waves per workgroup of the recurrence flavour (8 or 10).  Measured: 8 waves 139.7 us,
// 10 waves (5 waves per SIMD instead of 4) 166 us, 6 waves 204 us -- more waves in flight
// make every wave slower, so the default stays at 8
#ifndef CPOL_PSD_WAVES_U
#define CPOL_PSD_WAVES_U 8
#endif
#define CPOL_PSD_THREADS_U (CPOL_PSD_WAVES_U * CPOL_WAVE)

enum { PSD_MODE_GAMMA_EXP = 0, PSD_MODE_GAMMA_UNIFORM = 1, PSD_MODE_ICE = 2, PSD_MODE_MELTING = 3 };

struct PsdArgs {
    const int *unit_key;        // non-NULL: the work units are single items outside the integral tables, listed directly by
                                // k_classify / k_gate1 (unit u = {unit_key[u], start u, count 1}; `units` unused)
    const WorkUnit *units;
    const long long *totals;    // [1] = number of work units of this sweep; [2] != 0: k_psd_ice2 left
                                // units (lambda outside the tables) for k_psd<ICE>
    const int *perm;
    const double *par;          // [n_hydro][CPOL_MAX_PAR][n_sbg]
    double *res;                // [n_hydro][n_sbg][12]   sum_k sz[k][c] N[k] * dD
    double *vn;                 // [n_hydro][n_sbg][2]    integral of V N dD, of N dD (or NULL)
    double *par_w;              // = par, writable: 1-moment ice stores its normalised N0 in
                                // slot 3 for the Doppler-spectrum kernel (or NULL)
    long n_sbg;
    long long *clk;             // clock probe [2048][4]: s_memtime / s_memrealtime at the start and
                                // end of every workgroup (effective shader clock), or NULL
    int ice_force_sum;          // test hook (CPOL_ICE_FORCE_SUM=1): treat every lambda as outside the
                                // tabulated range, i.e. sum the ice normalisation integrals directly
    int ice_same_launch;        // k_psd_rare: the ice flavours run in ONE launch, so totals[2] (units k_psd_ice2 left to
                                // k_psd<ICE>, set by another workgroup) cannot be waited for: every unit is tested instead
};

// work unit u: from the sorted list of k_bucket_scatter, or one directly listed item
__device__ __forceinline__ void load_unit(const PsdArgs &a, int u, int &key, int &start, int &count)
{
    if (a.unit_key) {
        key = __builtin_amdgcn_readfirstlane(a.unit_key[u]);
        start = u;
        count = 1;
    } else {
        const WorkUnit *up = a.units + u;
        key = __builtin_amdgcn_readfirstlane(up->key);
        start = __builtin_amdgcn_readfirstlane(up->start);
        count = __builtin_amdgcn_readfirstlane(up->count);
    }
}

__device__ __forceinline__ int psd_mode_of(const cpol_hydro_desc &d)
{
    if (d.psd_family == CPOL_PSD_ICE_FIELD) return PSD_MODE_ICE;
    if (d.psd_family == CPOL_PSD_MELTING) return PSD_MODE_MELTING;
    return d.uniform_grid ? PSD_MODE_GAMMA_UNIFORM : PSD_MODE_GAMMA_EXP;
}

// x^y for x > 0 inside the per-bin loops of the ice / melting flavours: exp(y log x)
// is ~4x cheaper than the < 1 ulp ocml pow and accurate to a few 1e-16 here
// (|y log x| < 25), far inside the float32 store of the integrated sums.
__device__ __forceinline__ double powp(double x, double y) { return exp(y * log(x)); }

// 1-moment ice crystals (Field et al. 2005 double-moment normalised PSD)
__device__ __forceinline__ double phi23(double x)
{
    return 490.6 * exp(-20.78 * x) + 17.46 * powp(x, 0.6357) * exp(-3.290 * x);
}

// Melting snow / graupel (hydrometeors.py:333-478): per-item diameter grid.
struct MeltItem {
    double fw, fw2, phi;        // wet fraction, fw^2, Frick et al. velocity weight
    double lam_r;
};

__device__ __forceinline__ double powb(double x, double b)
{
    return (b == 2.0) ? x * x : powp(x, b);                 // NumPy: x**2 -> square
}

__device__ __forceinline__ double melt_mass(const cpol_hydro_desc &d, const MeltItem &it, double D,
                                            double D3)
{
    // hydrometeors.py:404-412 ; D3 = D**3 (shared with the volume)
    return it.fw2 * (d.r_a * D3) + (1.0 - it.fw2) * (d.a * powb(D, d.b));
}

__device__ __forceinline__ double melt_Dr(const cpol_hydro_desc &d, const MeltItem &it, double D)
{
    // hydrometeors.py:382-383
    const double D3 = D * D * D;
    const double rho = melt_mass(d, it, D, D3) / (3.14159265358979323846 / 6 * D3);
    return cbrt(rho / 1.0e-6) * D;                          // RHO_W = 1000/1000^3 kg mm-3
}

// Scattering tables and per-bin factors are immutable while a sweep runs: reading them
// through the CONSTANT address space guarantees the scalar data path (s_load -> SGPR
// operands) for wave-uniform addresses even when the kernel also stores to global memory.
typedef const double __attribute__((address_space(4))) *cdouble_p;
__device__ __forceinline__ cdouble_p as_const(const double *p) { return (cdouble_p)(uintptr_t)p; }

// 1-moment ice, lambda tables (hydrometeors.py::ice_norm_tables): position of lambda on the panel
// axis, and whether every item of a (up to 128-item) unit lies inside the tabulated range --
// evaluated identically by k_psd_ice2 (takes the unit if so) and k_psd<ICE> (takes it if not).
__device__ __forceinline__ double ice_panel_pos(const double *th, double lam)
{
    return (cp_log(lam) * 1.4426950408889634 - th[0]) * th[1];
}

__device__ __forceinline__ bool ice_unit_in_table(const HydroDev &h, const PsdArgs &a, int j, int start, int count)
{
    const double *th = h.aux + 12 * h.d.n_d + 8;
    bool ok = !a.ice_force_sum;
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
        const int sbg = a.perm[start + i];
        const double pf = ice_panel_pos(th, a.par[((long)j * CPOL_MAX_PAR) * a.n_sbg + sbg]);
        ok = ok && (pf >= 0.0 && pf < th[2]);                                  // NaN -> false
    }
    return __syncthreads_and(ok) != 0;
}

// LDS of the integrating kernels: one raw buffer per kernel, sized for the largest flavour, handed to the bodies (so
// that k_psd_rare can run every flavour from ONE launch)
#define CPOL_PSD_LDS_DOUBLES (8 * (CPOL_N_SZ + 3) * CPOL_WAVE)            /* 61 440 B: the generic flavours, 8 wave slots */
__device__ __forceinline__ double readlane_f64(double v, int src_uniform)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_uniform);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_uniform);
    return __hiloint2double(hi, lo);
}

template <int MODE, bool DOP2>
__device__ __forceinline__ void psd_body(const HydroSet &hs, const PsdArgs &a, double *lds_raw)
{
    // [wave][value][lane]; value 12 = normalisation sum (ice, melting)
    // the recurrence flavour needs no extra sums and combines its 8 partials with a
    // 3-level tree through 24 KB of LDS (4 wave slots): 8 waves per SIMD stay resident
    // ... and keeps TWO items per lane (128-item units): the wave-uniform table rows come
    // through the scalar cache, whose refill rate -- not the VALU -- bounds the kernel;
    // two items per fetched row halve that traffic
    constexpr int NV = (MODE == PSD_MODE_GAMMA_UNIFORM) ? 2 * CPOL_N_SZ + (DOP2 ? 4 : 0) : CPOL_N_SZ + 3;
    constexpr int NSLOT = (MODE == PSD_MODE_GAMMA_UNIFORM) ? 4 : CPOL_PSD_WAVES;
    static_assert(NSLOT * NV * CPOL_WAVE <= CPOL_PSD_LDS_DOUBLES, "LDS buffer of the integrating kernels too small");
    double (*const s_part)[NV][CPOL_WAVE] = reinterpret_cast<double (*)[NV][CPOL_WAVE]>(lds_raw);
#ifdef CPOL_CLOCK_PROBE
    // debug build only (make PROBE=1): the stores below cost the production kernel its
    // scalar loads of the unit list (the compiler can no longer prove them unclobbered)
    const bool probe = a.clk && MODE == PSD_MODE_GAMMA_UNIFORM && blockIdx.x < 2048 && threadIdx.x == 0;
    if (probe) {
        a.clk[blockIdx.x * 4 + 0] = (long long)__builtin_readcyclecounter();
        a.clk[blockIdx.x * 4 + 1] = (long long)wall_clock64();
    }
#endif
    // persistent workgroups: a fixed grid walks the unit list (no empty launches).  Units
    // are in slice order and consecutive workgroups land on different XCDs, so every XCD's
    // L2 sees every slice; an XCD-aware split (XCD x takes the x-th eighth of the units, its
    // L2 an eighth of the slices) was measured SLOWER in isolation (155-160 us vs 141 us):
    // all workgroups of an XCD then stream the same few slices at the same time.
    const int n_units = (int)a.totals[1];
    // (1-moment ice with lambda tables has 128-item units, taken by k_psd_ice2; a unit with a
    // lambda outside the tabulated range is summed here, as two 64-item halves: uu = 2 u + half)
    constexpr int SUB = (MODE == PSD_MODE_ICE) ? 2 : 1;
    for (int uu = blockIdx.x; uu < SUB * n_units; uu += gridDim.x) {
    int key, start, count;
    load_unit(a, uu / SUB, key, start, count);
    int j = 0;
    for (int q = 1; q < hs.n_hydro; ++q) if (key >= hs.h[q].key_base) j = q;
    const HydroDev &h = hs.h[j];
    const cpol_hydro_desc &d = h.d;
    if (psd_mode_of(d) != MODE) continue;                   // block-uniform
    if (MODE == PSD_MODE_MELTING && d.tab_degree == CPOL_MELT_DEGREE) continue;   // k_psd_melting_tab
    if (MODE == PSD_MODE_ICE) {
        if (d.tab_degree == CPOL_ICE_DEGREE) {
            if (!a.ice_same_launch && a.totals[2] == 0) continue;   // k_psd_ice2 took every unit (the rule)
            if (ice_unit_in_table(h, a, j, start, count)) continue;               // k_psd_ice2
            start += (uu % SUB) * CPOL_WAVE;
            count = min(CPOL_WAVE, count - (uu % SUB) * CPOL_WAVE);
            if (count <= 0) continue;
        } else if (uu % SUB) continue;
    }

    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n_d = d.n_d;
    constexpr int NW = (MODE == PSD_MODE_GAMMA_UNIFORM) ? CPOL_PSD_WAVES_U : CPOL_PSD_WAVES;
    const int chunk = (n_d + NW - 1) / NW;
    const int k0 = wave * chunk, k1 = min(k0 + chunk, n_d);
    const bool active = lane < count;
    const long n = a.n_sbg;
    const int sbg = a.perm[start + (active ? lane : 0)];
    const double *P = a.par + ((long)j * CPOL_MAX_PAR) * n + sbg;
    const cdouble_p slice = as_const(h.table + (long)(key - h.key_base) * n_d * CPOL_N_SZ);
    // Doppler scheme 2: (w rcs V, w rcs) per bin of this slice
    const cdouble_p rv = as_const(DOP2 ? h.rcsw + (long)(key - h.key_base) * n_d * 2 : h.table);

    double acc[CPOL_N_SZ];
#pragma unroll
    for (int c = 0; c < CPOL_N_SZ; ++c) acc[c] = 0.0;
    double msum = 0.0;       // normalisation integral (ice, melting)
    double vsum = 0.0, nsum = 0.0;   // fall-speed moments (numeric integrate_V: ice, melting)
    double scale = 1.0;      // applied by the combining thread

    if (MODE == PSD_MODE_GAMMA_EXP) {
        // N(D_k) = (N0 * pre[k]) * exp(-(lambda * dnu[k]))   (hydrometeors.py:143-147;
        // pre/dnu hold D^mu (x N0 for fixed intercepts) and D^nu exactly as NumPy
        // evaluates them on the float32 LUT diameter axis)
        const double lam = active ? P[0] : 0.0;
        const double N0 = active ? P[n] : 0.0;
        const cdouble_p pre = as_const(h.pre);
        const cdouble_p dnu = as_const(h.dnu);
        if (count == 1) {
            // ONE item in the unit (the rule for items listed directly, k_psd_rare): with the item on lane 0 the loop below
            // is a chain of 128 exponentials, one after the other.  The N(D_k) of the chunk do not depend on each other: 64 bins at a time, one per LANE,
            // then the sums in the order of the loop below -- the same values, the same order, the same bits.
            const double lam_u = readlane_f64(lam, 0), N0_u = readlane_f64(N0, 0);
            // lane q fetches the table row of bin kb + q (one round trip for 64 bins) and lays it down in LDS; then lane c < 12
            // sums column c, lanes 12 / 13 the two Doppler sums (scheme 2)
            constexpr int SW = CPOL_N_SZ + (DOP2 ? 2 : 0);               // doubles per staged bin
            static_assert(CPOL_PSD_WAVES * CPOL_WAVE * SW <= NSLOT * NV * CPOL_WAVE, "staging area of the single-item path: the LDS of k_psd<>");
            double *stage = lds_raw + (long)wave * (CPOL_WAVE * SW);
            const long sl0 = (long)(key - h.key_base) * n_d;
            const bool col = lane < CPOL_N_SZ + (DOP2 ? 2 : 0);
            double acc_l = 0.0;
            for (int kb = k0; kb < k1; kb += CPOL_WAVE) {
                const int kk = min(kb + lane, k1 - 1);
                const double nk_l = (N0_u * h.pre[kk]) * exp(-(lam_u * h.dnu[kk]));          // (lane q: bin kb + q)
                const double *rowv = h.table + (sl0 + kk) * CPOL_N_SZ;
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ; ++c) stage[lane * SW + c] = rowv[c];
                if (DOP2) { stage[lane * SW + CPOL_N_SZ] = h.rcsw[(sl0 + kk) * 2]; stage[lane * SW + CPOL_N_SZ + 1] = h.rcsw[(sl0 + kk) * 2 + 1]; }
                __builtin_amdgcn_wave_barrier();
                const int nq = min(CPOL_WAVE, k1 - kb);
#pragma unroll 8
                for (int q = 0; q < nq; ++q) {
                    const double nk = readlane_f64(nk_l, q);
                    const double x = col ? stage[q * SW + lane] : 0.0;
                    acc_l = fma(x, nk, acc_l);
                }
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();                                             // (the partial sums below take the same LDS)
#pragma unroll
            for (int c = 0; c < CPOL_N_SZ; ++c) acc[c] = readlane_f64(acc_l, c);
            if (DOP2) { vsum = readlane_f64(acc_l, CPOL_N_SZ); nsum = readlane_f64(acc_l, CPOL_N_SZ + 1); }
        } else {
#pragma unroll 2
        for (int k = k0; k < k1; ++k) {
            const double nk = (N0 * pre[k]) * exp(-(lam * dnu[k]));
            const cdouble_p row = slice + (long)k * CPOL_N_SZ;
#pragma unroll
            for (int c = 0; c < CPOL_N_SZ; ++c) acc[c] = fma(row[c], nk, acc[c]);
            if (DOP2) { vsum = fma(rv[2 * k], nk, vsum); nsum = fma(rv[2 * k + 1], nk, nsum); }
        }
        }
        if (!DOP2 && d.numeric_intv && a.vn) {
            // 2-moment ice: IceParticle.integrate_V on its own linspace grid
            const cdouble_p pn = as_const(h.aux), dn = as_const(h.aux + n_d);
            const cdouble_p Vn = as_const(h.aux + 2 * n_d);
#pragma unroll 1
            for (int k = k0; k < k1; ++k) {
                const double nk = (N0 * pn[k]) * exp(-(lam * dn[k]));
                vsum += nk * Vn[k];
                nsum += nk;
            }
        }
    } else if (MODE == PSD_MODE_GAMMA_UNIFORM) {
        // handled below (two items per lane)
    } else if (MODE == PSD_MODE_ICE) {
        //   aux[0..n_d)      D of the LUT axis (float64 of float32)
        //   aux[n_d..2n_d)   D of the normalisation grid  (hydrometeors.py:1331)
        //   aux[2n_d..3n_d)  a * D^b on the normalisation grid
        //   aux[3n_d]        dD of the normalisation grid
        const double lam = active ? P[0] : 1.0;
        const double N0 = active ? P[n] : 0.0;
        const cdouble_p Dl = as_const(h.aux);
        const cdouble_p Dn = as_const(h.aux + n_d);
        if (d.uniform_grid) {
            // PHI_23 on two uniform grids: both exponentials by geometric recurrence from an
            // exact exp at the chunk start, float32 node rounding restored to third order,
            // x^0.6357 = (lambda/1000)^0.6357 D^0.6357 (host table); see
            // hydrometeors.py::_ice_recurrence_aux.  msum carries no N0 (applied below).
            const cdouble_p hd = as_const(h.aux + 4 * n_d + 4);
            const cdouble_p ql = as_const(h.aux + 4 * n_d + 8);
            const cdouble_p qn = as_const(h.aux + 4 * n_d + 8 + 4 * n_d);
            const double lp = lam / 1000.0;
            const double a1 = 20.78 * lp, a2 = 3.290 * lp;
            const double B = 17.46 * powp(lp, 0.6357);
            const double a1s = a1 * a1, a1c = a1s * a1, a2s = a2 * a2, a2c = a2s * a2;
            const double dl0 = Dl[k0 < n_d ? k0 : 0], dn0 = Dn[k0 < n_d ? k0 : 0];
            double E1 = exp(-(a1 * dl0)), E2 = exp(-(a2 * dl0));
            double F1 = exp(-(a1 * dn0)), F2 = exp(-(a2 * dn0));
            const double r1 = exp(-(a1 * hd[0])), r2 = exp(-(a2 * hd[0]));
            const double s1 = exp(-(a1 * hd[1])), s2 = exp(-(a2 * hd[1]));
            if (count == 1) {
                // ONE item in the unit (an item listed directly, k_psd_rare: 68 of the 75 us of that kernel on a C4 volume were
                // one ice-crystal item walking its bins on one lane).  32 bins at a time: every lane walks the four geometric
                // recurrences (their roundings are part of the result) and lane q keeps the values of bin kb + q, forms the
                // bin's two PHI_23 factors with the statements of the loop below and lays the bin's table row, Doppler weights
                // and normalisation terms down in LDS; then lane c sums COLUMN c over the bins in the loop's order (0..11 the
                // table columns, 12 / 13 the sums of Doppler scheme 2, 14 the mass integral, 15 / 16 the fall-speed sums) --
                // the same values, the same order, the same bits; the sums go back to the item's lane at the end.
                constexpr int CH = 32, SW = CPOL_N_SZ + 5;
                static_assert(CPOL_PSD_WAVES * CH * SW <= NSLOT * NV * CPOL_WAVE, "staging area of the single-item path: the LDS of k_psd<ICE>");
                double *stage = lds_raw + (long)wave * (CH * SW);
                const long sl0 = (long)(key - h.key_base) * n_d;
                const double a1u = readlane_f64(a1, 0), a2u = readlane_f64(a2, 0), Bu = readlane_f64(B, 0);
                const double a1su = readlane_f64(a1s, 0), a1cu = readlane_f64(a1c, 0), a2su = readlane_f64(a2s, 0), a2cu = readlane_f64(a2c, 0);
                double E1u = readlane_f64(E1, 0), E2u = readlane_f64(E2, 0), F1u = readlane_f64(F1, 0), F2u = readlane_f64(F2, 0);
                const double r1u = readlane_f64(r1, 0), r2u = readlane_f64(r2, 0), s1u = readlane_f64(s1, 0), s2u = readlane_f64(s2, 0);
                const bool col = lane < SW;
                double acc_l = 0.0;
                for (int kb = k0; kb < k1; kb += CH) {
                    const int nq = min(CH, k1 - kb);
                    double E1_l = 0.0, E2_l = 0.0, F1_l = 0.0, F2_l = 0.0;
                    for (int q = 0; q < nq; ++q) {
                        if (lane == q) { E1_l = E1u; E2_l = E2u; F1_l = F1u; F2_l = F2u; }
                        F1u *= s1u;
                        F2u *= s2u;
                        E1u *= r1u;
                        E2u *= r2u;
                    }
                    double ph_l = 0.0, phn_l = 0.0;
                    if (lane < nq) {
                        const int k = kb + lane;
                        const double pwn = qn[4 * k], adb = qn[4 * k + 1], vk = qn[4 * k + 2];
                        const double x1 = a1u * qn[4 * k + 3], x2 = a2u * qn[4 * k + 3];
                        const double g1 = fma(x1, fma(x1, fma(x1, -1.0 / 6.0, 0.5), -1.0), 1.0);
                        const double g2 = fma(x2, fma(x2, fma(x2, -1.0 / 6.0, 0.5), -1.0), 1.0);
                        phn_l = fma(Bu * pwn, F2_l * g2, 490.6 * (F1_l * g1));
                        const double d1 = ql[4 * k], d2 = ql[4 * k + 1], d3 = ql[4 * k + 2], pw = ql[4 * k + 3];
                        const double c1 = fma(-a1u, d1, fma(a1su, d2, fma(-a1cu, d3, 1.0)));
                        const double c2 = fma(-a2u, d1, fma(a2su, d2, fma(-a2cu, d3, 1.0)));
                        ph_l = fma(Bu * pw, E2_l * c2, 490.6 * (E1_l * c1));
                        const double *rowv = h.table + (sl0 + k) * CPOL_N_SZ;
#pragma unroll
                        for (int c = 0; c < CPOL_N_SZ; ++c) stage[lane * SW + c] = rowv[c];
                        stage[lane * SW + CPOL_N_SZ] = DOP2 ? h.rcsw[(sl0 + k) * 2] : 0.0;
                        stage[lane * SW + CPOL_N_SZ + 1] = DOP2 ? h.rcsw[(sl0 + k) * 2 + 1] : 0.0;
                        stage[lane * SW + CPOL_N_SZ + 2] = adb;
                        stage[lane * SW + CPOL_N_SZ + 3] = vk;
                        stage[lane * SW + CPOL_N_SZ + 4] = 1.0;
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll 8
                    for (int q = 0; q < nq; ++q) {
                        const double ph = readlane_f64(ph_l, q), phn = readlane_f64(phn_l, q);
                        const double fct = lane < CPOL_N_SZ + 2 ? ph : phn;
                        const double x = col ? stage[q * SW + lane] : 0.0;
                        acc_l = fma(x, fct, acc_l);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                __syncthreads();                                         // (the partial sums below take the same LDS)
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ; ++c) acc[c] = readlane_f64(acc_l, c);
                msum = readlane_f64(acc_l, CPOL_N_SZ + 2);
                if (DOP2) { vsum = readlane_f64(acc_l, CPOL_N_SZ); nsum = readlane_f64(acc_l, CPOL_N_SZ + 1); }
                else { vsum = readlane_f64(acc_l, CPOL_N_SZ + 3); nsum = readlane_f64(acc_l, CPOL_N_SZ + 4); }
            } else
#pragma unroll 2
            for (int k = k0; k < k1; ++k) {
                const double pwn = qn[4 * k], adb = qn[4 * k + 1], vk = qn[4 * k + 2];
                const double x1 = a1 * qn[4 * k + 3], x2 = a2 * qn[4 * k + 3];
                const double g1 = fma(x1, fma(x1, fma(x1, -1.0 / 6.0, 0.5), -1.0), 1.0);
                const double g2 = fma(x2, fma(x2, fma(x2, -1.0 / 6.0, 0.5), -1.0), 1.0);
                const double phn = fma(B * pwn, F2 * g2, 490.6 * (F1 * g1));
                F1 *= s1;
                F2 *= s2;
                msum = fma(adb, phn, msum);
                if (!DOP2) {
                    vsum = fma(phn, vk, vsum);
                    nsum += phn;
                }
                const double d1 = ql[4 * k], d2 = ql[4 * k + 1], d3 = ql[4 * k + 2], pw = ql[4 * k + 3];
                const double c1 = fma(-a1, d1, fma(a1s, d2, fma(-a1c, d3, 1.0)));
                const double c2 = fma(-a2, d1, fma(a2s, d2, fma(-a2c, d3, 1.0)));
                const double ph = fma(B * pw, E2 * c2, 490.6 * (E1 * c1));
                E1 *= r1;
                E2 *= r2;
                const cdouble_p row = slice + (long)k * CPOL_N_SZ;
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ; ++c) acc[c] = fma(row[c], ph, acc[c]);
                if (DOP2) { vsum = fma(rv[2 * k], ph, vsum); nsum = fma(rv[2 * k + 1], ph, nsum); }
            }
            msum *= N0;
        } else {
        const cdouble_p aDb = as_const(h.aux + 2 * n_d);
        const cdouble_p Vn = as_const(h.aux + 3 * n_d + 1);   // alpha * Dn^beta (get_V)
#pragma unroll 1
        for (int k = k0; k < k1; ++k) {
            const double xn = lam * Dn[k] / 1000.0;
            const double phn = phi23(xn);
            msum += aDb[k] * (N0 * phn);                    // hydrometeors.py:1333-1337
            if (!DOP2) {
                vsum += phn * Vn[k];                        // hydrometeors.py:1267-1271
                nsum += phn;
            }
            const double xl = lam * Dl[k] / 1000.0;
            const double ph = phi23(xl);
            const cdouble_p row = slice + (long)k * CPOL_N_SZ;
#pragma unroll
            for (int c = 0; c < CPOL_N_SZ; ++c) acc[c] = fma(row[c], ph, acc[c]);
            if (DOP2) { vsum = fma(rv[2 * k], ph, vsum); nsum = fma(rv[2 * k + 1], ph, nsum); }
        }
        }
    } else {
        MeltItem it;
        it.fw = active ? P[n] : 0.5;
        it.lam_r = active ? P[2 * n] : 1.0;
        it.fw2 = it.fw * it.fw;
        it.phi = 0.246 * it.fw + (1 - 0.246) * pow(it.fw, 7.0);
        // hydrometeors.py:336-339 and utilities.py:158-173 (vlinspace)
        const double d_max = it.fw * d.r_dmax + (1 - it.fw) * d.s_dmax;
        const double d_min = it.fw * d.r_dmin + (1 - it.fw) * d.s_dmin;
        const double step = (d_max - d_min) / (double)(n_d - 1);
#pragma unroll 2
        for (int k = k0; k < k1; ++k) {
            const double D = d_min + step * (double)k;
            const double D2 = D + 0.01;                                         // :384
            // mass of the melting particle at D and D + 0.01 (:404-412); D^b, D^beta share
            // one logarithm when they are not plain squares / fourth roots
            double Db, D2b, Dbeta;
            if (d.b == 2.0 && d.beta == 0.25) {
                Db = D * D; D2b = D2 * D2; Dbeta = cp_fourth_root(D);
            } else {
                const double L = cp_log(D);
                Db = (d.b == 2.0) ? D * D : cp_exp(d.b * L);
                D2b = (d.b == 2.0) ? D2 * D2 : cp_exp(d.b * cp_log(D2));
                Dbeta = cp_exp(d.beta * L);
            }
            const double M = it.fw2 * (d.r_a * (D * D * D)) + (1.0 - it.fw2) * (d.a * Db);
            const double M2 = it.fw2 * (d.r_a * (D2 * D2 * D2)) + (1.0 - it.fw2) * (d.a * D2b);
            // D_r = (rho_m / rho_w)^(1/3) D with rho_m = M / (pi/6 D^3)  ==  cbrt(6e6/pi M)
            // (:382-383; the D^3 cancels exactly, rounding differs by ~2 ulp)
            const double c6 = 6.0e6 / 3.14159265358979323846;
            double Dr, sq;
            cp_cbrt_and_sixth(c6 * M, Dr, sq);                                 // D_r, sqrt(D_r)
            const double dDr = (cp_cbrt(c6 * M2) - Dr) * 100.0;                // /0.01, :384
            const double Nr = (d.r_n0 * sq) * cp_exp(-(it.lam_r * Dr));        // rain N(D_r)
            const double Vr = d.r_alpha * sq;                                  // rain V(D_r)
            const double Vd = d.alpha * Dbeta;
            const double V = it.phi * Vr + (1 - it.phi) * Vd;                  // :431-439
            const double Nraw = (Nr * Vr) * cp_rcp(V) * dDr;                   // :386-387
            msum += Nraw * M;                                                  // :478
            if (DOP2) {                        // rcs-weighted, per-item fall speed V(D)
                const double wr = rv[2 * k + 1];
                vsum = fma(wr * V, Nraw, vsum);
                nsum = fma(wr, Nraw, nsum);
            } else {
                vsum += Nraw * V;                                              // :457
                nsum += Nraw;                                                  // :458
            }
            const cdouble_p row = slice + (long)k * CPOL_N_SZ;
#pragma unroll
            for (int c = 0; c < CPOL_N_SZ; ++c) acc[c] = fma(row[c], Nraw, acc[c]);
        }
    }

    // ---- combine the 8 partial sums per (item, column), fixed order ----
    if (MODE == PSD_MODE_GAMMA_UNIFORM) {
        // nu == 1 on a (nearly) uniform diameter grid: exp(-lambda D_k) follows from
        // ONE exp per wave chunk and a geometric recurrence; the float32 rounding of
        // the grid nodes is restored to second order:
        //   D_k = D_k0 + (k-k0) h + dlt_k,  exp(-l D_k) = E_k0 r^(k-k0) (1 - x + x^2/2),
        //   x = l dlt_k, |x| < 2e-5  ->  truncation < 2e-15, drift < 128 ulp(double).
        const bool act1 = lane + CPOL_WAVE < count;
        const int sbg1 = a.perm[start + (act1 ? lane + CPOL_WAVE : 0)];
        const double *P1 = a.par + ((long)j * CPOL_MAX_PAR) * n + sbg1;
        const double lam0 = active ? P[0] : 0.0, lam1 = act1 ? P1[0] : 0.0;
        const double N00 = active ? P[n] : 0.0, N01 = act1 ? P1[n] : 0.0;
        const cdouble_p dnu = as_const(h.dnu);
        const double hstep = h.aux[0];
        const double d0 = dnu[k0 < n_d ? k0 : 0];
        double A0 = N00 * exp(-(lam0 * d0)), A1 = N01 * exp(-(lam1 * d0));
        const double r0 = exp(-(lam0 * hstep)), r1 = exp(-(lam1 * hstep));
        double acc1[CPOL_N_SZ];
#pragma unroll
        for (int c = 0; c < CPOL_N_SZ; ++c) acc1[c] = 0.0;
        double dv0 = 0.0, dn0 = 0.0, dv1 = 0.0, dn1 = 0.0;     // Doppler scheme 2 sums
        // per bin the host supplies (pre, q1 = pre*dlt, q2 = pre*dlt^2/2), so that
        //   N_k = A_k (pre_k - lambda q1_k + lambda^2 q2_k),  A_{k+1} = A_k r
        // costs 4 f64 ops per item on top of the 12 FMAs (one 32-byte scalar request)
        // NB the per-bin triple stays in its own small array: it is shared by every workgroup
        // and hits in the scalar cache, whereas table rows are streamed once each -- folding
        // it into 128-byte padded rows (two aligned 64-byte scalar loads per bin instead of
        // three unaligned ones, plus a third-order term) measured 142 us vs 139.7 us: neither
        // the number nor the alignment of the scalar requests bounds the kernel.  Also
        // measured and rejected: prefetching the next unit's slice into L2 with dummy vector
        // loads (160 us); staggering half of the waves by a few hundred cycles (no change).
        const cdouble_p pq = as_const(h.aux + 1 + n_d);          // [k][4] = (pre, q1, q2, .)
        if (count == 1) {
            // ONE item in the unit (the rule for items listed directly, k_psd_rare): the loop below then waits for one
            // scalar-cache round trip per bin (tools/rare_trace.py: ~60 of the 68 us k_psd_rare took for the single snow item
            // outside its table in a C4 volume share; 38 this way).
            // 64 bins at a time: lane q forms the polynomial factor of bin kb + q; then lane c sums COLUMN c over the bins in the
            // order of the loop below (the bin's factor by readlane, its own table value by a load that waits for nothing) --
            // the same values, the same order, the same bits; the sums go back to the item's lane at the end.
            const double lam_u = readlane_f64(lam0, 0), r_u = readlane_f64(r0, 0);
            double A_u = readlane_f64(A0, 0);
            // lane q fetches the table row of bin kb + q (one round trip for 64 bins) and lays it down in LDS; then lane c < 12
            // sums column c, lanes 12 / 13 the two Doppler sums (scheme 2)
            constexpr int SW = CPOL_N_SZ + (DOP2 ? 2 : 0);               // doubles per staged bin
            static_assert(CPOL_PSD_WAVES_U * CPOL_WAVE * SW <= NSLOT * NV * CPOL_WAVE, "staging area of the single-item path: the LDS of k_psd_uniform");
            double *stage = lds_raw + (long)wave * (CPOL_WAVE * SW);
            const long sl0 = (long)(key - h.key_base) * n_d;
            const bool col = lane < CPOL_N_SZ + (DOP2 ? 2 : 0);
            double acc_l = 0.0;
            for (int kb = k0; kb < k1; kb += CPOL_WAVE) {
                const int kk = min(kb + lane, k1 - 1);
                const double *pqv = h.aux + 1 + n_d + 4 * (long)kk;
                const double poly_l = fma(lam_u, fma(lam_u, pqv[2], -pqv[1]), pqv[0]);      // (lane q: bin kb + q)
                const double *rowv = h.table + (sl0 + kk) * CPOL_N_SZ;
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ; ++c) stage[lane * SW + c] = rowv[c];
                if (DOP2) { stage[lane * SW + CPOL_N_SZ] = h.rcsw[(sl0 + kk) * 2]; stage[lane * SW + CPOL_N_SZ + 1] = h.rcsw[(sl0 + kk) * 2 + 1]; }
                __builtin_amdgcn_wave_barrier();
                const int nq = min(CPOL_WAVE, k1 - kb);
#pragma unroll 8
                for (int q = 0; q < nq; ++q) {
                    const double n0 = A_u * readlane_f64(poly_l, q);
                    A_u *= r_u;
                    const double x = col ? stage[q * SW + lane] : 0.0;
                    acc_l = fma(x, n0, acc_l);
                }
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();                                             // (the partial sums below take the same LDS)
#pragma unroll
            for (int c = 0; c < CPOL_N_SZ; ++c) acc[c] = readlane_f64(acc_l, c);
            if (DOP2) { dv0 = readlane_f64(acc_l, CPOL_N_SZ); dn0 = readlane_f64(acc_l, CPOL_N_SZ + 1); }
        } else
#pragma unroll 2
        for (int k = k0; k < k1; ++k) {
            const double pk = pq[4 * k], q1 = pq[4 * k + 1], q2 = pq[4 * k + 2];
            const double n0 = A0 * fma(lam0, fma(lam0, q2, -q1), pk);
            const double n1 = A1 * fma(lam1, fma(lam1, q2, -q1), pk);
            A0 *= r0;
            A1 *= r1;
            const cdouble_p row = slice + (long)k * CPOL_N_SZ;
#pragma unroll
            for (int c = 0; c < CPOL_N_SZ; ++c) {
                const double sz = row[c];
                acc[c] = fma(sz, n0, acc[c]);
                acc1[c] = fma(sz, n1, acc1[c]);
            }
            if (DOP2) {
                const double wv = rv[2 * k], wr = rv[2 * k + 1];
                dv0 = fma(wv, n0, dv0); dn0 = fma(wr, n0, dn0);
                dv1 = fma(wv, n1, dv1); dn1 = fma(wr, n1, dn1);
            }
        }
        // waves beyond the eighth fold into waves 0.. first, then the tree
        // ((w0+w4)+(w2+w6)) + ((w1+w5)+(w3+w7)) through 4 LDS wave slots; the
        // wave that ends up with the total (wave 0) writes the results
        if (NW > 8) {
            if (wave >= 8) {
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ; ++c) {
                    s_part[wave - 8][c][lane] = acc[c];
                    s_part[wave - 8][CPOL_N_SZ + c][lane] = acc1[c];
                }
                if (DOP2) {
                    s_part[wave - 8][NV - 4][lane] = dv0; s_part[wave - 8][NV - 3][lane] = dn0;
                    s_part[wave - 8][NV - 2][lane] = dv1; s_part[wave - 8][NV - 1][lane] = dn1;
                }
            }
            __syncthreads();
            if (wave < NW - 8) {
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ; ++c) {
                    acc[c] += s_part[wave][c][lane];
                    acc1[c] += s_part[wave][CPOL_N_SZ + c][lane];
                }
                if (DOP2) {
                    dv0 += s_part[wave][NV - 4][lane]; dn0 += s_part[wave][NV - 3][lane];
                    dv1 += s_part[wave][NV - 2][lane]; dn1 += s_part[wave][NV - 1][lane];
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int half = 4; half >= 1; half >>= 1) {
            if (wave >= half && wave < 2 * half) {
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ; ++c) {
                    s_part[wave - half][c][lane] = acc[c];
                    s_part[wave - half][CPOL_N_SZ + c][lane] = acc1[c];
                }
                if (DOP2) {
                    s_part[wave - half][NV - 4][lane] = dv0; s_part[wave - half][NV - 3][lane] = dn0;
                    s_part[wave - half][NV - 2][lane] = dv1; s_part[wave - half][NV - 1][lane] = dn1;
                }
            }
            __syncthreads();
            if (wave < half) {
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ; ++c) {
                    acc[c] += s_part[wave][c][lane];
                    acc1[c] += s_part[wave][CPOL_N_SZ + c][lane];
                }
                if (DOP2) {
                    dv0 += s_part[wave][NV - 4][lane]; dn0 += s_part[wave][NV - 3][lane];
                    dv1 += s_part[wave][NV - 2][lane]; dn1 += s_part[wave][NV - 1][lane];
                }
            }
            __syncthreads();
        }
        if (wave == 0) {
            if (active) {
                double *o = a.res + ((long)j * n + sbg) * CPOL_N_SZ;
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ; ++c) o[c] = acc[c] * d.dD;
                if (DOP2) { a.vn[((long)j * n + sbg) * 2] = dv0; a.vn[((long)j * n + sbg) * 2 + 1] = dn0; }
            }
            if (act1) {
                double *o = a.res + ((long)j * n + sbg1) * CPOL_N_SZ;
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ; ++c) o[c] = acc1[c] * d.dD;
                if (DOP2) { a.vn[((long)j * n + sbg1) * 2] = dv1; a.vn[((long)j * n + sbg1) * 2 + 1] = dn1; }
            }
        }
        continue;
    }
#pragma unroll
    for (int c = 0; c < CPOL_N_SZ; ++c) s_part[wave][c][lane] = acc[c];
    if (MODE != PSD_MODE_GAMMA_UNIFORM) {
        s_part[wave][NV - 3][lane] = msum;
        s_part[wave][NV - 2][lane] = vsum;
        s_part[wave][NV - 1][lane] = nsum;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < CPOL_N_SZ * CPOL_WAVE; idx += CPOL_PSD_THREADS) {
        const int c = idx >> 6, l = idx & 63;
        if (l >= count) continue;
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < CPOL_PSD_WAVES; ++w) sum += s_part[w][c][l];
        const int sb = a.perm[start + l];
        const double *Pl = a.par + ((long)j * CPOL_MAX_PAR) * n + sb;
        if (MODE == PSD_MODE_ICE) {
            double m = 0.0;
#pragma unroll
            for (int w = 0; w < CPOL_PSD_WAVES; ++w) m += s_part[w][NV - 3][l];
            const double qm_est = m * h.aux[3 * n_d];
            scale = Pl[n] / qm_est * Pl[2 * n];             // N0 / QM_est * QM (:1339)
            if (a.par_w && c == 0) a.par_w[((long)j * CPOL_MAX_PAR + 3) * n + sb] = scale;
            sum = (scale * sum) * d.dD;
            if (a.vn && c < 2) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < CPOL_PSD_WAVES; ++w) t += s_part[w][NV - 2 + c][l];
                // scheme 1: integrate_V (x dD of its grid); scheme 2: unit-spaced trapezoid
                a.vn[((long)j * n + sb) * 2 + c] = DOP2 ? scale * t : (scale * t) * h.aux[3 * n_d];
            }
        } else if (MODE == PSD_MODE_MELTING) {
            double m = 0.0;
#pragma unroll
            for (int w = 0; w < CPOL_PSD_WAVES; ++w) m += s_part[w][NV - 3][l];
            // the item's own grid step (same expression as above, for lane l)
            const double fw = Pl[n];
            const double d_max = fw * d.r_dmax + (1 - fw) * d.s_dmax;
            const double d_min = fw * d.r_dmin + (1 - fw) * d.s_dmin;
            const double step = (d_max - d_min) / (double)(n_d - 1);
            const double dDl = (d_min + step * 1.0) - (d_min + step * 0.0);
            const double prop = Pl[0] / (m * dDl);          // q / integrate_M (:1428)
            sum = (prop * sum) * dDl;
            if (a.vn && c < 2) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < CPOL_PSD_WAVES; ++w) t += s_part[w][NV - 2 + c][l];
                a.vn[((long)j * n + sb) * 2 + c] = DOP2 ? prop * t : (prop * t) * dDl;
            }
        } else {
            sum = sum * d.dD;
            if (MODE == PSD_MODE_GAMMA_EXP && (DOP2 || d.numeric_intv) && a.vn && c < 2) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < CPOL_PSD_WAVES; ++w) t += s_part[w][NV - 2 + c][l];
                a.vn[((long)j * n + sb) * 2 + c] = DOP2 ? t : t * h.aux[3 * n_d];
            }
        }
        a.res[((long)j * n + sb) * CPOL_N_SZ + c] = sum;
    }
    __syncthreads();                                        // s_part is reused by the next unit
    }
#ifdef CPOL_CLOCK_PROBE
    if (probe) {
        a.clk[blockIdx.x * 4 + 2] = (long long)__builtin_readcyclecounter();
        a.clk[blockIdx.x * 4 + 3] = (long long)wall_clock64();
    }
#endif
}

template <int MODE, bool DOP2>
__global__ __launch_bounds__(CPOL_PSD_THREADS) void k_psd(HydroSet hs, PsdArgs a)
{
    constexpr int NVK = CPOL_N_SZ + 3;
    __shared__ double lds_raw[CPOL_PSD_WAVES * NVK * CPOL_WAVE];
    psd_body<MODE, DOP2>(hs, a, lds_raw);
}

// The recurrence flavour (the dominant kernel of a sweep).  The compiler's own allocation
// is 80 VGPRs, i.e. up to three resident 8-wave workgroups per CU (LDS 3 x 48 KB): 140 us on
// the bench sweep.  Every attempt to steer it measured slower in isolation:
// amdgpu_waves_per_eu(6,6) 190-220 us, amdgpu_waves_per_eu(4,5) 178 us, 8 KB of dynamic LDS
// (two workgroups per CU) 175 us.  Left alone on purpose.
template <bool DOP2>
__global__ __launch_bounds__(CPOL_PSD_THREADS_U) void k_psd_uniform(HydroSet hs, PsdArgs a)
{
    constexpr int NVU = 2 * CPOL_N_SZ + (DOP2 ? 4 : 0);
    __shared__ double lds_raw[4 * NVU * CPOL_WAVE];
    psd_body<PSD_MODE_GAMMA_UNIFORM, DOP2>(hs, a, lds_raw);
}

// The melting flavour (own kernel symbol so that its register allocation is reported and can be
// steered separately; see csrc/resource_usage.txt after `make asm`).
template <bool DOP2>
__global__ __launch_bounds__(CPOL_PSD_THREADS) void k_psd_melting(HydroSet hs, PsdArgs a)
{
    constexpr int NVK = CPOL_N_SZ + 3;
    __shared__ double lds_raw[CPOL_PSD_WAVES * NVK * CPOL_WAVE];
    psd_body<PSD_MODE_MELTING, DOP2>(hs, a, lds_raw);
}


// ---------------------------------------------------------------- melting flavour, table driven
// N(D_k) = G(fw, k) exp(-lambda_r D_r(fw, k)): the factors of the melting-particle PSD that depend
// on the wet fraction only -- two cube roots, a sixth / fourth root, two powers for graupel and
// a division per (item, bin): 214 VALU instructions per (item row, bin) in the direct form above
// (SQ_INSTS_VALU, C4 sweep) -- come from degree-10 polynomials in fw, tabulated per (wet-fraction
// bin of the unit's LUT slice, diameter bin) at staging time in extended precision
// (hydrometeors.py::melting_fw_tables, < 2e-15 relative); the rain slope lambda_r stays per item.
// Per bin the wave needs 44 coefficients + 12 table-row entries = 448 bytes of wave-uniform
// operands through the scalar path: measured with ONE item per lane the scalar data cache is
// 99 % busy (SQC_DCACHE_BUSY_CYCLES) and the kernel no faster than the direct form (10.1 vs
// 10.9 ms), whether the 8 waves split the bins or walk them together; with TWO items per lane
// (128-item units, as the recurrence flavour) every fetched operand feeds two items.
// Same work decomposition as k_psd_uniform: bins split over the 8 waves, fixed-order tree
// combine through 4 LDS wave slots.
// Experiment kept as a knob (make EXTRA=-DCPOL_PSD_L2_TOUCH=1): touch the cache lines of a wave's
// operands with vector loads before its bin loop so that the scalar loads hit L2.  Measured
// SLOWER on both ends: PSD stage of the C3 sweep 741 vs 660 us, of the C4 sweep 20.2 vs 18.5 ms.
#ifndef CPOL_PSD_L2_TOUCH
#define CPOL_PSD_L2_TOUCH 0
#endif
__device__ __forceinline__ int l2_touch(const void *p, long bytes, int lane)
{
    int t = 0;
#if CPOL_PSD_L2_TOUCH
    const char *b = (const char *)p;
#pragma unroll 4
    for (long off = (long)lane * 128; off < bytes; off += 64 * 128)
        t ^= __builtin_nontemporal_load((const int *)(b + off));
#endif
    return t;
}
__device__ __forceinline__ void psd_keep(int v) { asm volatile("" :: "v"(v)); }

#define CPOL_MELT_NV (2 * (CPOL_N_SZ + 3))
template <bool DOP2>
__device__ __forceinline__ void melt_store(const cpol_hydro_desc &d, const PsdArgs &a, int j, long n, int sb,
                                           double fw, double q, const double (&acc)[CPOL_N_SZ], double m,
                                           double v, double s)
{
    // the item's own grid step (hydrometeors.py:336-339, utilities.py:158-173)
    const double d_max = fw * d.r_dmax + (1 - fw) * d.s_dmax;
    const double d_min = fw * d.r_dmin + (1 - fw) * d.s_dmin;
    const double step = (d_max - d_min) / (double)(d.n_d - 1);
    const double dDl = (d_min + step * 1.0) - (d_min + step * 0.0);
    const double prop = q / (m * dDl);                              // q / integrate_M (:1428)
    double *o = a.res + ((long)j * n + sb) * CPOL_N_SZ;
#pragma unroll
    for (int c = 0; c < CPOL_N_SZ; ++c) o[c] = (prop * acc[c]) * dDl;
    if (a.vn) {
        a.vn[((long)j * n + sb) * 2] = DOP2 ? prop * v : (prop * v) * dDl;
        a.vn[((long)j * n + sb) * 2 + 1] = DOP2 ? prop * s : (prop * s) * dDl;
    }
}

template <bool DOP2>
__device__ __forceinline__ void psd_melting_tab_body(const HydroSet &hs, const PsdArgs &a, double *lds_raw)
{
    constexpr int NC = CPOL_MELT_DEGREE + 1;
    constexpr int NV = CPOL_MELT_NV;
    static_assert(4 * NV * CPOL_WAVE <= CPOL_PSD_LDS_DOUBLES, "LDS buffer of the integrating kernels too small");
    double (*const s_part)[NV][CPOL_WAVE] = reinterpret_cast<double (*)[NV][CPOL_WAVE]>(lds_raw);
    const int n_units = (int)a.totals[1];
    for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
        int key, start, count;
        load_unit(a, u, key, start, count);
        int j = 0;
        for (int q = 1; q < hs.n_hydro; ++q) if (key >= hs.h[q].key_base) j = q;
        const HydroDev &h = hs.h[j];
        const cpol_hydro_desc &d = h.d;
        if (d.psd_family != CPOL_PSD_MELTING || d.tab_degree != CPOL_MELT_DEGREE) continue;   // block-uniform

        const int lane = lane_id();
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const int n_d = d.n_d;
        const int chunk = (n_d + CPOL_PSD_WAVES - 1) / CPOL_PSD_WAVES;
        const int k0 = wave * chunk, k1 = min(k0 + chunk, n_d);
        const bool act0 = lane < count, act1 = lane + CPOL_WAVE < count;
        const long n = a.n_sbg;
        const int sbg0 = a.perm[start + (act0 ? lane : 0)];
        const int sbg1 = a.perm[start + (act1 ? lane + CPOL_WAVE : 0)];
        const double *P0 = a.par + ((long)j * CPOL_MAX_PAR) * n + sbg0;
        const double *P1 = a.par + ((long)j * CPOL_MAX_PAR) * n + sbg1;
        const int tb = (key - h.key_base) % d.n_t;                  // wet-fraction bin of the slice
        const cdouble_p head = as_const(h.aux);
        const double mid = head[2 * tb], inv_half = head[2 * tb + 1];
        const double fw0 = act0 ? P0[n] : mid, fw1 = act1 ? P1[n] : mid;
        const double lam0 = act0 ? P0[2 * n] : 1.0, lam1 = act1 ? P1[2 * n] : 1.0;
        const double u0 = (fw0 - mid) * inv_half, u1 = (fw1 - mid) * inv_half;
        const cdouble_p slice = as_const(h.table + (long)(key - h.key_base) * n_d * CPOL_N_SZ);
        const cdouble_p rv = as_const(DOP2 ? h.rcsw + (long)(key - h.key_base) * n_d * 2 : h.table);
        const cdouble_p co = as_const(h.aux + 2 * d.n_t + (long)tb * n_d * (CPOL_MELT_FUNCS * NC));

        double acc0[CPOL_N_SZ], acc1[CPOL_N_SZ];
#pragma unroll
        for (int c = 0; c < CPOL_N_SZ; ++c) { acc0[c] = 0.0; acc1[c] = 0.0; }
        double m0 = 0.0, v0 = 0.0, s0 = 0.0, m1 = 0.0, v1 = 0.0, s1 = 0.0;
        const int touched = l2_touch(h.aux + 2 * d.n_t + ((long)tb * n_d + k0) * (CPOL_MELT_FUNCS * NC),
                                     (long)(k1 - k0) * (CPOL_MELT_FUNCS * NC) * 8, lane)
                          ^ l2_touch(h.table + ((long)(key - h.key_base) * n_d + k0) * CPOL_N_SZ,
                                     (long)(k1 - k0) * CPOL_N_SZ * 8, lane);
#pragma unroll 1
        for (int k = k0; k < k1; ++k) {
            const cdouble_p c = co + (long)k * (CPOL_MELT_FUNCS * NC);
            double Dr0 = c[NC - 1], G0 = c[2 * NC - 1], GM0 = c[3 * NC - 1], GV0 = c[4 * NC - 1];
            double Dr1 = Dr0, G1 = G0, GM1 = GM0, GV1 = GV0;
#pragma unroll
            for (int q = NC - 2; q >= 0; --q) {                     // Horner, coefficient = scalar addend
                Dr0 = fma_sc(Dr0, u0, c[q]);           Dr1 = fma_sc(Dr1, u1, c[q]);
                G0 = fma_sc(G0, u0, c[NC + q]);        G1 = fma_sc(G1, u1, c[NC + q]);
                GM0 = fma_sc(GM0, u0, c[2 * NC + q]);  GM1 = fma_sc(GM1, u1, c[2 * NC + q]);
                GV0 = fma_sc(GV0, u0, c[3 * NC + q]);  GV1 = fma_sc(GV1, u1, c[3 * NC + q]);
            }
            const double E0 = cp_exp(-(lam0 * Dr0)), E1 = cp_exp(-(lam1 * Dr1));
            const double N0 = G0 * E0, N1 = G1 * E1;
            m0 = fma(GM0, E0, m0);                                                  // :478
            m1 = fma(GM1, E1, m1);
            if (DOP2) {
                const double wr = rv[2 * k + 1];
                v0 = fma(wr * GV0, E0, v0); s0 = fma(wr, N0, s0);
                v1 = fma(wr * GV1, E1, v1); s1 = fma(wr, N1, s1);
            } else {
                v0 = fma(GV0, E0, v0); s0 += N0;                                    // :457-458
                v1 = fma(GV1, E1, v1); s1 += N1;
            }
            const cdouble_p row = slice + (long)k * CPOL_N_SZ;
#pragma unroll
            for (int cc = 0; cc < CPOL_N_SZ; ++cc) {
                const double sz = row[cc];
                acc0[cc] = fma(sz, N0, acc0[cc]);
                acc1[cc] = fma(sz, N1, acc1[cc]);
            }
        }
        psd_keep(touched);
        // ((w0+w4)+(w2+w6)) + ((w1+w5)+(w3+w7)): fixed order, wave 0 ends up with the totals
#pragma unroll
        for (int half = 4; half >= 1; half >>= 1) {
            if (wave >= half && wave < 2 * half) {
                double (*sp)[CPOL_WAVE] = s_part[wave - half];
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ; ++c) { sp[c][lane] = acc0[c]; sp[CPOL_N_SZ + 3 + c][lane] = acc1[c]; }
                sp[CPOL_N_SZ][lane] = m0; sp[CPOL_N_SZ + 1][lane] = v0; sp[CPOL_N_SZ + 2][lane] = s0;
                sp[NV - 3][lane] = m1; sp[NV - 2][lane] = v1; sp[NV - 1][lane] = s1;
            }
            __syncthreads();
            if (wave < half) {
                double (*sp)[CPOL_WAVE] = s_part[wave];
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ; ++c) { acc0[c] += sp[c][lane]; acc1[c] += sp[CPOL_N_SZ + 3 + c][lane]; }
                m0 += sp[CPOL_N_SZ][lane]; v0 += sp[CPOL_N_SZ + 1][lane]; s0 += sp[CPOL_N_SZ + 2][lane];
                m1 += sp[NV - 3][lane]; v1 += sp[NV - 2][lane]; s1 += sp[NV - 1][lane];
            }
            __syncthreads();
        }
        if (wave == 0) {
            if (act0) melt_store<DOP2>(d, a, j, n, sbg0, fw0, P0[0], acc0, m0, v0, s0);
            if (act1) melt_store<DOP2>(d, a, j, n, sbg1, fw1, P1[0], acc1, m1, v1, s1);
        }
    }
}

// ---------------------------------------------------------------- 1-moment ice, two items per lane
// The Field et al. (2005) PSD on the LUT diameter axis by two geometric recurrences (as the ICE
// branch of psd_body), with the three normalisation sums of the item -- functions of its lambda
// only -- from degree-10 polynomials on 1/8-octave panels of lambda (ice_norm_tables): 25 f64
// operations per (item, bin) instead of 43.  With one item per lane that left the kernel bound by
// the scalar data cache (128 bytes of wave-uniform operands per 100 VALU cycles); two items per
// lane halve that, as in the recurrence and melting flavours.  Same decomposition: 128-item
// units, bins split over 8 waves, fixed-order tree combine.
#define CPOL_ICE_NV (2 * (CPOL_N_SZ + 2))
template <bool DOP2>
__device__ __forceinline__ void psd_ice2_body(const HydroSet &hs, const PsdArgs &a, double *lds_raw)
{
    constexpr int NC = CPOL_ICE_DEGREE + 1;
    constexpr int NV = CPOL_ICE_NV;                       // 2 x (12 columns + Doppler-2 v, n)
    static_assert(4 * NV * CPOL_WAVE <= CPOL_PSD_LDS_DOUBLES, "LDS buffer of the integrating kernels too small");
    double (*const s_part)[NV][CPOL_WAVE] = reinterpret_cast<double (*)[NV][CPOL_WAVE]>(lds_raw);
    const int n_units = (int)a.totals[1];
    for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
        int key, start, count;
        load_unit(a, u, key, start, count);
        int j = 0;
        for (int q = 1; q < hs.n_hydro; ++q) if (key >= hs.h[q].key_base) j = q;
        const HydroDev &h = hs.h[j];
        const cpol_hydro_desc &d = h.d;
        if (d.psd_family != CPOL_PSD_ICE_FIELD || !d.uniform_grid || d.tab_degree != CPOL_ICE_DEGREE) continue;
        if (!ice_unit_in_table(h, a, j, start, count)) {                // k_psd<ICE> sums this unit
            if (threadIdx.x == 0) atomicOr((unsigned long long *)&a.totals[2], 1ull);
            continue;
        }

        const int lane = lane_id();
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const int n_d = d.n_d;
        const int chunk = (n_d + CPOL_PSD_WAVES - 1) / CPOL_PSD_WAVES;
        const int k0 = wave * chunk, k1 = min(k0 + chunk, n_d);
        const bool act0 = lane < count, act1 = lane + CPOL_WAVE < count;
        const long n = a.n_sbg;
        const int sbg0 = a.perm[start + (act0 ? lane : 0)];
        const int sbg1 = a.perm[start + (act1 ? lane + CPOL_WAVE : 0)];
        const double *P0 = a.par + ((long)j * CPOL_MAX_PAR) * n + sbg0;
        const double *P1 = a.par + ((long)j * CPOL_MAX_PAR) * n + sbg1;
        const double lam0 = act0 ? P0[0] : 1.0, lam1 = act1 ? P1[0] : 1.0;
        const cdouble_p slice = as_const(h.table + (long)(key - h.key_base) * n_d * CPOL_N_SZ);
        const cdouble_p rv = as_const(DOP2 ? h.rcsw + (long)(key - h.key_base) * n_d * 2 : h.table);
        const cdouble_p Dl = as_const(h.aux);
        const cdouble_p hd = as_const(h.aux + 4 * n_d + 4);
        const cdouble_p ql = as_const(h.aux + 4 * n_d + 8);
        const double lp0 = lam0 / 1000.0, lp1 = lam1 / 1000.0;
        const double a10 = 20.78 * lp0, a20 = 3.290 * lp0, a11 = 20.78 * lp1, a21 = 3.290 * lp1;
        const double B0 = 17.46 * powp(lp0, 0.6357), B1 = 17.46 * powp(lp1, 0.6357);
        const double a10s = a10 * a10, a10c = a10s * a10, a20s = a20 * a20, a20c = a20s * a20;
        const double a11s = a11 * a11, a11c = a11s * a11, a21s = a21 * a21, a21c = a21s * a21;
        const double dl0 = Dl[k0 < n_d ? k0 : 0];
        double E10 = exp(-(a10 * dl0)), E20 = exp(-(a20 * dl0));
        double E11 = exp(-(a11 * dl0)), E21 = exp(-(a21 * dl0));
        const double r10 = exp(-(a10 * hd[0])), r20 = exp(-(a20 * hd[0]));
        const double r11 = exp(-(a11 * hd[0])), r21 = exp(-(a21 * hd[0]));

        double acc0[CPOL_N_SZ], acc1[CPOL_N_SZ];
#pragma unroll
        for (int c = 0; c < CPOL_N_SZ; ++c) { acc0[c] = 0.0; acc1[c] = 0.0; }
        double dv0 = 0.0, dn0 = 0.0, dv1 = 0.0, dn1 = 0.0;          // Doppler scheme 2 sums
        const int touched = l2_touch(h.table + ((long)(key - h.key_base) * n_d + k0) * CPOL_N_SZ,
                                     (long)(k1 - k0) * CPOL_N_SZ * 8, lane);
#pragma unroll 2
        for (int k = k0; k < k1; ++k) {
            const double d1 = ql[4 * k], d2 = ql[4 * k + 1], d3 = ql[4 * k + 2], pw = ql[4 * k + 3];
            const double c10 = fma(-a10, d1, fma(a10s, d2, fma(-a10c, d3, 1.0)));
            const double c20 = fma(-a20, d1, fma(a20s, d2, fma(-a20c, d3, 1.0)));
            const double c11 = fma(-a11, d1, fma(a11s, d2, fma(-a11c, d3, 1.0)));
            const double c21 = fma(-a21, d1, fma(a21s, d2, fma(-a21c, d3, 1.0)));
            const double ph0 = fma(B0 * pw, E20 * c20, 490.6 * (E10 * c10));
            const double ph1 = fma(B1 * pw, E21 * c21, 490.6 * (E11 * c11));
            E10 *= r10; E20 *= r20;
            E11 *= r11; E21 *= r21;
            const cdouble_p row = slice + (long)k * CPOL_N_SZ;
#pragma unroll
            for (int c = 0; c < CPOL_N_SZ; ++c) {
                const double sz = row[c];
                acc0[c] = fma(sz, ph0, acc0[c]);
                acc1[c] = fma(sz, ph1, acc1[c]);
            }
            if (DOP2) {
                const double wv = rv[2 * k], wr = rv[2 * k + 1];
                dv0 = fma(wv, ph0, dv0); dn0 = fma(wr, ph0, dn0);
                dv1 = fma(wv, ph1, dv1); dn1 = fma(wr, ph1, dn1);
            }
        }
        psd_keep(touched);
        // ((w0+w4)+(w2+w6)) + ((w1+w5)+(w3+w7)): fixed order, wave 0 ends up with the totals
#pragma unroll
        for (int half = 4; half >= 1; half >>= 1) {
            if (wave >= half && wave < 2 * half) {
                double (*sp)[CPOL_WAVE] = s_part[wave - half];
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ; ++c) { sp[c][lane] = acc0[c]; sp[CPOL_N_SZ + 2 + c][lane] = acc1[c]; }
                if (DOP2) { sp[CPOL_N_SZ][lane] = dv0; sp[CPOL_N_SZ + 1][lane] = dn0; sp[NV - 2][lane] = dv1; sp[NV - 1][lane] = dn1; }
            }
            __syncthreads();
            if (wave < half) {
                double (*sp)[CPOL_WAVE] = s_part[wave];
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ; ++c) { acc0[c] += sp[c][lane]; acc1[c] += sp[CPOL_N_SZ + 2 + c][lane]; }
                if (DOP2) { dv0 += sp[CPOL_N_SZ][lane]; dn0 += sp[CPOL_N_SZ + 1][lane]; dv1 += sp[NV - 2][lane]; dn1 += sp[NV - 1][lane]; }
            }
            __syncthreads();
        }
        if (wave == 0) {
            const double *th = h.aux + 12 * n_d + 8;
            const double dDn = h.aux[3 * n_d];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const bool act = it ? act1 : act0;
                if (!act) continue;
                const double lam = it ? lam1 : lam0;
                const double *P = it ? P1 : P0;
                const int sb = it ? sbg1 : sbg0;
                // the three normalisation sums of this lambda
                const double pf = ice_panel_pos(th, lam);
                const int pn = (int)pf;
                const double ut = 2.0 * (pf - (double)pn) - 1.0;
                const double *c = th + 4 + (long)pn * (CPOL_ICE_FUNCS * NC);
                double Sm = c[NC - 1], Sv = c[2 * NC - 1], Sn = c[3 * NC - 1];
#pragma unroll
                for (int q = NC - 2; q >= 0; --q) {
                    Sm = fma(Sm, ut, c[q]);
                    Sv = fma(Sv, ut, c[NC + q]);
                    Sn = fma(Sn, ut, c[2 * NC + q]);
                }
                const double N0 = P[n];
                const double qm_est = (N0 * Sm) * dDn;
                const double scale = N0 / qm_est * P[2 * n];         // N0 / QM_est * QM (:1339)
                if (a.par_w) a.par_w[((long)j * CPOL_MAX_PAR + 3) * n + sb] = scale;
                double *o = a.res + ((long)j * n + sb) * CPOL_N_SZ;
#pragma unroll
                for (int cc = 0; cc < CPOL_N_SZ; ++cc) o[cc] = (scale * (it ? acc1[cc] : acc0[cc])) * d.dD;
                if (a.vn) {
                    // scheme 1: integrate_V (x dD of its grid); scheme 2: unit-spaced trapezoid
                    a.vn[((long)j * n + sb) * 2] = DOP2 ? scale * (it ? dv1 : dv0) : (scale * Sv) * dDn;
                    a.vn[((long)j * n + sb) * 2 + 1] = DOP2 ? scale * (it ? dn1 : dn0) : (scale * Sn) * dDn;
                }
            }
        }
    }
}

template <bool DOP2>
__global__ __launch_bounds__(CPOL_PSD_THREADS) void k_psd_melting_tab(HydroSet hs, PsdArgs a)
{
    __shared__ double lds_raw[4 * CPOL_MELT_NV * CPOL_WAVE];
    psd_melting_tab_body<DOP2>(hs, a, lds_raw);
}

template <bool DOP2>
__global__ __launch_bounds__(CPOL_PSD_THREADS, 4) void k_psd_ice2(HydroSet hs, PsdArgs a)
{
    __shared__ double lds_raw[4 * CPOL_ICE_NV * CPOL_WAVE];
    psd_ice2_body<DOP2>(hs, a, lds_raw);
}

// Every integrating flavour from ONE launch, for the sweeps whose work units are single items listed directly (every slot on
// an integral table: the units are the handful of items outside the tables, usually none).  The flavours used to be up to
// five launches that found nothing to do (5-6 us apiece on the stream: 20 of the 157 us of a C3 sweep); the bodies are the
// kernels' own, run one after the other by every workgroup (each skips the units of the other flavours), so the results
// are the same bits.  `modes`: bit m = flavour m (PSD_MODE_*) is staged; bit 4: a melting slot without polynomial tables.
#ifdef CPOL_RARE_TRACE
__device__ unsigned long long g_rare_trace[64 * 8];      // measurement build: the 100-MHz clock between the flavours, per workgroup
#endif
template <bool DOP2>
__global__ __launch_bounds__(CPOL_PSD_THREADS) void k_psd_rare(HydroSet hs, PsdArgs a, int modes)
{
    static_assert(CPOL_PSD_THREADS_U == CPOL_PSD_THREADS, "k_psd_rare runs the recurrence flavour with the workgroup of the others");
    if (a.totals[1] == 0) return;                         // nothing outside the tables (the rule)
    __shared__ double lds_raw[CPOL_PSD_LDS_DOUBLES];
#ifdef CPOL_RARE_TRACE
#define RT(i) if (threadIdx.x == 0 && blockIdx.x < 64) g_rare_trace[blockIdx.x * 8 + (i)] = wall_clock64();
#else
#define RT(i)
#endif
    RT(0)
    if (modes & (1 << PSD_MODE_MELTING)) {
        psd_melting_tab_body<DOP2>(hs, a, lds_raw);
        __syncthreads();
        if (modes & 16) { psd_body<PSD_MODE_MELTING, DOP2>(hs, a, lds_raw); __syncthreads(); }
    }
    RT(1)
    if (modes & (1 << PSD_MODE_ICE)) {
        psd_ice2_body<DOP2>(hs, a, lds_raw);
        __syncthreads();
        RT(2)
        psd_body<PSD_MODE_ICE, DOP2>(hs, a, lds_raw);      // (a.ice_same_launch: tests every unit itself)
        __syncthreads();
    }
    RT(3)
    if (modes & (1 << PSD_MODE_GAMMA_UNIFORM)) { psd_body<PSD_MODE_GAMMA_UNIFORM, DOP2>(hs, a, lds_raw); __syncthreads(); }
    RT(4)
    if (modes & (1 << PSD_MODE_GAMMA_EXP)) psd_body<PSD_MODE_GAMMA_EXP, DOP2>(hs, a, lds_raw);
    RT(5)
#undef RT
}

// ---------------------------------------------------------------- integral tables (itab)
// Node lambda of (panel p, node q): the Chebyshev node x_q = cos(pi (q + 1/2) / NC) of the panel
// [2^(lo + p/PPO), 2^(lo + (p+1)/PPO)] on the log2(lambda) axis.
__device__ __forceinline__ double itab_node_lambda(double log2_lo, int p, int q)
{
    const double x = cos(3.14159265358979323846 * ((double)q + 0.5) / (double)CPOL_ITAB1_NC);
    return exp2(log2_lo + ((double)p + (x + 1.0) * 0.5) / (double)CPOL_ITAB_PPO);
}

// synthetic "items" of one hydrometeor slot: one per (slice, panel, node), with N0 = QM = 1
// (melting species: per (slice, panel) the 11 x 11 nodes in (lambda_r, fw) and one check point)
struct ItabBuildArgs {
    double *par;               // [CPOL_MAX_PAR][n_items] of THIS slot
    int *perm;                 // identity
    WorkUnit *units;
    long long *totals;         // [1] = n_units, [2] = 0
    long n_items;
    int n_slices, n_pan, key_base, unit_items;
    double log2_lo;
    int ppo;
    int two_d, n_t;            // melting species
    const double *head;        // [n_t][2]
};

__device__ __forceinline__ double cheb_node(int q)
{
    return cos(3.14159265358979323846 * ((double)q + 0.5) / (double)CPOL_ITAB_NC);
}

__global__ void k_itab_nodes(ItabBuildArgs b)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per_block = b.two_d ? CPOL_ITAB2_NODES : CPOL_ITAB1_NODES;
    const int per_slice = b.n_pan * per_block;
    const int upers = (per_slice + b.unit_items - 1) / b.unit_items;
    if (i == 0) { b.totals[0] = b.n_items; b.totals[1] = (long long)b.n_slices * upers; b.totals[2] = 0; }
    if (i < (long)b.n_slices * upers) {
        const int s = (int)(i / upers), uu = (int)(i % upers);
        WorkUnit w;
        w.key = b.key_base + s;
        w.start = s * per_slice + uu * b.unit_items;
        w.count = min(b.unit_items, per_slice - uu * b.unit_items);
        w.pad = 0;
        b.units[i] = w;
    }
    if (i >= b.n_items) return;
    const int r = (int)(i % per_slice);
    b.perm[i] = (int)i;
    if (b.two_d) {
        const int s = (int)(i / per_slice), tb = s % b.n_t;
        const int p = r / per_block, node = r % per_block;
        const bool chk = node == CPOL_ITAB2_NODES - 1;
        const double xw = chk ? CPOL_ITAB2_CHECK_W : cheb_node(node / CPOL_ITAB_NC);
        const double xu = chk ? CPOL_ITAB2_CHECK_U : cheb_node(node % CPOL_ITAB_NC);
        b.par[i] = 1.0;                                                        // QM
        b.par[b.n_items + i] = b.head[2 * tb] + xu / b.head[2 * tb + 1];       // fw
        b.par[2 * b.n_items + i] = exp2(b.log2_lo + ((double)p + (xw + 1.0) * 0.5) / (double)b.ppo);
        b.par[3 * b.n_items + i] = 0.0;
        return;
    }
    {
        const int p = r / CPOL_ITAB1_NODES, q = r % CPOL_ITAB1_NODES;      // q >= NC: the block's two check points
        const double uc = q == CPOL_ITAB1_NC ? CPOL_ITAB1_CHECK_U : CPOL_ITAB1_CHECK_U2;
        b.par[i] = q < CPOL_ITAB1_NC ? itab_node_lambda(b.log2_lo, p, q)
                                    : exp2(b.log2_lo + ((double)p + (uc + 1.0) * 0.5) / (double)CPOL_ITAB_PPO);
    }
    b.par[b.n_items + i] = 1.0;             // N0
    b.par[2 * b.n_items + i] = 1.0;         // QM (ice)
    b.par[3 * b.n_items + i] = 0.0;
}

// node values -> monomial coefficients of the degree-10 interpolant (matrix M = basis change x
// discrete Chebyshev transform, built on the host in extended precision)
struct ItabFitArgs {
    const double *res;         // [n_items][12]
    const double *vn;          // [n_items][2]
    const double *par;         // [CPOL_MAX_PAR][n_items] (slot 3: ice normalised N0)
    const double *M;           // [NC1][NC1] (NC1 = CPOL_ITAB1_NC)
    double *tab;               // [n_slices][n_pan][NC1][NFP]
    long n_items;
    int n_slices, n_pan;
    double log2_lo, d0;
    unsigned long long *worst; // check kernel: bits of the worst deviation | (block x NF + function) mod 2^24
                               // ([2]: bits of the worst deviation at the SECOND check point alone, near the panel edge)
    unsigned int *n_bad;       // check kernel: (block, function) pairs at or above max_dev
    double max_dev;
    unsigned long long *by_pan;   // check kernel: [n_pan] bits of the worst deviation per lambda panel, then [n_pan] at the edge point alone
    unsigned long long *by_fn;    // check kernel: [CPOL_ITAB_NF] ... per function
};

__global__ void k_itab_fit(ItabFitArgs f)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n_blocks = (long)f.n_slices * f.n_pan;
    if (t >= n_blocks * CPOL_ITAB_NF) return;
    const long blk = t / CPOL_ITAB_NF;
    const int fn = (int)(t % CPOL_ITAB_NF);
    const int p = (int)(blk % f.n_pan);
    auto value = [&](int q, double lambda) {
        const long i = blk * CPOL_ITAB1_NODES + q;
        double x = fn < CPOL_N_SZ ? f.res[i * CPOL_N_SZ + fn]
                 : fn < CPOL_N_SZ + 2 ? f.vn[i * 2 + (fn - CPOL_N_SZ)] : f.par[3 * f.n_items + i];
        // gamma family: tabulate exp(+lambda d0) x integral (no super-exponential decay left)
        if (f.d0 != 0.0) x *= exp(lambda * f.d0);
        return x;
    };
    double v[CPOL_ITAB1_NC];
    double scale = 0.0;
    bool nan = false;
#pragma unroll
    for (int q = 0; q < CPOL_ITAB1_NC; ++q) {
        v[q] = value(q, itab_node_lambda(f.log2_lo, p, q));
        nan = nan || !(v[q] == v[q]);
        scale = fmax(scale, fabs(v[q]));
    }
    double *o = f.tab + blk * (CPOL_ITAB1_NC * CPOL_ITAB_NFP) + fn;
    double c[CPOL_ITAB1_NC];
#pragma unroll
    for (int pw = 0; pw < CPOL_ITAB1_NC; ++pw) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < CPOL_ITAB1_NC; ++q) acc = fma(f.M[pw * CPOL_ITAB1_NC + q], v[q], acc);
        c[pw] = acc;
        o[pw * CPOL_ITAB_NFP] = acc;
        if (fn == CPOL_ITAB_NF - 1) o[pw * CPOL_ITAB_NFP + 1] = 0.0;     // padding column
    }
    // ---- the accuracy gate: the polynomial against the integrating kernel at the block's check point
    // (the 12th item of the block), on the scale of the function over the block -- the largest |value|
    // at the nodes and the check point -- so that a column that changes sign inside a panel raises no
    // false alarm, while node values that are rounding noise of a cancelling sum do ----
    const double lam_c = exp2(f.log2_lo + ((double)p + (CPOL_ITAB1_CHECK_U + 1.0) * 0.5) / (double)CPOL_ITAB_PPO);
    const double lam_e = exp2(f.log2_lo + ((double)p + (CPOL_ITAB1_CHECK_U2 + 1.0) * 0.5) / (double)CPOL_ITAB_PPO);
    const double ref = value(CPOL_ITAB1_NC, lam_c), ref_e = value(CPOL_ITAB1_NC + 1, lam_e);
    scale = fmax(scale, fmax(fabs(ref), fabs(ref_e)));
    double got = c[CPOL_ITAB1_NC - 1], got_e = c[CPOL_ITAB1_NC - 1];
#pragma unroll
    for (int q = CPOL_ITAB1_NC - 2; q >= 0; --q) {
        got = fma(got, CPOL_ITAB1_CHECK_U, c[q]);
        got_e = fma(got_e, CPOL_ITAB1_CHECK_U2, c[q]);
    }
    double err_e = scale > 0.0 ? fabs(got_e - ref_e) / scale : 0.0;
    if (nan || !(err_e == err_e) || !(ref_e == ref_e) || isinf(scale)) err_e = 1.0;
    double err = scale > 0.0 ? fabs(got - ref) / scale : 0.0;
    err = fmax(err, err_e);
    if (nan || !(err == err) || !(ref == ref) || isinf(scale)) err = 1.0;
    if (err >= f.max_dev) atomicAdd(f.n_bad, 1u);
    if (err > 0.0) {
        // running maxima: a plain load first -- only a thread that would raise the maximum issues the
        // atomic (5 M threads on ~200 addresses otherwise: 10 ms instead of 1)
        const unsigned long long eb = (unsigned long long)__double_as_longlong(err);   // (positive doubles order like their bits)
        const unsigned long long tagged = (eb & ~0xFFFFFFull) | ((unsigned long long)t & 0xFFFFFFull);
        if (tagged > __builtin_nontemporal_load(f.worst)) atomicMax(f.worst, tagged);
        const unsigned long long ee = (unsigned long long)__double_as_longlong(err_e);
        if (ee > __builtin_nontemporal_load(f.worst + 2)) atomicMax(f.worst + 2, ee);
        if (ee > __builtin_nontemporal_load(f.by_pan + f.n_pan + p)) atomicMax(f.by_pan + f.n_pan + p, ee);
        if (eb > __builtin_nontemporal_load(f.by_pan + p)) atomicMax(f.by_pan + p, eb);
        if (eb > __builtin_nontemporal_load(f.by_fn + fn)) atomicMax(f.by_fn + fn, eb);
    }
}

// Melting species: node values V[b][a] (b: lambda_r node, a: fw node) of one function of one
// block -> Chebyshev coefficients (Tm: discrete transform), terms of total degree > 10 dropped,
// -> monomial coefficients (C2M[n][p]: coefficient of x^p in T_n).  One thread per (block, function).
struct ItabFit2Args {
    const double *res;         // [n_items][12]
    const double *vn;          // [n_items][2]
    const double *M;           // [3][NC][NC]: M (1-D fit), Tm, C2M
    double *tab;               // [n_blocks][CPOL_ITAB2_NB][NFP]
    long n_blocks;
    unsigned long long *worst; // check kernel: bits of the worst relative deviation at the check points
};

__global__ void k_itab_fit2(ItabFit2Args f)
{
    constexpr int NC = CPOL_ITAB_NC, NFP = CPOL_ITAB_NFP;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= f.n_blocks * NFP) return;
    const int fn = (int)(t % NFP);
    const long blk = t / NFP;
    double *o = f.tab + blk * (CPOL_ITAB2_NB * NFP) + fn;
    if (fn >= CPOL_N_SZ + 2) {                                            // padding columns
        for (int row = 0; row < CPOL_ITAB2_NB; ++row) o[(long)row * NFP] = 0.0;
        return;
    }
    const double *Tm = f.M + NC * NC, *C2M = f.M + 2 * NC * NC;
    double A[NC][NC], B[NC][NC];
    for (int b = 0; b < NC; ++b)
        for (int a = 0; a < NC; ++a) {
            const long i = blk * CPOL_ITAB2_NODES + b * NC + a;
            A[b][a] = fn < CPOL_N_SZ ? f.res[i * CPOL_N_SZ + fn] : f.vn[i * 2 + (fn - CPOL_N_SZ)];
        }
    // Chebyshev coefficients in u (second index), then in w (first index)
    for (int b = 0; b < NC; ++b)
        for (int ca = 0; ca < NC; ++ca) {
            double c = 0.0;
            for (int a = 0; a < NC; ++a) c = fma(Tm[ca * NC + a], A[b][a], c);
            B[b][ca] = c;
        }
    for (int cb = 0; cb < NC; ++cb)
        for (int ca = 0; ca < NC; ++ca) {
            double c = 0.0;
            for (int b = 0; b < NC; ++b) c = fma(Tm[cb * NC + b], B[b][ca], c);
            A[cb][ca] = (ca + cb < NC) ? c : 0.0;                         // total degree <= 10
        }
    // monomials in u, then in w
    for (int cb = 0; cb < NC; ++cb)
        for (int pa = 0; pa < NC; ++pa) {
            double c = 0.0;
            for (int ca = pa; ca < NC; ++ca) c = fma(C2M[ca * NC + pa], A[cb][ca], c);
            B[cb][pa] = c;
        }
    for (int pb = 0; pb < NC; ++pb)
        for (int pa = 0; pa + pb < NC; ++pa) {
            double c = 0.0;
            for (int cb = pb; cb < NC; ++cb) c = fma(C2M[cb * NC + pb], B[cb][pa], c);
            o[(long)(CPOL_ITAB2_ROW(pb) + pa) * NFP] = c;
        }
}

// value of function fn of a 2-D block at (u, w)
__device__ __forceinline__ double itab2_eval(const double *blk, int fn, double u, double w)
{
    constexpr int NC = CPOL_ITAB_NC, NFP = CPOL_ITAB_NFP;
    double outer = 0.0;
    for (int pb = NC - 1; pb >= 0; --pb) {
        double inner = 0.0;
        for (int pa = NC - 1 - pb; pa >= 0; --pa) inner = fma(inner, u, blk[(long)(CPOL_ITAB2_ROW(pb) + pa) * NFP + fn]);
        outer = fma(outer, w, inner);
    }
    return outer;
}

// the fitted polynomial against the integrating kernel at the check point of every block
__global__ void k_itab_check2(ItabFit2Args f)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= f.n_blocks * (CPOL_N_SZ + 2)) return;
    const int fn = (int)(t % (CPOL_N_SZ + 2));
    const long blk = t / (CPOL_N_SZ + 2);
    const long i = blk * CPOL_ITAB2_NODES + (CPOL_ITAB2_NODES - 1);
    const double ref = fn < CPOL_N_SZ ? f.res[i * CPOL_N_SZ + fn] : f.vn[i * 2 + (fn - CPOL_N_SZ)];
    const double got = itab2_eval(f.tab + blk * (CPOL_ITAB2_NB * CPOL_ITAB_NFP), fn, CPOL_ITAB2_CHECK_U, CPOL_ITAB2_CHECK_W);
    // scale: the function's size over the block (its value at the first node), so that columns
    // that pass through zero do not raise a false alarm
    const long i0 = blk * CPOL_ITAB2_NODES;
    const double at0 = fn < CPOL_N_SZ ? f.res[i0 * CPOL_N_SZ + fn] : f.vn[i0 * 2 + (fn - CPOL_N_SZ)];
    const double scale = fmax(fabs(ref), fabs(at0));
    double err = scale > 0.0 ? fabs(got - ref) / scale : 0.0;
    if (!(err == err) || !(ref == ref)) err = 1.0;                        // NaN
    // 40 leading bits of the deviation | 24 bits of (block, function): one atomicMax finds both
    if (err > 0.0)
        atomicMax(f.worst, ((unsigned long long)__double_as_longlong(err) & ~0xFFFFFFull) | ((unsigned long long)t & 0xFFFFFFull));
}

#ifndef CPOL_LOOKUP_THREADS
#define CPOL_LOOKUP_THREADS 256
#endif
#ifndef CPOL_TILE_GATES_LOG2
#define CPOL_TILE_GATES_LOG2 2       // melting species: a wavefront = 16 neighbouring rays x 4 gates of one sub-beam index
#endif
// one thread per sub-beam gate: the items that k_classify placed on an integral table
struct LookupArgs {
    const int *key;             // [n_hydro][n_sbg] (valid where vmask says so)
    const unsigned char *vmask; // [n_sbg] bit j: hydrometeor j present
    const double2 *rec;         // [n_hydro][n_sbg] {panel position (-1: not on the table), scale} (k_classify)
    const double *par;          // [n_hydro][CPOL_MAX_PAR][n_sbg] (melting species: slot 1 = wet fraction)
    double *par_w;              // = par, writable (ice: normalised N0 for the spectrum kernels) or NULL
    double *res;                // [n_hydro][n_sbg][12]
    double *vn;                 // [n_hydro][n_sbg][2] or NULL
    long n_sbg;
    int tile;                   // lanes of a wavefront = 16 neighbouring rays x 4 gates of one sub-beam index
    int split;                  // wavefronts per tile (gridDim.y) sharing the distinct 2-D blocks and the 1-D species of the tile (1: large launches)
    int n_rays, n_sub, n_gates; // (melting species: neighbouring rays share the (slice, panel) block, gates do not)
    int vn_1d;                  // with skip_res_1d: the Doppler sums of the 1-D species are still written here (k_final
                                // evaluates the columns in place but reads vn[]; k_subbeam_sum writes them itself)
    int skip_res_1d;            // the 12 columns of the 1-D species are evaluated by k_subbeam_sum / k_final instead
                                // (nothing of them is stored); only vn / the ice intercept are written here
    // round 5 (tile_scan): 85 % of the tiles of a C4 volume hold nothing for this kernel, and a wavefront needs 2.3-2.5 us to
    // find that out.  A workgroup then owns the tiles blockIdx.x, blockIdx.x + gridDim.x, ...: its wavefronts first look at
    // the validity bytes of all of them (16 lanes per tile, one ray each, the 4 gates as one word; four tiles per wavefront
    // and pass, loads of different passes independent), list those with a species of `species2d` in LDS, and then take the
    // listed tiles in turn.  0: one wavefront per tile, as before.
    int tile_scan;
    unsigned species2d;         // bit j: species j has a 2-D table
    long n_tiles;
};

__device__ __forceinline__ double shfl_xor_f64(double v, int m)
{
    const int lo = __shfl_xor(__double2loint(v), m), hi = __shfl_xor(__double2hiint(v), m);
    return __hiloint2double(hi, lo);
}

// 2-D block: quarter r of the wavefront owns the rows of w^b, b = r, r+4, r+8 (u^0 .. u^(10-b) each);
// lane (r, f) keeps ITS 21 row values of column f in registers (rows beyond 10-b: zero) ...
#define CPOL_ITAB2_QROWS 21
__device__ __forceinline__ void itab2_quarter_rows(const double *c, int r, double (&rows)[CPOL_ITAB2_QROWS])
{
    constexpr int NC = CPOL_ITAB_NC, NFP = CPOL_ITAB_NFP;
    int idx = 0;
#pragma unroll
    for (int k = 2; k >= 0; --k) {
        const int b = r + 4 * k;                                        // power of w of this quarter's row set
        const int bb = min(b, NC - 1);
        const double *cb = c + (long)CPOL_ITAB2_ROW(bb) * NFP;
#pragma unroll
        for (int pa = NC - 1 - 4 * k; pa >= 0; --pa) rows[idx++] = pa <= NC - 1 - b ? cb[pa * NFP] : 0.0;
    }
}

// ... and sums  sum_b w^b sum_a rows u^a  over its b for one item: the quarters run the chain of
// the longest (r = 0) together, the shorter ones start with zeros
__device__ __forceinline__ double itab2_quarter_sum(const double (&rows)[CPOL_ITAB2_QROWS], int r, double ui, double wi)
{
    constexpr int NC = CPOL_ITAB_NC;
    const double w2 = wi * wi, w4 = w2 * w2;
    double acc = 0.0;
    int idx = 0;
#pragma unroll
    for (int k = 2; k >= 0; --k) {
        double inner = 0.0;
#pragma unroll
        for (int pa = NC - 1 - 4 * k; pa >= 0; --pa) inner = fma(inner, ui, rows[idx++]);
        acc = fma(acc, w4, inner);
    }
    return acc * ((r & 1 ? wi : 1.0) * (r & 2 ? w2 : 1.0));
}

// TWO items of a 2-D block per pass (CPOL_ITAB2_PAIR, round 5): lane = (item slot, half h, function f); half h owns the
// rows of the powers w^b with b = h (mod 2) -- 36 row values per lane in registers (h = 1: 30, padded with zeros) -- and
// sums  w^h sum_k (w^2)^k sum_a rows u^a  for ITS slot's item; one cross-lane add joins the halves.  The per-item work
// that is not arithmetic (operands brought to the lanes, the address, the store) is paid once per pair: ~40 wave
// instructions per item instead of ~70 with four quarters on one item.  NOT the default: the 36 row registers cost the
// kernel a wavefront per SIMD (128 VGPRs: 4 instead of 5) and that costs more than the instructions save -- C3 sweep
// 56.2 -> 54.1 us of PSD stage, the 225-ray C4 share 845 -> 866, the C4 volume 2 952 -> 3 107 (profiles/r5_variants.txt).
#ifndef CPOL_ITAB2_PAIR
#define CPOL_ITAB2_PAIR 0
#endif
#define CPOL_ITAB2_HROWS 36
__device__ __forceinline__ void itab2_half_rows(const double *c, int h, double (&rows)[CPOL_ITAB2_HROWS])
{
    constexpr int NC = CPOL_ITAB_NC, NFP = CPOL_ITAB_NFP;
    static_assert(NC == 11, "the half split is written for total degree 10");
    int idx = 0;
#pragma unroll
    for (int k = 5; k >= 0; --k) {
        const int b = h + 2 * k;                                        // power of w of this half's row set
        const int bb = min(b, NC - 1);
        const double *cb = c + (long)CPOL_ITAB2_ROW(bb) * NFP;
#pragma unroll
        for (int pa = NC - 1 - 2 * k; pa >= 0; --pa) rows[idx++] = pa <= NC - 1 - b ? cb[pa * NFP] : 0.0;
    }
}

__device__ __forceinline__ double itab2_half_sum(const double (&rows)[CPOL_ITAB2_HROWS], int h, double ui, double wi)
{
    constexpr int NC = CPOL_ITAB_NC;
    const double w2 = wi * wi;
    double acc = 0.0;
    int idx = 0;
#pragma unroll
    for (int k = 5; k >= 0; --k) {
        double inner = 0.0;
#pragma unroll
        for (int pa = NC - 1 - 2 * k; pa >= 0; --pa) inner = fma(inner, ui, rows[idx++]);
        acc = fma(acc, w2, inner);
    }
    return acc * (h ? wi : 1.0);
}

// The items of one block (`grp`: their lanes), two per pass; `emit(l, acc, live)` is called by EVERY lane with its slot's item l and
// the lane's function value (the two halves already joined, times the item's scale); `live` = the slot holds an item: lanes with
// h == 0 of a live slot store.
template <typename Emit>
__device__ __forceinline__ void itab2_walk_pairs(const double *blk_f, unsigned long long grp, double u, double w, double q, Emit emit)
{
    const int lane = lane_id();
    const int h = (lane >> 4) & 1, slot = lane >> 5;
    double rows[CPOL_ITAB2_HROWS];
    itab2_half_rows(blk_f, h, rows);
    while (grp) {
        const int la = (int)__ffsll((long long)grp) - 1;
        grp &= grp - 1;
        const bool two = grp != 0;
        const int lb = two ? (int)__ffsll((long long)grp) - 1 : la;
        if (two) grp &= grp - 1;
        const double ua = readlane_f64(u, la), ub = readlane_f64(u, lb);
        const double wa = readlane_f64(w, la), wb = readlane_f64(w, lb);
        const double qa = readlane_f64(q, la), qb = readlane_f64(q, lb);
        double acc = itab2_half_sum(rows, h, slot ? ub : ua, slot ? wb : wa);
        acc += shfl_xor_f64(acc, 16);
        acc *= slot ? qb : qa;
        emit(slot ? lb : la, acc, slot == 0 || two);                    // (every lane calls: the lambda may shuffle)
    }
}

// the 12 integrated columns of an item from its 1-D block (c) at panel position u, times its scale
__device__ __forceinline__ void itab1_columns(const double2 *c, double u, double scale, double2 (&v)[CPOL_N_SZ / 2])
{
    constexpr int NC = CPOL_ITAB1_NC, NFP = CPOL_ITAB_NFP;
#pragma unroll
    for (int f = 0; f < CPOL_N_SZ / 2; ++f) v[f] = c[(NC - 1) * (NFP / 2) + f];
#pragma unroll
    for (int q = NC - 2; q >= 0; --q) {
#pragma unroll
        for (int f = 0; f < CPOL_N_SZ / 2; ++f) {
            const double2 cq = c[q * (NFP / 2) + f];
            v[f].x = fma(v[f].x, u, cq.x);
            v[f].y = fma(v[f].y, u, cq.y);
        }
    }
#pragma unroll
    for (int f = 0; f < CPOL_N_SZ / 2; ++f) { v[f].x *= scale; v[f].y *= scale; }
}

// 1-D blocks (gamma family, 1-moment ice).  The coefficient block of an item is addressed by
// (LUT slice, lambda panel): 11 rows of 128 B, row q = the coefficients of u^q of all functions.
// Every lane reads the rows of ITS block as 16-B vector loads, all 12 Horner chains of an item
// advance together, and the loads of a row are independent of the arithmetic, so the compiler
// keeps whole rows in flight.  Lanes whose gates share a block read the same addresses.
// Measured alternatives on the bench sweep (195 751 items) / the C4 sweep at 3 degrees (15.1 M):
//   function-major table, one chain at a time (11 dependent 8-B loads each)   28.8 us / 1.69 ms
//   walk over the distinct blocks of the wavefront, rows by s_load into SGPRs,
//     v_fma_f64 with scalar addend (every row misses the scalar cache)        ~100 us / 1.6 ms
//   the same walk with the block staged in LDS and read back by broadcast      ~43 us / 1.5 ms
//     (a wavefront of 64 neighbouring gates spans ~7 distinct blocks, so the walk repeats)
//   this form                                                                  ~23 us / 1.1 ms
//   this form with the lanes of a wavefront on 8 neighbouring rays x 8 gates (2-3 distinct
//     blocks per wavefront instead of ~7): no change
//   the distinct blocks of the wavefront fetched once into LDS, rows gathered from LDS per lane
//     (bank-disjoint slots): no change (1.28 ms for the 15.1 M 1-D items of the C4 sweep)
//   keys of all species read first, parameters of species j + 1 requested before species j is
//     evaluated: no change (1.26 ms)
//   -> what is left is the issue rate of the 66-77 16-B gather instructions per wavefront and
//     species (~16 cycles each in the texture-address path, whatever the addresses), 84 ps per item
//
// 2-D blocks (melting species): 66 rows of 128 B per block and neighbouring gates hardly ever
// share one (the wet fraction crosses a 0.01-wide bin of the table per gate).  One lane per item
// -- with the full 11 x 11 tensor: 847 16-B loads per lane, 64 different cache lines per
// instruction -- took 3.0 ms for the 1.28 M melting items of the C4 sweep.  Here the WAVEFRONT takes the items of its 64 gates one
// after the other: lane = (quarter r, function f); quarter r sums the rows of the powers
// w^r, w^(r+4), w^(r+8), each row a full Horner chain in u, every row is read exactly once as one
// contiguous 128-B line per quarter; two cross-lane adds join the quarters.
// (Measured on the C4 volume, 6.4 M melting items, this form 1.35 ms: every block replaced by ONE cached
// block 1.32 ms -- the rows are not the cost; 2 / 3 / 4 items of a block in flight per trip 1.33 / 1.39 /
// 1.40 ms -- nor is the latency of an item's chain; four items per pass with lane = (item slot, function)
// and every lane running the whole 66-term polynomial from its own row loads, no cross-lane sums: 3.04 ms.
// Round 4: the rows of the NEXT distinct block requested before the items of the current one (two row sets in
// registers, 180 VGPRs, 2 wavefronts per SIMD): C3 sweep 66.6 -> 73.8 us, the 225-ray C4 share 245 -> 394 us,
// the C4 volume 1.36 -> 2.38 ms -- the walk does not wait for its rows at any of these sizes.
// The block staged in LDS (8.25 KB per wavefront, 4 wavefronts per SIMD) and its items four at a time, lane = (item slot,
// function) running the whole polynomial from the LDS copy -- the four quarter sums formed and joined exactly as here, the
// item's operands brought to its 16 lanes by ds_bpermute, no cross-lane sums: PSD stage of the C4 volume 3.05 -> 3.21 ms,
// of the 225-ray share 850 -> 877 us, of the C3 sweep 62.5 -> 59.8 us: ~10 items share a block, staging it costs what the
// 66 LDS reads per pass save.)
#ifndef CPOL_LOOKUP_GROUP_LOG2
#define CPOL_LOOKUP_GROUP_LOG2 2      // consecutive tiles per group of a workgroup's own tiles (tile scan)
#endif
#ifndef CPOL_LOOKUP_LIST_CAP
#define CPOL_LOOKUP_LIST_CAP 2048     // tiles a workgroup of k_psd_lookup owns at most when it scans for those with work (8 KB of LDS)
#endif
#ifndef CPOL_LOOKUP_WPE
#define CPOL_LOOKUP_WPE 5             // wavefronts per SIMD asked of the register allocator (102 VGPRs: 5 fit since round 4; C4 volume 4 / 5 / 6 / 8: 1.36 / 1.22 / 1.51 / 2.24 ms, share 245 / 227 / 278 us, C3 sweep 66.6 / 67.1 us)
#endif
__global__ __launch_bounds__(CPOL_LOOKUP_THREADS, CPOL_LOOKUP_WPE) void k_psd_lookup(HydroSet hs, ItabSet its, LookupArgs a)
{
    constexpr int NC = CPOL_ITAB1_NC, NFP = CPOL_ITAB_NFP, NB = NC * NFP;      // (1-D blocks; the 2-D walk has its own constants)
    long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = lane_id();
#ifdef CPOL_LOOKUP_TRACE
    const unsigned long long trace_t0 = wall_clock64();
    const unsigned long trace_w = ((unsigned long)blockIdx.y * gridDim.x * blockDim.x + (unsigned long)i0) >> 6;
    unsigned trace_work = 0;
    struct TraceEnd {
        unsigned long long t0; unsigned long w; unsigned *work; int lane;
        __device__ ~TraceEnd() {
            if (lane == 0 && w < CPOL_SUBSUM_TRACE_N) {
                g_subsum_trace[CPOL_SUBSUM_TRACE_W * w] = t0;
                g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + 1] = wall_clock64();
                g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + 2] = *work;
                g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                                              (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32;
            }
        }
    } trace_end{trace_t0, trace_w, &trace_work, lane};
#endif
    // a.tile_scan: the workgroup lists the tiles among its own that hold a species with a 2-D table, then walks the list
    __shared__ int s_list[CPOL_LOOKUP_LIST_CAP];
    __shared__ int s_n;
    constexpr int WAVES = CPOL_LOOKUP_THREADS / CPOL_WAVE;
    int n_list = 1;
    if (a.tile_scan) {
        constexpr int TG = CPOL_TILE_GATES_LOG2, GATES = 1 << TG, RAYS = 64 >> TG;
        static_assert(GATES == 4 && RAYS == 16, "the tile scan reads the 4 gates of a ray as one word and joins the 16 rays of a tile");
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        const int r = lane & (RAYS - 1), q = lane >> 4, wv = (int)(threadIdx.x >> 6);
        const int n_gt = (a.n_gates + GATES - 1) >> TG;
        // the workgroup owns groups of 2^GL consecutive tiles (gate tiles of one sub-beam index: neighbours share table blocks in L2),
        // group g of blockIdx.x = blockIdx.x + g gridDim.x
        constexpr int GL = CPOL_LOOKUP_GROUP_LOG2, GS = 1 << GL;
        const long G4 = (a.n_tiles + GS - 1) >> GL;
        const long K = G4 > (long)blockIdx.x ? GS * ((G4 - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;   // tiles of this workgroup
        const bool aligned = (a.n_gates & 3) == 0 && ((unsigned long long)a.vmask & 3) == 0;
#pragma unroll 4
        for (long k0 = 0; k0 < K; k0 += 4 * WAVES) {
            const long k = k0 + wv * 4 + q;
            const long W = (((long)blockIdx.x + (k >> GL) * gridDim.x) << GL) + (k & (GS - 1));
            bool has = false;
            if (k < K && W < a.n_tiles) {
                const int gt = (int)(W % n_gt), sub = (int)((W / n_gt) % a.n_sub);
                const int ray = (int)(W / ((long)n_gt * a.n_sub)) * RAYS + r;
                if (ray < a.n_rays) {
                    const long ib = ((long)ray * a.n_sub + sub) * a.n_gates + (long)gt * GATES;
                    unsigned bits = 0;
                    if (aligned) {                                      // (n_gates a multiple of 4: the word lies inside the ray)
                        const unsigned w4 = *reinterpret_cast<const unsigned *>(a.vmask + ib);
                        bits = w4 | w4 >> 8 | w4 >> 16 | w4 >> 24;
                    } else {
                        for (int g = 0; g < GATES; ++g)
                            if (gt * GATES + g < a.n_gates) bits |= a.vmask[ib + g];
                    }
                    has = (bits & a.species2d & 0xFFu) != 0;
                }
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(has);
            unsigned tiles = 0;                                         // bit t: tile t of this wavefront's four holds something (uniform)
            for (int t = 0; t < 4; ++t) tiles |= ((m >> (16 * t)) & 0xFFFFull) ? 1u << t : 0u;
            if (!tiles) continue;
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_n, __popc(tiles));
            base = __builtin_amdgcn_readfirstlane(base);
            if (r == 0 && ((tiles >> q) & 1u)) s_list[base + __popc(tiles & ((1u << q) - 1u))] = (int)W;
        }
        __syncthreads();
        n_list = s_n;
    }
    for (int entry = a.tile_scan ? (int)(threadIdx.x >> 6) : 0; entry < n_list; entry += a.tile_scan ? WAVES : 1) {
    bool in = i0 < a.n_sbg;
    if (a.tile) {
        long W = i0 >> 6;                                               // tile = wavefront
        if (a.tile_scan) W = s_list[entry];
        constexpr int TG = CPOL_TILE_GATES_LOG2, GATES = 1 << TG, RAYS = 64 >> TG;    // tile = RAYS rays x GATES gates
        const int n_gt = (a.n_gates + GATES - 1) >> TG;
        const int gt = (int)(W % n_gt), sub = (int)((W / n_gt) % a.n_sub);
        const int ray = (int)(W / ((long)n_gt * a.n_sub)) * RAYS + (lane >> TG), gate = gt * GATES + (lane & (GATES - 1));
        in = ray < a.n_rays && gate < a.n_gates;
        i0 = ((long)ray * a.n_sub + sub) * a.n_gates + gate;
    }
    const long i = in ? i0 : 0;
    const long n = a.n_sbg;
    const unsigned present = in ? a.vmask[i] : 0u;
    for (int j = 0; j < hs.n_hydro; ++j) {
        const ItabDev &t = its.t[j];
        if (!t.tab) continue;                                           // uniform
        // (k_subbeam_sum evaluates the columns AND the Doppler sums of the 1-D species: nothing to do
        // here for a species that does not need the ice intercept of the spectrum kernels)
        if (a.skip_res_1d && !t.two_d &&
            !(a.par_w && hs.h[j].d.psd_family == CPOL_PSD_ICE_FIELD) && !(a.vn_1d && a.vn && t.writes_vn)) continue;
        const int key = ((present >> j) & 1u) ? a.key[(long)j * n + i] : -1;
        const double *P = a.par + ((long)j * CPOL_MAX_PAR) * n + i;
        const double2 rc = key >= 0 ? a.rec[(long)j * n + i] : make_double2(-1.0, 0.0);
        const double pf = rc.x;
        const bool item = pf >= 0.0;                                    // NaN -> false
        if (t.two_d) {
            // melting species: QM x F_c(slice, fw, lambda_r), one item of the wavefront at a time
            int blk = 0;
            double u = 0.0, w = 0.0, q = 0.0;
            if (item) {
                const int pn = min((int)pf, t.n_pan - 1);
                w = 2.0 * (pf - (double)pn) - 1.0;
                const int sl = key - hs.h[j].key_base;
                const int tb = sl % t.n_t;
                u = (P[n] - t.head[2 * tb]) * t.head[2 * tb + 1];
                q = rc.y;
                blk = sl * t.n_pan + pn;
            }
            // the wavefront walks over the DISTINCT blocks of its items (16 neighbouring rays x 4
            // gates: ~6): the rows of a block are fetched once (21 values per lane) and serve
            // every item of the wavefront that lies on it
            // a.split > 1 (small launches): `split` wavefronts per tile (blockIdx.y), each takes every split-th distinct
            // block (and every split-th 1-D species below) -- with few wavefronts per SIMD the kernel lasts as long as
            // its busiest wavefront, a tile inside the melting layer with 64 items of both melting species evaluated one
            // after the other.  C3 sweep at 3 deg, 1 / 2 / 4 / 8 wavefronts per tile: 66.5 / 53 / 50.5 / 52 us; the 225-ray
            // C4 share 846 / 861 / 937 us of PSD stage and the C4 volume with 2: 3.03 -> 3.41 ms: only for small launches.
            // (Measured and dropped: key, record and wet fraction of EVERY species requested right after the validity
            // byte, 150 VGPRs: 50.5 -> 59.9 us, although the skeleton alone -- both table parts compiled out -- takes 23 us.)
            unsigned long long todo = __ballot(item);
            const int f = lane & (NFP - 1), r = lane >> 4;
            int turn = 0;
            while (todo) {
                const int lead = (int)__ffsll((long long)todo) - 1;
                const int cur = __builtin_amdgcn_readlane(blk, lead);
                unsigned long long grp = __ballot(item && blk == cur);
                todo &= ~grp;
                const bool mine = turn == (int)blockIdx.y;
                turn = turn + 1 == a.split ? 0 : turn + 1;
                if (!mine) continue;
#ifdef CPOL_LOOKUP_TRACE
                ++trace_work;
#endif
#if CPOL_ITAB2_PAIR
                itab2_walk_pairs(t.tab + (long)cur * (CPOL_ITAB2_NB * NFP) + f, grp, u, w, q, [&](int l, double acc, bool live) {
                    // (l: lane-varying between the two slots -- the item's sub-beam gate by bpermute, not readlane)
                    const long il = ((long)__shfl((int)(i >> 32), l) << 32) | (unsigned)__shfl((int)i, l);
                    const long o = (long)j * n + il;
                    if (!live || (lane & 16)) return;                       // (half 1 holds the same sum)
                    if (f < CPOL_N_SZ) a.res[o * CPOL_N_SZ + f] = acc;
                    else if (f < CPOL_N_SZ + 2 && a.vn) a.vn[o * 2 + (f - CPOL_N_SZ)] = acc;
                });
#else
                double rows[CPOL_ITAB2_QROWS];
                itab2_quarter_rows(t.tab + (long)cur * (CPOL_ITAB2_NB * NFP) + f, r, rows);
                while (grp) {
                    const int l = (int)__ffsll((long long)grp) - 1;
                    grp &= grp - 1;
                    double acc = itab2_quarter_sum(rows, r, readlane_f64(u, l), readlane_f64(w, l));
                    acc += shfl_xor_f64(acc, 16);
                    acc += shfl_xor_f64(acc, 32);
                    acc *= readlane_f64(q, l);
                    const long o = (long)j * n + (((long)__builtin_amdgcn_readlane((int)(i >> 32), l) << 32)
                                                 | (unsigned)__builtin_amdgcn_readlane((int)i, l));
                    if (lane < CPOL_N_SZ) a.res[o * CPOL_N_SZ + lane] = acc;
                    else if (lane < CPOL_N_SZ + 2 && a.vn) a.vn[o * 2 + (lane - CPOL_N_SZ)] = acc;
                }
#endif
            }
            continue;
        }
        if (!item || (j % a.split) != (int)blockIdx.y) continue;        // (the 1-D items: one lane each; the species dealt to the tile's wavefronts)
        const cpol_hydro_desc &d = hs.h[j].d;
        const bool ice = d.psd_family == CPOL_PSD_ICE_FIELD;
        const int pn = min((int)pf, t.n_pan - 1);
        const double u = 2.0 * (pf - (double)pn) - 1.0;
        const double2 *c = reinterpret_cast<const double2 *>(
            t.tab + ((long)(key - hs.h[j].key_base) * t.n_pan + pn) * NB);
        // gamma: N0 x exp(-lambda d0) x G(lambda);  ice: QM x R(lambda)  (the scale comes with the record)
        const double scale = rc.y;
        const bool want_vn = a.vn && t.writes_vn && (!a.skip_res_1d || a.vn_1d);     // uniform (k_subbeam_sum writes them otherwise)
        const bool want_n0 = ice && a.par_w;                            // uniform
        if (!a.skip_res_1d) {
            double2 v[CPOL_N_SZ / 2];
            itab1_columns(c, u, scale, v);
            double2 *o = reinterpret_cast<double2 *>(a.res + ((long)j * n + i) * CPOL_N_SZ);
#pragma unroll
            for (int f = 0; f < CPOL_N_SZ / 2; ++f) o[f] = v[f];
        }
        if (want_vn || want_n0) {
            double2 w = c[(NC - 1) * (NFP / 2) + CPOL_N_SZ / 2];
            double w2 = c[(NC - 1) * (NFP / 2) + CPOL_N_SZ / 2 + 1].x;
#pragma unroll
            for (int q = NC - 2; q >= 0; --q) {
                const double2 cq = c[q * (NFP / 2) + CPOL_N_SZ / 2];
                w.x = fma(w.x, u, cq.x);
                w.y = fma(w.y, u, cq.y);
                w2 = fma(w2, u, c[q * (NFP / 2) + CPOL_N_SZ / 2 + 1].x);
            }
            if (want_vn)
                *reinterpret_cast<double2 *>(a.vn + ((long)j * n + i) * 2) = make_double2(w.x * scale, w.y * scale);
            if (want_n0) a.par_w[((long)j * CPOL_MAX_PAR + 3) * n + i] = w2 * scale;
        }
    }
    }
}
