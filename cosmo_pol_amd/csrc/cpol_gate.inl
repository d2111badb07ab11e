// cpol_gate.inl -- the single-beam fast path: ONE kernel from the interpolated model values of a
// gate to its polarimetric variables.
//
// With one sub-beam per radial (the reference's default quadrature nh = nv = 1 is rare, but C2, C3 and
// the GPM swaths of BASELINE.json are single-beam) a sub-beam gate IS an output gate and nothing is
// accumulated across sub-beams, so the stages that the general launch sequence keeps apart -- PSD
// parameters and table positions (k_classify), evaluation of the integral tables (k_psd_lookup), sum
// over the hydrometeors and get_pol_from_sz (the per-gate half of k_final) -- can run in the registers
// of one thread: no key / record / result arrays between them (on the C2 sweep 46 MB of `res` written
// and read back, 11 MB of keys and records), one launch instead of three to five.  What is left to the
// per-ray kernel (k_final) are the three sequential float32 range scans and the sensitivity cut.
//
// Reference functions replaced: those of k_classify (melting.py:19-90, hydrometeors.py set_psd of every
// species, lut.py:336-341), of k_psd_lookup (doppler_scatter.py:246-251 through the integral tables) and of
// final_gate (doppler_scatter.py:133-134, 259-268, 400-401, 491-544, 276-281, 313-333, 472-477;
// interpolation.py:36-89).  Same device functions, same operand order: bit-identical to the general sequence
// (tests/test_gpu_parity.py runs every golden radial both ways).
//
// Items OUTSIDE the integral tables (lambda beyond the tabulated panels: a handful per volume) still need the
// integrating kernels.  The thread that meets one appends it to the work-unit list directly (one unit per item:
// no counting sort, no scan / scatter launches), marks its gate `deferred` and leaves it alone; after the
// integrating kernels k_final recomputes the deferred gates with the general per-gate function (final_gate)
// from the item records this kernel stored for exactly those gates.  A species whose fall-speed sums are
// totals over the ray (1-moment ice, hydrometeors.py:1256-1275) makes the gate that receives the total depend on
// the whole ray: its vn[] goes to k_ice_first as before and k_final recomputes that one gate per ray.

struct GateArgs {
    float *sk, *sh, *sv;          // [n_rg] operands of the three range scans (2 KDP; two-way attenuation factors)
    unsigned char *defer;         // [n_rg] 1: the gate holds an item outside the integral tables (k_final recomputes it)
    int *unit_key;                // [n_hydro * n_rg] LUT slice of every item outside the tables (one work unit each)
    int *perm;                    // [n_hydro * n_rg] its gate
    unsigned long long *totals;   // [0] items, [1] units handed to the integrating kernels (cleared by k_interp_sweep)
    double *res;                  // [n_hydro][n_rg][12] melting items (read again only for recomputed gates)
    int store_items;              // some species' fall-speed sums are per ray: vmask / key / rec / vn of EVERY gate are
                                  // stored (k_ice_first and the recomputed first-ice gate read them), not only of deferred gates
    int analytic_vn;              // Doppler scheme 1: the gamma species take their analytic fall-speed moments
};

#define CPOL_GATE1_THREADS 256
#ifndef CPOL_GATE1_ROW_UNROLL
#define CPOL_GATE1_ROW_UNROLL 2      // coefficient rows of a 1-D block requested together
#endif

// key / record / moments of the items of one gate, stored for k_final's final_gate (second pass: rare)
__device__ __forceinline__ void gate1_store_item(const ClassifyArgs &a, const ItabDev &tj, int j, long n, long i,
                                                 const ClassItem &it, bool analytic_vn)
{
    if (!it.valid) return;
    a.key[(long)j * n + i] = it.key;
    if (tj.tab) a.rec[(long)j * n + i] = make_double2(it.pf, it.scale);
    if (it.has_vn && analytic_vn && a.vn)
        *reinterpret_cast<double2 *>(a.vn + ((long)j * n + i) * 2) = make_double2(it.dv, it.dn);
}

#ifndef CPOL_GATE1_WPE
#define CPOL_GATE1_WPE 3             // wavefronts per SIMD asked of the register allocator: 170 VGPRs, so that the 2 812 wavefronts of a
                                     // 360 x 500 sweep are resident at once (C2 sweep, whole sequence: its own choice 216 VGPRs / 2
                                     // per SIMD 103 us, 3: 97.6, 4: 108.6; rows of a block requested 1 / 2 / 5 at a time: 97.8 / 97.6 / 114.5)
#endif
#if CPOL_GATE1_WPE
#define CPOL_GATE1_ATTR __attribute__((amdgpu_waves_per_eu(CPOL_GATE1_WPE, CPOL_GATE1_WPE)))
#else
#define CPOL_GATE1_ATTR
#endif
// TWO_D: a melting species is present (CPOL_GATE1=2): the 2-D walk, its 21 row registers per lane and the 28 KB of LDS that
// bring its results to their lanes are compiled into that instantiation only
// INTERP (k_interp_gate1): the gate is interpolated here as well -- grid and thread mapping of k_interp_sweep, the
// variables of the gate in the thread's column of the workgroup's LDS array instead of vals[] (as in
// k_interp_classify, cpol_fused.inl): the whole sweep is this kernel and the range scans of k_final.
template <bool TWO_D, bool INTERP>
__device__ __forceinline__ void gate1_body(const ModelDev &m, const InterpArgs &ia, float *sv, const HydroSet &hs, const ItabSet &its,
                                           const ClassifyArgs &a, const FinalArgs &f, const GateArgs &g)
{
    constexpr int NC = CPOL_ITAB1_NC, NFP = CPOL_ITAB_NFP, NB = NC * NFP;      // (1-D blocks)
    __shared__ double s_walk[TWO_D ? CPOL_GATE1_THREADS / CPOL_WAVE : 1][TWO_D ? CPOL_WAVE : 1][CPOL_N_SZ + 2];   // 2-D walk: item -> its lane
    __shared__ int s_lookup;
    if (threadIdx.x == 0) s_lookup = 0;
    __syncthreads();
    long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    bool in = i0 < a.n_sbg;
    float e_interp = 0.f;
    if (INTERP) {
        long sbg = 0;
        const int status = interp_gate<true>(m, ia, sv, sbg, e_interp);
        in = status != 3;
        i0 = sbg;
        if (in && status != 0)                       // (no model values here: NaN, as k_interp_sweep leaves them in vals[])
            for (int v = 0; v < m.n_vars; ++v) sv[v * blockDim.x] = __builtin_nanf("");
    }
    const long n = a.n_sbg;
    const long i = in ? i0 : 0;
    // the model variables of the gate: vals[var * vn + vi]
    float *const vals = INTERP ? sv : a.vals;
    const long vn = INTERP ? (long)blockDim.x : n, vi = INTERP ? 0 : i;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int ray = (int)(i / f.n_gates), gate = (int)(i % f.n_gates);
    int my_lookup = 0;

    // ---- melting.py:34-83 (as k_classify) ----
    float qms = 0.f, qmg = 0.f;
    double fws = 0.0, fwg = 0.0;
    if (a.with_melting && in) {
        float qr = vals[a.var_qr * vn + vi], qs = vals[a.var_qs * vn + vi], qg = vals[a.var_qg * vn + vi];
        float qsg = qs + qg;
        if (qr > 0.f && qsg > 0.f) {
            qms = qs + qr * (qs / qsg);
            qmg = qg + qr * (qg / qsg);
            if (qms > 0.f || qmg > 0.f) {
                vals[a.var_qr * vn + vi] = 0.f;
                vals[a.var_qs * vn + vi] = 0.f;
                vals[a.var_qg * vn + vi] = 0.f;
            }
            fws = (double)(qr * qs / qsg) / (double)qms;
            fwg = (double)(qr * qg / qsg) / (double)qmg;
        }
    }
    if (INTERP && in)                                // what later kernels (and gate_finish below) read of the gate
        for (int v = 0; v < m.n_vars; ++v)
            if ((ia.store_mask >> v) & 1u) ia.vals[(long)v * n + i] = sv[v * blockDim.x];
    const float e = !in ? 0.f : INTERP ? e_interp : a.elev[i];
    const int var_t0 = hs.h[0].d.var_t;
    const float T0 = in ? vals[var_t0 * vn + vi] : 0.f;
    float q_ahead = (in && hs.h[0].d.q_source == CPOL_Q_MODEL) ? vals[hs.h[0].d.var_q * vn + vi] : 0.f;

    float tot[CPOL_N_SZ];
#pragma unroll
    for (int c = 0; c < CPOL_N_SZ; ++c) tot[c] = 0.f;
    double mom_v = 0.0, mom_n = 0.0;        // fall-speed sums over the species present (subbeam_proj's v, nn)
    unsigned vbits = 0;
    bool deferred = false;
    const double w0 = f.sub_w[0];
    const bool want_rvel = f.RVEL != nullptr;

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
            q_ahead = vals[hs.h[j + 1].d.var_q * vn + vi];
        const ItabDev &t = its.t[j];
        ClassItem it;
        classify_item(h, t, a, vals, vn, vi, i, in, qm, fw, T0, var_t0, e, it);
        const bool want_vn = want_rvel && t.writes_vn;                      // uniform: the table carries the Doppler sums
        double2 v[CPOL_N_SZ / 2];
#pragma unroll
        for (int c = 0; c < CPOL_N_SZ / 2; ++c) v[c] = make_double2(0.0, 0.0);
        double2 wv = make_double2(0.0, 0.0);
        bool have = false;
        if (TWO_D && t.two_d) {
            // ---- melting species: the wavefront walks over the distinct 2-D blocks of its items (k_psd_lookup) ----
            int blk = 0;
            double u = 0.0, w = 0.0, q = 0.0;
            if (it.lookup) {
                const int pn = min((int)it.pf, t.n_pan - 1);
                w = 2.0 * (it.pf - (double)pn) - 1.0;
                const int sl = it.key - h.key_base;
                const int tb = sl % t.n_t;
                u = (it.p1 - t.head[2 * tb]) * t.head[2 * tb + 1];
                q = it.scale;
                blk = sl * t.n_pan + pn;
            }
            unsigned long long todo = __ballot(it.lookup);
            const int fn = lane & (NFP - 1), r = lane >> 4;
            while (todo) {
                const int lead = (int)__ffsll((long long)todo) - 1;
                const int cur = __builtin_amdgcn_readlane(blk, lead);
                unsigned long long grp = __ballot(it.lookup && blk == cur);
                todo &= ~grp;
                double rows[CPOL_ITAB2_QROWS];
                itab2_quarter_rows(t.tab + (long)cur * (CPOL_ITAB2_NB * NFP) + fn, r, rows);
                while (grp) {
                    const int l = (int)__ffsll((long long)grp) - 1;
                    grp &= grp - 1;
                    double acc = itab2_quarter_sum(rows, r, readlane_f64(u, l), readlane_f64(w, l));
                    acc += shfl_xor_f64(acc, 16);
                    acc += shfl_xor_f64(acc, 32);
                    acc *= readlane_f64(q, l);
                    if (lane < CPOL_N_SZ + 2) s_walk[wave][l][lane] = acc;
                    const long o = (long)j * n + (((long)__builtin_amdgcn_readlane((int)(i >> 32), l) << 32)
                                                 | (unsigned)__builtin_amdgcn_readlane((int)i, l));
                    if (lane < CPOL_N_SZ) g.res[o * CPOL_N_SZ + lane] = acc;      // (for a recomputed gate)
                    else if (lane < CPOL_N_SZ + 2 && a.vn) a.vn[o * 2 + (lane - CPOL_N_SZ)] = acc;        // (k_ice_first, a recomputed gate)
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (it.lookup) {
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ / 2; ++c) v[c] = make_double2(s_walk[wave][lane][2 * c], s_walk[wave][lane][2 * c + 1]);
                wv = make_double2(s_walk[wave][lane][CPOL_N_SZ], s_walk[wave][lane][CPOL_N_SZ + 1]);
                have = true;
            }
            __builtin_amdgcn_wave_barrier();
        } else if (t.tab && it.lookup) {
            // ---- 1-D block: the lane gathers the rows of its (slice, panel) block (itab1_columns) ----
            const int pn = min((int)it.pf, t.n_pan - 1);
            const double u = 2.0 * (it.pf - (double)pn) - 1.0;
            const double2 *blk = reinterpret_cast<const double2 *>(t.tab + ((long)(it.key - h.key_base) * t.n_pan + pn) * NB);
#pragma unroll
            for (int c = 0; c < CPOL_N_SZ / 2; ++c) v[c] = blk[(NC - 1) * (NFP / 2) + c];
            if (want_vn) wv = blk[(NC - 1) * (NFP / 2) + CPOL_N_SZ / 2];
#pragma unroll CPOL_GATE1_ROW_UNROLL
            for (int q = NC - 2; q >= 0; --q) {
#pragma unroll
                for (int c = 0; c < CPOL_N_SZ / 2; ++c) {
                    const double2 cq = blk[q * (NFP / 2) + c];
                    v[c].x = fma(v[c].x, u, cq.x);
                    v[c].y = fma(v[c].y, u, cq.y);
                }
                if (want_vn) {
                    const double2 cq = blk[q * (NFP / 2) + CPOL_N_SZ / 2];
                    wv.x = fma(wv.x, u, cq.x);
                    wv.y = fma(wv.y, u, cq.y);
                }
            }
#pragma unroll
            for (int c = 0; c < CPOL_N_SZ / 2; ++c) { v[c].x *= it.scale; v[c].y *= it.scale; }
            wv.x *= it.scale; wv.y *= it.scale;
            if (want_vn && a.vn && g.store_items)
                *reinterpret_cast<double2 *>(a.vn + ((long)j * n + i) * 2) = wv;
            have = true;
        }
        my_lookup += it.lookup ? 1 : 0;
        if (it.valid) vbits |= 1u << j;
        if (it.valid && !it.lookup) {
            // ---- an item outside the integral table: a work unit of its own for the integrating kernels ----
            deferred = true;
            const unsigned long long idx = atomicAdd(g.totals + 1, 1ull);
            atomicAdd(g.totals, 1ull);
            g.unit_key[idx] = it.key;
            g.perm[idx] = (int)i;
            double *P = a.par + ((long)j * CPOL_MAX_PAR) * n + i;
            P[0] = it.p0;
            P[n] = it.p1;
            if (h.n_par >= 3) P[2 * n] = it.p2;
        }
        // ---- sub-beam accumulation of ONE sub-beam (nansum([NaN, y]) stored as float32), sum over the species ----
        float acc[CPOL_N_SZ];
#pragma unroll
        for (int c = 0; c < CPOL_N_SZ; ++c) {
            double y = ((c & 1) ? v[c / 2].y : v[c / 2].x) * w0;
            if (!(y == y)) y = 0.0;
            acc[c] = have ? (float)(0.0 + y) : 0.f;                          // (an absent species adds 0: nansum)
        }
#pragma unroll
        for (int c = 0; c < CPOL_N_SZ; ++c) tot[c] = (j == 0) ? acc[c] : tot[c] + acc[c];
        // ---- fall-speed moments of the species (subbeam_proj) ----
        if (want_rvel && it.valid && f.vsrc[j] == 1) {
            double vj, nj;
            if (want_vn || (TWO_D && t.two_d)) { vj = wv.x; nj = wv.y; }
            else { vj = it.dv; nj = it.dn; }
            if (vj == vj) mom_v += vj;
            if (nj == nj) mom_n += nj;
        }
        // (a species summed over the ray, vsrc == 2, adds nothing here: the gate that receives its total is recomputed)
        if (g.store_items) gate1_store_item(a, t, j, n, i, it, g.analytic_vn != 0);
    }

    // ---- items of a deferred gate, stored for final_gate (rare: second pass over the species) ----
    if (__ballot(deferred && !g.store_items)) {
        float q2 = (in && hs.h[0].d.q_source == CPOL_Q_MODEL) ? vals[hs.h[0].d.var_q * vn + vi] : 0.f;
        for (int j = 0; j < hs.n_hydro; ++j) {
            const HydroDev &h = hs.h[j];
            const cpol_hydro_desc &d = h.d;
            float qm = 0.f;
            double fw = 0.0;
            if (in) {
                if (d.q_source == CPOL_Q_MODEL) qm = q2;
                else if (d.q_source == CPOL_Q_MELT_SNOW) { qm = qms; fw = fws; }
                else { qm = qmg; fw = fwg; }
            }
            if (in && j + 1 < hs.n_hydro && hs.h[j + 1].d.q_source == CPOL_Q_MODEL)
                q2 = vals[hs.h[j + 1].d.var_q * vn + vi];
            ClassItem it;
            classify_item(h, its.t[j], a, vals, vn, vi, i, in && deferred, qm, fw, T0, var_t0, e, it);
            if (deferred) {
                gate1_store_item(a, its.t[j], j, n, i, it, g.analytic_vn != 0);
                // (the Doppler sums of a table item of this gate: evaluated once more from its block)
                const ItabDev &t = its.t[j];
                if (want_rvel && t.writes_vn && t.tab && !t.two_d && it.lookup && a.vn) {
                    const int pn = min((int)it.pf, t.n_pan - 1);
                    const double u = 2.0 * (it.pf - (double)pn) - 1.0;
                    const double2 *blk = reinterpret_cast<const double2 *>(t.tab + ((long)(it.key - h.key_base) * t.n_pan + pn) * NB);
                    double2 wv = blk[(NC - 1) * (NFP / 2) + CPOL_N_SZ / 2];
                    for (int q = NC - 2; q >= 0; --q) {
                        const double2 cq = blk[q * (NFP / 2) + CPOL_N_SZ / 2];
                        wv.x = fma(wv.x, u, cq.x);
                        wv.y = fma(wv.y, u, cq.y);
                    }
                    *reinterpret_cast<double2 *>(a.vn + ((long)j * n + i) * 2) = make_double2(wv.x * it.scale, wv.y * it.scale);
                }
            }
        }
    }
    if (in && (g.store_items || deferred)) a.vmask[i] = (unsigned char)vbits;
    if (in) g.defer[i] = deferred ? 1 : 0;

    // ---- items on integral tables (cpol_counters) ----
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) my_lookup += __shfl_xor(my_lookup, off);
    if (lane == 0 && my_lookup) atomicAdd(&s_lookup, my_lookup);
    __syncthreads();
    if (threadIdx.x == 0) count_table_items(a.n_lookup, s_lookup);

    // ---- get_pol_from_sz, RVEL, mask, model variables; the operands of the range scans ----
    if (!in || deferred) return;
    float k2, fh, fv;
    gate_finish(f, ray, gate, tot, want_rvel, mom_v, mom_n, 0.0, k2, fh, fv);
    g.sk[i] = k2;
    g.sh[i] = fh;
    g.sv[i] = fv;
}

template <bool TWO_D>
__global__ __launch_bounds__(CPOL_GATE1_THREADS) CPOL_GATE1_ATTR void k_gate1(HydroSet hs, ItabSet its, ClassifyArgs a, FinalArgs f, GateArgs g)
{
    gate1_body<TWO_D, false>(ModelDev{}, InterpArgs{}, nullptr, hs, its, a, f, g);
}

// grid = (n_rays, ceil(n_gates / 256)), dynamic LDS = n_vars * 256 floats (the mapping of k_interp_sweep with one sub-beam)
template <bool TWO_D>
__global__ __launch_bounds__(CPOL_GATE1_THREADS) CPOL_GATE1_ATTR void k_interp_gate1(ModelDev m, InterpArgs ia, HydroSet hs, ItabSet its,
                                                                                    ClassifyArgs a, FinalArgs f, GateArgs g)
{
    extern __shared__ float s_vals[];                 // [n_vars][blockDim.x]
    clear_counters(ia.zero_buf, ia.zero_n, ia.zero_buf2, ia.zero_n2);      // (the NEXT sweep's set)
    gate1_body<TWO_D, true>(m, ia, s_vals + threadIdx.x, hs, its, a, f, g);
}


// k_gate1_species: the same work with ONE WAVEFRONT PER SPECIES -- a workgroup takes 64 consecutive gates, its wavefront j
// the hydrometeor j of those gates (classification, block gather, Horner chains); the per-species columns and fall-speed
// sums meet in LDS and wavefront 0 adds them in species order (the float32 sums of k_gate1, term by term) and finishes
// the gates.  k_gate1 is bound by the latency of ONE thread's chain through all species at 3 wavefronts per SIMD; here a
// thread's chain is one species long, the species is wave-uniform (no divergent rule switch) and the registers of a
// thread hold one block instead of the bookkeeping of a loop over species.  Not for melting species (k_gate1<true>) nor
// when a species' fall-speed sums are per ray (store_items): those keep k_gate1.
// grid = ceil(n_rg / 64), block = 64 * n_hydro, dynamic LDS = n_hydro * 64 * GATE1S_BYTES
#define GATE1S_BYTES (CPOL_N_SZ * 4 + 8 + 8 + 4)
#ifndef CPOL_GATE1S_ROW_UNROLL
#define CPOL_GATE1S_ROW_UNROLL 2     // coefficient rows of the block requested together
#endif
#ifndef CPOL_GATE1S_WPE
#define CPOL_GATE1S_WPE 0
#endif
#if CPOL_GATE1S_WPE
#define CPOL_GATE1S_ATTR __attribute__((amdgpu_waves_per_eu(CPOL_GATE1S_WPE, CPOL_GATE1S_WPE)))
#else
#define CPOL_GATE1S_ATTR
#endif
__global__ __launch_bounds__(64 * CPOL_MAX_HYDRO) CPOL_GATE1S_ATTR void k_gate1_species(HydroSet hs, ItabSet its, ClassifyArgs a, FinalArgs f, GateArgs g)
{
    constexpr int NC = CPOL_ITAB1_NC, NFP = CPOL_ITAB_NFP, NB = NC * NFP;
    extern __shared__ unsigned char s_raw[];
    const int n_h = hs.n_hydro;
    float *s_acc = reinterpret_cast<float *>(s_raw);                                   // [n_h][12][64]
    double *s_mv = reinterpret_cast<double *>(s_raw + (size_t)n_h * CPOL_N_SZ * 64 * 4);   // [n_h][64]
    double *s_mn = s_mv + n_h * 64;                                                    // [n_h][64]
    unsigned *s_flag = reinterpret_cast<unsigned *>(s_mn + n_h * 64);                  // [n_h][64]: 1 valid, 2 off the tables, 4 moments
    __shared__ int s_lookup;
    if (threadIdx.x == 0) s_lookup = 0;
    const int lane = lane_id();
    const int j = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                    // the wavefront's hydrometeor
    const long i0 = (long)blockIdx.x * 64 + lane;
    const bool in = i0 < a.n_sbg;
    const long n = a.n_sbg;
    const long i = in ? i0 : 0;
    const HydroDev &h = hs.h[j];
    const cpol_hydro_desc &d = h.d;
    const ItabDev &t = its.t[j];
    const bool want_rvel = f.RVEL != nullptr;
    const bool want_vn = want_rvel && t.writes_vn;                                     // uniform: the table carries the Doppler sums
    const double w0 = f.sub_w[0];

    const float e = in ? a.elev[i] : 0.f;
    const float T = in ? a.vals[d.var_t * n + i] : 0.f;
    const float qm = in ? a.vals[d.var_q * n + i] : 0.f;                               // (q_source == CPOL_Q_MODEL: no melting species here)
    ClassItem it;
    classify_item(h, t, a, a.vals, n, i, i, in, qm, 0.0, T, d.var_t, e, it);
    double2 v[CPOL_N_SZ / 2];
#pragma unroll
    for (int c = 0; c < CPOL_N_SZ / 2; ++c) v[c] = make_double2(0.0, 0.0);
    double2 wv = make_double2(0.0, 0.0);
    bool have = false;
    if (t.tab && it.lookup) {
        // ---- 1-D block: the lane gathers the rows of its (slice, panel) block (as k_gate1) ----
        const int pn = min((int)it.pf, t.n_pan - 1);
        const double u = 2.0 * (it.pf - (double)pn) - 1.0;
        const double2 *blk = reinterpret_cast<const double2 *>(t.tab + ((long)(it.key - h.key_base) * t.n_pan + pn) * NB);
#pragma unroll
        for (int c = 0; c < CPOL_N_SZ / 2; ++c) v[c] = blk[(NC - 1) * (NFP / 2) + c];
        if (want_vn) wv = blk[(NC - 1) * (NFP / 2) + CPOL_N_SZ / 2];
#pragma unroll CPOL_GATE1S_ROW_UNROLL
        for (int q = NC - 2; q >= 0; --q) {
#pragma unroll
            for (int c = 0; c < CPOL_N_SZ / 2; ++c) {
                const double2 cq = blk[q * (NFP / 2) + c];
                v[c].x = fma(v[c].x, u, cq.x);
                v[c].y = fma(v[c].y, u, cq.y);
            }
            if (want_vn) {
                const double2 cq = blk[q * (NFP / 2) + CPOL_N_SZ / 2];
                wv.x = fma(wv.x, u, cq.x);
                wv.y = fma(wv.y, u, cq.y);
            }
        }
#pragma unroll
        for (int c = 0; c < CPOL_N_SZ / 2; ++c) { v[c].x *= it.scale; v[c].y *= it.scale; }
        wv.x *= it.scale; wv.y *= it.scale;
        have = true;
    }
    int my_lookup = it.lookup ? 1 : 0;
    const bool off_table = it.valid && !it.lookup;
    if (off_table) {
        // ---- an item outside the integral table: a work unit of its own for the integrating kernels (as k_gate1) ----
        const unsigned long long idx = atomicAdd(g.totals + 1, 1ull);
        atomicAdd(g.totals, 1ull);
        g.unit_key[idx] = it.key;
        g.perm[idx] = (int)i;
        double *P = a.par + ((long)j * CPOL_MAX_PAR) * n + i;
        P[0] = it.p0;
        P[n] = it.p1;
        if (h.n_par >= 3) P[2 * n] = it.p2;
    }
    // ---- this species' term of the ONE sub-beam (nansum([NaN, y]) stored as float32) and its fall-speed moments ----
#pragma unroll
    for (int c = 0; c < CPOL_N_SZ; ++c) {
        double y = ((c & 1) ? v[c / 2].y : v[c / 2].x) * w0;
        if (!(y == y)) y = 0.0;
        s_acc[(j * CPOL_N_SZ + c) * 64 + lane] = have ? (float)(0.0 + y) : 0.f;
    }
    const bool moments = want_rvel && it.valid && f.vsrc[j] == 1;
    double vj = 0.0, nj = 0.0;
    if (moments) {
        if (want_vn) { vj = wv.x; nj = wv.y; }
        else { vj = it.dv; nj = it.dn; }
    }
    s_mv[j * 64 + lane] = vj;
    s_mn[j * 64 + lane] = nj;
    s_flag[j * 64 + lane] = (it.valid ? 1u : 0u) | (off_table ? 2u : 0u) | (moments ? 4u : 0u);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) my_lookup += __shfl_xor(my_lookup, off);
    __syncthreads();                                  // (s_lookup = 0 is visible; the species' terms are in LDS)
    if (lane == 0 && my_lookup) atomicAdd(&s_lookup, my_lookup);

    // ---- the gate: deferred if any species is off its table; the items of a deferred gate are stored for final_gate ----
    unsigned vbits = 0;
    bool deferred = false;
    for (int q = 0; q < n_h; ++q) {
        const unsigned fl = s_flag[q * 64 + lane];
        vbits |= (fl & 1u) << q;
        deferred = deferred || (fl & 2u);
    }
    if (deferred) {
        gate1_store_item(a, t, j, n, i, it, g.analytic_vn != 0);
        if (want_vn && t.tab && !t.two_d && it.lookup && a.vn)
            *reinterpret_cast<double2 *>(a.vn + ((long)j * n + i) * 2) = wv;
    }
    __syncthreads();
    if (threadIdx.x == 0) count_table_items(a.n_lookup, s_lookup);
    if (j != 0) return;
    // ---- wavefront 0: sum over the species in order, get_pol_from_sz, RVEL, mask; the operands of the range scans ----
    if (in && deferred) a.vmask[i] = (unsigned char)vbits;
    if (in) g.defer[i] = deferred ? 1 : 0;
    if (!in || deferred) return;
    float tot[CPOL_N_SZ];
    double mom_v = 0.0, mom_n = 0.0;
    for (int q = 0; q < n_h; ++q) {
#pragma unroll
        for (int c = 0; c < CPOL_N_SZ; ++c) {
            const float acc = s_acc[(q * CPOL_N_SZ + c) * 64 + lane];
            tot[c] = (q == 0) ? acc : tot[c] + acc;
        }
        if (s_flag[q * 64 + lane] & 4u) {
            const double vq = s_mv[q * 64 + lane], nq = s_mn[q * 64 + lane];
            if (vq == vq) mom_v += vq;
            if (nq == nq) mom_n += nq;
        }
    }
    const int ray = (int)(i / f.n_gates), gate = (int)(i % f.n_gates);
    float k2, fh, fv;
    gate_finish(f, ray, gate, tot, want_rvel, mom_v, mom_n, 0.0, k2, fh, fv);
    g.sk[i] = k2;
    g.sh[i] = fh;
    g.sv[i] = fv;
}
