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
    int *ticket;                  // [n_rays] k_gate1_ray: workgroups of the ray that have finished (0 between sweeps)
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
#if CPOL_ITAB2_PAIR
                itab2_walk_pairs(t.tab + (long)cur * (CPOL_ITAB2_NB * NFP) + fn, grp, u, w, q, [&](int l, double acc, bool live) {
                    const long il = ((long)__shfl((int)(i >> 32), l) << 32) | (unsigned)__shfl((int)i, l);
                    const long o = (long)j * n + il;
                    if (!live || (lane & 16)) return;
                    if (fn < CPOL_N_SZ + 2) s_walk[wave][l][fn] = acc;
                    if (fn < CPOL_N_SZ) g.res[o * CPOL_N_SZ + fn] = acc;          // (for a recomputed gate)
                    else if (fn < CPOL_N_SZ + 2 && a.vn) a.vn[o * 2 + (fn - CPOL_N_SZ)] = acc;   // (k_ice_first, a recomputed gate)
                });
#else
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
#endif
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
// ---- an item outside its integral table, integrated by the wavefront that meets it (k_gate1_ray) ----
// The wavefront plays the integrating kernels' workgroup (psd_body): its eight wavefronts each sum a chunk of the diameter
// bins, column by column.  Here lane (w, c) = (chunk, column) sums column c over the bins of chunk w -- 4 chunks x 12
// columns at a time, two passes -- with the same statements in the same order (GAMMA_UNIFORM: exp(-lambda D_k) by the
// geometric recurrence from one exp at the chunk start; GAMMA_EXP: one exp per bin), and the eight partial sums of a column
// are joined as there: ((w0+w4)+(w2+w6)) + ((w1+w5)+(w3+w7)), resp. in wave order -- the 12 columns are the bits
// k_psd_uniform / k_psd<GAMMA_EXP> leave in res[] (tests/test_gpu_edges.py compares the launch sequences bit for bit).
// One accumulator per lane: the function is inlined and costs the kernel no register it does not have anyway, and no
// call.  (Round 5, first form: lanes 0..7 = the chunks with 12 accumulators each, behind a real call -- 126 VGPRs instead of
// 97, and a kernel that calls sets up a stack: c2 with three sweeps side by side ran 11 % slower for a path that is never
// taken there, 4.69-4.84 against 5.30e9 gates/s.)  Items outside the tables are a handful per volume (lambda beyond the
// last panel: a mass density of 1e-17 kg m-3), ~40 us of one wavefront each; the host takes this path only when every slot
// is a gamma-family species without Doppler-scheme-2 sums whose table kept (all but the tail of) its panels.
// `lam`, `n0`, `slice_index`: wave-uniform.  The 12 columns (x dD) are left in out[] (LDS).
__device__ __forceinline__ void integrate_gamma_item_wave(const double *table, const double *pre, const double *dnu, const double *aux,
                                                          int n_d, int uniform_grid, double dD, int slice_index,
                                                          double lam, double n0, double *out /* LDS, [CPOL_N_SZ] */)
{
    const int lane = lane_id();
    constexpr int NW = 8;                                  // CPOL_PSD_WAVES = CPOL_PSD_WAVES_U = 8 chunks
    static_assert(CPOL_PSD_WAVES == 8 && CPOL_PSD_WAVES_U == 8, "integrate_gamma_item_wave mirrors the 8-wave split of psd_body");
    static_assert(4 * CPOL_N_SZ <= CPOL_WAVE, "four chunks x 12 columns per pass");
    const int chunk = (n_d + NW - 1) / NW;
    const int wl = lane / CPOL_N_SZ, c = lane - wl * CPOL_N_SZ;         // chunk of the pass (0..3; 4, 5: idle lanes), column
    const bool on = wl < 4;
    const double *slice = table + (long)slice_index * n_d * CPOL_N_SZ;
    double part[2];                                        // column c over chunk wl (pass 0) and wl + 4 (pass 1)
    if (uniform_grid) {
        // (psd_body, PSD_MODE_GAMMA_UNIFORM)
        const double hstep = aux[0];
        const double r0 = exp(-(lam * hstep));
        const double *pq = aux + 1 + n_d;                       // [k][4] = (pre, q1, q2, .)
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const int w = (on ? wl : 0) + 4 * pass;
            const int k0 = w * chunk, k1 = on ? min(k0 + chunk, n_d) : k0;
            double A0 = n0 * exp(-(lam * dnu[k0 < n_d ? k0 : 0]));
            double acc = 0.0;
#pragma unroll 1
            for (int k = k0; k < k1; ++k) {
                const double pk = pq[4 * k], q1 = pq[4 * k + 1], q2 = pq[4 * k + 2];
                const double nk = A0 * fma(lam, fma(lam, q2, -q1), pk);
                A0 *= r0;
                acc = fma(slice[(long)k * CPOL_N_SZ + c], nk, acc);
            }
            if (pass == 0) part[0] = acc; else part[1] = acc;
        }
    } else {
        // (psd_body, PSD_MODE_GAMMA_EXP)
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const int w = (on ? wl : 0) + 4 * pass;
            const int k0 = w * chunk, k1 = on ? min(k0 + chunk, n_d) : k0;
            double acc = 0.0;
#pragma unroll 1
            for (int k = k0; k < k1; ++k) {
                const double nk = (n0 * pre[k]) * exp(-(lam * dnu[k]));
                acc = fma(slice[(long)k * CPOL_N_SZ + c], nk, acc);
            }
            if (pass == 0) part[0] = acc; else part[1] = acc;
        }
    }
    double x;
    if (uniform_grid) {
        // the tree of the kernel: (w += w + 4), (w += w + 2), (w0 += w1)
        x = part[0] + part[1];
        x += shfl_f64(x, (lane + 2 * CPOL_N_SZ) & 63);
        x += shfl_f64(x, (lane + CPOL_N_SZ) & 63);
    } else {
        // wave order: ((((((w0 + w1) + w2) + w3) + w4) + w5) + w6) + w7, from 0.0
        x = 0.0;
#pragma unroll
        for (int q = 0; q < NW; ++q) x += shfl_f64(part[q >> 2], ((q & 3) * CPOL_N_SZ + c) & 63);
    }
    if (lane < CPOL_N_SZ) out[lane] = x * dD;
    __builtin_amdgcn_wave_barrier();
}

// RAY = false: k_gate1_species as described above (grid = ceil(n_rg / 64)).
// RAY = true (k_gate1_ray): the workgroup's 64 gates are gates [64 bx, 64 bx + 64) of ray `by` (grid = (ceil(n_gates / 64),
// n_rays)), an item outside its table is integrated on the spot (integrate_gamma_item_wave: no deferred gates, no
// integrating launch behind this kernel), and the workgroup that finishes LAST of its ray -- a ticket per ray, taken behind
// a device-scope fence -- runs the ray's three sequential float32 range scans, PHIDP, the attenuated ZDR and the
// sensitivity cut (the second half of k_final): the whole sweep is k_interp_sweep + this kernel.
#ifndef CPOL_EXP_ONE_BLOCK
#define CPOL_EXP_ONE_BLOCK 0
#endif
#ifndef CPOL_GATE1_TILE_SWIZZLE
#define CPOL_GATE1_TILE_SWIZZLE 1  // k_gate1_ray: workgroup (x, y) takes gate tile (x + y) mod n_tiles of ray y (0: tile x).  Workgroups go to the 8 XCDs
                                   // round robin by their linear index, x fastest: with 8 tiles per 500-gate ray tile x of EVERY ray landed on XCD x -- and
                                   // the tiles differ threefold in work (the first two hold no hydrometeor at 1 deg, tile 4 the melting region): the XCDs of the
                                   // empty tiles were idle after 14.6 us of a 31.7-us launch (tools/gate1_trace.py).  Rotated, every XCD ends within 2 us of 27.4
#endif
#ifndef CPOL_GATE1_LAST_WAVE
#define CPOL_GATE1_LAST_WAVE 1    // k_gate1_ray: the wavefront that finishes last finishes the gates, the others leave early (0: two workgroup barriers, wavefront 0 finishes)
#endif
#ifndef CPOL_GATE1_LDS
#define CPOL_GATE1_LDS 0          // 1 (build knob, measured and rejected in round 6): the single-beam species kernels take the coefficient blocks of a
                                  // wavefront's gates through LDS, each distinct block once, instead of every lane gathering its own.  Same bits
                                  // (tests/test_gpu_edges.py, test_gpu_parity.py, test_gpu_headline.py pass with it), but the wavefront's 4-6
                                  // distinct blocks become 4-6 SERIAL global -> LDS round trips with a few lanes busy in each Horner round:
                                  // k_gate1_ray 36 -> 81 us, the pipelined C2 sweep 35 -> 70 us (profiles/r6_variants.txt, item 7)
#endif
// dynamic LDS of k_gate1_species / k_gate1_ray behind the species' terms: two block buffers per wavefront
#define GATE1S_BLK_BYTES (CPOL_GATE1_LDS ? 2 * ((CPOL_ITAB1_NC * CPOL_ITAB_NFP / 2) > CPOL_WAVE ? (CPOL_ITAB1_NC * CPOL_ITAB_NFP / 2) : CPOL_WAVE) * 16 : 0)
template <bool RAY, bool TICKET>
__device__ __forceinline__ void gate1_species_body(const HydroSet &hs, const ItabSet &its, const ClassifyArgs &a, const FinalArgs &f,
                                                   const GateArgs &g, const ScanRayArgs &r)
{
    constexpr int NC = CPOL_ITAB1_NC, NFP = CPOL_ITAB_NFP, NB = NC * NFP;
    extern __shared__ unsigned char s_raw[];
    const int n_h = hs.n_hydro;
    float *s_acc = reinterpret_cast<float *>(s_raw);                                   // [n_h][12][64]
    double *s_mv = reinterpret_cast<double *>(s_raw + (size_t)n_h * CPOL_N_SZ * 64 * 4);   // [n_h][64]
    double *s_mn = s_mv + n_h * 64;                                                    // [n_h][64]
    unsigned *s_flag = reinterpret_cast<unsigned *>(s_mn + n_h * 64);                  // [n_h][64]: 1 valid, 2 off the tables, 4 moments
    __shared__ int s_lookup;
    __shared__ int s_last;
    __shared__ int s_done;                       // LAST_WAVE: wavefronts whose terms are in LDS
    // LAST_WAVE (k_gate1_ray, round 6): no workgroup barrier behind the species' work.  60 % of the wavefronts of a C2 sweep hold no
    // item at all (a species absent from the tile's 64 gates) and used to sit at the barrier -- each holding one of the 4 wave
    // slots per SIMD the kernel's registers allow -- until the slowest wavefront of their workgroup had gathered and evaluated
    // its blocks.  Now a wavefront that has put its terms into LDS adds to a counter and LEAVES; the one whose add comes last
    // finishes the 64 gates (the sum over the species in order, get_pol_from_sz, RVEL, mask: the same statements on the same
    // LDS values, whoever runs them).  The only barrier left stands at the start, where every wavefront arrives at once.
    constexpr bool LAST_WAVE = RAY && !TICKET && CPOL_GATE1_LAST_WAVE;
    if (threadIdx.x == 0) { s_lookup = 0; s_done = 0; }
    if (LAST_WAVE) __syncthreads();
    const int lane = lane_id();
    const int j = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                    // the wavefront's hydrometeor
    const int ray_b = RAY ? (int)blockIdx.y : 0;
#if CPOL_GATE1_TILE_SWIZZLE
    // (workgroup (x, y) takes gate tile (x + y) mod n_tiles of ray y, so that every XCD -- workgroups go to the XCDs round robin
    // by their linear index, x fastest -- sees every range of the ray instead of ONE gate tile of all rays: see the knob)
    const int gate_b = (RAY ? (int)((blockIdx.x + blockIdx.y) % gridDim.x) : (int)blockIdx.x) * 64 + lane;
#else
    const int gate_b = (int)blockIdx.x * 64 + lane;
#endif
    const long i0 = RAY ? (long)ray_b * f.n_gates + gate_b : (long)blockIdx.x * 64 + lane;
    const bool in = RAY ? gate_b < f.n_gates : i0 < a.n_sbg;
    const long n = a.n_sbg;
    const long i = in ? i0 : 0;
    const HydroDev &h = hs.h[j];
    const cpol_hydro_desc &d = h.d;
    const ItabDev &t = its.t[j];
    const bool want_rvel = f.RVEL != nullptr;
    const bool want_vn = want_rvel && t.writes_vn;                                     // uniform: the table carries the Doppler sums
    const double w0 = f.sub_w[0];

    // (round 6) k_interp_sweep's word of this tile: bit q clear = no gate of the tile has a positive mass density of slot q, so the slot
    // has no item here (classify_item: valid needs qm > 0).  The wavefront of such a slot -- 60 % of them on the C2 sweep, 2.4 us of
    // life each for two loads and 15 LDS columns of zeros -- loads nothing, writes nothing, takes its ticket and leaves; the
    // finishing wavefront reads the same word and leaves the slot's LDS columns alone (x + 0.0f = x: the same bits).
    unsigned pmask = ~0u;
#if CPOL_GATE1_PRESENT
    if (LAST_WAVE && a.present)
        pmask = a.present[(long)ray_b * gridDim.x + (gate_b >> 6)];                    // (wave-uniform: one scalar load)
#endif
    const bool here = (pmask >> j) & 1u;                                               // (wave-uniform)
    const bool inl = in && here;
    const float e = inl ? a.elev[i] : 0.f;
    const float T = inl ? a.vals[d.var_t * n + i] : 0.f;
    const float qm = inl ? a.vals[d.var_q * n + i] : 0.f;                              // (q_source == CPOL_Q_MODEL: no melting species here)
#ifdef CPOL_SUBSUM_TRACE
    // (-DCPOL_SUBSUM_TRACE build, tools/gate1_trace.py: the phases of every wavefront of k_gate1_ray on the 100-MHz clock)
    unsigned long long g1t[6], g1_valid = 0, g1_hw = 0;
    g1t[0] = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    g1t[1] = wall_clock64();                           // the gate's model values have arrived
#endif
    ClassItem it;
    classify_item(h, t, a, a.vals, n, i, i, inl, qm, 0.0, T, d.var_t, e, it);
#ifdef CPOL_SUBSUM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    g1t[2] = wall_clock64();                           // PSD parameters, table position
#endif
    double2 v[CPOL_N_SZ / 2];
#pragma unroll
    for (int c = 0; c < CPOL_N_SZ / 2; ++c) v[c] = make_double2(0.0, 0.0);
    double2 wv = make_double2(0.0, 0.0);
    bool have = false;
#if CPOL_GATE1_LDS
    {
        // ---- 1-D blocks through LDS (round 6): the 64 gates of a wavefront -- neighbours along one ray, one species -- sit on 4 to 6
        // distinct (slice, panel) blocks on average (12 at most on the C2 sweep), yet every lane gathered its own 77 double2.  Here
        // the wavefront walks its DISTINCT blocks (k_subbeam_sum_lds's scheme): the 88 16-byte pieces of a block go straight from
        // global memory into LDS (two global_load_lds_dwordx4), the lanes on that block run their Horner chains on coefficients all
        // read from ONE address (ds_read_b128 broadcast), while the pieces of the next block are on their way into the second
        // buffer.  Same coefficients, same order of the chains: the same bits as the gather.
        constexpr int PIECES = NC * NFP / 2;                 // 16-byte pieces of a block (88 at degree 10)
        constexpr int PBUF = PIECES > CPOL_WAVE ? PIECES : CPOL_WAVE;
        constexpr int REST = PIECES - CPOL_WAVE;
        double2 *my_blk = reinterpret_cast<double2 *>(s_raw + (size_t)n_h * 64 * GATE1S_BYTES) + (size_t)j * 2 * PBUF;
        const bool on_tab = t.tab && it.lookup;
        const int pn = on_tab ? min((int)it.pf, t.n_pan - 1) : 0;
        const double u = 2.0 * (it.pf - (double)pn) - 1.0;
        const int blk_id = on_tab ? (it.key - h.key_base) * t.n_pan + pn : -1;
        auto request = [&](int blk, int bufi) {
            const double2 *src = reinterpret_cast<const double2 *>(t.tab + (long)blk * NB);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + (PIECES >= CPOL_WAVE ? lane : min(lane, PIECES - 1))),
                                             (void __attribute__((address_space(3))) *)(my_blk + bufi * PBUF), 16, 0, 0);
            if (lane < REST)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + CPOL_WAVE + lane),
                                                 (void __attribute__((address_space(3))) *)(my_blk + bufi * PBUF + CPOL_WAVE), 16, 0, 0);
        };
        unsigned long long todo = __builtin_amdgcn_ballot_w64(on_tab);
        if (todo) {
            int cur = 0;
            int b = __builtin_amdgcn_readlane(blk_id, __ffsll((long long)todo) - 1);
            request(b, 0);
            while (todo) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const bool mine = blk_id == b;
                todo &= ~__builtin_amdgcn_ballot_w64(mine);
                if (todo) {
                    b = __builtin_amdgcn_readlane(blk_id, __ffsll((long long)todo) - 1);
                    request(b, cur ^ 1);                    // in flight during the chains below
                }
                const double2 *sb = my_blk + cur * PBUF;
                if (mine) {
#pragma unroll
                    for (int c = 0; c < CPOL_N_SZ / 2; ++c) v[c] = sb[(NC - 1) * (NFP / 2) + c];
                    if (want_vn) wv = sb[(NC - 1) * (NFP / 2) + CPOL_N_SZ / 2];
#pragma unroll CPOL_GATE1S_ROW_UNROLL
                    for (int q = NC - 2; q >= 0; --q) {
#pragma unroll
                        for (int c = 0; c < CPOL_N_SZ / 2; ++c) {
                            const double2 cq = sb[q * (NFP / 2) + c];
                            v[c].x = fma(v[c].x, u, cq.x);
                            v[c].y = fma(v[c].y, u, cq.y);
                        }
                        if (want_vn) {
                            const double2 cq = sb[q * (NFP / 2) + CPOL_N_SZ / 2];
                            wv.x = fma(wv.x, u, cq.x);
                            wv.y = fma(wv.y, u, cq.y);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < CPOL_N_SZ / 2; ++c) { v[c].x *= it.scale; v[c].y *= it.scale; }
                    wv.x *= it.scale; wv.y *= it.scale;
                    have = true;
                }
                cur ^= 1;
            }
        }
    }
#else
    if (t.tab && it.lookup) {
        // ---- 1-D block: the lane gathers the rows of its (slice, panel) block (as k_gate1) ----
        const int pn = min((int)it.pf, t.n_pan - 1);
        const double u = 2.0 * (it.pf - (double)pn) - 1.0;
#if CPOL_EXP_ONE_BLOCK            // (timing experiment only: every lane reads the same block -- no gather)
        const double2 *blk = reinterpret_cast<const double2 *>(t.tab);
#else
        const double2 *blk = reinterpret_cast<const double2 *>(t.tab + ((long)(it.key - h.key_base) * t.n_pan + pn) * NB);
#endif
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
#endif
#ifdef CPOL_SUBSUM_TRACE
    {
        double keep = 0.0;
#pragma unroll
        for (int c = 0; c < CPOL_N_SZ / 2; ++c) keep += v[c].x + v[c].y;
        asm volatile("" :: "v"(keep));
        g1t[3] = wall_clock64();                       // coefficient gather + Horner chains
    }
#endif
    int my_lookup = it.lookup ? 1 : 0;
    bool off_table = it.valid && !it.lookup;
    if (RAY) {
        // (items outside the integral table: integrated further down, when the columns of the others have left the registers)
    } else if (off_table) {
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
    if (here) {
#pragma unroll
    for (int c = 0; c < CPOL_N_SZ; ++c) {
        double y = ((c & 1) ? v[c / 2].y : v[c / 2].x) * w0;
        if (!(y == y)) y = 0.0;
        s_acc[(j * CPOL_N_SZ + c) * 64 + lane] = have ? (float)(0.0 + y) : 0.f;
    }
    }
    if (RAY) {
        // ---- items outside the integral table: integrated here, one after the other (rare), their terms straight into LDS ----
        unsigned long long todo = __ballot(off_table);
        if (todo && lane == 0) {                                  // (cpol_counters: items integrated bin by bin, one "unit" each)
            atomicAdd(g.totals, (unsigned long long)__popcll(todo));
            atomicAdd(g.totals + 1, (unsigned long long)__popcll(todo));
        }
        while (todo) {
            const int l = (int)__ffsll((long long)todo) - 1;
            todo &= todo - 1;
            double *col = s_mv + j * 64;                          // (this wavefront's own piece of LDS, written for good further down)
            integrate_gamma_item_wave(h.table, h.pre, h.dnu, h.aux, d.n_d, d.uniform_grid, d.dD,
                                      __builtin_amdgcn_readlane(it.key, l) - h.key_base, readlane_f64(it.p0, l), readlane_f64(it.p1, l), col);
            if (lane < CPOL_N_SZ) {                               // (lane c: column c of the item of lane l, as the loop above forms it)
                double y = col[lane] * w0;
                if (!(y == y)) y = 0.0;
                s_acc[(j * CPOL_N_SZ + lane) * 64 + l] = (float)(0.0 + y);
            }
            __builtin_amdgcn_wave_barrier();
        }
        off_table = false;                                        // (nothing is deferred)
    }
    const bool moments = want_rvel && it.valid && f.vsrc[j] == 1;
    double vj = 0.0, nj = 0.0;
    if (moments) {
        if (want_vn) { vj = wv.x; nj = wv.y; }
        else { vj = it.dv; nj = it.dn; }
    }
    if (here) {
    s_mv[j * 64 + lane] = vj;
    s_mn[j * 64 + lane] = nj;
    s_flag[j * 64 + lane] = (it.valid ? 1u : 0u) | (off_table ? 2u : 0u) | (moments ? 4u : 0u);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) my_lookup += __shfl_xor(my_lookup, off);
    if (LAST_WAVE) {
        // this wavefront's share of the sweep's table-item count, then its ticket: release (its LDS stores are complete),
        // acquire for the wavefront that finds every other ticket taken
        if (lane == 0 && my_lookup) {
            const unsigned slot = ((blockIdx.x + blockIdx.y * gridDim.x) * 4u + (unsigned)j) & (CPOL_COUNT_SLOTS - 1);
            atomicAdd(a.n_lookup + 2 + slot, my_lookup);
        }
        int old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(&s_done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        old = __builtin_amdgcn_readfirstlane(old);
#ifdef CPOL_SUBSUM_TRACE
        g1t[4] = wall_clock64();                       // terms in LDS, ticket taken
        g1_valid = (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(it.valid)) | ((unsigned long long)j << 32);
        g1_hw = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32;
        if (old != n_h - 1) {
            const unsigned long w = ((unsigned long)blockIdx.y * gridDim.x + blockIdx.x) * n_h + j;
            if (lane == 0 && w < CPOL_SUBSUM_TRACE_N) {
                for (int q = 0; q < 5; ++q) g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + q] = g1t[q];
                g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + 5] = 0;
                g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + 6] = g1_valid;
                g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + 7] = g1_hw;
            }
        }
#endif
        if (old != n_h - 1) return;                   // (not the last: the slot is free for the next workgroup's wavefronts)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
    __syncthreads();                                  // (s_lookup = 0 is visible; the species' terms are in LDS)
    if (lane == 0 && my_lookup) atomicAdd(&s_lookup, my_lookup);
    }

    // ---- the gate: deferred if any species is off its table; the items of a deferred gate are stored for final_gate ----
    unsigned vbits = 0;
    bool deferred = false;
    for (int q = 0; q < n_h; ++q) {
        const unsigned fl = ((pmask >> q) & 1u) ? s_flag[q * 64 + lane] : 0u;          // (a slot absent from the tile wrote nothing)
        vbits |= (fl & 1u) << q;
        deferred = deferred || (fl & 2u);
    }
    if (!RAY && deferred) {                              // (RAY: nothing is ever deferred -- the item's parameters are dead by now)
        gate1_store_item(a, t, j, n, i, it, g.analytic_vn != 0);
        if (want_vn && t.tab && !t.two_d && it.lookup && a.vn)
            *reinterpret_cast<double2 *>(a.vn + ((long)j * n + i) * 2) = wv;
    }
    if (!LAST_WAVE) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned slot = ((blockIdx.x + blockIdx.y * gridDim.x) * 4u) & (CPOL_COUNT_SLOTS - 1);
        if (s_lookup) atomicAdd(a.n_lookup + 2 + slot, s_lookup);
    }
    }
    if (j == 0 || LAST_WAVE) {
        // ---- wavefront 0: sum over the species in order, get_pol_from_sz, RVEL, mask; the operands of the range scans ----
        if (in && deferred) a.vmask[i] = (unsigned char)vbits;
        if (in && !RAY) g.defer[i] = deferred ? 1 : 0;
        if (in && !deferred) {
            float tot[CPOL_N_SZ];
            double mom_v = 0.0, mom_n = 0.0;
            for (int q = 0; q < n_h; ++q) {
                if (!((pmask >> q) & 1u)) {                 // (wave-uniform) absent from the tile: the column of +0.0f it would have written
#pragma unroll                                              // (added all the same: -0.0f + 0.0f = +0.0f, the bits of the other forms)
                    for (int c = 0; c < CPOL_N_SZ; ++c) tot[c] = (q == 0) ? 0.f : tot[c] + 0.f;
                    continue;
                }
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
    }
#ifdef CPOL_SUBSUM_TRACE
    if (LAST_WAVE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long w = ((unsigned long)blockIdx.y * gridDim.x + blockIdx.x) * n_h + j;
        if (lane == 0 && w < CPOL_SUBSUM_TRACE_N) {
            for (int q = 0; q < 5; ++q) g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + q] = g1t[q];
            g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + 5] = wall_clock64();      // the 64 gates finished (this wavefront took the last ticket)
            g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + 6] = g1_valid;
            g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + 7] = g1_hw;
        }
    }
#endif
    if (!TICKET) return;

    // ---- the ray's ticket: the workgroup that finishes LAST of the ray scans it ----
    // (hand-off between workgroups as MI355X_MICROARCH.md prescribes: the XCDs' L2s are not coherent with each other and
    // a CU's L1 is never refreshed by another CU's stores.  Producer: plain stores, every storing wave waits for them,
    // workgroup barrier, ONE lane releases at agent scope -- writes the XCD L2's dirty lines back -- and adds to the ray's
    // ticket.  Consumer: the workgroup whose add came last, told by the value the add returned; the adding lane acquires at
    // agent scope -- invalidates this CU's L1 -- waits for the invalidate, the workgroup barrier holds the other waves
    // until then; plain loads behind it.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int old = __hip_atomic_fetch_add(g.ticket + ray_b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == (int)gridDim.x - 1;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // (ready for the next sweep: nobody else touches the ray's ticket any more)
            __hip_atomic_store(g.ticket + ray_b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    const int ng = f.n_gates, tid = threadIdx.x, nthr = blockDim.x;
    const int wave = j, n_waves = n_h;
    const long base = (long)ray_b * ng;
    float *lds = reinterpret_cast<float *>(s_raw);    // [3][n_gates] (the species' terms above are consumed)
    float *s_k = lds, *s_h = lds + ng, *s_v = lds + 2 * ng;
    for (int gg = tid; gg < ng; gg += nthr) {
        s_k[gg] = g.sk[base + gg];
        s_h[gg] = g.sh[base + gg];
        s_v[gg] = g.sv[base + gg];
    }
    __syncthreads();
    // strictly sequential float32 scans (np.cumsum / np.cumprod order is part of the numerical contract), a wavefront per scan
    // (with fewer than three species a wavefront takes several): scan_lds_wave_exact / scan_lds_sequential (cpol_final.inl)
    {
        const int n_scan = f.with_attenuation ? 3 : 1;
        for (int sc = wave; sc < n_scan; sc += n_waves) {          // (wave-uniform)
            float *sv = (sc == 0) ? s_k : (sc == 1 ? s_h : s_v);
#if CPOL_SCAN_FORM == 1
            if (sc == 0) scan_lds_wave_exact<false>(sv, ng, lane);
            else scan_lds_wave_exact<true>(sv, ng, lane);
#elif CPOL_SCAN_FORM == 0
            if (lane == 0) {
                if (sc == 0) scan_lds_sequential<false>(sv, ng);
                else scan_lds_sequential<true>(sv, ng);
            }
#endif
        }
    }
    __syncthreads();
    double *rvel = f.RVEL ? f.RVEL : r.RVEL;
    for (int gg = tid; gg < ng; gg += nthr) {
        const long ii = base + gg;
        // (written by other workgroups of this kernel: read behind the acquire fence)
        const float zh = f.ZH[ii], zv = f.ZV[ii];
        float phidp = s_k[gg] * r.radial_res / 1000.0f + f.DELTA_HV[ii];
        float zdr = f.ZDR[ii];
        if (f.with_attenuation) zdr = (zh * s_h[gg]) / (zv * s_v[gg]);
        bool cut = false;
        if (r.sens_thr) {
            // 10*np.log10(ZH) (float32) < threshold(r) (float64)
            const float dbz = 10.0f * (float)log10((double)zh);
            cut = (double)dbz < r.sens_thr[gg];
        }
        if (cut) {
            const float qnan = __builtin_nanf("");
            f.ZH[ii] = qnan; f.ZV[ii] = qnan; f.KDP[ii] = qnan; f.RHOHV[ii] = qnan;
            zdr = qnan; phidp = qnan;
            if (rvel) rvel[ii] = __builtin_nan("");
        }
        r.PHIDP[ii] = phidp;
        f.ZDR[ii] = zdr;
    }
}

__global__ __launch_bounds__(64 * CPOL_MAX_HYDRO) CPOL_GATE1S_ATTR void k_gate1_species(HydroSet hs, ItabSet its, ClassifyArgs a, FinalArgs f, GateArgs g)
{
    gate1_species_body<false, false>(hs, its, a, f, g, ScanRayArgs{});
}

// grid = (ceil(n_gates / 64), n_rays), block = 64 * n_hydro, dynamic LDS = max(n_hydro * 64 * GATE1S_BYTES, 3 * n_gates * 4)
__global__ __launch_bounds__(64 * CPOL_MAX_HYDRO) CPOL_GATE1S_ATTR void k_gate1_ray(HydroSet hs, ItabSet its, ClassifyArgs a, FinalArgs f, GateArgs g,
                                                                                  ScanRayArgs r)
{
    gate1_species_body<true, false>(hs, its, a, f, g, r);
}

// the same with the ray's scans inside (the ticket: see gate1_species_body); CPOL_GATE1_RAY=3.  Measured: the agent-scope
// release every workgroup needs in front of its ticket costs more than the launch of k_scan_rays saves (profiles/r5_variants.txt)
__global__ __launch_bounds__(64 * CPOL_MAX_HYDRO) CPOL_GATE1S_ATTR void k_gate1_ray_scan(HydroSet hs, ItabSet its, ClassifyArgs a, FinalArgs f, GateArgs g,
                                                                                       ScanRayArgs r)
{
    gate1_species_body<true, true>(hs, its, a, f, g, r);
}

// The range scans of a single-beam sweep whose gates k_gate1_ray has finished (no deferred gates: nothing of final_gate):
// one workgroup per ray, the operands of the three strictly sequential float32 scans through LDS, lane 0 of three
// wavefronts runs one scan each, then PHIDP, the attenuated ZDR and the sensitivity cut -- the second half of k_final
// without the per-gate function's 243 registers (k_final<256> on the C2 sweep: 15-19 us for this work).
__global__ __launch_bounds__(256) void k_scan_rays(FinalArgs f, GateArgs g, ScanRayArgs r)
{
    extern __shared__ float lds_scan[];                // [3][n_gates]
    const int ray = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ng = f.n_gates;
    const long base = (long)ray * ng;
    float *s_k = lds_scan, *s_h = lds_scan + ng, *s_v = lds_scan + 2 * ng;
    for (int gg = tid; gg < ng; gg += 256) {
        s_k[gg] = g.sk[base + gg];
        s_h[gg] = g.sh[base + gg];
        s_v[gg] = g.sv[base + gg];
    }
    // what the last loop needs of the gates, requested before the scans so that it arrives behind them (two gates per
    // thread at 500 gates; a ray of more than 512 gates reads the rest in the loop itself)
    constexpr int PRE = 2;
    float p_zh[PRE], p_zv[PRE], p_dhv[PRE], p_zdr[PRE];
    double p_thr[PRE];
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
        const int gg = tid + q * 256;
        const bool ok = gg < ng;
        const long ii = base + (ok ? gg : 0);
        p_zh[q] = f.ZH[ii]; p_zv[q] = f.ZV[ii]; p_dhv[q] = f.DELTA_HV[ii]; p_zdr[q] = f.ZDR[ii];
        p_thr[q] = r.sens_thr ? r.sens_thr[ok ? gg : 0] : 0.0;
    }
    __syncthreads();
#if CPOL_SCAN_FORM == 1
    if (wave == 0 || (f.with_attenuation && wave < 3)) {          // (wave-uniform)
        if (wave == 0) scan_lds_wave_exact<false>(s_k, ng, lane);
        else scan_lds_wave_exact<true>(wave == 1 ? s_h : s_v, ng, lane);
    }
#elif CPOL_SCAN_FORM == 0
    if (lane == 0 && (wave == 0 || (f.with_attenuation && wave < 3))) {
        if (wave == 0) scan_lds_sequential<false>(s_k, ng);
        else scan_lds_sequential<true>(wave == 1 ? s_h : s_v, ng);
    }
#endif
    __syncthreads();
    double *rvel = f.RVEL ? f.RVEL : r.RVEL;
    for (int gg = tid, q = 0; gg < ng; gg += 256, ++q) {
        const long ii = base + gg;
        float zh, zv, dhv, zdr;
        double thr;
        if (q < PRE) { zh = p_zh[q < PRE ? q : 0]; zv = p_zv[q < PRE ? q : 0]; dhv = p_dhv[q < PRE ? q : 0]; zdr = p_zdr[q < PRE ? q : 0]; thr = p_thr[q < PRE ? q : 0]; }
        else { zh = f.ZH[ii]; zv = f.ZV[ii]; dhv = f.DELTA_HV[ii]; zdr = f.ZDR[ii]; thr = r.sens_thr ? r.sens_thr[gg] : 0.0; }
        float phidp = s_k[gg] * r.radial_res / 1000.0f + dhv;
        if (f.with_attenuation) zdr = (zh * s_h[gg]) / (zv * s_v[gg]);
        bool cut = false;
        if (r.sens_thr) {
            // 10*np.log10(ZH) (float32) < threshold(r) (float64)
            const float dbz = 10.0f * (float)log10((double)zh);
            cut = (double)dbz < thr;
        }
        if (cut) {
            const float qnan = __builtin_nanf("");
            f.ZH[ii] = qnan; f.ZV[ii] = qnan; f.KDP[ii] = qnan; f.RHOHV[ii] = qnan;
            zdr = qnan; phidp = qnan;
            if (rvel) rvel[ii] = __builtin_nan("");
        }
        r.PHIDP[ii] = phidp;
        f.ZDR[ii] = zdr;
    }
}
