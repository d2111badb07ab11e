// k_interp_classify: the gate kernel and the classification of the gate's hydrometeors in ONE pass -- the
// launch sequences whose slots all have integral tables (the C3 and C4 volumes; the single-beam sequence without melting
// species has its own fused kernel, k_gate1).
//
// k_interp_sweep leaves n_vars float32 per sub-beam gate in vals[] and k_classify reads temperature, mass
// densities and elevation back: 4.8 + 7.1 GB on the C4 volume, and two kernels that each wait -- the first for its
// float64 pipe (VALU 0.81), the second for its dependent float64 chains at 4 wavefronts per SIMD (VALU 0.47).
// Here the variables of a gate stay with the thread that interpolated them (its column of the workgroup's LDS
// array: classify_item picks variables by runtime index, which registers cannot do), the melting scheme edits them
// there, and only what later kernels read goes to vals[] (a.store_mask: U, V, W for the Doppler terms; everything
// when the integrated model variables are asked for).  Same device functions as the two kernels -- interp_gate<>,
// classify_item -- on the same values: identical results (tests/test_gpu_fullsize.py compares the sequences).
// The counters the classification adds to were cleared by the sweep before this one (two sets used in turn: a
// kernel cannot clear what its own workgroups count into).  Not with the debug reads, integration scheme 'ml'
// (k_ml_weights runs between the two kernels) or Doppler scheme 3 (the spectrum kernels read vals[]).
#ifndef CPOL_FUSED_WPE
#define CPOL_FUSED_WPE 5          // measured on the C4 volume: 3 / 4 (the allocator's choice) / 5 / 6 wavefronts per SIMD: 4.31 / 3.51 / 3.34 / 4.34 ms
#endif
#if CPOL_FUSED_WPE
#define CPOL_FUSED_ATTR __attribute__((amdgpu_waves_per_eu(CPOL_FUSED_WPE, CPOL_FUSED_WPE)))
#else
#define CPOL_FUSED_ATTR
#endif
__global__ __launch_bounds__(256) CPOL_FUSED_ATTR void k_interp_classify(ModelDev m, InterpArgs ia, HydroSet hs, ItabSet its,
                                                                        ClassifyArgs a)
{
    extern __shared__ float s_vals[];                 // [n_vars][blockDim.x]
    clear_counters(ia.zero_buf, ia.zero_n, ia.zero_buf2, ia.zero_n2);      // (the NEXT sweep's set)
    float *sv = s_vals + threadIdx.x;
    const long ls = blockDim.x;
    long sbg = 0;
    float e = 0.f;
    const int status = interp_gate<true>(m, ia, sv, sbg, e);
    const bool in = status == 0;                      // (the other gates hold NaN: no item, k_classify finds the same)
    const long n = a.n_sbg, i = sbg;
    float qms = 0.f, qmg = 0.f;
    double fws = 0.0, fwg = 0.0;
    if (a.with_melting && in) {
        // melting.py:34-83, float32 arithmetic as NumPy evaluates it (as k_classify)
        float qr = sv[a.var_qr * ls], qs = sv[a.var_qs * ls], qg = sv[a.var_qg * ls];
        float qsg = qs + qg;
        if (qr > 0.f && qsg > 0.f) {
            qms = qs + qr * (qs / qsg);
            qmg = qg + qr * (qg / qsg);
            if (qms > 0.f || qmg > 0.f) {
                sv[a.var_qr * ls] = 0.f;
                sv[a.var_qs * ls] = 0.f;
                sv[a.var_qg * ls] = 0.f;
            }
            fws = (double)(qr * qs / qsg) / (double)qms;
            fwg = (double)(qr * qg / qsg) / (double)qmg;
        }
    }
    if (in)
        for (int v = 0; v < m.n_vars; ++v)
            if ((ia.store_mask >> v) & 1u) ia.vals[(long)v * n + i] = sv[v * ls];
    const int var_t0 = hs.h[0].d.var_t;
    const float T0 = in ? sv[var_t0 * ls] : 0.f;
    int my_lookup = 0;
    unsigned vbits = 0;
    for (int j = 0; j < hs.n_hydro; ++j) {
        const HydroDev &h = hs.h[j];
        const cpol_hydro_desc &d = h.d;
        float qm = 0.f;
        double fw = 0.0;
        if (in) {
            if (d.q_source == CPOL_Q_MODEL) qm = sv[d.var_q * ls];
            else if (d.q_source == CPOL_Q_MELT_SNOW) { qm = qms; fw = fws; }
            else { qm = qmg; fw = fwg; }
        }
        const ItabDev &tj = its.t[j];
        ClassItem it;
        classify_item(h, tj, a, sv, ls, 0, i, in, qm, fw, T0, var_t0, e, it);
        if (it.valid) {
            double *P = a.par + ((long)j * CPOL_MAX_PAR) * n + i;
            my_lookup += it.lookup ? 1 : 0;
            if (!it.lookup) {                                 // read by the integrating kernels
                P[0] = it.p0;
                P[n] = it.p1;
                if (h.n_par >= 3) P[2 * n] = it.p2;
                if (tj.tab) P[4 * n] = it.pf;
            } else if (tj.two_d) {
                P[n] = it.p1;                                 // wet fraction: position inside the 2-D block
            }
            if (tj.tab) a.rec[(long)j * n + i] = make_double2(it.pf, it.scale);
            if (it.has_vn && a.vn)
                *reinterpret_cast<double2 *>(a.vn + ((long)j * n + i) * 2) = make_double2(it.dv, it.dn);
            vbits |= 1u << j;
            if (!it.lookup) {                                 // a handful per volume: listed as one-item work units
                const unsigned long long idx = atomicAdd(a.rare_totals + 1, 1ull);
                atomicAdd(a.rare_totals, 1ull);
                atomicAdd(a.n_lookup + 1, 1);
                a.rare_key[idx] = it.key;
                a.rare_perm[idx] = (int)i;
            }
            a.key[(long)j * n + i] = it.key;
        }
    }
    if (status != 3) a.vmask[i] = (unsigned char)vbits;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) my_lookup += __shfl_xor(my_lookup, off);
    if (lane_id() == 0) count_table_items(a.n_lookup, my_lookup);
}
