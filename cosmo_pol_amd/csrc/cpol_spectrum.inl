// cpol_spectrum.inl -- Doppler scheme 3: the Doppler spectrum of every gate.
//
// Reference functions replaced (wolfidan/cosmo_pol):
//   get_doppler_spectrum            scatter/doppler_scatter.py:603-716
//   get_diameter_from_rad_vel       scatter/doppler_scatter.py:545-600
//   get_refl                        scatter/doppler_c.c:11-32
//   per-sub-beam attenuation, weights, accumulation   scatter/doppler_scatter.py:297-305, 353-391
//   RVEL from the spectrum, aliasing                  scatter/doppler_scatter.py:422-437
// Hydrometeors with power-law fall speeds only (R, S, G, H, I); the host refuses scheme 3
// with the melting species.  float32 / float64 follow NumPy-2 promotion of the reference
// statements (see oracle/cosmo_pol_oracle/spectrum.py); float32 libm values are float64
// results rounded once.  Quirks kept: the diameter clamp of hydrometeor j also re-clamps
// the columns before it (:583-586); the radar constant carries K^2 squared (:709); the
// attenuation of the r-th VALID gate of a hydrometeor is added to gate r (:305).
// Departure: table indices outside [0, n_d] (possible through that clamp quirk; the C
// reference then reads out of bounds) are clamped.

#define CPOL_SPEC_THREADS 256
#define CPOL_SPEC_MAX_H CPOL_MAX_HYDRO

struct SpecArgs {
    const float *vals;          // [n_vars][n_sbg]
    const signed char *mask;    // [n_sbg]
    const float *elev;          // folded elevation [n_sbg]
    const int *key;             // [n_hydro][n_sbg]
    const double *par;          // [n_hydro][CPOL_MAX_PAR][n_sbg]
    const double *wgate;        // [n_sbg] or NULL
    const double *geo;          // [n_rays][n_h][8]
    const int *sub_h;
    const double *varray;       // [n_v]
    float *beam;                // [n_sbg][n_v] float32 spectrum of every sub-beam gate
    long n_sbg;
    int n_gates, n_sub, n_h, n_v;
    int var_u, var_v, var_w, var_rho, var_t;
    float c_spec;               // (float)(wavelength^4 / (pi^5 K^2 K^2))
};

__device__ __forceinline__ float clamp_pair(float D, float d_min, float d_max)
{
    if (D >= d_max) D = d_max;
    if (D <= d_min) D = d_min;
    return D;
}

// one workgroup per sub-beam gate
__global__ __launch_bounds__(CPOL_SPEC_THREADS) void k_spec_gate(HydroSet hs, SpecSet ss, SpecArgs a)
{
    extern __shared__ float lds_spec[];
    const long sbg = blockIdx.x;
    const int tid = threadIdx.x;
    const long n = a.n_sbg;
    float *out = a.beam + sbg * a.n_v;
    for (int v = tid; v < a.n_v; v += CPOL_SPEC_THREADS) out[v] = 0.0f;
    if (a.mask[sbg] != 0) return;
    if (a.wgate && a.wgate[sbg] == 0.0) return;

    // hydrometeors present at this gate, in slot order
    int present[CPOL_SPEC_MAX_H], n_p = 0;
    for (int j = 0; j < hs.n_hydro; ++j)
        if (a.key[(long)j * n + sbg] >= 0) present[n_p++] = j;
    if (n_p == 0) return;
    const int n_d = hs.h[present[0]].d.n_d;
    float *prodL = lds_spec;                            // [n_p][n_d]
    float *DL = lds_spec + (long)hs.n_hydro * n_d;        // [n_p][n_v]

    // ---- N(D) x rcs per diameter bin (float32 product of float32 operands) ----
    for (int p = 0; p < n_p; ++p) {
        const int j = present[p];
        const HydroDev &h = hs.h[j];
        const cpol_hydro_desc &d = h.d;
        const int key = a.key[(long)j * n + sbg];
        const SpecDev &sd = ss.s[j];
        const float *rcs = sd.rcs32 + (long)(key - h.key_base) * d.n_d;
        const double *P = a.par + ((long)j * CPOL_MAX_PAR) * n + sbg;
        const double lam = P[0];
        const bool fixed_n0 = d.rule == CPOL_RULE_RAIN_1MOM || d.rule == CPOL_RULE_GRAUPEL_1MOM;
        const double N0 = (d.psd_family == CPOL_PSD_ICE_FIELD) ? P[3 * n] : (fixed_n0 ? d.n0_fixed : P[n]);
        for (int k = tid; k < d.n_d; k += CPOL_SPEC_THREADS) {
            double nk;
            if (d.psd_family == CPOL_PSD_ICE_FIELD) {
                const double x = lam * (double)sd.dgrid[k] / 1000.0;
                nk = N0 * phi23(x);
            } else {
                nk = (N0 * (double)sd.dgrid[d.n_d + k]) * exp(-(lam * (double)sd.dgrid[2 * d.n_d + k]));
            }
            prodL[(long)p * n_d + k] = (float)nk * rcs[k];
        }
    }

    // ---- diameters of the velocity-bin edges ----
    const int rs = (int)(sbg / a.n_gates);              // ray * n_sub + sub
    const int ray = rs / a.n_sub, sub = rs % a.n_sub;
    const double *gc = a.geo + ((long)ray * a.n_h + a.sub_h[sub]) * 8;
    const float th = a.elev[sbg] * 0.017453292f;        // np.deg2rad of a float32
    const float tan32 = (float)tan((double)th), sin32 = (float)sin((double)th);
    const float rho = a.vals[(long)a.var_rho * n + sbg];
    const float rho0 = a.vals[(long)a.var_rho * n + (long)rs * a.n_gates];
    const float inv_rho = 1.0f / sqrtf(rho / rho0);
    const double U = (double)a.vals[(long)a.var_u * n + sbg], V = (double)a.vals[(long)a.var_v * n + sbg],
                 W = (double)a.vals[(long)a.var_w * n + sbg];
    const double horiz = (U * gc[0] + V * gc[1]) / (double)tan32;
    for (int v = tid; v < a.n_v; v += CPOL_SPEC_THREADS) {
        const double wh = (double)inv_rho * (W + horiz - a.varray[v] / (double)sin32);
        const bool ok = wh >= 0.0;
        for (int p = 0; p < n_p; ++p) {
            const cpol_hydro_desc &d = hs.h[present[p]].d;
            float D = ok ? (float)pow(wh / d.alpha, 1.0 / d.beta) : __builtin_nanf("");
            if (ok)
                for (int q = p; q < n_p; ++q) {           // clamps of this and the later species
                    const SpecDev &sq = ss.s[present[q]];
                    D = clamp_pair(D, sq.dgrid[0], sq.dgrid[hs.h[present[q]].d.n_d - 1]);
                }
            DL[(long)p * a.n_v + v] = D;
        }
    }
    __syncthreads();

    // ---- rows = consecutive valid edges; sum N x rcs over [Da, Db) per species ----
    for (int v = tid; v + 1 < a.n_v; v += CPOL_SPEC_THREADS) {
        const float e0 = DL[v], e1 = DL[v + 1];
        if (!(e0 == e0) || !(e1 == e1)) continue;         // wh < 0 or NaN at one of the edges
        bool keep = false;
        float refl = 0.0f;
        for (int p = 0; p < n_p; ++p) {
            const SpecDev &h = ss.s[present[p]];
            const float x0 = DL[(long)p * a.n_v + v], x1 = DL[(long)p * a.n_v + v + 1];
            const float Da = fminf(x0, x1), Db = fmaxf(x0, x1);
            if (Db - Da != 0.0f) keep = true;
            int ia = (int)((Da - h.dgrid[0]) / h.step32);
            int ib = (int)((Db - h.dgrid[0]) / h.step32);
            ia = max(0, min(ia, n_d));
            ib = max(0, min(ib, n_d));
            float sum = 0.0f;
            for (int k = ia; k < ib; ++k) sum += prodL[(long)p * n_d + k];
            refl += sum * h.step32;
        }
        if (keep) out[v] = refl * a.c_spec;
    }
}

// ---- per (ray, sub-beam): attenuation of the spectrum (doppler_scatter.py:297-305, 372-384)
struct SpecAttenArgs {
    const int *key;             // [n_hydro][n_sbg]
    const double *res;          // [n_hydro][n_sbg][12]
    float *beam;                // [n_sbg][n_v]
    long n_sbg;
    int n_gates, n_v, n_hydro;
    double c_att;               // 4.343e-3 * 2 * wavelength
    double res_km;              // radial_res / 1000.
};

__global__ __launch_bounds__(64) void k_spec_atten(SpecAttenArgs a)
{
    extern __shared__ double lds_att[];                 // [n_gates]
    const long rs = blockIdx.x;
    const int lane = threadIdx.x;
    const long base = rs * a.n_gates;
    for (int g = lane; g < a.n_gates; g += 64) lds_att[g] = 0.0;
    __syncthreads();
    for (int j = 0; j < a.n_hydro; ++j) {
        int count = 0;
        for (int g0 = 0; g0 < a.n_gates; g0 += 64) {
            const int g = g0 + lane;
            const bool valid = g < a.n_gates && a.key[(long)j * a.n_sbg + base + g] >= 0;
            const unsigned long long m = __ballot(valid);
            if (valid) {
                const int rank = count + __popcll(m & ((1ull << lane) - 1ull));
                double ah = a.c_att * a.res[((long)j * a.n_sbg + base + g) * CPOL_N_SZ + 11];
                ah *= a.res_km;
                if (ah == ah) lds_att[rank] += ah;        // nansum_arr: NaN counts as 0
            }
            count += __popcll(m);
        }
        __syncthreads();
    }
    if (lane == 0) {
        double c = 0.0;
        for (int g = 0; g < a.n_gates; ++g) { c += lds_att[g]; lds_att[g] = c; }
    }
    __syncthreads();
    for (int g = 0; g < a.n_gates; ++g) {
        float *row = a.beam + (base + g) * a.n_v;
        float s = 0.0f;
        for (int v = lane; v < a.n_v; v += 64) { const float x = row[v]; if (x == x) s += x; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        if (!(s > 0.0f)) continue;                        // wave-uniform
        const float db = 10.0f * (float)log10((double)s);
        const double frac = (double)s / pow(10.0, 0.1 * ((double)db - lds_att[g]));
        for (int v = lane; v < a.n_v; v += 64) row[v] = (float)((double)row[v] / frac);
    }
}

// ---- accumulation over sub-beams, RVEL (doppler_scatter.py:386-391, 422-437)
struct SpecFinalArgs {
    const float *beam;          // [n_sbg][n_v]
    const double *sub_w;
    const double *wgate;        // per-gate weights (scheme 'ml') or NULL
    const double *varray;
    const double *nyquist;      // [n_rays] or NULL
    const double *sens_thr;     // [n_gates] dBZ threshold per gate, or NULL (no sensitivity cut)
    double *spectrum;           // [n_rg][n_v]
    double *RVEL;               // [n_rg]
    int n_rays, n_gates, n_sub, n_v;
};

__global__ __launch_bounds__(64) void k_spec_final(SpecFinalArgs a)
{
    const long rg = blockIdx.x;
    const int lane = threadIdx.x;
    const int ray = (int)(rg / a.n_gates), gate = (int)(rg % a.n_gates);
    const long sbg0 = (long)ray * a.n_sub * a.n_gates + gate;
    double num = 0.0, den = 0.0;
    for (int v = lane; v < a.n_v; v += 64) {
        double acc = 0.0;
        for (int s = 0; s < a.n_sub; ++s) {
            const long sbg = sbg0 + (long)s * a.n_gates;
            const float b = a.beam[sbg * a.n_v + v];
            // beam_spectrum *= quad_weight: float32 x python float -> float32; an array of
            // float64 weights multiplies in float64 and stores float32
            const float wb = a.wgate ? (float)((double)b * a.wgate[sbg]) : b * (float)a.sub_w[s];
            acc += (double)wb;
        }
        // cut_at_sensitivity as get_PPI / get_RHI call it (a list of lists of radials,
        // doppler_scatter.py:839-850) censors the spectrum BIN BY BIN with
        // 10 log10(spectrum) < threshold(r); the gate-level ZH mask does not touch it.  (The cut
        // runs after RVEL was derived from the uncensored spectrum, :422-437.)
        a.spectrum[rg * a.n_v + v] =
            (a.sens_thr && 10.0 * log10(acc) < a.sens_thr[gate]) ? __builtin_nan("") : acc;
        if (acc == acc) { num += a.varray[v] * acc; den += acc; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        num += shfl_f64(num, lane ^ off);
        den += shfl_f64(den, lane ^ off);
    }
    if (lane == 0) {
        double rv = num / den;
        if (a.nyquist) {
            const double nyq = a.nyquist[ray], pi = 3.14159265358979323846;
            const double theta = (rv + nyq) / (2 * nyq) * pi - pi / 2.;
            const double fold = atan(tan(theta));
            rv = (fold + pi / 2) * (2 * nyq) / pi - nyq;
        }
        a.RVEL[rg] = rv;
    }
}

