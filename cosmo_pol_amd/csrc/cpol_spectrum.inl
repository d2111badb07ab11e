// cpol_spectrum.inl -- Doppler scheme 3: the Doppler spectrum of every gate.
//
// Reference functions replaced (wolfidan/cosmo_pol):
//   get_doppler_spectrum            scatter/doppler_scatter.py:603-716
//   get_diameter_from_rad_vel       scatter/doppler_scatter.py:545-600
//   get_refl                        scatter/doppler_c.c:11-32
//   per-sub-beam attenuation, weights, accumulation   scatter/doppler_scatter.py:297-305, 353-391
//   RVEL from the spectrum, aliasing                  scatter/doppler_scatter.py:422-437
// Hydrometeors with power-law fall speeds (R, S, G, H, I) and, since round 6, the melting species (mS, mG): their fall
// speed V(D) = phi V_rain(D_r(D)) + (1 - phi) V_dry(D) has no closed inverse, the reference inverts it at EVERY gate by
// linear interpolation over the n_d nodes (V(D_k), D_k), D_k = linspace(d_min(fw), d_max(fw), n_d) -- `set_psd` clears the
// interpolator, hydrometeors.py:1426,1474; get_D_from_V :480-500 -- and evaluates N(D) on the FLOAT32 copy of that grid with
// NumPy's promotion rules (float32 powers of the float32 diameters, float64 wherever the float64 wet fraction enters:
// melt_psd32 below follows the statements of hydrometeors.py:372-439 operand by operand).
// float32 / float64 follow NumPy-2 promotion of the reference
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
    int n_melt_rows;            // 0, or 2 when a melting species is staged: rows of n_d float64 fall speeds in front of the float32 part of the LDS image
};

__device__ __forceinline__ float clamp_pair(float D, float d_min, float d_max)
{
    if (D >= d_max) D = d_max;
    if (D <= d_min) D = d_min;
    return D;
}

// a float32 matrix compared with the float64 limits of a melting species (shape (1,) arrays: the comparison promotes to float64,
// the assignment rounds the limit to float32; doppler_scatter.py:585-586)
__device__ __forceinline__ float clamp_pair64(float D, double d_min, double d_max)
{
    if ((double)D >= d_max) D = (float)d_max;
    if ((double)D <= d_min) D = (float)d_min;
    return D;
}

// NumPy's float32 power of a float32 array by a Python float (weak scalar -> float32): powf, taken as the float64 result
// rounded once; x**2 is NumPy's square fast path
__device__ __forceinline__ float powf32(float x, double y)
{
    if (y == 2.0) return x * x;
    return (float)pow((double)x, (double)(float)y);
}

// The melting species at ONE gate (hydrometeors.py:303-478).  fw, lam_r, q: float64 as the classification left them.
struct MeltGate {
    double fw, fw2, phi, lam_r, q;
    double d_min, d_max, step;       // the gate's diameter range and (d_max - d_min) / (n_d - 1)
};

// float64 diameter D: mass, D_r, fall speed; `Nraw` = rain N(D_r) V_rain(D_r) / V(D) dD_r/dD (get_N without prop_factor)
__device__ __forceinline__ void melt_terms64(const cpol_hydro_desc &d, const MeltGate &g, double D, double &M, double &V, double &Nraw)
{
    const double pi6 = 3.14159265358979323846 / 6;
    auto mass = [&](double x) { return g.fw2 * (d.r_a * pow(x, d.r_b)) + (1.0 - g.fw2) * (d.a * (d.b == 2.0 ? x * x : pow(x, d.b))); };
    auto d_r = [&](double x) { return pow((mass(x) / (pi6 * pow(x, 3.0))) / 1.0e-6, 1 / 3.) * x; };     // RHO_W = 1e-6 kg mm-3
    M = mass(D);
    const double Dr = d_r(D);
    const double dDr = (d_r(D + 0.01) - Dr) / 0.01;
    const double sq = sqrt(Dr);                                             // rain: mu = beta = 0.5, nu = 1
    const double Nr = (d.r_n0 * sq) * exp(-(g.lam_r * Dr));
    const double Vr = d.r_alpha * sq;
    const double Vd = d.alpha * pow(D, d.beta);
    V = g.phi * Vr + (1 - g.phi) * Vd;
    Nraw = Nr * Vr / V * dDr;
}

// float32 diameter D32 (a node of the float32 copy of the gate's grid): get_N as NumPy evaluates it on a float32 array --
// a D^b, pi/6 D^3, alpha_dry D^beta_dry and D + 0.01 in float32; everything the float64 wet fraction touches in float64
__device__ __forceinline__ double melt_psd32(const cpol_hydro_desc &d, const MeltGate &g, float D32, double prop)
{
    const float pi6_32 = (float)(3.14159265358979323846 / 6);
    auto d_r = [&](float x) {
        const float m_rain = (float)d.r_a * powf32(x, d.r_b), m_dry = (float)d.a * powf32(x, d.b);
        const double M = g.fw2 * (double)m_rain + (1.0 - g.fw2) * (double)m_dry;
        const float vol = pi6_32 * powf32(x, 3.0);
        return pow((M / (double)vol) / 1.0e-6, 1 / 3.) * (double)x;
    };
    const double Dr = d_r(D32);
    const double dDr = (d_r(D32 + 0.01f) - Dr) / 0.01;
    const double sq = sqrt(Dr);
    const double Nr = (d.r_n0 * sq) * exp(-(g.lam_r * Dr));
    const double Vr = d.r_alpha * sq;
    const float Vd = (float)d.alpha * powf32(D32, d.beta);
    const double V = g.phi * Vr + (1 - g.phi) * (double)Vd;
    return (prop * Nr) * Vr / V * dDr;
}

// one workgroup per sub-beam gate
__global__ __launch_bounds__(CPOL_SPEC_THREADS) void k_spec_gate(HydroSet hs, SpecSet ss, SpecArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds_spec[];
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
    // LDS: [2][n_d] float64 fall speeds of the (at most two) melting species + [CPOL_SPEC_THREADS] float64 scratch, then
    // [n_hydro][n_d] float32 N x rcs, [n_hydro][n_v] float32 edge diameters
    double *VL = reinterpret_cast<double *>(lds_spec);                        // [n_melt_rows][n_d]
    double *redL = VL + (long)a.n_melt_rows * n_d;                            // [CPOL_SPEC_THREADS] (with melting species only)
    float *prodL = reinterpret_cast<float *>(redL + (a.n_melt_rows ? CPOL_SPEC_THREADS : 0));       // [n_p][n_d]
    float *DL = prodL + (long)hs.n_hydro * n_d;                               // [n_p][n_v]
    MeltGate mg[2];
    int melt_of[CPOL_SPEC_MAX_H];                       // present species -> 0 / 1 (its row of VL), or -1
    float dmin32[CPOL_SPEC_MAX_H], step32[CPOL_SPEC_MAX_H];
    int n_melt = 0;

    // ---- N(D) x rcs per diameter bin (float32 product of float32 operands) ----
    for (int p = 0; p < n_p; ++p) {
        const int j = present[p];
        const HydroDev &h = hs.h[j];
        const cpol_hydro_desc &d = h.d;
        const int key = a.key[(long)j * n + sbg];
        const SpecDev &sd = ss.s[j];
        const float *rcs = sd.rcs32 + (long)(key - h.key_base) * d.n_d;
        const double *P = a.par + ((long)j * CPOL_MAX_PAR) * n + sbg;
        melt_of[p] = -1;
        if (d.psd_family == CPOL_PSD_MELTING && n_melt < a.n_melt_rows) {
            // ---- a melting species: its grid, fall speeds and PSD belong to THIS gate's wet fraction ----
            MeltGate &g = mg[n_melt];
            g.q = P[0]; g.fw = P[n]; g.lam_r = P[2 * n];
            g.fw2 = g.fw * g.fw;                                            // fw ** 2: NumPy's square
            g.phi = 0.246 * g.fw + (1 - 0.246) * pow(g.fw, 7.0);
            g.d_max = g.fw * d.r_dmax + (1 - g.fw) * d.s_dmax;              // hydrometeors.py:336-339
            g.d_min = g.fw * d.r_dmin + (1 - g.fw) * d.s_dmin;
            g.step = (g.d_max - g.d_min) / (double)(d.n_d - 1);
            double *Vrow = VL + (long)n_melt * n_d;
            // integrate_M on the float64 grid (vlinspace, utilities.py:158-173; :462-478) and, on the same nodes, the fall
            // speeds of get_D_from_V (np.linspace: the same nodes but for the last, which is d_max itself)
            double msum = 0.0;
            for (int k = tid; k < d.n_d; k += CPOL_SPEC_THREADS) {
                const double D = g.d_min + g.step * (double)k;
                double M, V, Nraw;
                melt_terms64(d, g, D, M, V, Nraw);
                msum += Nraw * M;
                if (k == d.n_d - 1 && D != g.d_max) { double M1, N1; melt_terms64(d, g, g.d_max, M1, V, N1); }
                Vrow[k] = V;
            }
            redL[tid] = msum;
            __syncthreads();
            for (int off = CPOL_SPEC_THREADS / 2; off >= 1; off >>= 1) {
                if (tid < off) redL[tid] += redL[tid + off];
                __syncthreads();
            }
            const double dD = (g.d_min + g.step * 1.0) - (g.d_min + g.step * 0.0);
            const double prop = g.q / (redL[0] * dD);                       // q / integrate_M (:1428, :1475)
            __syncthreads();                                                // (redL is reused by the second melting species)
            // the float32 grid of the spectrum: D[:, j] = np.linspace(d_min, d_max, n_d) stored as float32
            const float D0 = (float)g.d_min, D1 = (float)(1.0 * g.step + g.d_min);
            dmin32[p] = D0; step32[p] = D1 - D0;
            for (int k = tid; k < d.n_d; k += CPOL_SPEC_THREADS) {
                const float D32 = (k == d.n_d - 1) ? (float)g.d_max : (float)((double)k * g.step + g.d_min);
                prodL[(long)p * n_d + k] = (float)melt_psd32(d, g, D32, prop) * rcs[k];
            }
            melt_of[p] = n_melt++;
            continue;
        }
        dmin32[p] = sd.dgrid[0]; step32[p] = sd.step32;
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
    float *okL = DL + (long)hs.n_hydro * a.n_v;         // [n_v] 1: wh >= 0 at this velocity (a row of the reference's D matrix)
    __syncthreads();                                    // (VL complete)
    for (int v = tid; v < a.n_v; v += CPOL_SPEC_THREADS) {
        const double wh = (double)inv_rho * (W + horiz - a.varray[v] / (double)sin32);
        const bool ok = wh >= 0.0;
        okL[v] = ok ? 1.0f : 0.0f;
        for (int p = 0; p < n_p; ++p) {
            const cpol_hydro_desc &d = hs.h[present[p]].d;
            float D = __builtin_nanf("");
            if (ok && melt_of[p] < 0) {
                D = (float)pow(wh / d.alpha, 1.0 / d.beta);
            } else if (ok) {
                // scipy.interpolate.interp1d(V_all, D_all, bounds_error=False, fill_value=nan) at wh (hydrometeors.py:494-500): for 1-D
                // float64 data SciPy hands the evaluation to np.interp -- j with V_j <= wh < V_{j+1} by binary search, D_j when wh == V_j
                // or j is the last node, else slope * (wh - V_j) + D_j -- and fills NaN outside [V_0, V_{n-1}]
                const MeltGate &g = mg[melt_of[p]];
                const double *Vrow = VL + (long)melt_of[p] * n_d;
                if (!(wh < Vrow[0]) && !(wh > Vrow[d.n_d - 1])) {
                    int lo_i = 0, hi_i = d.n_d;                            // first index with V > wh
                    while (lo_i < hi_i) {
                        const int mid = (lo_i + hi_i) >> 1;
                        if (Vrow[mid] <= wh) lo_i = mid + 1; else hi_i = mid;
                    }
                    const int j = max(0, lo_i - 1);
                    const double y_j = (j == d.n_d - 1) ? g.d_max : (double)j * g.step + g.d_min;
                    if (j == d.n_d - 1 || Vrow[j] == wh) {
                        D = (float)y_j;
                    } else {
                        const double y_n = (j + 1 == d.n_d - 1) ? g.d_max : (double)(j + 1) * g.step + g.d_min;
                        const double slope = (y_n - y_j) / (Vrow[j + 1] - Vrow[j]);
                        D = (float)(slope * (wh - Vrow[j]) + y_j);
                    }
                }
            }
            if (ok)
                for (int q = p; q < n_p; ++q) {           // clamps of this and the later species (a NaN stays a NaN)
                    if (melt_of[q] >= 0) {
                        D = clamp_pair64(D, mg[melt_of[q]].d_min, mg[melt_of[q]].d_max);
                    } else {
                        const SpecDev &sq = ss.s[present[q]];
                        D = clamp_pair(D, sq.dgrid[0], sq.dgrid[hs.h[present[q]].d.n_d - 1]);
                    }
                }
            DL[(long)p * a.n_v + v] = D;
        }
    }
    __syncthreads();

    // ---- rows = consecutive valid edges; sum N x rcs over [Da, Db) per species ----
    for (int v = tid; v + 1 < a.n_v; v += CPOL_SPEC_THREADS) {
        if (okL[v] == 0.0f || okL[v + 1] == 0.0f) continue;   // wh < 0 at one of the edges: not a row of the D matrix
        bool keep = false;
        float refl = 0.0f;
        for (int p = 0; p < n_p; ++p) {
            const float x0 = DL[(long)p * a.n_v + v], x1 = DL[(long)p * a.n_v + v + 1];
            // np.minimum / np.maximum propagate a NaN (a melting species outside its interpolator) to BOTH edges; (Db - Da) == 0
            // is then false -- the row is kept -- and the two (int) conversions of doppler_c.c give the same index: an empty bin
            const bool nan_edge = !(x0 == x0) || !(x1 == x1);
            const float Da = fminf(x0, x1), Db = fmaxf(x0, x1);
            if (nan_edge || Db - Da != 0.0f) keep = true;
            if (nan_edge) continue;
            int ia = (int)((Da - dmin32[p]) / step32[p]);
            int ib = (int)((Db - dmin32[p]) / step32[p]);
            ia = max(0, min(ia, n_d));
            ib = max(0, min(ib, n_d));
            float sum = 0.0f;
            for (int k = ia; k < ib; ++k) sum += prodL[(long)p * n_d + k];
            refl += sum * step32[p];
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

