// cpol_interp.inl -- beam geometry + gate interpolation kernels.
//
// Reference functions replaced (wolfidan/cosmo_pol):
//   _ref_4_3                 interpolation/atm_refraction.py:181-220
//   trilin_interp_radial     interpolation/interpolation.py:498-597 (geodesic per
//                            gate, rotated-pole transform, domain check)
//   get_all_radar_pts        interpolation/interpolation_c.c:12-104
//   binary_search            interpolation/interpolation_c.c:108-135
//   trilinear_interp         interpolation/interpolation_c.c:138-164
//   mask coding              interpolation/interpolation.py:398-411
// All float32 arithmetic below is written in the reference's operand order and
// the TU is compiled with -ffp-contract=off: gate values are bit-identical to
// the x86-64 SSE build of the reference C.

// ---------------------------------------------------------------- staging
// [nz][ny][nx] -> H[cell][nz], and the pair (model top, lowest level) of every column
__global__ void k_stage_heights(const float *__restrict__ src, float *__restrict__ dst,
                                float2 *__restrict__ top_low, int nz, long ncell)
{
    long cell = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= ncell) return;
    for (int k = 0; k < nz; ++k) dst[cell * nz + k] = src[(long)k * ncell + cell];
    top_low[cell] = make_float2(src[cell], src[(long)(nz - 1) * ncell + cell]);
}

// [nz][ny][nx] of variable v -> V[cell][nz][n_vars]
__global__ void k_stage_variable(const float *__restrict__ src, float *__restrict__ dst,
                                 int nz, long ncell, int n_vars, int v)
{
    long cell = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= ncell) return;
    for (int k = 0; k < nz; ++k)
        dst[(cell * nz + k) * n_vars + v] = src[(long)k * ncell + cell];
}

// ---------------------------------------------------------------- trajectory
// ray_traj: [n_rays][n_v][4] = (el_rad, sin el, cos el, el_deg); site: [n_rays][8] or NULL
// traj_out: float32 [n_rays][n_vnodes][3][n_gates]  (s, h, e_deg)
struct TrajArgs {
    const double *ray_traj;
    const double *site;         // per-ray (sin U1, cos U1, lon, alt, re, first gate, n kept, -)
    float *traj_out;
    int n_rays, n_v, n_gates, mode;
    double range0, range_step, ke, re, alt;
    // constants of a ray that k_interp_sweep would otherwise evaluate once per sub-beam gate (NULL: not wanted)
    const double *geo;          // [n_rays][n_h][8]
    double *ray_const;          // [n_rays * n_h][2] sin / cos (2 sigma1), then [n_rays][2] sin / cos (site longitude)
    int n_h;
    double lon1;
    // rotated coordinates along a ray as polynomials of the arc distance (k_trajectory): NULL: not wanted
    double *poly;               // [n_rays * n_h][2][CPOL_GEO_NP] monomial coefficients of rlat(x), rlon(x) [deg], x = s * poly_scale - 1
    const double *poly_M;       // [CPOL_GEO_NP][CPOL_GEO_NP] Chebyshev-node values -> monomial coefficients
    double poly_scale;          // 2 / s_max
    double sin_u1, cos_u1;      // reduced latitude of the (one) radar site
};

// height of candidate gate k of a downward-looking (spaceborne) ray
// (atm_refraction.py:254-255 with KE = 1; float64)
__device__ __forceinline__ double spaceborne_height(double r, double re, double sin_el, double alt)
{
    double temp = sqrt(r * r + re * re + 2.0 * r * 1.0 * re * sin_el);
    return -(temp - re) + alt;
}

// (s, h, e_deg) of gate g of (ray, vertical node) rv as float32, exactly as the reference
// casts them: shared by k_trajectory and by k_interp_sweep (which evaluates it in place,
// one launch less per sweep).  NaN = no gate here (spaceborne rays shorter than the batch).
struct RayPathArgs {
    const double *ray_traj;     // [n_rays][n_v][4]
    const double *site;         // [n_rays][8] or NULL
    int n_v, mode;
    double range0, range_step, ke, re, alt;
};

__device__ __forceinline__ void ray_path(const RayPathArgs &a, int ray, int rv, int g,
                                         float &s32, float &h32, float &e32)
{
    const double el = a.ray_traj[rv * 4 + 0];
    const double sin_el = a.ray_traj[rv * 4 + 1];
    const double cos_el = a.ray_traj[rv * 4 + 2];
    if (a.mode == CPOL_GEOM_SPACEBORNE) {
        const double *st = a.site + (long)ray * 8;
        const double alt = st[3], re = st[4];
        const int k0 = (int)st[5], n_kept = (int)st[6];
        if (g >= n_kept) {
            s32 = h32 = e32 = __builtin_nanf("");
            return;
        }
        const double r = a.range0 + (double)(k0 + g) * a.range_step;
        const double h = spaceborne_height(r, re, sin_el, alt);
        const double s = re * asin((r * cos_el) / (re + h));
        const double e = a.ray_traj[rv * 4 + 3]
            - atan(r * cos_el / (r * sin_el + re + alt)) * (180.0 / 3.14159265358979323846);
        s32 = (float)s;
        h32 = (float)h;
        e32 = (float)e;                       // degrees (the reference's second rad2deg is a bug)
        return;
    }
    const double alt = a.site ? a.site[(long)ray * 8 + 3] : a.alt;
    const double re = a.site ? a.site[(long)ray * 8 + 4] : a.re;
    const double ke = a.ke;
    const double r = a.range0 + (double)g * a.range_step;
    const double ke_re = ke * re;
    // atm_refraction.py:206-215, same operand order
    double temp = sqrt(r * r + ke_re * ke_re + 2.0 * r * ke * re * sin_el);
    double h = temp - ke_re + alt;
    double s = ke_re * asin((r * cos_el) / (ke_re + h));
    double e = el + atan(r * cos_el / (r * sin_el + ke_re + alt));
    s32 = (float)s;
    h32 = (float)h;
    // np.rad2deg on float32: x * (180.0f / float(pi)), constant formed in float32
    const float rad2deg_f = 180.0f / 3.14159265358979323846f;
    e32 = (float)e * rad2deg_f;
}

// the counters of a sweep start at zero: the gate kernel of the sweep BEFORE it cleared them (two sets used in turn)
__device__ __forceinline__ void clear_counters(int *zero_buf, int zero_n, int *zero_buf2, int zero_n2)
{
    if (!zero_buf) return;
    const long total = (long)gridDim.x * gridDim.y * blockDim.x;
    const long first = ((long)blockIdx.x * gridDim.y + blockIdx.y) * blockDim.x + threadIdx.x;
    for (long i = first; i < zero_n; i += total)
        zero_buf[i] = 0;
    if (zero_buf2 && first < zero_n2) zero_buf2[first] = 0;      // (zero_n2 <= one workgroup)
}

// ---- rotated coordinates of the NON-CENTRAL sub-beams as polynomials of the arc distance (round 5) ----
// The gates of one (ray, horizontal quadrature node) lie on ONE geodesic from the radar: their rotated coordinates are
// two smooth functions of the arc distance s alone, evaluated ~3 500 times per volume ray (7 vertical nodes x 500
// gates) with ~300 float64 instructions each (Vincenty passes, two inverse trigonometric functions, the rotation).  Over
// the 150 km of a radar ray -- 0.024 rad of arc -- a degree-8 polynomial through 9 Chebyshev nodes reproduces them to
// rounding (next coefficient ~ (0.024 / 4)^9 / 9! of the function), so k_trajectory evaluates the LONG form (the central
// sub-beam's: 5 Vincenty passes, atan2 / asin, a true division) at the 9 nodes once per (ray, node) and the gate kernel
// runs two Horner chains: 17 FMAs per sub-beam gate.  The central sub-beam, whose float64 latitude / longitude are
// outputs, keeps the long form.  Measured against the long form on the C4 volume (tools/fast_sub_check.py,
// profiles/r5_fast_sub_check.json): see DESIGN.md section 4.  Ground radars with the 4/3-earth ray paths only.
#ifndef CPOL_GEO_NP
#define CPOL_GEO_NP 9
#endif
__device__ __forceinline__ void exact_rotated_coords(const ModelDev &m, const double *gc, double sin_u1, double cos_u1, double lon1,
                                                     double s2s1, double c2s1, double s, double &rlat_deg, double &rlon_deg)
{
    // (the statements of interp_gate's long form)
    const double sin_a1 = gc[0], cos_a1 = gc[1], sin_alpha = gc[3];
    const double bA = gc[4], B = gc[5], C = gc[6];
    const double f = CPOL_WGS84_F;
    const double sigma0 = s / bA;
    double sigma = sigma0, cos2sm = 0.0, sin_s = 0.0, cos_s = 1.0;
#pragma unroll 1
    for (int it = 0; it <= CPOL_VINCENTY_ITERS; ++it) {
        sincos(sigma, &sin_s, &cos_s);
        cos2sm = c2s1 * cos_s - s2s1 * sin_s;
        if (it == CPOL_VINCENTY_ITERS) break;
        const double dsig = B * sin_s * (cos2sm + B / 4.0 * (cos_s * (-1.0 + 2.0 * cos2sm * cos2sm)
                            - B / 6.0 * cos2sm * (-3.0 + 4.0 * sin_s * sin_s) * (-3.0 + 4.0 * cos2sm * cos2sm)));
        sigma = sigma0 + dsig;
    }
    const double tmp = sin_u1 * sin_s - cos_u1 * cos_s * cos_a1;
    const double lat_num = sin_u1 * cos_s + cos_u1 * sin_s * cos_a1;
    const double lat_den = (1.0 - f) * sqrt(sin_alpha * sin_alpha + tmp * tmp);
    const double lam_y = sin_s * sin_a1, lam_x = cos_u1 * cos_s - sin_u1 * sin_s * cos_a1;
    const double dlon = (1.0 - C) * f * sin_alpha * (sigma + C * sin_s * (cos2sm + C * cos_s * (-1.0 + 2.0 * cos2sm * cos2sm)));
    const double lat2 = atan2(lat_num, lat_den);
    const double L = atan2(lam_y, lam_x) - dlon;
    const double lat_deg = lat2 / CPOL_DEG, lon_deg = lon1 + L / CPOL_DEG;
    double sl, cl, slon, clon;
    sincos(lat_deg * CPOL_DEG, &sl, &cl);
    sincos(lon_deg * CPOL_DEG, &slon, &clon);
    const double x = clon * cl, y = slon * cl, z = sl;
    const double x_new = m.ctcp * x + m.ctsp * y + m.st * z;
    const double y_new = m.nsp * x + m.cp * y;
    const double z_new = m.nstcp * x - m.stsp * y + m.ct * z;
    rlon_deg = atan2(y_new, x_new) / CPOL_DEG;
    rlat_deg = asin(z_new) / CPOL_DEG;
}

// Ray paths ahead of the sweep kernel: with several horizontal quadrature nodes the sub-beams of one
// vertical node share their path (7 x 7 nodes: 6 of 7 evaluations of the refraction formulas saved, ~300
// float64 instructions per sub-beam gate); also the parity access to the paths (cpol_debug_read "traj").
// Same device function as the in-place evaluation: identical values.  The blocks of the first gate tile
// also evaluate the per-ray constants of the geodesic (sin / cos of 2 sigma1 per (ray, horizontal node),
// of the site longitude per ray) with the same OCML calls k_interp_sweep would make per gate.
// grid = (n_rays * n_v, ceil(n_gates / 256))
__global__ __launch_bounds__(256) void k_trajectory(ModelDev m, TrajArgs a)
{
    int g = blockIdx.y * blockDim.x + threadIdx.x;
    int rv = blockIdx.x;                       // ray * n_v + vnode
    if (a.poly && blockIdx.y == 0) {
        // the coordinate polynomials of (ray, horizontal node) e: NP threads per e, one node each (the long form of the
        // geodesic once per thread instead of NP times in a row: the share of one of 8 GPUs has 1 575 e and waited 36 us for
        // this kernel), the node values meet in LDS and thread (e, pw) forms monomial coefficient pw of rlat(x), rlon(x) from them
        constexpr int NP = CPOL_GEO_NP, EPB = 256 / NP;
        __shared__ double s_node[2][EPB][NP];
        const long n_geo = (long)a.n_rays * a.n_h;
        const int el = threadIdx.x / NP, q = threadIdx.x % NP;
        for (long e0 = (long)blockIdx.x * EPB; e0 < n_geo; e0 += (long)gridDim.x * EPB) {
            const long e = e0 + el;
            const bool ok = el < EPB && e < n_geo;
            if (ok) {
                const double *gc = a.geo + e * 8;
                double s2s1, c2s1;
                sincos(2.0 * gc[2], &s2s1, &c2s1);
                const double xq = cos(3.14159265358979323846 * ((double)q + 0.5) / (double)NP);
                double vlat, vlon;
                exact_rotated_coords(m, gc, a.sin_u1, a.cos_u1, a.lon1, s2s1, c2s1, (xq + 1.0) / a.poly_scale, vlat, vlon);
                s_node[0][el][q] = vlat;
                s_node[1][el][q] = vlon;
            }
            __syncthreads();
            if (ok) {
                double c0 = 0.0, c1 = 0.0;
                // (round 6, advisor: a fit nobody compares with the long form must not be badly conditioned -- a rotated longitude that
                // wraps at +-180 deg between two nodes, a point next to the rotated pole.  Node values more than a degree apart over a
                // fraction of a 150-km ray, or not finite: the coefficients become NaN, the gate kernel's guard reads NaN as "neighbour
                // in another cell" and every gate of this (ray, node) takes the long form)
                bool sane = true;
                for (int k = 0; k < NP; ++k) {
                    c0 = fma(a.poly_M[q * NP + k], s_node[0][el][k], c0);
                    c1 = fma(a.poly_M[q * NP + k], s_node[1][el][k], c1);
                    if (k > 0) sane = sane && fabs(s_node[0][el][k] - s_node[0][el][k - 1]) <= 1.0 && fabs(s_node[1][el][k] - s_node[1][el][k - 1]) <= 1.0;
                }
                if (!sane) c0 = c1 = __builtin_nan("");
                double *o = a.poly + e * (2 * NP);
                o[q] = c0;
                o[NP + q] = c1;
            }
            __syncthreads();
        }
    }
    if (a.ray_const && blockIdx.y == 0) {
        const long n_geo = (long)a.n_rays * a.n_h;
        for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_geo + a.n_rays;
             e += (long)gridDim.x * blockDim.x) {
            double sn, cs;
            if (e < n_geo) {
                sincos(2.0 * a.geo[e * 8 + 2], &sn, &cs);
            } else {
                const long ray = e - n_geo;
                sincos((a.site ? a.site[ray * 8 + 2] : a.lon1) * CPOL_DEG, &sn, &cs);
            }
            a.ray_const[2 * e] = sn;
            a.ray_const[2 * e + 1] = cs;
        }
    }
    if (g >= a.n_gates || !a.traj_out) return;
    RayPathArgs rp;
    rp.ray_traj = a.ray_traj; rp.site = a.site; rp.n_v = a.n_v; rp.mode = a.mode;
    rp.range0 = a.range0; rp.range_step = a.range_step; rp.ke = a.ke; rp.re = a.re; rp.alt = a.alt;
    float s32, h32, e32;
    ray_path(rp, rv / a.n_v, rv, g, s32, h32, e32);
    float *o = a.traj_out + (long)rv * 3 * a.n_gates;
    o[g] = s32;
    o[a.n_gates + g] = h32;
    o[2 * a.n_gates + g] = e32;
}

// first candidate gate whose height is below `ceiling` (heights decrease along a
// downward ray): one thread per (ray, vertical node)
__global__ void k_spaceborne_first_gate(const double *__restrict__ ray_traj,
                                        const double *__restrict__ site,
                                        const int *__restrict__ n_cand, int *__restrict__ first,
                                        int n_rays, int n_v, double range0, double range_step,
                                        double ceiling)
{
    int rv = blockIdx.x * blockDim.x + threadIdx.x;
    if (rv >= n_rays * n_v) return;
    const int ray = rv / n_v;
    const double sin_el = ray_traj[rv * 4 + 1];
    const double alt = site[(long)ray * 8 + 3], re = site[(long)ray * 8 + 4];
    int lo = 0, hi = n_cand[ray];              // answer in [lo, hi]
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        double h = spaceborne_height(range0 + (double)mid * range_step, re, sin_el, alt);
        if (h < ceiling) hi = mid; else lo = mid + 1;
    }
    first[rv] = lo;
}

// ---------------------------------------------------------------- gate kernel
#ifdef CPOL_INTERP_TRACE
// (measurement build: phase k of this wavefront on the 100-MHz clock, after everything issued before it has arrived)
#define ITRACE(tr, k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); (tr)[k] = wall_clock64(); } while (0)
#define ITRACE_ARG , unsigned long long *itr
#define ITRACE_PASS , itr
#else
#define ITRACE(tr, k) do { } while (0)
#define ITRACE_ARG
#define ITRACE_PASS
#endif
// largest i in [0, n-2] with col[i] >= key (col strictly descending, col[0] >= key)
__device__ __forceinline__ int level_search(const float *__restrict__ col, int n, float key)
{
    int step = 1;
    while ((step << 1) <= n - 2) step <<= 1;    // wave-uniform
    int i = 0;
#pragma unroll 1
    for (; step >= 1; step >>= 1) {
        int j = i + step;
        if (j <= n - 2 && col[j] >= key) i = j;
    }
    return i;
}

// ---- IEEE float32 quotients n / d as float64 products (round 6) ----
// The compiler's correctly rounded float32 division is 11 instructions (v_div_scale x2, v_rcp, 2 + 4 FMA steps, v_div_fmas,
// v_div_fixup), and a gate divides 36 + 6 times -- by FOUR distinct column thicknesses and the two grid resolutions: a fifth of
// the gate kernel's instructions.  RN32(RN64(n * r)) with r within one float64 ulp of 1 / d IS RN32(n / d) for every pair of
// float32 operands: the exact quotient of two 24-bit significands is never a float32 rounding boundary (a 25-bit number m with
// m d = n would need 25 significant bits in n) and keeps a relative distance >= 2^-49 from the nearest one, while the product
// is within 2^-53 (its own rounding) + 2^-52 (r) of it -- subnormal quotients have coarser boundaries, overflow is the boundary
// 2^128 (1 - 2^-25), zeros / infinities / NaN follow from r = 1 / d as IEEE gives it (the last branch below).  One reciprocal
// per column (v_rcp_f64, 2^-23, and two Newton steps: 2^-46, then the final FMA's rounding) and 3 instructions per quotient.
// Pinned by tests/test_gpu_math.py (10^8 random operand pairs of every class against `/`, bit for bit) and by every bit-exact
// test of the gate kernel (test_gate_kernel_bit_exact: uint32 views against the gcc-compiled reference C).
#ifndef CPOL_DIV_AS_PRODUCT
#define CPOL_DIV_AS_PRODUCT 1
#endif
__device__ __forceinline__ double rcp_for_div32(float d)
{
    const double x = (double)d;
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    if (!(fabsf(d) > 0.0f) || !(fabsf(d) < __builtin_inff())) r = 1.0 / x;         // (0, inf, NaN: what IEEE gives)
    return r;
}
__device__ __forceinline__ float div32_by(float n, double r) { return (float)((double)n * r); }

struct __attribute__((packed, aligned(4))) F2 { float v[2]; };
struct __attribute__((packed, aligned(4))) F4 { float v[4]; };
#ifndef CPOL_LEVEL_SEARCH_WIDE
#define CPOL_LEVEL_SEARCH_WIDE 1
#endif
#ifndef CPOL_INTERP_XCD_RAYS
#define CPOL_INTERP_XCD_RAYS 1
#endif

struct GateGeom {
    int status;            // 0 ok, +1 above model top, -1 below topography
    float x, y, dx, dy;    // fractional position (x along rows/lat, y along cols/lon)
    long cell[4];          // the 4 neighbour columns (i0,i1) (i0,i1+1) (i0+1,i1) (i0+1,i1+1)
    int c1[4];             // upper level of the bracketing pair (c2 = c1+1)
    float z1[4], z2[4];
    double rz[4];          // 1 / (z1 - z2) for div32_by (CPOL_DIV_AS_PRODUCT)
};

__device__ __forceinline__ void gate_geometry(const ModelDev &m, float rlat, float rlon, float h,
                                              GateGeom &g ITRACE_ARG)
{
    // interpolation_c.c:43-57
#if CPOL_DIV_AS_PRODUCT
    float p0 = div32_by(rlat - m.llc1, m.rres1);
    float p1 = div32_by(rlon - m.llc0, m.rres0);
#else
    float p0 = (rlat - m.llc1) / m.res1;
    float p1 = (rlon - m.llc0) / m.res0;
#endif
    int i0 = (int)floor((double)p0);
    int i1 = (int)floor((double)p1);
    // fmodf(p, 1.0f) = p - trunc(p) with the sign of p: exact (the difference of a float and its integer part
    // is representable), so the same bits as the C library's loop (OCML: ~25 instructions each, these: 3)
    g.x = copysignf(p0 - truncf(p0), p0);
    g.y = copysignf(p1 - truncf(p1), p1);
    g.dx = 1.0f - g.x;
    g.dy = 1.0f - g.y;
    // the reference does not range-check; callers pass the domain check first.
    // Clamp so that a gate exactly on the upper domain edge cannot fault.
    int i0a = min(max(i0, 0), m.ny - 1), i0b = min(max(i0 + 1, 0), m.ny - 1);
    int i1a = min(max(i1, 0), m.nx - 1), i1b = min(max(i1 + 1, 0), m.nx - 1);
    g.cell[0] = (long)i0a * m.nx + i1a;
    g.cell[1] = (long)i0a * m.nx + i1b;
    g.cell[2] = (long)i0b * m.nx + i1a;
    g.cell[3] = (long)i0b * m.nx + i1b;
    const int nz = m.nz;
    float t[4], top[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float2 tl = m.HT[g.cell[k]]; top[k] = tl.x; t[k] = tl.y; }
    // interpolation_c.c:61
    float topo = g.dx * g.dy * t[0] + g.x * t[2] * g.dy + g.dx * t[1] * g.y + g.x * g.y * t[3];
    ITRACE(itr, 2);                                        // topography of the four columns
    if (!(topo < h)) { g.status = -1; return; }
    g.status = 0;
    // Level search (interpolation_c.c:108-135: the largest i in [0, nz - 2] with col[i] >= h, 0 if
    // none; columns descend strictly with the level index).  Column 0 by bisection: 7 dependent
    // 4-byte gathers at nz = 80.  The three neighbour columns start from its answer -- terrain-following
    // levels put them within a level or two -- and check the bracket col[i] >= h > col[i + 1] with
    // ONE 8-byte load, stepping up or down while it fails: the same index, ~3.5 gather instructions
    // instead of 21, and the bracketing heights (z1, z2) come with it.  (The kernel is bound by the
    // cache lines its gathers touch, not by arithmetic.)
    const float *col[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) col[k] = m.H + g.cell[k] * nz;
    int idx[4];
    float za[4], zb[4];                 // col[idx], col[idx + 1]
    {
        int i = 0;
#if CPOL_LEVEL_SEARCH_WIDE
        // (round 6: the bisection is 7 DEPENDENT gathers -- 2.2 us of a wavefront's 13 on the C2 sweep, 4.2 of 21 on the C4 volume,
        // tools/interp_trace.py -- for 7 loads.  Strictly descending columns make the index a COUNT: the number of levels 1 .. nz - 2
        // at or above the gate.  Two rounds of independent loads: every 16th level, then the 15 levels behind the last of those at
        // or above -- 4 + 4 gathers, the same index.)
        if (nz >= 4 && nz <= 16 * 9) {
            float cv[8];
#pragma unroll
            for (int q = 1; q <= 4; ++q) cv[q - 1] = col[0][min(16 * q, nz - 1)];   // (independent loads, no branch between them)
#pragma unroll
            for (int q = 5; q <= 8; ++q) cv[q - 1] = 0.0f;
            if (nz - 2 >= 16 * 5) {                                                  // (wave-uniform)
#pragma unroll
                for (int q = 5; q <= 8; ++q) cv[q - 1] = col[0][min(16 * q, nz - 1)];
            }
            int c = 0;
#pragma unroll
            for (int q = 1; q <= 8; ++q) c += (16 * q <= nz - 2 && cv[q - 1] >= h) ? 1 : 0;
            const int base = 16 * c;
            // levels base + 1 .. base + 16 (the last is the next of the sixteenths -- below the gate, or past nz - 2 -- and counts nothing)
            F4 w[4];
            int s0[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s0[r] = min(base + 1 + 4 * r, nz - 4);                               // (window kept inside the column)
                w[r] = *(const F4 *)(col[0] + s0[r]);
            }
            int cnt = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int lev = s0[r] + t;
                    cnt += (lev >= base + 1 + 4 * r && lev <= nz - 2 && w[r].v[t] >= h) ? 1 : 0;
                }
            i = base + cnt;
        } else
#endif
        {
            int step = 1;
            while ((step << 1) <= nz - 2) step <<= 1;        // wave-uniform
#pragma unroll 1
            for (; step >= 1; step >>= 1) {
                const int j = min(i + step, nz - 2);
                const float v = col[0][j];
                if (i + step <= nz - 2 && v >= h) i = j;
            }
        }
        idx[0] = i;
        const F2 p = *(const F2 *)(col[0] + i);
        za[0] = p.v[0]; zb[0] = p.v[1];
    }
    ITRACE(itr, 3);                                        // column 0 bisected
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        int i = idx[0];
        F2 p = *(const F2 *)(col[k] + i);
        float a = p.v[0], b = p.v[1];
#pragma unroll 1
        for (;;) {
            if (a >= h) {
                if (i == nz - 2 || !(b >= h)) break;
                ++i; a = b; b = col[k][i + 1];
            } else {
                if (i == 0) break;
                --i; b = a; a = col[k][i];
            }
        }
        idx[k] = i; za[k] = a; zb[k] = b;
    }
    ITRACE(itr, 4);                                        // the three neighbour columns bracketed
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (h > top[k]) { g.status = 1; return; }            // interpolation_c.c:70-73
        int c1 = (h < t[k]) ? nz - 3 : min(idx[k], nz - 3);   // :74-81
        g.c1[k] = c1;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (g.c1[k] == idx[k]) {
            g.z1[k] = za[k]; g.z2[k] = zb[k];
        } else {                                             // below the lowest level / at nz - 2: rare
            g.z1[k] = col[k][g.c1[k]];
            g.z2[k] = col[k][g.c1[k] + 1];
        }
    }
#if CPOL_DIV_AS_PRODUCT
#pragma unroll
    for (int k = 0; k < 4; ++k) g.rz[k] = rcp_for_div32(g.z1[k] - g.z2[k]);
#endif
}

__device__ __forceinline__ float gate_value(const ModelDev &m, const GateGeom &g, float h, int v)
{
    float val[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float *p = m.V + ((g.cell[k] * m.nz + g.c1[k]) * m.n_vars + v);
        float v1 = p[0], v2 = p[m.n_vars];
        // interpolation_c.c:151
#if CPOL_DIV_AS_PRODUCT
        val[k] = v2 - div32_by(v2 - v1, g.rz[k]) * (h - g.z2[k]);
#else
        val[k] = v2 - (v2 - v1) / (g.z1[k] - g.z2[k]) * (h - g.z2[k]);
#endif
    }
    // interpolation_c.c:162
    return g.dx * g.dy * val[0] + g.x * val[2] * g.dy + g.dx * val[1] * g.y + g.x * g.y * val[3];
}

// Four consecutive variables of one gate at once: the variables of a (cell, level) are
// contiguous in V, so each of the 8 neighbours is ONE 16-byte load per lane instead of four
// 4-byte ones (every lane gathers from its own cache lines: the load COUNT is what the
// texture path pays for).  Same arithmetic, statement by statement, as gate_value.
// (Round 3 measured and dropped the 36 float32 divisions of a gate as ONE float64 reciprocal per column and a
// float64 product rounded to float32: 2.17 -> 2.11 ms on the C4 volume at the same 5 wavefronts per SIMD, 2.52 ms
// at the 3 the allocator then chose.  Round 6 has it -- div32_by above, with the proof and the device comparison --
// inside the fused kernel's fixed register budget.)

__device__ __forceinline__ void gate_value4(const ModelDev &m, const GateGeom &g, float h, int v0,
                                            float out[4])
{
    float val[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float *p = m.V + ((g.cell[k] * m.nz + g.c1[k]) * m.n_vars + v0);
        const F4 a = *(const F4 *)p, b = *(const F4 *)(p + m.n_vars);
        const float dh = h - g.z2[k];
#if CPOL_DIV_AS_PRODUCT
#pragma unroll
        for (int j = 0; j < 4; ++j) val[k][j] = b.v[j] - div32_by(b.v[j] - a.v[j], g.rz[k]) * dh;
#else
        const float dz = g.z1[k] - g.z2[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) val[k][j] = b.v[j] - (b.v[j] - a.v[j]) / dz * dh;
#endif
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        out[j] = g.dx * g.dy * val[0][j] + g.x * val[2][j] * g.dy + g.dx * val[1][j] * g.y
                 + g.x * g.y * val[3][j];
}

// explicit points -> all variables, reference sentinels (cpol_interp_points)
__global__ void k_interp_points(ModelDev m, const float *__restrict__ coords,
                                const float *__restrict__ heights, float *__restrict__ out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    GateGeom g;
    float h = heights[i];
#ifdef CPOL_INTERP_TRACE
    unsigned long long itr[8];
#endif
    gate_geometry(m, coords[2 * i], coords[2 * i + 1], h, g ITRACE_PASS);
    for (int v = 0; v < m.n_vars; ++v) {
        float r;
        if (g.status == 1) r = -9999.0f;
        else if (g.status == -1) r = __builtin_nanf("");
        else r = gate_value(m, g, h, v);
        out[(long)v * n + i] = r;
    }
}

// ---------------------------------------------------------------- sweep kernel
// One thread per sub-beam gate; a wave = 64 consecutive gates of ONE sub-beam
// (gate-stride coalesced stores, neighbouring gates share grid columns).
// (Measured and dropped: wavefronts of 7 vertical quadrature nodes x 9 gates -- the same model columns at
// different heights, a quarter of the distinct columns per gather -- 2.18 -> 2.40 ms on the C4 volume: since
// the hinted level search the kernel is bound by its float64 arithmetic, 1 660 VALU instructions per
// sub-beam gate = 1.87 ms at full issue rate, and the scattered 36-byte store segments cost more than the
// gathers save.)
// grid = (n_rays * n_sub, ceil(n_gates/256)): no 65535 limit on the number of rays
struct InterpArgs {
    const float *traj;          // [n_rays][n_v][3][n_gates] ray paths: host-supplied (CPOL_GEOM_HOST_PATHS) or
                                // k_trajectory's (sub-beams sharing a vertical node); NULL: ray_path() in place
    const double *ray_const;    // k_trajectory's per-ray constants or NULL: evaluated per gate
    RayPathArgs rp;
    int *zero_buf;              // the NEXT sweep's bucket counters, cleared here (no fill kernel)
    int zero_n;
    int *zero_buf2;             // ... and the work-unit totals of the single-beam fast path (k_gate1 counts into them) or NULL
    int zero_n2;
    const double *geo;          // [n_rays][n_h][8]
    const int *sub_h, *sub_v;
    float *vals;                // [n_vars][n_sbg]
    signed char *mask;          // [n_sbg]
    float *elev;                // [n_sbg] folded elevation (doppler_scatter.py:173-176)
    float *coords;              // [n_sbg][2] rotated (lat, lon)  (debug / parity)
    double *lats, *lons;        // [n_rays*n_gates] central sub-beam
    float *dist, *heights;      // [n_rays*n_gates] central sub-beam
    int *error_flag;
    unsigned store_mask;        // k_interp_classify: bit v = variable v is read by a later kernel and goes to vals[]
    int n_rays, n_gates, n_sub, n_h, n_v, central_sub;
    double sin_u1, cos_u1, lon1;
    const double *site;         // per-ray site or NULL
    int exact_sub;              // debug (cpol_sweep_params.debug_flags & CPOL_DEBUG_EXACT_SUBBEAMS): every sub-beam takes the central one's long form
    const double *poly;         // k_trajectory's coordinate polynomials [n_rays * n_h][2][CPOL_GEO_NP] or NULL
    double poly_scale;          // x = s * poly_scale - 1
    int poly_central;           // the central sub-beam takes the polynomials too (its float64 latitude / longitude are not asked for)
    // k_gate1_ray's sweeps (round 6, CPOL_GATE1_PRESENT builds only): one word per (ray, 64-gate tile) -- bit s set if ANY gate of the tile has pres_var[s] > 0, the
    // necessary condition of an item of hydrometeor slot s (classify_item: valid = inside && qm > 0 ...).  The tile is this kernel's
    // wavefront AND k_gate1_ray's workgroup: the wavefront of a species absent from the tile (60 % of them on the C2 sweep) then
    // neither loads nor classifies nor writes terms, it takes its ticket and leaves.  NULL: not produced.
    unsigned *present;          // [n_rays][ceil(n_gates / 64)]
    int n_pres;
    int pres_var[CPOL_MAX_HYDRO];
};

#ifndef CPOL_RAY_PREP_MIN_SUB
#define CPOL_RAY_PREP_MIN_SUB 4      // sub-beams per radial from which k_trajectory runs ahead of the sweep kernel
#endif
#ifndef CPOL_INTERP_FAST_SUB
#define CPOL_INTERP_FAST_SUB 1       // 0: every sub-beam through atan2 -> degrees -> sincos like the central one
#endif
#ifndef CPOL_INTERP_WAVES
#define CPOL_INTERP_WAVES 0          // > 0: ask for that many wavefronts per SIMD (register budget 512 / waves)
#endif
#if CPOL_INTERP_WAVES > 0
#define CPOL_INTERP_ATTR __attribute__((amdgpu_waves_per_eu(CPOL_INTERP_WAVES, CPOL_INTERP_WAVES)))
#else
#define CPOL_INTERP_ATTR
#endif
// 1 / sqrt(x) for x of order 1: v_rsq_f64 (~27 bits) and two Newton steps y <- y (3/2 - x/2 y^2); within 2 ulp
__device__ __forceinline__ double rsqrt_newton(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}

// a * b + c with a wave-uniform (literal) addend from a scalar register pair: left to itself the compiler forms
// v_fmac_f64 and first moves every coefficient into vector registers (two v_mov_b32 per Horner step)
__device__ __forceinline__ double fma_kc(double a, double b, double c_uniform)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_uniform));
    return r;
}

// atan(t) and asin(z) for |t|, |z| <= 0.2 (the rotated coordinates of a limited-area domain: +-11 degrees) as their
// Maclaurin series, 13 / 12 terms (next term < 7e-19 / 5e-18 relative): within 2 ulp of the C library's value, in ~15
// instructions instead of the ~90 / ~60 of the general atan2 / asin.
__device__ __forceinline__ double atan_small(double t)
{
    const double z = t * t;
    double p = 1.0 / 25.0;
    p = fma_kc(p, z, -1.0 / 23.0);
    p = fma_kc(p, z, 1.0 / 21.0);
    p = fma_kc(p, z, -1.0 / 19.0);
    p = fma_kc(p, z, 1.0 / 17.0);
    p = fma_kc(p, z, -1.0 / 15.0);
    p = fma_kc(p, z, 1.0 / 13.0);
    p = fma_kc(p, z, -1.0 / 11.0);
    p = fma_kc(p, z, 1.0 / 9.0);
    p = fma_kc(p, z, -1.0 / 7.0);
    p = fma_kc(p, z, 1.0 / 5.0);
    p = fma_kc(p, z, -1.0 / 3.0);
    p = fma(p, z, 1.0);
    return t * p;
}

__device__ __forceinline__ double asin_small(double x)
{
    const double z = x * x;
    double p = 88179.0 / 12058624.0;
    p = fma_kc(p, z, 46189.0 / 5505024.0);
    p = fma_kc(p, z, 12155.0 / 1245184.0);
    p = fma_kc(p, z, 6435.0 / 557056.0);
    p = fma_kc(p, z, 143.0 / 10240.0);
    p = fma_kc(p, z, 231.0 / 13312.0);
    p = fma_kc(p, z, 63.0 / 2816.0);
    p = fma_kc(p, z, 35.0 / 1152.0);
    p = fma_kc(p, z, 5.0 / 112.0);
    p = fma_kc(p, z, 3.0 / 40.0);
    p = fma_kc(p, z, 1.0 / 6.0);
    p = fma(p, z, 1.0);
    return x * p;
}

// One sub-beam gate: ray path, geodesic, rotated-pole coordinates, level search, the variables.  Returns the
// gate's status (0: inside the model, values valid; 1 / -1 / 2: above / below / outside, values NaN; 3: no such
// gate in this launch) and its index.  KEEP = false: every variable goes to a.vals[] (k_interp_sweep).
// KEEP = true (k_interp_classify): the values of a status-0 gate stay with the thread -- sv[v * blockDim.x],
// its column of the workgroup's LDS array -- and the caller stores what later kernels read; of the NaN of the
// other gates only the variables in a.store_mask are written.
template <bool KEEP>
__device__ __forceinline__ int interp_gate(const ModelDev &m, const InterpArgs &a, float *sv, long &sbg_out, float &elev_out)
{
    // ---- which (ray, sub-beam, block of gates) this workgroup takes ----
    // The hardware deals workgroups to the 8 XCDs round robin by their linear index, and every XCD has its own L2.  With
    // blockIdx.x = ray * n_sub + sub the 49 sub-beams of a ray -- which read the same model columns, a few kilometres apart --
    // went to all eight L2s, each fetching the columns for itself.  Here XCD k takes the rays k, k + 8, ... whole, in order
    // (round 6; the rays beyond a multiple of 8 in the plain order): the sub-beams of a ray, and both halves of its gates, share one L2.
    int bx = blockIdx.x, by = blockIdx.y;
#if CPOL_INTERP_XCD_RAYS
    if (a.n_sub > 1 && gridDim.x == (unsigned)(a.n_rays * a.n_sub)) {                            // (wave-uniform)
        const unsigned L = blockIdx.y * gridDim.x + blockIdx.x;                                   // (decoded from scratch: any bijection will do)
        const unsigned per_ray = (unsigned)a.n_sub * gridDim.y;
        const unsigned n8 = (unsigned)a.n_rays & ~7u, T8 = n8 * per_ray;                         // (the rays beyond a multiple of 8: plain order)
        unsigned ray_, w;
        if (L < T8) {
            const unsigned xcd = L & 7u, q = L >> 3;
            ray_ = (q / per_ray) * 8u + xcd;
            w = q % per_ray;
        } else {
            ray_ = n8 + (L - T8) / per_ray;
            w = (L - T8) % per_ray;
        }
        bx = (int)(ray_ * (unsigned)a.n_sub + w % (unsigned)a.n_sub);
        by = (int)(w / (unsigned)a.n_sub);
    }
#endif
    const int gate = by * blockDim.x + threadIdx.x;
    const int sub = bx % a.n_sub, ray = bx / a.n_sub;
#ifdef CPOL_INTERP_TRACE
    unsigned long long itr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    itr[0] = wall_clock64();
#endif
    if (gate >= a.n_gates) return 3;
    const int ih = a.sub_h[sub], jv = a.sub_v[sub];
    const long sbg = ((long)ray * a.n_sub + sub) * a.n_gates + gate;
    const long n_sbg = (long)a.n_rays * a.n_sub * a.n_gates;
    sbg_out = sbg;
    elev_out = 0.0f;

    float s32, h32, e32;
    if (a.traj) {
        const float *tr = a.traj + ((long)(ray * a.n_v + jv) * 3) * a.n_gates;
        s32 = tr[gate]; h32 = tr[a.n_gates + gate]; e32 = tr[2 * a.n_gates + gate];
    } else {
        ray_path(a.rp, ray, ray * a.n_v + jv, gate, s32, h32, e32);
    }
    const float qnan = __builtin_nanf("");
    if (!(s32 == s32) || !(h32 == h32)) {
        // no gate here (ray shorter than the batch: spaceborne / host paths): counts
        // as "above the model", produces no item
        a.mask[sbg] = 1;
        for (int v = 0; v < m.n_vars; ++v)
            if (!KEEP || ((a.store_mask >> v) & 1u)) a.vals[(long)v * n_sbg + sbg] = qnan;
        a.elev[sbg] = 0.0f;
        if (a.coords) { a.coords[2 * sbg] = qnan; a.coords[2 * sbg + 1] = qnan; }
        if (sub == a.central_sub) {
            const long rg = (long)ray * a.n_gates + gate;
            if (a.lats) a.lats[rg] = __builtin_nan("");
            if (a.lons) a.lons[rg] = __builtin_nan("");
            if (a.dist) a.dist[rg] = qnan;
            if (a.heights) a.heights[rg] = qnan;
        }
        return 1;
    }
    float rlon, rlat;
    double lat_deg = 0.0, lon_deg = 0.0;
    // (wave-uniform: a wavefront walks ONE sub-beam; exact_sub is a kernel argument)
    const bool use_poly = a.poly && (sub != a.central_sub || a.poly_central) && !a.exact_sub;
    const bool short_form = sub != a.central_sub && !a.exact_sub && !a.poly;      // (the closed short form of round 4: CPOL_GEO_POLY=0)
    // (round 6: the form no longer depends on what the caller asks for.  A single-beam sweep whose float64 latitude / longitude ARE
    // outputs runs the long form for those two arrays and still takes its float32 grid coordinates from the guarded polynomials,
    // so that the first sweep of a table set -- which fetches the gate coordinates -- and every later one give the same bits)
    const bool want_latlon = use_poly && sub == a.central_sub && (a.lats || a.lons);
    bool long_form = !use_poly;
    float rlat_p = 0.0f, rlon_p = 0.0f;
    if (use_poly) {
        // ---- the rotated coordinates from the (ray, horizontal node)'s polynomials (k_trajectory / k_geo_poly) ----
        constexpr int NP = CPOL_GEO_NP;
        const double *pc = a.poly + (long)(ray * a.n_h + ih) * (2 * NP);          // (wave-uniform: scalar loads)
        const double x = fma((double)s32, a.poly_scale, -1.0);
        double la = pc[NP - 1], lo = pc[2 * NP - 1];
#pragma unroll
        for (int q = NP - 2; q >= 0; --q) {
            la = fma(la, x, pc[q]);
            lo = fma(lo, x, pc[NP + q]);
        }
        rlat = rlat_p = (float)la;
        rlon = rlon_p = (float)lo;
        // GUARD: the polynomials follow the long form to ~2e-13 deg (degree 8 through 9 Chebyshev nodes over 0.024 rad of arc;
        // float32 ulp at 3 deg: 2.4e-7), so the two float32 coordinates are the same number or neighbours.  If the coordinate's
        // neighbours on both sides (c -/+ 1..2 ulp) fall into the SAME cell of the model grid and inside the domain -- the
        // reference's own float32 expressions, monotonic in c -- the cell index, the domain check and the corner columns are
        // the long form's whatever it rounds to; only the weights inside the cell can move (by <= 1 ulp of the cell
        // coordinate, in ~6e-7 of the gates: profiles/r5_fast_sub_check.json).  Otherwise (one gate in ~5 000) the gate takes
        // the long form after all: index work is the long form's by construction.
        auto spans = [](float c, float llc, float urc, float res, double rres) {
            // 2^-23 max(|c|, |llc|): between 1 and 2 ulps of c, and never below the rounding of `c - llc` (round 6, advisor: the rotated
            // coordinates of a limited-area model sit near 0, where an ulp of c falls below the polynomial's own ~2e-13 deg)
            const float d = fmaxf(fmaxf(fabsf(c), fabsf(llc)), 1.0e-30f) * 1.1920929e-7f;
            const float lo = c - d, hi = c + d;
#if CPOL_DIV_AS_PRODUCT
            return floorf(div32_by(lo - llc, rres)) != floorf(div32_by(hi - llc, rres)) || !(lo >= llc) || !(hi <= urc);   // (NaN: true)
#else
            return floorf((lo - llc) / res) != floorf((hi - llc) / res) || !(lo >= llc) || !(hi <= urc);   // (NaN: true)
#endif
        };
        long_form = spans(rlat, m.llc1, m.urc1, m.res1, m.rres1) || spans(rlon, m.llc0, m.urc0, m.res0, m.rres0);
    }
    const bool poly_ok = use_poly && !long_form;
    if (long_form || want_latlon) {
    const double sin_u1 = a.site ? a.site[(long)ray * 8 + 0] : a.sin_u1;
    const double cos_u1 = a.site ? a.site[(long)ray * 8 + 1] : a.cos_u1;
    const double lon1 = a.site ? a.site[(long)ray * 8 + 2] : a.lon1;

    // ---- WGS84 direct geodesic (Vincenty, fixed iteration count) ----
    const double *gc = a.geo + (long)(ray * a.n_h + ih) * 8;
    const double sin_a1 = gc[0], cos_a1 = gc[1], sigma1 = gc[2], sin_alpha = gc[3];
    const double bA = gc[4], B = gc[5], C = gc[6];
    const double f = CPOL_WGS84_F;
    const double sigma0 = (double)s32 / bA;
    double sigma = sigma0, cos2sm, sin_s, cos_s;
    // cos(2 sigma1 + sigma) by the addition theorem from one sincos(sigma) per iteration;
    // sigma = arc / b is a few 1e-2 rad at radar ranges, so sin / cos are short Taylor sums
    // (next terms sigma^13/13!, sigma^12/12! < 1e-19 for |sigma| < 0.1), OCML beyond
    double s2s1, c2s1;
    if (a.ray_const) {                          // (wave-uniform: scalar loads)
        const double *rc = a.ray_const + 2 * (long)(ray * a.n_h + ih);
        s2s1 = rc[0]; c2s1 = rc[1];
    } else {
        sincos(2.0 * sigma1, &s2s1, &c2s1);
    }
    // (the correction converges by a factor B ~ 1.7e-3 per pass: after 4 passes sigma is within 4e-16 rad -- 1e-14
    // relative -- of the fixed point, after 5 within rounding.  The central sub-beam, whose float64 coordinates
    // are outputs, takes 5; the others, of which only the float32 grid coordinates are used, 4.)
#if CPOL_INTERP_FAST_SUB
    const int n_iter = short_form ? CPOL_VINCENTY_ITERS - 1 : CPOL_VINCENTY_ITERS;
#else
    const int n_iter = CPOL_VINCENTY_ITERS;
#endif
#pragma unroll 1
    for (int it = 0; it <= n_iter; ++it) {
        if (fabs(sigma) < 0.1) {
            const double z = sigma * sigma;
            sin_s = sigma * fma(z, fma(z, fma(z, fma(z, fma(z, -1.0 / 39916800.0, 1.0 / 362880.0),
                                                     -1.0 / 5040.0), 1.0 / 120.0), -1.0 / 6.0), 1.0);
            cos_s = fma(z, fma(z, fma(z, fma(z, fma(z, -1.0 / 3628800.0, 1.0 / 40320.0), -1.0 / 720.0),
                                      1.0 / 24.0), -0.5), 1.0);
        } else {
            sincos(sigma, &sin_s, &cos_s);
        }
        cos2sm = c2s1 * cos_s - s2s1 * sin_s;
        if (it == n_iter) break;                        // values of the converged sigma
        double dsig = B * sin_s * (cos2sm + B / 4.0 * (cos_s * (-1.0 + 2.0 * cos2sm * cos2sm)
                      - B / 6.0 * cos2sm * (-3.0 + 4.0 * sin_s * sin_s)
                      * (-3.0 + 4.0 * cos2sm * cos2sm)));
        sigma = sigma0 + dsig;
    }
    const double tmp = sin_u1 * sin_s - cos_u1 * cos_s * cos_a1;
    const double lat_num = sin_u1 * cos_s + cos_u1 * sin_s * cos_a1;
    const double lat_den = (1.0 - f) * sqrt(sin_alpha * sin_alpha + tmp * tmp);
    const double lam_y = sin_s * sin_a1, lam_x = cos_u1 * cos_s - sin_u1 * sin_s * cos_a1;
    // L = lam - dlon: the ellipsoidal correction of the longitude difference (|dlon| < f sigma ~ 1e-4 rad)
    const double dlon = (1.0 - C) * f * sin_alpha *
        (sigma + C * sin_s * (cos2sm + C * cos_s * (-1.0 + 2.0 * cos2sm * cos2sm)));
    double sl, cl, slon, clon;                    // sin / cos of the geographic latitude and longitude
#if CPOL_INTERP_FAST_SUB
    if (short_form) {
        // Sub-beams other than the central one (wave-uniform: a wavefront walks ONE sub-beam) need the
        // geographic coordinates only as sin / cos for the rotated-pole transform: taken straight from the
        // arguments of the two atan2 (sin = y / hypot, cos = x / hypot) and the addition theorem, instead of
        // atan2 -> degrees -> radians -> sincos.  Same values to a few ulp (float64) before the float32 cast
        // of the rotated coordinates, i.e. the same float32 coordinate in all but ~1e-8 of the cases, like
        // the device's Taylor sums above; 4 OCML calls of ~100 instructions fewer per sub-beam gate.
        // (1 / sqrt as the hardware's reciprocal square root + two Newton steps -- the arguments are ~1 -- instead
        // of a correctly rounded square root and a correctly rounded division: ~10 instructions instead of ~45)
        const double hl = rsqrt_newton(lat_num * lat_num + lat_den * lat_den);
        sl = lat_num * hl; cl = lat_den * hl;
        const double hm = rsqrt_newton(lam_y * lam_y + lam_x * lam_x);
        const double sm = lam_y * hm, cm = lam_x * hm;                      // sin / cos of lam
        const double d2 = dlon * dlon;
        const double sd = dlon * fma(d2, -1.0 / 6.0, 1.0);                   // |dlon| < 2e-4: next terms < 1e-21
        const double cd = fma(d2, fma(d2, 1.0 / 24.0, -0.5), 1.0);
        const double sL = sm * cd - cm * sd, cL = cm * cd + sm * sd;        // lam - dlon
        double s1, c1;
        if (a.ray_const) {
            const double *rc = a.ray_const + 2 * ((long)a.n_rays * a.n_h + ray);
            s1 = rc[0]; c1 = rc[1];
        } else {
            sincos(lon1 * CPOL_DEG, &s1, &c1);
        }
        slon = s1 * cL + c1 * sL; clon = c1 * cL - s1 * sL;
    } else
#endif
    {
        const double lat2 = atan2(lat_num, lat_den);
        const double lam = atan2(lam_y, lam_x);
        const double L = lam - dlon;
        lat_deg = lat2 / CPOL_DEG;
        lon_deg = lon1 + L / CPOL_DEG;
        const double latr = lat_deg * CPOL_DEG, lonr = lon_deg * CPOL_DEG;
        sincos(latr, &sl, &cl);
        sincos(lonr, &slon, &clon);
    }

    // ---- rotated-pole transform (float64) -> float32 grid coordinates ----
    const double x = clon * cl, y = slon * cl, z = sl;
    const double x_new = m.ctcp * x + m.ctsp * y + m.st * z;
    const double y_new = m.nsp * x + m.cp * y;
    const double z_new = m.nstcp * x - m.stsp * y + m.ct * z;
#if CPOL_INTERP_FAST_SUB
    if (short_form) {
        // (radians -> degrees as a product with 180 / pi: a float64 division by a constant is ~25 instructions;
        // the last float64 bit may differ from the quotient's, the float32 cast hides it but for ~1e-8 of the gates)
        // (and the two angles from their short series where the rotated coordinates are small -- every limited-area
        // domain -- with 1 / x_new by Newton from v_rcp_f64: ~115 instructions fewer per sub-beam gate)
        double rx = __builtin_amdgcn_rcp(x_new);
        rx = rx * fma(-x_new, rx, 2.0);
        rx = rx * fma(-x_new, rx, 2.0);
        const double tq = y_new * rx;
        if (x_new > 0.5 && fabs(tq) <= 0.2 && fabs(z_new) <= 0.2) {
            rlon = (float)(atan_small(tq) * (1.0 / CPOL_DEG));
            rlat = (float)(asin_small(z_new) * (1.0 / CPOL_DEG));
        } else {
            rlon = (float)(atan2(y_new, x_new) * (1.0 / CPOL_DEG));
            rlat = (float)(asin(z_new) * (1.0 / CPOL_DEG));
        }
    } else
#endif
    {
        rlon = (float)(atan2(y_new, x_new) / CPOL_DEG);
        rlat = (float)(asin(z_new) / CPOL_DEG);
    }

    }
    if (poly_ok) { rlat = rlat_p; rlon = rlon_p; }     // (want_latlon: the long form ran for lat_deg / lon_deg alone)

    // interpolation.py:572-575 (IndexError in the reference)
    if (rlon < m.llc0 || rlat < m.llc1 || rlon > m.urc0 || rlat > m.urc1 ||
        !(rlon == rlon) || !(rlat == rlat)) {
        atomicOr(a.error_flag, 1);          // sticky until reported (cpol_synchronize / cpol_counters)
        a.mask[sbg] = 2;
        for (int v = 0; v < m.n_vars; ++v)
            if (!KEEP || ((a.store_mask >> v) & 1u)) a.vals[(long)v * n_sbg + sbg] = __builtin_nanf("");
        a.elev[sbg] = e32;
        elev_out = e32;
        if (a.coords) { a.coords[2 * sbg] = rlat; a.coords[2 * sbg + 1] = rlon; }
        if (sub == a.central_sub) {         // no stale data in the caller's buffers
            const long rg = (long)ray * a.n_gates + gate;
            if (a.lats) a.lats[rg] = __builtin_nan("");
            if (a.lons) a.lons[rg] = __builtin_nan("");
            if (a.dist) a.dist[rg] = qnan;
            if (a.heights) a.heights[rg] = qnan;
        }
        return 2;
    }

    GateGeom g;
    ITRACE(itr, 1);                                        // trajectory + grid coordinates
    gate_geometry(m, rlat, rlon, h32, g ITRACE_PASS);
#if CPOL_GATE1_PRESENT
    unsigned pres = 0;                                     // (KEEP = false: bit s = this gate has pres_var[s] > 0)
#endif
    if (g.status == 0) {
        int v = 0;
        for (; v + 4 <= m.n_vars; v += 4) {
            float o[4];
            gate_value4(m, g, h32, v, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (KEEP) sv[(v + j) * blockDim.x] = o[j];
                else a.vals[(long)(v + j) * n_sbg + sbg] = o[j];
#if CPOL_GATE1_PRESENT
                if (!KEEP && a.present)
                    for (int s = 0; s < a.n_pres; ++s) pres |= (a.pres_var[s] == v + j && o[j] > 0.f) ? 1u << s : 0u;
#endif
            }
        }
        for (; v < m.n_vars; ++v) {
            const float o = gate_value(m, g, h32, v);
            if (KEEP) sv[v * blockDim.x] = o;
            else a.vals[(long)v * n_sbg + sbg] = o;
#if CPOL_GATE1_PRESENT
            if (!KEEP && a.present)
                for (int s = 0; s < a.n_pres; ++s) pres |= (a.pres_var[s] == v && o > 0.f) ? 1u << s : 0u;
#endif
        }
    } else {
        for (int v = 0; v < m.n_vars; ++v)
            if (!KEEP || ((a.store_mask >> v) & 1u)) a.vals[(long)v * n_sbg + sbg] = qnan;
    }
    ITRACE(itr, 5);                                        // the variables gathered and interpolated (KEEP: in LDS)
    a.mask[sbg] = (signed char)g.status;

    // elevation folded into [0, 90] for the LUT (doppler_scatter.py:173-176, in place)
    if (e32 > 90.0f) e32 = 180.0f - e32;
    if (e32 < 0.0f) e32 = -e32;
    a.elev[sbg] = e32;
    elev_out = e32;
    if (a.coords) { a.coords[2 * sbg] = rlat; a.coords[2 * sbg + 1] = rlon; }

    if (sub == a.central_sub) {
        const long rg = (long)ray * a.n_gates + gate;
        if (a.lats) a.lats[rg] = lat_deg;
        if (a.lons) a.lons[rg] = lon_deg;
        if (a.dist) a.dist[rg] = s32;
        if (a.heights) a.heights[rg] = h32;
    }
#if CPOL_GATE1_PRESENT
    if (!KEEP && a.present) {
        // the tile's word: OR over the lanes that came this far (the others -- no gate, outside the domain -- hold no item); written by
        // the first of them.  A wavefront none of whose lanes comes here leaves its word as it was: whatever it says, the tile has no item.
        unsigned word = 0;
        for (int s = 0; s < a.n_pres; ++s)
            word |= __builtin_amdgcn_ballot_w64((pres >> s) & 1u) ? 1u << s : 0u;
        const int lane = (int)(threadIdx.x & 63);
        if (lane == __builtin_amdgcn_readfirstlane(lane))
            a.present[(long)ray * ((a.n_gates + 63) / 64) + gate / 64] = word;
    }
#endif
#ifdef CPOL_INTERP_TRACE
    {
        ITRACE(itr, 6);                                    // everything stored
        const unsigned long w = ((unsigned long)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x / 64) + threadIdx.x / 64;
        const unsigned long long n_ok = (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(g.status == 0));
        if ((threadIdx.x & 63) == 0 && w < CPOL_SUBSUM_TRACE_N) {
            for (int q = 0; q < 7; ++q) g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + q] = itr[q];
            g_subsum_trace[CPOL_SUBSUM_TRACE_W * w + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15ull) << 32 | n_ok << 40;
        }
    }
#endif
    return g.status;
}

__global__ __launch_bounds__(256) CPOL_INTERP_ATTR void k_interp_sweep(ModelDev m, InterpArgs a)
{
    clear_counters(a.zero_buf, a.zero_n, a.zero_buf2, a.zero_n2);
    long sbg;
    float e;
    interp_gate<false>(m, a, nullptr, sbg, e);
}
