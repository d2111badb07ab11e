// How a wavefront gets ONE 1-D table block's coefficients to all 64 lanes (k_subbeam_sum): hipcc --offload-arch=gfx950 -O3
//   A  every coefficient pair read by all lanes from one LDS address (ds_read_b128 broadcast): round 4's form
//   B  the block's rows read ONCE (lane l of a row of 16 holds coefficient l of table row q), every coefficient
//      then broadcast inside the rows of 16 lanes by v_mov_b64_dpp row_newbcast, Horner order unchanged
//   C  as B, the polynomial as a sum of c_q u^q: the broadcast rides in v_fmac_f64_dpp (no separate move)
// time per (wavefront, block) with 16 wavefronts per CU resident, all CUs busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define NC 11
#define NFP 16
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int N> __device__ __forceinline__ double bcast16(double x)
{
    double r;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(N));
    return r;
}
template <int N> __device__ __forceinline__ void fmac_bcast16(double &acc, double c, double p)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "v"(p), "n"(N));
}

template <int FORM>
__global__ __launch_bounds__(64) void k(const double *tab, const double *us, double *out, int n_blocks, int reps)
{
    __shared__ double2 sb[NC * NFP / 2];
    const int lane = threadIdx.x;
    const double u = us[lane];
    double acc[12];
    for (int c = 0; c < 12; ++c) acc[c] = 0.0;
    for (int r = 0; r < reps; ++r) {
        const double2 *src = (const double2 *)(tab + (long)((blockIdx.x + r) % n_blocks) * NC * NFP);
        sb[lane] = src[lane];
        if (lane < NC * NFP / 2 - 64) sb[64 + lane] = src[64 + lane];
        __syncthreads();
        double2 v[6];
        if (FORM == 0) {
#pragma unroll
            for (int f = 0; f < 6; ++f) v[f] = sb[(NC - 1) * (NFP / 2) + f];
#pragma unroll 2
            for (int q = NC - 2; q >= 0; --q)
#pragma unroll
                for (int f = 0; f < 6; ++f) {
                    const double2 cq = sb[q * (NFP / 2) + f];
                    v[f].x = fma(v[f].x, u, cq.x);
                    v[f].y = fma(v[f].y, u, cq.y);
                }
        } else {
            double C[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) C[q] = ((const double *)sb)[q * NFP + (lane & 15)];
            if (FORM == 1) {
#define ROW(q, OP) { \
    OP(0, bcast16<0>(C[q]), bcast16<1>(C[q])) OP(1, bcast16<2>(C[q]), bcast16<3>(C[q])) OP(2, bcast16<4>(C[q]), bcast16<5>(C[q])) \
    OP(3, bcast16<6>(C[q]), bcast16<7>(C[q])) OP(4, bcast16<8>(C[q]), bcast16<9>(C[q])) OP(5, bcast16<10>(C[q]), bcast16<11>(C[q])) }
#define SET(f, a, b) v[f].x = a; v[f].y = b;
#define HOR(f, a, b) v[f].x = fma(v[f].x, u, a); v[f].y = fma(v[f].y, u, b);
                ROW(NC - 1, SET)
#pragma unroll
                for (int q = NC - 2; q >= 0; --q) ROW(q, HOR)
            } else {
                double p[NC];
                p[0] = 1.0;
#pragma unroll
                for (int q = 1; q < NC; ++q) p[q] = p[q - 1] * u;
#pragma unroll
                for (int f = 0; f < 6; ++f) v[f] = make_double2(0.0, 0.0);
#define ACC(q) { fmac_bcast16<0>(v[0].x, C[q], p[q]); fmac_bcast16<1>(v[0].y, C[q], p[q]); fmac_bcast16<2>(v[1].x, C[q], p[q]); fmac_bcast16<3>(v[1].y, C[q], p[q]); \
                 fmac_bcast16<4>(v[2].x, C[q], p[q]); fmac_bcast16<5>(v[2].y, C[q], p[q]); fmac_bcast16<6>(v[3].x, C[q], p[q]); fmac_bcast16<7>(v[3].y, C[q], p[q]); \
                 fmac_bcast16<8>(v[4].x, C[q], p[q]); fmac_bcast16<9>(v[4].y, C[q], p[q]); fmac_bcast16<10>(v[5].x, C[q], p[q]); fmac_bcast16<11>(v[5].y, C[q], p[q]); }
#pragma unroll
                for (int q = NC - 1; q >= 0; --q) ACC(q)
            }
        }
#pragma unroll
        for (int f = 0; f < 6; ++f) { acc[2 * f] += v[f].x; acc[2 * f + 1] += v[f].y; }
        __syncthreads();
    }
    for (int c = 0; c < 12; ++c) out[((long)blockIdx.x * 64 + lane) * 12 + c] = acc[c];
}

int main()
{
    const int n_blocks = 512, reps = 400, waves = 256 * 16;
    std::vector<double> tab((size_t)n_blocks * NC * NFP), us(64);
    for (size_t i = 0; i < tab.size(); ++i) tab[i] = 1.0 / (1.0 + (double)(i % 977)) - 0.3 / (1.0 + (double)(i % 31));
    for (int i = 0; i < 64; ++i) us[i] = -1.0 + 2.0 * i / 63.0;
    double *d_tab, *d_us, *d_out;
    CHECK(hipMalloc(&d_tab, tab.size() * 8)); CHECK(hipMalloc(&d_us, 64 * 8)); CHECK(hipMalloc(&d_out, (size_t)waves * 64 * 12 * 8));
    CHECK(hipMemcpy(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_us, us.data(), 64 * 8, hipMemcpyHostToDevice));
    std::vector<double> ref, got((size_t)waves * 64 * 12);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int form = 0; form < 3; ++form) {
        float best = 1e30f;
        for (int it = 0; it < 4; ++it) {
            CHECK(hipEventRecord(e0, 0));
            if (form == 0) hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, d_tab, d_us, d_out, n_blocks, reps);
            if (form == 1) hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, d_tab, d_us, d_out, n_blocks, reps);
            if (form == 2) hipLaunchKernelGGL(k<2>, dim3(waves), dim3(64), 0, 0, d_tab, d_us, d_out, n_blocks, reps);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        CHECK(hipMemcpy(got.data(), d_out, got.size() * 8, hipMemcpyDeviceToHost));
        size_t diff = 0; double worst = 0.0;
        if (form == 0) ref = got;
        else for (size_t i = 0; i < got.size(); ++i) if (got[i] != ref[i]) { ++diff; double r = fabs(got[i] - ref[i]) / (fabs(ref[i]) + 1e-300); if (r > worst) worst = r; }
        // per CU: 16 wavefronts x reps blocks
        printf("form %c: %.3f ms  %.1f ns per (wavefront, block) per CU  values that differ from A: %zu (worst rel %.2e)\n", 'A' + form, best,
               best * 1e6 / (16.0 * reps), diff, worst);
    }
    return 0;
}
