// Microbenchmark: issue rate of the f64 MFMA instructions of gfx950 and of v_fma_f64, alone and
// interleaved (one wave per SIMD and four waves per SIMD).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void k(double *out, long long *cyc, int iters)
{
    double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 + threadIdx.x * 1e-4;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, v0 = 1, v1 = 1, v2 = 1, v3 = 1, v4 = 1, v5 = 1, v6 = 1, v7 = 1;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0 || MODE == 3) {          // 4 independent 16x16x4
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        if (MODE == 1 || MODE == 4) {          // 4 independent 4x4x4 (4 blocks)
            s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s2, 0, 0, 0);
            s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s3, 0, 0, 0);
        }
        if (MODE == 2 || MODE == 3 || MODE == 4) {   // 8 independent v_fma_f64
            v0 = fma(v0, a, b); v1 = fma(v1, a, b); v2 = fma(v2, a, b); v3 = fma(v3, a, b);
            v4 = fma(v4, a, b); v5 = fma(v5, a, b); v6 = fma(v6, a, b); v7 = fma(v7, a, b);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + s0 + s1 + s2 + s3
                                                + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)cyc, (unsigned long long)(t1 - t0));
}

template <int MODE>
void run(const char *name, int waves_per_simd)
{
    double *out; long long *cyc, h = 0;
    const int iters = 2000, threads = 64 * 4 * waves_per_simd;   // one workgroup on one CU
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
    k<MODE><<<1, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipMemset(cyc, 0, 8);
    k<MODE><<<1, threads>>>(out, cyc, iters);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s waves/SIMD %d: %.1f cycles per iteration (slowest wave)\n", name, waves_per_simd, (double)h / iters);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("4 x mfma_f64_16x16x4", w);
        run<1>("4 x mfma_f64_4x4x4_4b", w);
        run<2>("8 x v_fma_f64", w);
        run<3>("4 x 16x16x4 + 8 x v_fma_f64", w);
        run<4>("4 x 4x4x4 + 8 x v_fma_f64", w);
    }
    return 0;
}
