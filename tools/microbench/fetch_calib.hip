// fetch_calib.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access widths
// the cosmo_pol kernels use (MI355X_MICROARCH.md: "FETCH_SIZE reports 1/2 of the bytes of a wide
// coalesced streaming read (16 B / lane); other access widths are uncalibrated: calibrate on a known
// byte count in your own access pattern").  Every kernel reads (or writes) exactly N_BYTES once:
//   read4 / read8 / read16   coalesced streaming loads of 4 / 8 / 16 B per lane
//   gather16                 16-B loads at pseudo-random 128-B-aligned places (a table-row gather)
//   gather4                  4-B loads at pseudo-random places (the float32 T-function tables)
//   write4 / write8 / write16 coalesced streaming stores
// build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
// run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./fetch_calib   (and WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define N_BYTES (1ull << 30)

__global__ void k_read4(const float *__restrict__ p, size_t n, float *__restrict__ sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n; i += stride) acc += p[i];
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ void k_read8(const double *__restrict__ p, size_t n, double *__restrict__ sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (; i < n; i += stride) acc += p[i];
    if (acc == 12345.678) sink[0] = acc;
}
__global__ void k_read16(const double2 *__restrict__ p, size_t n, double *__restrict__ sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (; i < n; i += stride) { double2 v = p[i]; acc += v.x + v.y; }
    if (acc == 12345.678) sink[0] = acc;
}
// one 16-B load per lane at a pseudo-random 128-B line of the buffer: n_lines lines, each touched once
__global__ void k_gather16(const double2 *__restrict__ p, size_t n_lines, double *__restrict__ sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (; i < n_lines; i += stride) {
        const size_t line = (i * 2654435761ull) % n_lines;      // (n_lines a power of two, odd multiplier: a permutation)
        double2 v = p[line * 8];                                  // 8 double2 per 128-B line
        acc += v.x + v.y;
    }
    if (acc == 12345.678) sink[0] = acc;
}
__global__ void k_gather4(const float *__restrict__ p, size_t n_lines, float *__restrict__ sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n_lines; i += stride) {
        const size_t line = (i * 2654435761ull) % n_lines;
        acc += p[line * 32];                                      // 32 floats per 128-B line
    }
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ void k_write4(float *__restrict__ p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = (float)i;
}
__global__ void k_write8(double *__restrict__ p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = (double)i;
}
__global__ void k_write16(double2 *__restrict__ p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = make_double2((double)i, 1.0);
}

// scalar loads (k_subbeam_sum<true>): every wavefront reads 96 B of each of its 128-B rows through the
// scalar cache (three s_load_dwordx8), rows at pseudo-random places: n_lines rows, each touched once
typedef unsigned int __attribute__((address_space(4))) sconst_u32;
__global__ void k_sread96(const unsigned int *__restrict__ p, size_t n_lines, unsigned int *__restrict__ sink)
{
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    unsigned int acc = 0;
    for (size_t i = wave; i < n_lines; i += n_waves) {
        const size_t line = (i * 2654435761ull) % n_lines;
        const unsigned long long addr = __builtin_amdgcn_readfirstlane((unsigned int)(line & 0xffffffffu))
            | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned int)(line >> 32)) << 32);
        const sconst_u32 *q = (const sconst_u32 *)((unsigned long long)p + addr * 128ull);
#pragma unroll
        for (int k = 0; k < 24; ++k) acc ^= q[k];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    void *buf = nullptr, *sink = nullptr;
    if (hipMalloc(&buf, N_BYTES) != hipSuccess || hipMalloc(&sink, 256) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(buf, 0, N_BYTES);
    const dim3 grid(256 * 32), block(256);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_read4, grid, block, 0, 0, (const float *)buf, N_BYTES / 4, (float *)sink);
        hipLaunchKernelGGL(k_read8, grid, block, 0, 0, (const double *)buf, N_BYTES / 8, (double *)sink);
        hipLaunchKernelGGL(k_read16, grid, block, 0, 0, (const double2 *)buf, N_BYTES / 16, (double *)sink);
        hipLaunchKernelGGL(k_gather16, grid, block, 0, 0, (const double2 *)buf, N_BYTES / 128, (double *)sink);
        hipLaunchKernelGGL(k_gather4, grid, block, 0, 0, (const float *)buf, N_BYTES / 128, (float *)sink);
        hipLaunchKernelGGL(k_sread96, grid, block, 0, 0, (const unsigned int *)buf, N_BYTES / 128, (unsigned int *)sink);
        hipLaunchKernelGGL(k_write4, grid, block, 0, 0, (float *)buf, N_BYTES / 4);
        hipLaunchKernelGGL(k_write8, grid, block, 0, 0, (double *)buf, N_BYTES / 8);
        hipLaunchKernelGGL(k_write16, grid, block, 0, 0, (double2 *)buf, N_BYTES / 16);
    }
    hipDeviceSynchronize();
    printf("fetch_calib: every streaming kernel moved %llu bytes; the gathers touched %llu lines of 128 B\n",
           (unsigned long long)N_BYTES, (unsigned long long)(N_BYTES / 128));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
