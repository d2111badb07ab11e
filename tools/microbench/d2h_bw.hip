// Device-to-host bandwidth of one sweep's result window (13.7 MB): how the copy is issued.
//   hipcc --offload-arch=gfx950 -O2 -o d2h_bw d2h_bw.hip && ./d2h_bw
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t bytes = 360ull * 500 * 76, reps = 60;
    const int NS = 3;
    void *dev[NS], *host[NS];
    hipStream_t st[NS];
    for (int i = 0; i < NS; ++i) {
        CK(hipMalloc(&dev[i], bytes));
        CK(hipMemset(dev[i], i + 1, bytes));
        CK(hipHostMalloc(&host[i], bytes, hipHostMallocDefault));
        CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    }
    CK(hipDeviceSynchronize());
    auto report = [&](const char *what, double t, size_t total) { printf("%-62s %7.2f GB/s  (%.3f ms per window)\n", what, total / t / 1e9, 1e3 * t / reps); };
    for (int pass = 0; pass < 2; ++pass) {
        // (a) one copy per window, one stream
        double t0 = now();
        for (size_t r = 0; r < reps; ++r) CK(hipMemcpyAsync(host[0], dev[0], bytes, hipMemcpyDeviceToHost, st[0]));
        CK(hipDeviceSynchronize());
        if (pass) report("one hipMemcpyAsync per window, one stream", now() - t0, bytes * reps);
        // (b) windows round robin over three streams (what the lanes do)
        t0 = now();
        for (size_t r = 0; r < reps; ++r) CK(hipMemcpyAsync(host[r % NS], dev[r % NS], bytes, hipMemcpyDeviceToHost, st[r % NS]));
        CK(hipDeviceSynchronize());
        if (pass) report("one copy per window, three streams round robin", now() - t0, bytes * reps);
        // (c) every window split in two halves on two streams
        t0 = now();
        for (size_t r = 0; r < reps; ++r) {
            size_t h = bytes / 2 & ~(size_t)4095;
            CK(hipMemcpyAsync(host[0], dev[0], h, hipMemcpyDeviceToHost, st[0]));
            CK(hipMemcpyAsync((char *)host[0] + h, (char *)dev[0] + h, bytes - h, hipMemcpyDeviceToHost, st[1]));
        }
        CK(hipDeviceSynchronize());
        if (pass) report("window split in two halves on two streams", now() - t0, bytes * reps);
        // (d) a kernel storing into the page-locked block (zero copy), grids of 64 / 256 / 1024 workgroups
        for (int wgs : {16, 64, 256, 1024}) {
            t0 = now();
            for (size_t r = 0; r < reps; ++r)
                hipLaunchKernelGGL(k_copy16, dim3(wgs), dim3(256), 0, st[0], (const uint4 *)dev[0], (uint4 *)host[0], bytes / 16);
            CK(hipDeviceSynchronize());
            char what[96];
            snprintf(what, sizeof what, "kernel stores into the host block, %d workgroups of 256", wgs);
            if (pass) report(what, now() - t0, bytes * reps);
        }
        // (e) kernel copies on three streams
        t0 = now();
        for (size_t r = 0; r < reps; ++r)
            hipLaunchKernelGGL(k_copy16, dim3(64), dim3(256), 0, st[r % NS], (const uint4 *)dev[r % NS], (uint4 *)host[r % NS], bytes / 16);
        CK(hipDeviceSynchronize());
        if (pass) report("kernel stores, 64 workgroups, three streams", now() - t0, bytes * reps);
        // (f) one big copy (60 windows' worth would not fit the test: 8 windows)
    }
    unsigned char *h0 = (unsigned char *)host[0];
    printf("check: %d %d\n", h0[0], h0[bytes - 1]);
    return 0;
}
