// What does `wave_shr:1` do on gfx950?  One step, then the 63-step running sum / product against the sequential loop.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o /tmp/dpp_scan tools/microbench/dpp_scan.hip && /tmp/dpp_scan
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>

__global__ void one_step(float *out)
{
    float s = (float)(threadIdx.x + 1), x = 100.0f;
    asm volatile("s_nop 4\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+&v"(s) : "v"(x));
    out[threadIdx.x] = s;
}

template <bool MUL>
__global__ void scan(const float *in, float *out)
{
    const float x = in[threadIdx.x];
    float s = x;
    if (MUL) asm volatile("s_nop 4\n\t.rept 63\n\ts_nop 1\n\tv_mul_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t.endr" : "+&v"(s) : "v"(x));
    else     asm volatile("s_nop 4\n\t.rept 63\n\ts_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t.endr" : "+&v"(s) : "v"(x));
    out[threadIdx.x] = s;
}

int main()
{
    float *d_in, *d_out, h[64], in[64];
    hipMalloc(&d_in, 256); hipMalloc(&d_out, 256);
    hipLaunchKernelGGL(one_step, dim3(1), dim3(64), 0, 0, d_out);
    hipMemcpy(h, d_out, 256, hipMemcpyDeviceToHost);
    printf("one step of s[l] = l + 1 with x = 100 (expected: 1, 101, 102, 103 ...):\n");
    for (int l = 0; l < 64; ++l) printf("%g ", h[l]);
    printf("\n");
    srand(7);
    for (int mul = 0; mul < 2; ++mul) {
        for (int l = 0; l < 64; ++l) in[l] = mul ? 0.9f + 0.2f * (rand() / (float)RAND_MAX) : (rand() / (float)RAND_MAX - 0.3f);
        hipMemcpy(d_in, in, 256, hipMemcpyHostToDevice);
        if (mul) hipLaunchKernelGGL(scan<true>, dim3(1), dim3(64), 0, 0, d_in, d_out);
        else hipLaunchKernelGGL(scan<false>, dim3(1), dim3(64), 0, 0, d_in, d_out);
        hipMemcpy(h, d_out, 256, hipMemcpyDeviceToHost);
        float c = in[0];
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            if (l) c = mul ? c * in[l] : c + in[l];
            if (memcmp(&c, &h[l], 4)) { if (bad < 5) printf("  lane %d: sequential %.9g, dpp %.9g\n", l, c, h[l]); ++bad; }
        }
        printf("%s: %d of 64 lanes differ from the sequential loop\n", mul ? "product" : "sum", bad);
    }
    return 0;
}
