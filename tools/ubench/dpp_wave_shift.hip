// Does gfx950 execute the GFX9 whole-wave DPP shifts (wave_shr:1 / wave_shl:1) as documented?  The Lucas-Kanade sweep kernel
// (csrc/lk_sweep.hip) takes its horizontal window sums with them (no LDS).  Prints PASS/FAIL and the cost per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ float wshr1(float v) { return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float wshl1(float v) { return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x130, 0xf, 0xf, true)); }
__global__ void k_check(const float* in, float* outR, float* outL)
{
    const float v = in[threadIdx.x];
    float acc = v;
#pragma unroll
    for (int i = 0; i < 6; i++) acc = v + wshr1(acc);   // sum of v[l-6..l]
    outR[threadIdx.x] = acc;
    outL[threadIdx.x] = wshl1(wshl1(v));                // v[l+2]
}
__global__ void k_rate(float* out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, v = 0.001f * threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            a0 = v + wshr1(a0);
            a1 = v + wshr1(a1);
            a2 = v + wshr1(a2);
            a3 = v + wshr1(a3);
            a4 = v + wshr1(a4);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4;
}
int main()
{
    float *in, *r, *l;
    hipMalloc(&in, 256); hipMalloc(&r, 256); hipMalloc(&l, 256);
    std::vector<float> h(64), hr(64), hl(64);
    for (int i = 0; i < 64; i++) h[i] = (float)(i * i % 17) + 0.25f * i;
    hipMemcpy(in, h.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, in, r, l);
    hipMemcpy(hr.data(), r, 256, hipMemcpyDeviceToHost);
    hipMemcpy(hl.data(), l, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) {
        float acc = i >= 6 ? h[i - 6] : 0.0f;   // Horner from the far end, zeros shifted in at lane 0
        for (int d = 5; d >= 0; d--) acc = h[i] * 0 + ((i - d >= 0 ? h[i - d] : 0.0f) + acc);
        // the kernel's association: acc_k(l) = v(l) + acc_{k-1}(l-1)
        float ref;
        {
            float a[64];
            for (int q = 0; q < 64; q++) a[q] = h[q];
            for (int k = 0; k < 6; k++) {
                float n[64];
                for (int q = 0; q < 64; q++) n[q] = h[q] + (q > 0 ? a[q - 1] : 0.0f);
                for (int q = 0; q < 64; q++) a[q] = n[q];
            }
            ref = a[i];
        }
        (void)acc;
        if (hr[i] != ref) bad++;
        const float wantL = i + 2 < 64 ? h[i + 2] : 0.0f;
        if (hl[i] != wantL) bad++;
    }
    printf("wave_shr:1 / wave_shl:1 across all 64 lanes: %s\n", bad ? "FAIL" : "PASS");
    float* out;
    const int blocks = 256 * 4 * 4, iters = 2000;
    hipMalloc(&out, blocks * 64 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(64), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)iters * 40;  // per wave
    printf("v_add_f32_dpp wave_shr:1: %.2f ns per wave-instruction per SIMD with 4 waves resident (%.3f ms)\n", ms * 1e6 / insts / 4.0, ms);
    return bad ? 1 : 0;
}
