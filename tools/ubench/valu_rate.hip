// Micro-benchmark: VALU issue rate per SIMD vs waves per SIMD on gfx950.
// Each wave runs N independent-ish v_add / v_cndmask / v_mul ops; blocks of 64*W threads, one block per CU-SIMD slot.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void k(float* out, int iters, float a, float b)
{
    float x0 = threadIdx.x * 0.001f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {  // v_fma_f32 chain x8 independent
            x0 = x0 * a + b; x1 = x1 * a + b; x2 = x2 * a + b; x3 = x3 * a + b;
            x4 = x4 * a + b; x5 = x5 * a + b; x6 = x6 * a + b; x7 = x7 * a + b;
        } else if (KIND == 1) {  // v_cndmask with per-lane condition
            bool c = (i + threadIdx.x) & 1;
            x0 = c ? x1 : x0; x1 = c ? x2 : x1; x2 = c ? x3 : x2; x3 = c ? x4 : x3;
            x4 = c ? x5 : x4; x5 = c ? x6 : x5; x6 = c ? x7 : x6; x7 = c ? x0 : x7;
        } else {  // v_exp
            x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1); x2 = __builtin_amdgcn_exp2f(x2); x3 = __builtin_amdgcn_exp2f(x3);
            x4 = __builtin_amdgcn_exp2f(x4); x5 = __builtin_amdgcn_exp2f(x5); x6 = __builtin_amdgcn_exp2f(x6); x7 = __builtin_amdgcn_exp2f(x7);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int KIND>
void run(const char* name)
{
    float* d;
    hipMalloc(&d, 256 * 1024 * 64 * sizeof(float));
    const int iters = 4096;
    for (int wavesPerSimd = 1; wavesPerSimd <= 8; wavesPerSimd *= 2) {
        const int threads = 256;                      // 4 waves per block = 1 per SIMD
        const int blocks = 256 * wavesPerSimd;        // blocks per CU = wavesPerSimd
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.999f, 0.001f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.999f, 0.001f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 8 * wavesPerSimd;   // wave-instructions issued on one SIMD
        printf("%s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, wavesPerSimd, ms,
               ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
    hipFree(d);
}

int main()
{
    run<0>("v_fma    ");
    run<1>("v_cndmask");
    run<2>("v_exp    ");
    return 0;
}
