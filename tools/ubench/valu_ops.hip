// Micro-benchmark: cycles per wave-instruction per SIMD for the VALU ops the fuse kernel is made of
// (4 waves per SIMD, 16 independent chains per wave, fully unrolled inline asm).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters)
{
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3;
    float a = 1.0001f, b = 0.5f;
    unsigned long long m = 0x5555555555555555ull;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));) }
        if (KIND == 1) { REP16(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));) }
        if (KIND == 2) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %4, %5\n v_cndmask_b32_e64 %1, %1, %4, %5\n v_cndmask_b32_e64 %2, %2, %4, %5\n v_cndmask_b32_e64 %3, %3, %4, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b), "s"(m));) }
        if (KIND == 3) { REP16(asm volatile("v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b) : "vcc");) }
        if (KIND == 4) { REP16(asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %4\n v_cmp_lt_f32_e64 s[22:23], %1, %4\n v_cmp_lt_f32_e64 s[24:25], %2, %4\n v_cmp_lt_f32_e64 s[26:27], %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (KIND == 5) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));) }
        if (KIND == 6) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (KIND == 7) { REP16(asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3;
}

template <int KIND>
void run(const char* name, float* d)
{
    const int iters = 256, blocks = 256 * 4;  // 4 blocks of 4 waves per CU = 4 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 64 * 4;  // 64 instr per iteration, 4 waves per SIMD
    printf("%-22s %.3f ms -> %.3f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / instr_per_simd);
}

int main()
{
    float* d; hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    for (int rep = 0; rep < 2; rep++) {
        run<0>("v_add_f32", d); run<1>("v_mul_f32", d); run<5>("v_fma_f32", d); run<2>("v_cndmask_b32_e64 (sgpr)", d);
        run<3>("v_cndmask_b32_e32 (vcc)", d); run<4>("v_cmp_lt_f32_e64", d); run<6>("v_exp_f32", d); run<7>("v_cvt_f32_u32", d);
    }
    return 0;
}
