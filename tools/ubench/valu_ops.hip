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
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 q0 = {r0, r1}, q1 = {r1, r2}, q2 = {r2, r3}, q3 = {r3, r0}, qa = {a, a}, qb = {b, b};
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));) }
        if (KIND == 1) { REP16(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));) }
        if (KIND == 2) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %4, %5\n v_cndmask_b32_e64 %1, %1, %4, %5\n v_cndmask_b32_e64 %2, %2, %4, %5\n v_cndmask_b32_e64 %3, %3, %4, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b), "s"(m));) }
        if (KIND == 3) { REP16(asm volatile("v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b) : "vcc");) }
        if (KIND == 4) { REP16(asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %4\n v_cmp_lt_f32_e64 s[22:23], %1, %4\n v_cmp_lt_f32_e64 s[24:25], %2, %4\n v_cmp_lt_f32_e64 s[26:27], %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (KIND == 5) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));) }
        if (KIND == 6) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (KIND == 8) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(qa), "v"(qb));) }
        if (KIND == 9) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(qa));) }
        if (KIND == 10) { REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(qa));) }
        if (KIND == 11) { REP16(asm volatile("v_sub_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_fma_f32 %2, %2, %4, %4\n v_add_f32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));) }
        if (KIND == 12) { REP16(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 13) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 14) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 1, %4\n v_lshl_or_b32 %1, %1, 1, %4\n v_lshl_or_b32 %2, %2, 1, %4\n v_lshl_or_b32 %3, %3, 1, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 15) { REP16(asm volatile("v_bitop3_b32 %0, %0, %4, %5 bitop3:0xca\n v_bitop3_b32 %1, %1, %4, %5 bitop3:0xca\n v_bitop3_b32 %2, %2, %4, %5 bitop3:0xca\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0xca" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 16) { REP16(asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 17) { REP16(asm volatile("v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte0 %1, %1\n v_cvt_f32_ubyte0 %2, %2\n v_cvt_f32_ubyte0 %3, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 18) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 19) { REP16(asm volatile("v_trunc_f32 %0, %0\n v_trunc_f32 %1, %1\n v_trunc_f32 %2, %2\n v_trunc_f32 %3, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 20) { REP16(asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 21) { REP16(asm volatile("v_bfe_u32 %0, %0, 1, 8\n v_bfe_u32 %1, %1, 1, 8\n v_bfe_u32 %2, %2, 1, 8\n v_bfe_u32 %3, %3, 1, 8" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 22) { REP16(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 23) { REP16(asm volatile("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 24) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 25) { REP16(asm volatile("v_mul_f32 %0, s4, %0\n v_mul_f32 %1, s4, %1\n v_mul_f32 %2, s4, %2\n v_mul_f32 %3, s4, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 26) { REP16(asm volatile("v_mul_f32 %0, 0x3f800347, %0\n v_mul_f32 %1, 0x3f800347, %1\n v_mul_f32 %2, 0x3f800347, %2\n v_mul_f32 %3, 0x3f800347, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 27) { REP16(asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 28) { REP16(asm volatile("v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 29) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 30) { REP16(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 31) { REP16(asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "s4");) }
        if (KIND == 40) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));) }
        if (KIND == 41) { REP16(asm volatile("v_mad_i64_i32 %0, s[20:21], %4, %4, %0\n v_mad_i64_i32 %1, s[20:21], %4, %4, %1\n v_mad_i64_i32 %2, s[20:21], %4, %4, %2\n v_mad_i64_i32 %3, s[20:21], %4, %4, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(a) : "s20", "s21");) }
        if (KIND == 42) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 1, %4\n v_lshl_add_u64 %1, %1, 1, %4\n v_lshl_add_u64 %2, %2, 1, %4\n v_lshl_add_u64 %3, %3, 1, %4" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(qa));) }
        if (KIND == 43) { REP16(asm volatile("v_div_scale_f32 %0, vcc, %0, %4, %0\n v_div_scale_f32 %1, vcc, %1, %4, %1\n v_div_scale_f32 %2, vcc, %2, %4, %2\n v_div_scale_f32 %3, vcc, %3, %4, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a) : "vcc");) }
        if (KIND == 44) { REP16(asm volatile("v_div_fixup_f32 %0, %0, %4, %5\n v_div_fixup_f32 %1, %1, %4, %5\n v_div_fixup_f32 %2, %2, %4, %5\n v_div_fixup_f32 %3, %3, %4, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));) }
        if (KIND == 45) { REP16(asm volatile("v_div_fmas_f32 %0, %0, %4, %5\n v_div_fmas_f32 %1, %1, %4, %5\n v_div_fmas_f32 %2, %2, %4, %5\n v_div_fmas_f32 %3, %3, %4, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "vcc");) }
        if (KIND == 46) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));) }
        if (KIND == 47) { REP16(asm volatile("v_min_i32 %0, %0, %4\n v_min_i32 %1, %1, %4\n v_min_i32 %2, %2, %4\n v_min_i32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));) }
        if (KIND == 48) { REP16(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (KIND == 7) { REP16(asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + q0.x + q0.y + q1.x + q1.y + q2.x + q2.y + q3.x + q3.y;
}

template <int KIND>
void run(const char* name, float* d)
{
    const int iters = 4096, blocks = 256 * 4;  // 4 blocks of 4 waves per CU = 4 waves per SIMD
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
        run<12>("v_and_b32", d);
        run<13>("v_add_u32", d);
        run<14>("v_lshl_or_b32", d);
        run<15>("v_bitop3_b32", d);
        run<16>("v_lshlrev_b32", d);
        run<17>("v_cvt_f32_ubyte0", d);
        run<18>("v_lshl_add_u32", d);
        run<19>("v_trunc_f32", d);
        run<20>("v_cvt_i32_f32", d);
        run<21>("v_bfe_u32", d);
        run<22>("v_xor_b32", d);
        run<23>("v_add3_u32", d);
        run<24>("v_mad_u32_u24", d);
        run<25>("v_mul_f32 sgpr", d);
        run<26>("v_mul_f32 literal", d);
        run<27>("v_max_f32", d);
        run<28>("v_floor_f32", d);
        run<29>("v_rcp_f32", d);
        run<30>("v_mov_b32", d);
        run<31>("v_cvt_f32_u32_sdwa", d);
        run<40>("v_mul_lo_u32", d); run<41>("v_mad_i64_i32", d); run<42>("v_lshl_add_u64", d); run<43>("v_div_scale_f32", d);
        run<44>("v_div_fixup_f32", d); run<45>("v_div_fmas_f32", d); run<46>("v_mul_u32_u24", d); run<47>("v_min_i32", d); run<48>("v_sqrt_f32", d);
        run<8>("v_pk_fma_f32", d); run<9>("v_pk_mul_f32", d); run<10>("v_pk_add_f32", d); run<11>("mixed sub/mul/fma/add", d);
    }
    return 0;
}
