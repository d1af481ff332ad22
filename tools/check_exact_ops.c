/* Exhaustive host checks (all 2^32 float bit patterns) behind two device-side identities:
 *   round2i: trunc(x + copysign(0.49999997f, x)) == roundf(x)                      (common.hpp)
 *   divN   : fma(fma(-d, x*r, x), r, x*r) == x / d with r = RN(1/d), finite x      (robustness.hip: d = 9, 3)
 * gcc -O2 -fopenmp -ffp-contract=off tools/check_exact_ops.c -o /tmp/check_exact_ops -lm && /tmp/check_exact_ops
 * Expected: 0 bad for round2i, d = 9 and d = 3; non-zero for 12 and sqrt(2) (which therefore keep the IEEE division). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static void check_div(float d)
{
    const float r = (float)(1.0 / (double)d);
    uint64_t bad = 0, n = 0;
#pragma omp parallel for reduction(+ : bad, n)
    for (int64_t i = 0; i < (1LL << 32); i++) {
        uint32_t u = (uint32_t)i;
        float x;
        memcpy(&x, &u, 4);
        if (isnan(x) || isinf(x)) continue;
        const float q = x * r;
        const float q2 = fmaf(fmaf(-d, q, x), r, q);
        n++;
        if (!(q2 == x / d)) bad++;
    }
    printf("div by %.9g: checked %llu, bad %llu\n", d, (unsigned long long)n, (unsigned long long)bad);
}

int main(void)
{
    uint64_t bad = 0, n = 0;
#pragma omp parallel for reduction(+ : bad, n)
    for (int64_t i = 0; i < (1LL << 32); i++) {
        uint32_t u = (uint32_t)i;
        float x;
        memcpy(&x, &u, 4);
        if (isnan(x)) continue;
        n++;
        if (!(truncf(x + copysignf(0.49999997f, x)) == roundf(x))) bad++;
    }
    printf("round2i: checked %llu, bad %llu\n", (unsigned long long)n, (unsigned long long)bad);
    check_div(9.0f);
    check_div(3.0f);
    check_div(12.0f);
    check_div(sqrtf(2.0f));
    return 0;
}
