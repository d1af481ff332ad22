// lk_math.hpp -- closed-form 2x2 pseudo-inverse of the Lucas-Kanade normal
// matrix, shared by the straight D4 kernel and the fused LK iteration.
// Behavioural spec: reference test_opencv/opticalFlow.cu:236-292.
#pragma once
#include "common.hpp"

// cos and sin of t = atan2(y, x) / 2 without the three libm calls of the reference (:236-238,
// :262-264): t is in [-pi/2, pi/2], so cos t >= 0 and sin t has the sign of y; the half-angle
// identities are taken in their cancellation-free form.  Agrees with cosf/sinf(0.5f * atan2f()) to a
// few ulp; (0, 0) gives t = 0 like atan2f, NaN propagates.
__device__ __forceinline__ void lk_half_angle(float y, float x, float& ct, float& st, float* rOut = nullptr)
{
    // v_sqrt_f32 / v_rcp_f32 / v_rsq_f32 (1 ulp) instead of the IEEE-exact expansions: this routine is
    // a few-ulp replacement of libm calls to begin with
    const float r = __builtin_amdgcn_sqrtf(x * x + y * y);
    if (rOut) *rOut = r;
    if (r == 0.0f) {
        ct = 1.0f;
        st = 0.0f;
        return;
    }
    const float h = 0.5f * __builtin_amdgcn_rcpf(r);  // 1 / (2 r)
    if (x >= 0.0f) {
        ct = __builtin_amdgcn_sqrtf((r + x) * h);
        st = y * h * __builtin_amdgcn_rcpf(ct);
    } else {
        const float sa = __builtin_amdgcn_sqrtf((r - x) * h);  // |sin t|
        st = copysignf(sa, y);
        ct = fabsf(y) * h * __builtin_amdgcn_rcpf(sa);
    }
}

// 2x2 pseudo-inverse by closed-form SVD, shared with the fused kernel.
// Returns false when the pixel is rejected (sigma1 < minDet, :255-257 quirk kept).
__device__ __forceinline__ bool lk_pinv(float a, float b, float d, float minDet, float inv[4])
{
    const float c = b;  // matMul[2] = matMul[1] (:234)
    float ct, st;
    float S2;
    lk_half_angle(2.0f * a * c + 2.0f * b * d, a * a + b * b - c * c - d * d, ct, st, &S2);  // theta = atan2(..)/2 (:236-238)
    const float UT0 = ct, UT2 = -st, UT1 = st, UT3 = ct;
    const float S1 = a * a + b * b + c * c + d * d;
    // S2 (:244-245) = sqrt((a a + b b - c c - d d)^2 + 4 (a c + b d)^2) is the modulus r the half-angle routine has just taken
    // of (x, y) = (a a + b b - c c - d d, 2 a c + 2 b d): y = 2 (a c + b d) exactly (a scaling by two commutes with the
    // roundings), so y y = 4 (a c + b d)^2 bit for bit -- the same number, one square root instead of two.
    // v_sqrt_f32 / v_rcp_f32 (1 ulp) for sqrtf and 1/x: sigma2 comes out of a cancellation (S1 - S2)
    // whose error dwarfs an ulp, so the IEEE-exact expansions buy nothing here
    float sigma1 = __builtin_amdgcn_sqrtf((S1 + S2) * 0.5f);
    float sigma2 = __builtin_amdgcn_sqrtf((S1 - S2) * 0.5f);
    const float smin = fminf(sigma1, sigma1);  // (sic) :255
    if (smin < minDet) return false;
    sigma1 = sigma1 != 0 ? __builtin_amdgcn_rcpf(sigma1) : 0;
    sigma2 = sigma2 != 0 ? __builtin_amdgcn_rcpf(sigma2) : 0;
    const float S0 = sigma1, Sb = 0, Sc = 0, S3 = sigma2;
    float ce, se;
    lk_half_angle(2.0f * a * b + 2.0f * c * d, a * a - b * b + c * c - d * d, ce, se);  // epsilon (:262-264)
    float s11 = (a * ct + c * st) * ce + (b * ct + d * st) * se;
    float s22 = (a * st - c * ct) * se + (-b * st + d * ct) * ce;
    s11 = s11 > 0.0f ? 1.0f : s11 < 0 ? -1.0f : 0.0f;
    s22 = s22 > 0.0f ? 1.0f : s22 < 0 ? -1.0f : 0.0f;
    const float V0 = s11 * ce, V1 = -s22 * se, V2 = s11 * se, V3 = s22 * ce;
    const float m0 = S0 * UT0 + Sb * UT2;
    const float m1 = S0 * UT1 + Sb * UT3;
    const float m2 = Sc * UT0 + S3 * UT2;
    const float m3 = Sc * UT1 + S3 * UT3;
    inv[0] = V0 * m0 + V1 * m2;
    inv[1] = V0 * m1 + V1 * m3;
    inv[2] = V2 * m0 + V3 * m2;
    inv[3] = V2 * m1 + V3 * m3;
    return true;
}

