// lk_math.hpp -- closed-form 2x2 pseudo-inverse of the Lucas-Kanade normal
// matrix, shared by the straight D4 kernel and the fused LK iteration.
// Behavioural spec: reference test_opencv/opticalFlow.cu:236-292.
#pragma once
#include "common.hpp"

// 2x2 pseudo-inverse by closed-form SVD, shared with the fused kernel.
// Returns false when the pixel is rejected (sigma1 < minDet, :255-257 quirk kept).
__device__ __forceinline__ bool lk_pinv(float a, float b, float d, float minDet, float inv[4])
{
    const float c = b;  // matMul[2] = matMul[1] (:234)
    const float theta = 0.5f * atan2f(2.0f * a * c + 2.0f * b * d, a * a + b * b - c * c - d * d);
    const float ct = cosf(theta);
    const float st = sinf(theta);
    const float UT0 = ct, UT2 = -st, UT1 = st, UT3 = ct;
    const float S1 = a * a + b * b + c * c + d * d;
    const float S2 =
        sqrtf((a * a + b * b - c * c - d * d) * (a * a + b * b - c * c - d * d) + 4 * (a * c + b * d) * (a * c + b * d));
    float sigma1 = sqrtf((S1 + S2) / 2);
    float sigma2 = sqrtf((S1 - S2) / 2);
    const float smin = fminf(sigma1, sigma1);  // (sic) :255
    if (smin < minDet) return false;
    sigma1 = sigma1 != 0 ? 1.0f / sigma1 : 0;
    sigma2 = sigma2 != 0 ? 1.0f / sigma2 : 0;
    const float S0 = sigma1, Sb = 0, Sc = 0, S3 = sigma2;
    const float epsilon = 0.5f * atan2f(2.0f * a * b + 2.0f * c * d, a * a - b * b + c * c - d * d);
    const float ce = cosf(epsilon);
    const float se = sinf(epsilon);
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

