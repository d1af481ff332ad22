// accumulate_common.hpp -- tap arithmetic shared by the accumulate kernels
// (reference test_opencv/DeBayerKernels.cu:288-468).
#pragma once
#include <type_traits>
#include "common.hpp"

struct Levels3 {
    float white[3], black[3];
};

static inline Levels3 make_levels(mfsr_float3 white, mfsr_float3 black)
{
    Levels3 l;
    l.white[0] = white.x;
    l.white[1] = white.y;
    l.white[2] = white.z;
    l.black[0] = black.x;
    l.black[1] = black.y;
    l.black[2] = black.z;
    return l;
}

template <bool FAST>
__device__ __forceinline__ float tap_weight(int px, int py, float kx, float ky, float kz)
{
    // DeBayerKernels.cu:335-338 / :427-430
    float w = (float)(px * px) * kx + (float)(2 * px * py) * kz + (float)(py * py) * ky;
    if (FAST) {
        const float t = w * -0.72134752044448170368f;  // -0.5 * log2(e)
        w = __builtin_amdgcn_exp2f(t);
    } else {
        w = expf(-0.5f * w);
    }
    if (!finitef(w)) w = (px * py == 0) ? 1.0f : 0.0f;
    return w;
}

__device__ __forceinline__ void tap_accumulate(float raw, float w, int color, const float4& cert4, const Levels3& lv,
                                               pix3& pixel, pix3& totalWeight)
{
    // DeBayerKernels.cu:342-370 / :434-462
    if (color == MFSR_GREEN) {
        raw = (raw - lv.black[1]) / lv.white[1];
        float certainty = cert4.y;
        if (!finitef(certainty)) certainty = 0.0f;
        pixel.y += raw * w * certainty;
        totalWeight.y += w * certainty;
    } else if (color == MFSR_RED) {
        raw = (raw - lv.black[0]) / lv.white[0];
        float certainty = cert4.x;
        if (!finitef(certainty)) certainty = 0.0f;
        pixel.x += raw * w * certainty;
        totalWeight.x += w * certainty;
    } else if (color == MFSR_BLUE) {
        raw = (raw - lv.black[2]) / lv.white[2];
        float certainty = cert4.z;
        if (!finitef(certainty)) certainty = 0.0f;
        pixel.z += raw * w * certainty;
        totalWeight.z += w * certainty;
    }
}


// tap_accumulate for the FAST kernels: the white level enters as a reciprocal (one v_rcp per channel and
// pixel instead of an IEEE division per tap); inside the FAST kernels' tolerance, not bit-identical to
// tap_accumulate.  (A branch-free variant -- operands picked with selects, masked adds to all three
// channels -- was measured slower: 36 instead of 28 us for the margin kernel.)
__device__ __forceinline__ void tap_accumulate_fast(float raw, float w, int color, const float4& cert4, const Levels3& lv,
                                                    const float (&invWhite)[3], pix3& pixel, pix3& totalWeight)
{
    if (color == MFSR_GREEN) {
        raw = (raw - lv.black[1]) * invWhite[1];
        float certainty = cert4.y;
        if (!finitef(certainty)) certainty = 0.0f;
        pixel.y += raw * w * certainty;
        totalWeight.y += w * certainty;
    } else if (color == MFSR_RED) {
        raw = (raw - lv.black[0]) * invWhite[0];
        float certainty = cert4.x;
        if (!finitef(certainty)) certainty = 0.0f;
        pixel.x += raw * w * certainty;
        totalWeight.x += w * certainty;
    } else if (color == MFSR_BLUE) {
        raw = (raw - lv.black[2]) * invWhite[2];
        float certainty = cert4.z;
        if (!finitef(certainty)) certainty = 0.0f;
        pixel.z += raw * w * certainty;
        totalWeight.z += w * certainty;
    }
}

// Branch-free tap for the kernels that run the straight arithmetic on whole wavefronts (frame margin): the three colour
// branches of tap_accumulate_fast all execute when the lanes of a wave sit on different colours (they always do on a
// Bayer row).  Here the tap adds (raw / wm) * w * c and w * c to per-COLOUR sums with lane masks (wm = the largest white
// level: a colour-independent scale that keeps the terms <= w * c -- kernel parameters that are not positive
// semi-definite give weights up to 1e38, and sums of un-scaled raw * w * c would overflow where the reference's do not);
// black and white level enter once per pixel: sum((raw - b) / wl * w c) = sum(raw / wm * w c) * (wm / wl) - (b / wl) * sum(w c).
__device__ __forceinline__ void tap_accumulate_sums(float raw, float w, int color, const float4& cert4, float invWm, float (&S)[3],
                                                    float (&W)[3])
{
    float certainty = color == MFSR_GREEN ? cert4.y : (color == MFSR_RED ? cert4.x : cert4.z);
    if (!finitef(certainty)) certainty = 0.0f;
    const float t = w * certainty;
    const float v = (raw * invWm) * t;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const uint32_t m = 0u - (uint32_t)(color == c);  // MFSR_RED = 0, MFSR_GREEN = 1, MFSR_BLUE = 2
        S[c] += __uint_as_float(m & __float_as_uint(v));
        W[c] += __uint_as_float(m & __float_as_uint(t));
    }
}

// GEOM_CROP: the reference geometry (x2, output grid dimX x dimY over the central
//            half of the frame).
// GEOM_FULL: scale s, output grid (s*dimX) x (s*dimY) over the whole frame.
enum { GEOM_CROP = 0, GEOM_FULL = 1 };

__device__ __forceinline__ int floordiv_pos(int a, int s)
{
    // floor(a / s) for s > 0 and any a
    int q = a / s;
    return (a % s < 0) ? q - 1 : q;
}

// One output pixel of accumulateImagesSuperRes (:379-468) / its full-frame
// generalisation, straight from the reference: read-modify-write of imgOut and
// totalWeights at (x, y).  Caller guarantees 1 <= x < outW-1, 1 <= y < outH-1.
// core: accumulates the 25 taps of output pixel (x, y) into the register values pixel / totalWeight
template <int GEOM, bool FAST, bool SUMS = false>
__device__ __forceinline__ void accumulate_pixel_core(int x, int y, const uint16_t* __restrict__ dataIn,
                                                      const float4* __restrict__ certaintyMask,
                                                      const mfsr_tex2d& kernelParam, const mfsr_tex2d& shifts,
                                                      const Levels3& lv, int dimX, int dimY, int scale, int strideMask,
                                                      int cfa, pix3& pixel, pix3& totalWeight)
{
    const int outW = (GEOM == GEOM_CROP) ? dimX : dimX * scale;
    const int outH = (GEOM == GEOM_CROP) ? dimY : dimY * scale;

    float posX, posY, fscale;
    if (GEOM == GEOM_CROP) {
        posX = ((float)x + 0.5f + (float)(dimX / 2)) / 2.0f / (float)dimX;  // :398
        posY = ((float)y + 0.5f + (float)(dimY / 2)) / 2.0f / (float)dimY;
        fscale = 2.0f;
    } else {
        posX = ((float)x + 0.5f) / (float)outW;
        posY = ((float)y + 0.5f) / (float)outH;
        fscale = (float)scale;
    }
    const float4 kernel = tex4<ADDR_CLAMP>(kernelParam, posX, posY);
    const float2 shift = tex2<ADDR_CLAMP>(shifts, posX, posY);
    const int sx = round2i(shift.x * fscale);  // :403-406
    const int sy = round2i(shift.y * fscale);

    // FAST: the 13 distinct weights (w(px,py) == w(-px,-py), and so is the non-finite rule) and the
    // reciprocal white levels, once per pixel
    float wSym[13];
    float invWhite[3] = {0.0f, 0.0f, 0.0f};
    if (FAST) {
#pragma unroll
        for (int n = 0; n < 13; n++) wSym[n] = tap_weight<true>(n % 5 - 2, n / 5 - 2, kernel.x, kernel.y, kernel.z);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) invWhite[ch] = __builtin_amdgcn_rcpf(lv.white[ch]);
    }
    float sumS[3] = {0.0f, 0.0f, 0.0f}, sumW[3] = {0.0f, 0.0f, 0.0f};  // SUMS: per-colour sums of this call
    const float wm = fmaxf(fmaxf(lv.white[0], lv.white[1]), lv.white[2]);
    const float invWm = __builtin_amdgcn_rcpf(wm);
    int ppsxA[5], ppxA[5];
#pragma unroll
    for (int px = -2; px <= 2; px++) {
        if (GEOM == GEOM_CROP) {
            ppsxA[px + 2] = clampi((x + px + sx + dimX / 2) / 2, dimX / 4, dimX / 2 - 1 + dimX / 4);  // :419
            ppxA[px + 2] = clampi((x + px + dimX / 2) / 2, dimX / 4, dimX / 2 - 1 + dimX / 4);        // :422
        } else {
            ppsxA[px + 2] = clampi(floordiv_pos(x + px + sx, scale), 0, dimX - 1);
            ppxA[px + 2] = clampi(floordiv_pos(x + px, scale), 0, dimX - 1);
        }
    }
    auto taps = [&](auto useSums) {
#pragma unroll
        for (int py = -2; py <= 2; py++) {
            int ppsy, ppy;
            if (GEOM == GEOM_CROP) {
                ppsy = clampi((y + py + sy + dimY / 2) / 2, dimY / 4, dimY / 2 - 1 + dimY / 4);
                ppy = clampi((y + py + dimY / 2) / 2, dimY / 4, dimY / 2 - 1 + dimY / 4);
            } else {
                ppsy = clampi(floordiv_pos(y + py + sy, scale), 0, dimY - 1);
                ppy = clampi(floordiv_pos(y + py, scale), 0, dimY - 1);
            }
            const uint16_t* rawRow = dataIn + (size_t)ppsy * dimX;
            const float4* maskRow = row_ptr(certaintyMask, strideMask, ppy / 2);
#pragma unroll
            for (int px = -2; px <= 2; px++) {
                const int ppsx = ppsxA[px + 2];
                const int color = cfa_at(cfa, ppsy, ppsx);
                const float raw = (float)rawRow[ppsx];
                const float4 cert4 = maskRow[ppxA[px + 2] / 2];
                if constexpr (decltype(useSums)::value) {
                    const int n = (py + 2) * 5 + (px + 2);
                    tap_accumulate_sums(raw, wSym[n <= 12 ? n : 24 - n], color, cert4, invWm, sumS, sumW);
                } else if (FAST) {
                    const int n = (py + 2) * 5 + (px + 2);
                    tap_accumulate_fast(raw, wSym[n <= 12 ? n : 24 - n], color, cert4, lv, invWhite, pixel, totalWeight);
                } else {
                    const float w = tap_weight<false>(px, py, kernel.x, kernel.y, kernel.z);
                    tap_accumulate(raw, w, color, cert4, lv, pixel, totalWeight);
                }
            }
        }
    };
    // per-colour sums with deferred levels only where every weight is in [0, 1] (positive semi-definite kernel parameters):
    // other pixels (weights up to 1e38 -- hostile input, rare) keep the per-tap normalisation, whose overflow behaviour is
    // the reference's
    const bool sums = FAST && SUMS && kernel.x >= 0.0f && kernel.y >= 0.0f && kernel.z * kernel.z <= kernel.x * kernel.y &&
                      kernel.x < 1e30f && kernel.y < 1e30f;
    if (sums)
        taps(std::true_type{});
    else
        taps(std::false_type{});
    if (sums) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float v = __builtin_fmaf(sumS[ch], wm * invWhite[ch], -(lv.black[ch] * invWhite[ch]) * sumW[ch]);
            if (ch == 0) pixel.x += v;
            if (ch == 1) pixel.y += v;
            if (ch == 2) pixel.z += v;
        }
        totalWeight.x += sumW[0];
        totalWeight.y += sumW[1];
        totalWeight.z += sumW[2];
    }
    (void)outW;
    (void)outH;
}

template <int GEOM, bool FAST, bool SUMS = false>
__device__ __forceinline__ void accumulate_pixel_generic(int x, int y, const uint16_t* __restrict__ dataIn,
                                                         pix3* __restrict__ imgOut, pix3* __restrict__ totalWeights,
                                                         const float4* __restrict__ certaintyMask,
                                                         const mfsr_tex2d& kernelParam, const mfsr_tex2d& shifts,
                                                         const Levels3& lv, int dimX, int dimY, int scale, int strideOut,
                                                         int strideMask, int cfa)
{
    pix3 pixel = row_ptr(imgOut, strideOut, y)[x];
    pix3 totalWeight = row_ptr(totalWeights, strideOut, y)[x];
    accumulate_pixel_core<GEOM, FAST, SUMS>(x, y, dataIn, certaintyMask, kernelParam, shifts, lv, dimX, dimY, scale, strideMask,
                                      cfa, pixel, totalWeight);
    row_ptr(imgOut, strideOut, y)[x] = pixel;
    row_ptr(totalWeights, strideOut, y)[x] = totalWeight;
}
