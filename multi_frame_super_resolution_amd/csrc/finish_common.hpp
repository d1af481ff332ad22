// finish_common.hpp -- H1 (ApplyWeighting, kernel.cu:426-481, with the fallback image resampled on the fly) + H2 (GammasRGB,
// :393-422) + quantisation for ONE pixel: shared by k_finishFused (glue.hip) and by the epilogue of the warp+fuse tile
// kernels (accumulate_fast.hip: the burst's last launch normalises the pixels it has just accumulated instead of leaving
// that to a pass of its own) -- one piece of code, so the two give the same bits.
#pragma once
#include "common.hpp"

// bilinear fetch of a float3 image (clamp)
__device__ __forceinline__ pix3 sample_pix3(const pix3* __restrict__ in, int inPitch, int inW, int inH, float u, float v)
{
    const TexCoord c = tex_coord<ADDR_CLAMP>(inW, inH, u, v);
    const pix3* r0 = row_ptr(in, inPitch, c.j0);
    const pix3* r1 = row_ptr(in, inPitch, c.j1);
    const pix3 t00 = r0[c.i0], t10 = r0[c.i1], t01 = r1[c.i0], t11 = r1[c.i1];
    pix3 o;
    o.x = lerp4(t00.x, t10.x, t01.x, t11.x, c.a, c.b);
    o.y = lerp4(t00.y, t10.y, t01.y, t11.y, c.a, c.b);
    o.z = lerp4(t00.z, t10.z, t01.z, t11.z, c.a, c.b);
    return o;
}

__device__ __forceinline__ float apply_weight_f(float inout, float val, float w, float threshold)
{
    // kernel.cu:447-456
    if (w < threshold) {
        val += inout;
        w += 1;
    }
    inout = 0;
    if (w != 0) inout = val / w;
    return inout;
}

__device__ __forceinline__ float gamma_f(float v)
{
    // kernel.cu:380-390, :407-420
    if (isnan(v)) v = 0;
    v = fmaxf(fminf(v, 1.0f), 0.0f);
    if (v <= 0.0031308f) return 12.92f * v;
    return (1.0f + 0.055f) * powf(v, 1.0f / 2.4f) - 0.055f;
}

__device__ __forceinline__ int quantize1(float f, float maxOut)
{
    if (isnan(f)) f = 0;
    f = fmaxf(fminf(f, 1.0f), 0.0f);
    return (int)(f * maxOut + 0.5f);
}

// what a finish needs besides the two accumulators (by value in kernel arguments)
struct FinishArgs {
    const pix3* fallback;   // may be null
    int fbPitch, fbW, fbH;
    float u0, u1, v0, v1;   // window of the fallback image the full output image covers (the pipeline: 0, 1, 0, 1)
    float threshold;
    int applyGamma;
    float maxOut;
    int width, fullHeight;  // of the full output image
    pix3* outImg;           // full-image pointers; either may be null
    int outPitch;
    uint16_t* out16;
};

// H1 of pixel (x, yFull) of the full image from its accumulated value `val` and weight `w` (before the gamma)
__device__ __forceinline__ pix3 finish_weighted(const FinishArgs& f, int x, int yFull, pix3 val, pix3 w)
{
    pix3 inout = {0.0f, 0.0f, 0.0f};
    // ApplyWeighting reads the fallback only where a weight is under the threshold (kernel.cu:444-462): the resample (12
    // loads, two divisions, the bilinear mix) is skipped by the waves that need none
    if (f.fallback && (w.x < f.threshold || w.y < f.threshold || w.z < f.threshold)) {
        const float u = f.u0 + (f.u1 - f.u0) * (((float)x + 0.5f) / (float)f.width);
        const float v = f.v0 + (f.v1 - f.v0) * (((float)yFull + 0.5f) / (float)f.fullHeight);
        inout = sample_pix3(f.fallback, f.fbPitch, f.fbW, f.fbH, u, v);
    }
    inout.x = apply_weight_f(inout.x, val.x, w.x, f.threshold);
    inout.y = apply_weight_f(inout.y, val.y, w.y, f.threshold);
    inout.z = apply_weight_f(inout.z, val.z, w.z, f.threshold);
    return inout;
}

// finished value (H1, then H2 if asked for) of the pixel
__device__ __forceinline__ pix3 finish_value(const FinishArgs& f, int x, int yFull, pix3 val, pix3 w)
{
    pix3 inout = finish_weighted(f, x, yFull, val, w);
    if (f.applyGamma) {
        inout.x = gamma_f(inout.x);
        inout.y = gamma_f(inout.y);
        inout.z = gamma_f(inout.z);
    }
    return inout;
}

// host side (glue.hip): finish of the frame margins of HR rows [rowBegin, rowEnd) -- the top and bottom bands of M rows and
// the M outermost columns on either side of the other rows -- i.e. of every pixel the tile kernels' epilogue leaves out.
// imgOut / totalWeights / fin's output pointers are those of the FULL image.  0 on success.
int mfsr_finish_margins(const pix3* imgOut, const pix3* totalWeights, int pitch, const FinishArgs& fin, int rowBegin, int rowEnd, int M,
                        hipStream_t st);
