// accumulate.hip -- the warp+fuse stage: the
// kernel the headline metric is quoted on.
// Behavioural spec: reference test_opencv/DeBayerKernels.cu:288-468.
//
// Structure (reference structure kept: one launch per frame, accumulators are
// read-modify-written in HBM, 48 B per HR pixel per frame):
//   * one thread per HR pixel, 64 x 4 workgroups, rows of 64 consecutive pixels
//     per wavefront so the four 12-byte accumulator streams (2 loads + 2 stores)
//     are contiguous 768-byte wave accesses;
//   * the per-pixel fields (kernel parameters, flow) are fetched with the exact
//     bilinear arithmetic of the oracle so that roundf(s*flow) -- which decides
//     the tap geometry -- is bit-identical;
//   * the 5x5 taps touch only a 3x3 raw neighbourhood and <= 2x2 certainty
//     texels: raw u16 / mask reads hit L1/L2 (the frame is 16.6 MB at 4K);
//   * weight exponent: px*px, 2*px*py, py*py are exact small integers, so the
//     quadratic form costs two rounded adds per tap exactly as in the reference;
//   * FAST=true evaluates exp(-w/2) as v_exp_f32(w * (-0.5*log2 e)) (<= 2 ulp of
//     the correctly rounded result, well inside the +-1 LSB output budget);
//     FAST=false calls the ocml expf for tight parity tests.
#include "accumulate_common.hpp"

// ---- G1: accumulateImages (DeBayerKernels.cu:290-376) -------------------------
__global__ void __launch_bounds__(256)
    k_accumulateImages(const uint16_t* __restrict__ dataIn, pix3* __restrict__ imgOut, pix3* __restrict__ totalWeights,
                       const float4* __restrict__ certaintyMask, const pix3* __restrict__ kernelParam,
                       const float2* __restrict__ shifts, Levels3 lv, int dimX, int dimY, int strideOut, int strideMask,
                       int strideShift, int cfa)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x < 1 || y < 1 || x >= dimX - 1 || y >= dimY - 1) return;
    pix3 pixel = row_ptr(imgOut, strideOut, y)[x];
    pix3 totalWeight = row_ptr(totalWeights, strideOut, y)[x];
    const pix3 kernel = row_ptr(kernelParam, strideOut, y)[x];  // strideOut: reference quirk (:308)
    const float2 shift = row_ptr(shifts, strideShift, y)[x];
    const int sx = round2i(shift.x);
    const int sy = round2i(shift.y);
#pragma unroll
    for (int py = -2; py <= 2; py++) {
        const int ppsy = clampi(y + py + sy, 0, dimY - 1);
        const int ppy = clampi(y + py, 0, dimY - 1);
#pragma unroll
        for (int px = -2; px <= 2; px++) {
            const int ppsx = clampi(x + px + sx, 0, dimX - 1);
            const int ppx = clampi(x + px, 0, dimX - 1);
            const int color = cfa_at(cfa, ppsy, ppsx);
            const float w = tap_weight<false>(px, py, kernel.x, kernel.y, kernel.z);
            const float raw = (float)dataIn[(size_t)ppsy * dimX + ppsx];
            const float4 cert4 = row_ptr(certaintyMask, strideMask, ppy / 2)[ppx / 2];
            tap_accumulate(raw, w, color, cert4, lv, pixel, totalWeight);
        }
    }
    row_ptr(imgOut, strideOut, y)[x] = pixel;
    row_ptr(totalWeights, strideOut, y)[x] = totalWeight;
}

extern "C" int mfsr_accumulateImages(const uint16_t* dataIn, mfsr_float3* imgOut, mfsr_float3* totalWeights,
                                     const mfsr_float4* certaintyMask, const mfsr_float3* kernelParam,
                                     const mfsr_float2* shifts, mfsr_float3 whiteLevel, mfsr_float3 blackLevel, int dimX,
                                     int dimY, int strideOut, int strideMask, int strideShift, mfsr_stream_t stream)
{
    MFSR_REQUIRE(dataIn && imgOut && totalWeights && certaintyMask && kernelParam && shifts);
    MFSR_REQUIRE(dimX > 2 && dimY > 2);
    MFSR_REQUIRE((long long)strideOut >= 12LL * dimX && (strideOut & 3) == 0);
    MFSR_REQUIRE((long long)strideShift >= 8LL * dimX && (strideShift & 7) == 0 && ((uintptr_t)shifts & 7) == 0);
    MFSR_REQUIRE((long long)strideMask >= 16LL * ((dimX + 1) / 2) && (strideMask & 15) == 0 &&
                 ((uintptr_t)certaintyMask & 15) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(dimX, 64), mfsr_cdiv(dimY, 4));
    hipLaunchKernelGGL(k_accumulateImages, grid, block, 0, mfsr_s(stream), dataIn, (pix3*)imgOut, (pix3*)totalWeights,
                       (const float4*)certaintyMask, (const pix3*)kernelParam, (const float2*)shifts,
                       make_levels(whiteLevel, blackLevel), dimX, dimY, strideOut, strideMask, strideShift,
                       mfsr_cfa_packed());
    return mfsr_launch_status("accumulateImages");
}

// ---- G2: accumulateImagesSuperRes (DeBayerKernels.cu:379-468) and its
//      full-frame generalisation ------------------------------------------------
// GEOM_CROP: the reference geometry (x2, output grid dimX x dimY over the central
//            half of the frame).
// GEOM_FULL: scale s, output grid (s*dimX) x (s*dimY) over the whole frame.
template <int GEOM, bool FAST>
__global__ void __launch_bounds__(256)
    k_accumulateSuperRes(const uint16_t* __restrict__ dataIn, pix3* __restrict__ imgOut, pix3* __restrict__ totalWeights,
                         const float4* __restrict__ certaintyMask, mfsr_tex2d kernelParam, mfsr_tex2d shifts, Levels3 lv,
                         int dimX, int dimY, int scale, int strideOut, int strideMask, int cfa, int rowBegin, int rowEnd)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y + rowBegin;  // [rowBegin, rowEnd): output row window of the launch
    const int outW = (GEOM == GEOM_CROP) ? dimX : dimX * scale;
    const int outH = (GEOM == GEOM_CROP) ? dimY : dimY * scale;
    if (x < 1 || y < 1 || x >= outW - 1 || y >= outH - 1 || y >= rowEnd) return;
    accumulate_pixel_generic<GEOM, FAST>(x, y, dataIn, imgOut, totalWeights, certaintyMask, kernelParam, shifts, lv, dimX,
                                         dimY, scale, strideOut, strideMask, cfa);
}

// 0: straight kernel with the ocml expf (tight parity tests); 1: straight kernel with
// v_exp_f32; 2 (default): additionally the x2 strip kernel of accumulate_fast.hip where
// it applies.
static int g_accumulate_fast = 2;
extern "C" int mfsr_set_accumulate_fast_exp(int enable)
{
    g_accumulate_fast = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
    return MFSR_OK;
}

int mfsr_try_launch_accumulate2x_strip(int nFrames, const uint16_t* const* dataIn, mfsr_float3* imgOut,
                                       mfsr_float3* totalWeights, const mfsr_float4* const* certaintyMask,
                                       mfsr_tex2d kernelParam, const mfsr_tex2d* shifts, mfsr_float3 whiteLevel,
                                       mfsr_float3 blackLevel, int dimX, int dimY, int strideOut, int strideMask,
                                       int fresh, int rowBegin, int rowEnd, mfsr_stream_t stream);  // accumulate_fast.hip

int mfsr_try_launch_accumulate4x_tile(int nFrames, const uint16_t* const* dataIn, mfsr_float3* imgOut,
                                      mfsr_float3* totalWeights, const mfsr_float4* const* certaintyMask,
                                      mfsr_tex2d kernelParam, const mfsr_tex2d* shifts, mfsr_float3 whiteLevel,
                                      mfsr_float3 blackLevel, int dimX, int dimY, int strideOut, int strideMask,
                                      int fresh, int rowBegin, int rowEnd, mfsr_stream_t stream);  // accumulate_fast.hip

static int check_superres_args(const uint16_t* dataIn, mfsr_float3* imgOut, mfsr_float3* totalWeights,
                               const mfsr_float4* certaintyMask, const mfsr_tex2d& kernelParam, const mfsr_tex2d& shifts,
                               int dimX, int dimY, int outW, int strideOut, int strideMask)
{
    MFSR_REQUIRE(dataIn && imgOut && totalWeights && certaintyMask);
    MFSR_REQUIRE(dimX >= 8 && dimY >= 8);
    MFSR_REQUIRE(mfsr_tex_ok(kernelParam, 16) && ((uintptr_t)kernelParam.ptr & 15) == 0 && (kernelParam.pitch & 15) == 0);
    MFSR_REQUIRE(mfsr_tex_ok(shifts, 8) && ((uintptr_t)shifts.ptr & 7) == 0 && (shifts.pitch & 7) == 0);
    MFSR_REQUIRE((long long)strideOut >= 12LL * outW && (strideOut & 3) == 0);
    MFSR_REQUIRE((long long)strideMask >= 16LL * ((dimX + 1) / 2) && (strideMask & 15) == 0 &&
                 ((uintptr_t)certaintyMask & 15) == 0);
    return MFSR_OK;
}

extern "C" int mfsr_accumulateImagesSuperRes(const uint16_t* dataIn, mfsr_float3* imgOut, mfsr_float3* totalWeights,
                                             const mfsr_float4* certaintyMask, mfsr_tex2d kernelParam, mfsr_tex2d shifts,
                                             mfsr_float3 whiteLevel, mfsr_float3 blackLevel, int dimX, int dimY,
                                             int strideOut, int strideMask, mfsr_stream_t stream)
{
    int rc = check_superres_args(dataIn, imgOut, totalWeights, certaintyMask, kernelParam, shifts, dimX, dimY, dimX,
                                 strideOut, strideMask);
    if (rc) return rc;
    dim3 block(64, 4), grid(mfsr_cdiv(dimX, 64), mfsr_cdiv(dimY, 4));
    const Levels3 lv = make_levels(whiteLevel, blackLevel);
    if (g_accumulate_fast)
        hipLaunchKernelGGL((k_accumulateSuperRes<GEOM_CROP, true>), grid, block, 0, mfsr_s(stream), dataIn, (pix3*)imgOut,
                           (pix3*)totalWeights, (const float4*)certaintyMask, kernelParam, shifts, lv, dimX, dimY, 2,
                           strideOut, strideMask, mfsr_cfa_packed(), 0, dimY);
    else
        hipLaunchKernelGGL((k_accumulateSuperRes<GEOM_CROP, false>), grid, block, 0, mfsr_s(stream), dataIn,
                           (pix3*)imgOut, (pix3*)totalWeights, (const float4*)certaintyMask, kernelParam, shifts, lv, dimX,
                           dimY, 2, strideOut, strideMask, mfsr_cfa_packed(), 0, dimY);
    return mfsr_launch_status("accumulateImagesSuperRes");
}

// one frame, HR row window [rowBegin, rowEnd) (whole frame: 0, scale*dimY)
static int accumulate_full_rows(const uint16_t* dataIn, mfsr_float3* imgOut, mfsr_float3* totalWeights,
                                const mfsr_float4* certaintyMask, mfsr_tex2d kernelParam, mfsr_tex2d shifts,
                                mfsr_float3 whiteLevel, mfsr_float3 blackLevel, int dimX, int dimY, int scale, int strideOut,
                                int strideMask, int rowBegin, int rowEnd, mfsr_stream_t stream)
{
    MFSR_REQUIRE(scale >= 1 && scale <= 8);
    int rc = check_superres_args(dataIn, imgOut, totalWeights, certaintyMask, kernelParam, shifts, dimX, dimY,
                                 dimX * scale, strideOut, strideMask);
    if (rc) return rc;
    if (g_accumulate_fast == 2 && scale == 2 &&
        mfsr_try_launch_accumulate2x_strip(1, &dataIn, imgOut, totalWeights, &certaintyMask, kernelParam, &shifts, whiteLevel,
                                           blackLevel, dimX, dimY, strideOut, strideMask, 0, rowBegin, rowEnd, stream) == 1)
        return mfsr_launch_status("accumulateSuperResFull(strip)");
    if (g_accumulate_fast == 2 && scale == 4 &&
        mfsr_try_launch_accumulate4x_tile(1, &dataIn, imgOut, totalWeights, &certaintyMask, kernelParam, &shifts, whiteLevel,
                                          blackLevel, dimX, dimY, strideOut, strideMask, 0, rowBegin, rowEnd, stream) == 1)
        return mfsr_launch_status("accumulateSuperResFull(x4 tile)");
    dim3 block(64, 4), grid(mfsr_cdiv((long long)dimX * scale, 64), mfsr_cdiv(rowEnd - rowBegin, 4));
    const Levels3 lv = make_levels(whiteLevel, blackLevel);
    if (g_accumulate_fast)
        hipLaunchKernelGGL((k_accumulateSuperRes<GEOM_FULL, true>), grid, block, 0, mfsr_s(stream), dataIn, (pix3*)imgOut,
                           (pix3*)totalWeights, (const float4*)certaintyMask, kernelParam, shifts, lv, dimX, dimY, scale,
                           strideOut, strideMask, mfsr_cfa_packed(), rowBegin, rowEnd);
    else
        hipLaunchKernelGGL((k_accumulateSuperRes<GEOM_FULL, false>), grid, block, 0, mfsr_s(stream), dataIn,
                           (pix3*)imgOut, (pix3*)totalWeights, (const float4*)certaintyMask, kernelParam, shifts, lv, dimX,
                           dimY, scale, strideOut, strideMask, mfsr_cfa_packed(), rowBegin, rowEnd);
    return mfsr_launch_status("accumulateSuperResFull");
}

extern "C" int mfsr_accumulateSuperResFull(const uint16_t* dataIn, mfsr_float3* imgOut, mfsr_float3* totalWeights,
                                           const mfsr_float4* certaintyMask, mfsr_tex2d kernelParam, mfsr_tex2d shifts,
                                           mfsr_float3 whiteLevel, mfsr_float3 blackLevel, int dimX, int dimY, int scale,
                                           int strideOut, int strideMask, mfsr_stream_t stream)
{
    return accumulate_full_rows(dataIn, imgOut, totalWeights, certaintyMask, kernelParam, shifts, whiteLevel, blackLevel, dimX, dimY,
                                scale, strideOut, strideMask, 0, scale * dimY, stream);
}

extern "C" int mfsr_accumulateSuperResFullN(int nFrames, const uint16_t* const* dataIn, mfsr_float3* imgOut,
                                            mfsr_float3* totalWeights, const mfsr_float4* const* certaintyMask,
                                            mfsr_tex2d kernelParam, const mfsr_tex2d* shifts, mfsr_float3 whiteLevel,
                                            mfsr_float3 blackLevel, int dimX, int dimY, int scale, int strideOut,
                                            int strideMask, int accumulatorsUndefined, mfsr_stream_t stream);

// Two frames onto the same accumulators in one call: what two successive
// mfsr_accumulateSuperResFull calls compute (frame 0 then frame 1), with the accumulators read and
// written once when the x2 LDS tile kernel serves the geometry (24 instead of 48 B per HR pixel and
// frame).  The per-pixel sums of the two frames are added to each other before they are added to the
// accumulator, so results equal the two-call sequence to fp32 rounding, not bit for bit.
extern "C" int mfsr_accumulateSuperResFull2(const uint16_t* dataIn0, const uint16_t* dataIn1, mfsr_float3* imgOut,
                                            mfsr_float3* totalWeights, const mfsr_float4* certaintyMask0,
                                            const mfsr_float4* certaintyMask1, mfsr_tex2d kernelParam, mfsr_tex2d shifts0,
                                            mfsr_tex2d shifts1, mfsr_float3 whiteLevel, mfsr_float3 blackLevel, int dimX,
                                            int dimY, int scale, int strideOut, int strideMask, mfsr_stream_t stream)
{
    MFSR_REQUIRE(scale >= 1 && scale <= 8);
    int rc = check_superres_args(dataIn0, imgOut, totalWeights, certaintyMask0, kernelParam, shifts0, dimX, dimY, dimX * scale,
                                 strideOut, strideMask);
    if (rc) return rc;
    rc = check_superres_args(dataIn1, imgOut, totalWeights, certaintyMask1, kernelParam, shifts1, dimX, dimY, dimX * scale,
                             strideOut, strideMask);
    if (rc) return rc;
    const uint16_t* raws[2] = {dataIn0, dataIn1};
    const mfsr_float4* masks[2] = {certaintyMask0, certaintyMask1};
    const mfsr_tex2d sh[2] = {shifts0, shifts1};
    return mfsr_accumulateSuperResFullN(2, raws, imgOut, totalWeights, masks, kernelParam, sh, whiteLevel, blackLevel, dimX, dimY,
                                        scale, strideOut, strideMask, 0, stream);
}

// nFrames (1 .. MFSR_MAX_FUSE_GROUP) frames in one call; accumulatorsUndefined != 0: the planes are overwritten as if
// they had been zeroed before the call (the first launch of a burst: saves the memset and the read of
// both planes -- 0 + x == x, so the result equals the zeroed-and-accumulated one bit for bit).
// Only HR rows [rowBegin, rowEnd) are touched (stripe-sharded bursts): rowBegin a multiple of 16, rowEnd a multiple
// of 16 or the frame's last row + 1; every pixel of the window gets exactly what the whole-frame call gives it.
extern "C" int mfsr_accumulateSuperResFullRows(int nFrames, const uint16_t* const* dataIn, mfsr_float3* imgOut,
                                               mfsr_float3* totalWeights, const mfsr_float4* const* certaintyMask,
                                               mfsr_tex2d kernelParam, const mfsr_tex2d* shifts, mfsr_float3 whiteLevel,
                                               mfsr_float3 blackLevel, int dimX, int dimY, int scale, int strideOut,
                                               int strideMask, int accumulatorsUndefined, int rowBegin, int rowEnd,
                                               mfsr_stream_t stream)
{
    MFSR_REQUIRE(nFrames >= 1 && nFrames <= MFSR_MAX_FUSE_GROUP && dataIn && certaintyMask && shifts);
    MFSR_REQUIRE(scale >= 1 && scale <= 8);
    const int hrH = scale * dimY;
    MFSR_REQUIRE(rowBegin >= 0 && rowBegin < rowEnd && rowEnd <= hrH && (rowBegin % 16) == 0 && ((rowEnd % 16) == 0 || rowEnd == hrH));
    for (int n = 0; n < nFrames; n++) {
        const int rc = check_superres_args(dataIn[n], imgOut, totalWeights, certaintyMask[n], kernelParam, shifts[n], dimX, dimY,
                                           dimX * scale, strideOut, strideMask);
        if (rc) return rc;
    }
    const int fresh = accumulatorsUndefined ? 1 : 0;
    if (g_accumulate_fast == 2 && scale == 2) {
        const int r = mfsr_try_launch_accumulate2x_strip(nFrames, dataIn, imgOut, totalWeights, certaintyMask, kernelParam, shifts,
                                                         whiteLevel, blackLevel, dimX, dimY, strideOut, strideMask, fresh, rowBegin,
                                                         rowEnd, stream);
        if (r == 1) return mfsr_launch_status("accumulateSuperResFullN(strip)");
        if (r < 0) return MFSR_E_INVALID;
    }
    if (g_accumulate_fast == 2 && scale == 4) {
        const int r = mfsr_try_launch_accumulate4x_tile(nFrames, dataIn, imgOut, totalWeights, certaintyMask, kernelParam, shifts,
                                                        whiteLevel, blackLevel, dimX, dimY, strideOut, strideMask, fresh, rowBegin,
                                                        rowEnd, stream);
        if (r == 1) return mfsr_launch_status("accumulateSuperResFullN(x4 tile)");
        if (r < 0) return MFSR_E_INVALID;
    }
    if (nFrames > 2) {
        // no kernel of this geometry takes the whole group: two frames, then the rest
        const int rc = mfsr_accumulateSuperResFullRows(2, dataIn, imgOut, totalWeights, certaintyMask, kernelParam, shifts, whiteLevel,
                                                       blackLevel, dimX, dimY, scale, strideOut, strideMask, accumulatorsUndefined,
                                                       rowBegin, rowEnd, stream);
        if (rc) return rc;
        return mfsr_accumulateSuperResFullRows(nFrames - 2, dataIn + 2, imgOut, totalWeights, certaintyMask + 2, kernelParam,
                                               shifts + 2, whiteLevel, blackLevel, dimX, dimY, scale, strideOut, strideMask, 0,
                                               rowBegin, rowEnd, stream);
    }
    if (fresh) {
        const size_t off = (size_t)rowBegin * strideOut, bytes = (size_t)(rowEnd - rowBegin) * strideOut;
        MFSR_HIP_TRY(hipMemsetAsync((char*)imgOut + off, 0, bytes, mfsr_s(stream)));
        MFSR_HIP_TRY(hipMemsetAsync((char*)totalWeights + off, 0, bytes, mfsr_s(stream)));
    }
    for (int n = 0; n < nFrames; n++) {
        const int rc = accumulate_full_rows(dataIn[n], imgOut, totalWeights, certaintyMask[n], kernelParam, shifts[n], whiteLevel,
                                            blackLevel, dimX, dimY, scale, strideOut, strideMask, rowBegin, rowEnd, stream);
        if (rc) return rc;
    }
    return MFSR_OK;
}

extern "C" int mfsr_accumulateSuperResFullN(int nFrames, const uint16_t* const* dataIn, mfsr_float3* imgOut,
                                            mfsr_float3* totalWeights, const mfsr_float4* const* certaintyMask,
                                            mfsr_tex2d kernelParam, const mfsr_tex2d* shifts, mfsr_float3 whiteLevel,
                                            mfsr_float3 blackLevel, int dimX, int dimY, int scale, int strideOut,
                                            int strideMask, int accumulatorsUndefined, mfsr_stream_t stream)
{
    MFSR_REQUIRE(scale >= 1 && scale <= 8 && dimY > 0);
    return mfsr_accumulateSuperResFullRows(nFrames, dataIn, imgOut, totalWeights, certaintyMask, kernelParam, shifts, whiteLevel,
                                           blackLevel, dimX, dimY, scale, strideOut, strideMask, accumulatorsUndefined, 0,
                                           scale * dimY, stream);
}
