// tile_tracker.hip -- tile-based shift tracker, kernel-shape and finishing
// stages.
// Behavioural spec: reference test_opencv/kernel.cu:116-891.
//
// MI355X notes:
//  * squaredSum / findMinimum are "one THREAD per tile, serial loop" in the
//    reference (kernel.cu:126-141, :521-541).  Here one 64-lane WAVEFRONT owns a
//    tile and reduces with DPP/shuffle butterflies; findMinimum's argmin keeps
//    the serial semantics (first strict minimum) by reducing (value, index)
//    pairs with index as tie-break, so its result is bit-identical.
//  * trackTilesFused evaluates the whole B1..B7 chain for one tile inside one
//    workgroup from LDS (no FFT, no tile stacks in HBM).
#include <cfloat>

#include <cstdlib>

#include <cstring>
#include "common.hpp"

// ---- wavefront reductions -----------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// (value, index) minimum; ties -> lower index (== first strict minimum of a serial scan)
__device__ __forceinline__ void wave_argmin(float& v, int& idx)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        const bool take = (ov < v) || (ov == v && oi < idx);
        v = take ? ov : v;
        idx = take ? oi : idx;
    }
}

// ---- B3: squaredSum (kernel.cu:119-143) ---------------------------------------
// one wavefront per tile; lane-strided partial sums + butterfly.  Summation
// order differs from the serial reference loop: equal within fp32 rounding.
__global__ void __launch_bounds__(256) k_squaredSum(const float* __restrict__ inTiles, float* __restrict__ outValues,
                                                   int maxShift, int tileSize, int tileCount)
{
    const int lane = threadIdx.x & 63;
    const int tileIdx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tileIdx >= tileCount) return;
    const int L = tileSize + 2 * maxShift;
    const float* t = inTiles + (size_t)tileIdx * L * L;
    float sum = 0;
    for (int i = lane; i < tileSize * tileSize; i += 64) {
        const int y = i / tileSize, x = i - y * tileSize;
        const float p = t[(y + maxShift) * L + x + maxShift];
        sum += p * p;
    }
    sum = wave_sum(sum);
    if (lane == 0) outValues[tileIdx] = sum;
}

extern "C" int mfsr_squaredSum(const float* inTiles, float* outValues, int maxShift, int tileSize, int tileCount,
                               mfsr_stream_t stream)
{
    MFSR_REQUIRE(inTiles && outValues && maxShift >= 0 && tileSize > 0 && tileCount > 0);
    hipLaunchKernelGGL(k_squaredSum, dim3(mfsr_cdiv(tileCount, 4)), dim3(256), 0, mfsr_s(stream), inTiles, outValues,
                       maxShift, tileSize, tileCount);
    return mfsr_launch_status("squaredSum");
}

// ---- B4: boxFilterWithBorderX / Y (kernel.cu:149-218) --------------------------
// One workgroup per (tile, row|column); the line is staged in LDS once and
// every lane sums its window from LDS in the reference's ascending order, so the
// result is bit-identical.
template <bool ALONG_X>
__global__ void __launch_bounds__(128) k_boxFilterWithBorder(const float* __restrict__ inTiles, float* __restrict__ outTiles,
                                                            int maxShift, int tileSize, int tileCount)
{
    extern __shared__ __attribute__((aligned(16))) float s_line[];
    const int L = tileSize + 2 * maxShift;
    const int line = blockIdx.x;  // row (X filter) or column (Y filter)
    const int tileIdx = blockIdx.y;
    if (tileIdx >= tileCount) return;
    const float* t = inTiles + (size_t)tileIdx * L * L;
    float* o = outTiles + (size_t)tileIdx * L * L;
    for (int i = threadIdx.x; i < L; i += blockDim.x) s_line[i] = ALONG_X ? t[line * L + i] : t[i * L + line];
    __syncthreads();
    for (int p = threadIdx.x; p < L; p += blockDim.x) {
        float outVal = 0;
        if (p >= tileSize / 2 && p <= maxShift * 2 + tileSize / 2) {
            for (int shift = -tileSize / 2; shift < tileSize / 2; shift++) {
                const float v = s_line[p + shift];
                outVal += ALONG_X ? v * v : v;  // X squares its input (:177), Y does not (:214)
            }
        }
        if (ALONG_X)
            o[line * L + p] = outVal;
        else
            o[p * L + line] = outVal;
    }
}

static int box_filter_launch(bool alongX, const float* inTiles, float* outTiles, int maxShift, int tileSize, int tileCount,
                             mfsr_stream_t stream)
{
    MFSR_REQUIRE(inTiles && outTiles && maxShift >= 0 && tileSize > 0 && tileCount > 0 && tileCount <= 65535);
    const int L = tileSize + 2 * maxShift;
    dim3 grid(L, tileCount), block(128);
    if (alongX)
        hipLaunchKernelGGL(k_boxFilterWithBorder<true>, grid, block, L * sizeof(float), mfsr_s(stream), inTiles, outTiles,
                           maxShift, tileSize, tileCount);
    else
        hipLaunchKernelGGL(k_boxFilterWithBorder<false>, grid, block, L * sizeof(float), mfsr_s(stream), inTiles, outTiles,
                           maxShift, tileSize, tileCount);
    return mfsr_launch_status(alongX ? "boxFilterWithBorderX" : "boxFilterWithBorderY");
}

extern "C" int mfsr_boxFilterWithBorderX(const float* inTiles, float* outTiles, int maxShift, int tileSize, int tileCount,
                                         mfsr_stream_t stream)
{
    return box_filter_launch(true, inTiles, outTiles, maxShift, tileSize, tileCount, stream);
}
extern "C" int mfsr_boxFilterWithBorderY(const float* inTiles, float* outTiles, int maxShift, int tileSize, int tileCount,
                                         mfsr_stream_t stream)
{
    return box_filter_launch(false, inTiles, outTiles, maxShift, tileSize, tileCount, stream);
}

// ---- B6: normalizedCC (kernel.cu:227-259) -------------------------------------
__global__ void __launch_bounds__(256)
    k_normalizedCC(const float* __restrict__ ccImage, const float* __restrict__ squaredTemplate,
                   const float* __restrict__ boxFilteredImage, float* __restrict__ shiftImage, int maxShift, int tileSize,
                   int tileCount)
{
    const int R = 2 * maxShift + 1;
    const int L = tileSize + 2 * maxShift;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int tileIdx = blockIdx.y;
    if (i >= R * R || tileIdx >= tileCount) return;
    const int pxY = i / R, pxX = i - pxY * R;  // pxX,pxY <= 2*maxShift: the '>' guard of :240
    const int shiftX = pxX - maxShift, shiftY = pxY - maxShift;
    const int fftShiftX = shiftX < 0 ? L + shiftX : shiftX;
    const int fftShiftY = shiftY < 0 ? L + shiftY : shiftY;
    const size_t base = (size_t)tileIdx * L * L;
    const float cc = ccImage[base + (size_t)fftShiftY * L + fftShiftX];
    const float bf = boxFilteredImage[base + (size_t)(L / 2 + shiftY) * L + (L / 2 + shiftX)];
    shiftImage[(size_t)tileIdx * R * R + i] = squaredTemplate[tileIdx] + bf - 2 * cc;
}

extern "C" int mfsr_normalizedCC(const float* ccImage, const float* squaredTemplate, const float* boxFilteredImage,
                                 float* shiftImage, int maxShift, int tileSize, int tileCount, mfsr_stream_t stream)
{
    MFSR_REQUIRE(ccImage && squaredTemplate && boxFilteredImage && shiftImage);
    MFSR_REQUIRE(maxShift >= 0 && tileSize > 0 && tileCount > 0 && tileCount <= 65535);
    const int R = 2 * maxShift + 1;
    hipLaunchKernelGGL(k_normalizedCC, dim3(mfsr_cdiv(R * R, 256), tileCount), dim3(256), 0, mfsr_s(stream), ccImage,
                       squaredTemplate, boxFilteredImage, shiftImage, maxShift, tileSize, tileCount);
    return mfsr_launch_status("normalizedCC");
}

// ---- B1/B2: convertToTilesOverlapBorder / PreShift (kernel.cu:265-378) ---------
// source pixel of tile-local (pxX,pxY): base shift + rotation about the image
// centre, roundf, float clamp (:299-313 / :358-372)
// the whole-pixel offset every pixel of a tile is gathered at (the same for all its pixels)
__device__ __forceinline__ int2 tile_shift_cs(int imgWidth, int imgHeight, int tileSize, int tileIdxX, int tileIdxY, float2 shift,
                                              float2 baseShift, float cf, float sf)
{
    shift.x += cf * -baseShift.x - sf * -baseShift.y;
    shift.y += sf * -baseShift.x + cf * -baseShift.y;
    const float patchCenterX = (float)(tileIdxX * tileSize + tileSize / 2 - imgWidth / 2);
    const float patchCenterY = (float)(tileIdxY * tileSize + tileSize / 2 - imgHeight / 2);
    shift.x += cf * patchCenterX - sf * patchCenterY - patchCenterX;
    shift.y += sf * patchCenterX + cf * patchCenterY - patchCenterY;
    return make_int2(round2i(shift.x), round2i(shift.y));
}

// pixel (pxX, pxY) of the tile whose origin + whole-pixel offset is (originX, originY), clamped to the image
__device__ __forceinline__ float tile_fetch_at(const float* __restrict__ inImg, int imgWidth, int imgHeight, int imgPitch, int originX,
                                               int originY, int pxX, int pxY)
{
    int pxInImgX = originX + pxX;
    int pxInImgY = originY + pxY;
    pxInImgX = f2i(fminf(fmaxf((float)pxInImgX, 0.0f), (float)(imgWidth - 1)));
    pxInImgY = f2i(fminf(fmaxf((float)pxInImgY, 0.0f), (float)(imgHeight - 1)));
    return row_ptr(inImg, imgPitch, pxInImgY)[pxInImgX];
}

__device__ __forceinline__ float tile_fetch_cs(const float* __restrict__ inImg, int imgWidth, int imgHeight, int imgPitch,
                                               int tileSize, int tileIdxX, int tileIdxY, int pxX, int pxY, float2 shift,
                                               float2 baseShift, float cf, float sf)
{
    const int2 o = tile_shift_cs(imgWidth, imgHeight, tileSize, tileIdxX, tileIdxY, shift, baseShift, cf, sf);
    return tile_fetch_at(inImg, imgWidth, imgHeight, imgPitch, tileIdxX * tileSize + o.x, tileIdxY * tileSize + o.y, pxX, pxY);
}

__device__ __forceinline__ float tile_fetch(const float* __restrict__ inImg, int imgWidth, int imgHeight, int imgPitch,
                                            int tileSize, int tileIdxX, int tileIdxY, int pxX, int pxY, float2 shift,
                                            float2 baseShift, float baseRotation)
{
    return tile_fetch_cs(inImg, imgWidth, imgHeight, imgPitch, tileSize, tileIdxX, tileIdxY, pxX, pxY, shift, baseShift,
                         cosf(baseRotation), sinf(baseRotation));
}

template <bool PRESHIFT>
__global__ void __launch_bounds__(256)
    k_convertToTiles(const float* __restrict__ inImg, float* __restrict__ outTiles, const float2* __restrict__ preShift,
                     int preShiftPitch, int imgWidth, int imgHeight, int imgPitch, int maxShift, int tileSize,
                     int tileCountX, int tileCountY, float2 baseShift, float baseRotation)
{
    const int L = tileSize + 2 * maxShift;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int tileIdx = blockIdx.y;
    if (i >= L * L || tileIdx >= tileCountX * tileCountY) return;
    const int pxY = i / L, pxX = i - pxY * L;
    const size_t o = (size_t)tileIdx * L * L + i;
    if (!PRESHIFT && (pxX < maxShift || pxY < maxShift || pxX >= tileSize + maxShift || pxY >= tileSize + maxShift)) {
        outTiles[o] = 0;  // :286-290
        return;
    }
    const int tileIdxY = tileIdx / tileCountX;
    const int tileIdxX = tileIdx - tileIdxY * tileCountX;
    float2 shift = make_float2(0.0f, 0.0f);
    if (PRESHIFT) shift = row_ptr(preShift, preShiftPitch, tileIdxY)[tileIdxX];
    outTiles[o] = tile_fetch(inImg, imgWidth, imgHeight, imgPitch, tileSize, tileIdxX, tileIdxY, pxX, pxY, shift,
                             baseShift, baseRotation);
}

extern "C" int mfsr_convertToTilesOverlapBorder(const float* inImg, float* outTiles, int imgWidth, int imgHeight,
                                                int imgPitch, int maxShift, int tileSize, int tileCountX, int tileCountY,
                                                mfsr_float2 baseShift, float baseRotation, mfsr_stream_t stream)
{
    MFSR_REQUIRE(inImg && outTiles && imgWidth > 0 && imgHeight > 0 && (long long)imgPitch >= 4LL * imgWidth);
    MFSR_REQUIRE((imgPitch & 3) == 0 && maxShift >= 0 && tileSize > 0 && tileCountX > 0 && tileCountY > 0);
    MFSR_REQUIRE((long long)tileCountX * tileCountY <= 65535);
    const int L = tileSize + 2 * maxShift;
    hipLaunchKernelGGL(k_convertToTiles<false>, dim3(mfsr_cdiv(L * L, 256), tileCountX * tileCountY), dim3(256), 0,
                       mfsr_s(stream), inImg, outTiles, (const float2*)nullptr, 0, imgWidth, imgHeight, imgPitch, maxShift,
                       tileSize, tileCountX, tileCountY, make_float2(baseShift.x, baseShift.y), baseRotation);
    return mfsr_launch_status("convertToTilesOverlapBorder");
}

extern "C" int mfsr_convertToTilesOverlapPreShift(const float* inImg, float* outTiles, const mfsr_float2* preShift,
                                                  int preShiftPitch, int imgWidth, int imgHeight, int imgPitch,
                                                  int maxShift, int tileSize, int tileCountX, int tileCountY,
                                                  mfsr_float2 baseShift, float baseRotation, mfsr_stream_t stream)
{
    MFSR_REQUIRE(inImg && outTiles && preShift && imgWidth > 0 && imgHeight > 0 && (long long)imgPitch >= 4LL * imgWidth);
    MFSR_REQUIRE((imgPitch & 3) == 0 && maxShift >= 0 && tileSize > 0 && tileCountX > 0 && tileCountY > 0);
    MFSR_REQUIRE((long long)preShiftPitch >= 8LL * tileCountX && (preShiftPitch & 7) == 0 && ((uintptr_t)preShift & 7) == 0);
    MFSR_REQUIRE((long long)tileCountX * tileCountY <= 65535);
    const int L = tileSize + 2 * maxShift;
    hipLaunchKernelGGL(k_convertToTiles<true>, dim3(mfsr_cdiv(L * L, 256), tileCountX * tileCountY), dim3(256), 0,
                       mfsr_s(stream), inImg, outTiles, (const float2*)preShift, preShiftPitch, imgWidth, imgHeight,
                       imgPitch, maxShift, tileSize, tileCountX, tileCountY, make_float2(baseShift.x, baseShift.y),
                       baseRotation);
    return mfsr_launch_status("convertToTilesOverlapPreShift");
}

// ---- B5: conjugateComplexMulKernel (kernel.cu:485-501) -------------------------
__global__ void __launch_bounds__(256) k_conjugateComplexMul(const float2* __restrict__ aIn, float2* __restrict__ bInOut,
                                                            int maxElem)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= maxElem) return;
    float2 valA = aIn[idx];
    valA.y = -valA.y;
    const float2 valB = bInOut[idx];
    float2 res;
    res.x = valA.x * valB.x - valA.y * valB.y;
    res.y = valA.x * valB.y + valA.y * valB.x;
    bInOut[idx] = res;
}

extern "C" int mfsr_conjugateComplexMulKernel(const mfsr_float2* aIn, mfsr_float2* bInOut, int maxElem,
                                              mfsr_stream_t stream)
{
    MFSR_REQUIRE(aIn && bInOut && maxElem > 0 && ((uintptr_t)aIn & 7) == 0 && ((uintptr_t)bInOut & 7) == 0);
    hipLaunchKernelGGL(k_conjugateComplexMul, dim3(mfsr_cdiv(maxElem, 256)), dim3(256), 0, mfsr_s(stream),
                       (const float2*)aIn, (float2*)bInOut, maxElem);
    return mfsr_launch_status("conjugateComplexMulKernel");
}

// ---- B7: findMinimum (kernel.cu:503-636) ---------------------------------------
__constant__ float c_FA11[9] = {1.0f / 4.0f, -2.0f / 4.0f, 1.0f / 4.0f, 2.0f / 4.0f, -4.0f / 4.0f,
                                2.0f / 4.0f, 1.0f / 4.0f,  -2.0f / 4.0f, 1.0f / 4.0f};
__constant__ float c_FA22[9] = {1.0f / 4.0f,  2.0f / 4.0f, 1.0f / 4.0f, -2.0f / 4.0f, -4.0f / 4.0f,
                                -2.0f / 4.0f, 1.0f / 4.0f, 2.0f / 4.0f, 1.0f / 4.0f};
__constant__ float c_FA12[9] = {1.0f / 4.0f, 0.0f / 4.0f,  -1.0f / 4.0f, 0.0f / 4.0f, 0.0f / 4.0f,
                                0.0f / 4.0f, -1.0f / 4.0f, 0.0f / 4.0f,  1.0f / 4.0f};
__constant__ float c_Fb1[9] = {-1.0f / 8.0f, 0.0f / 8.0f,  1.0f / 8.0f, -2.0f / 8.0f, 0.0f / 8.0f,
                               2.0f / 8.0f,  -1.0f / 8.0f, 0.0f / 8.0f, 1.0f / 8.0f};
__constant__ float c_Fb2[9] = {-1.0f / 8.0f, -2.0f / 8.0f, -1.0f / 8.0f, 0.0f / 8.0f, 0.0f / 8.0f,
                               0.0f / 8.0f,  1.0f / 8.0f,  2.0f / 8.0f,  1.0f / 8.0f};

// The wavefront-wide part: min / argmin / max over a (2S+1)^2 distance image that
// `img` points at (global or LDS).  Every lane returns the same values.
template <typename Ptr>
__device__ __forceinline__ void wave_min_argmin_max(Ptr img, int count, int lane, float& minVal, int& minIdx,
                                                    float& maxVal)
{
    minVal = FLT_MAX;
    maxVal = -FLT_MAX;
    minIdx = 0x7fffffff;
    for (int i = lane; i < count; i += 64) {
        const float val = img[i];
        maxVal = fmaxf(maxVal, val);
        if (val < minVal) {
            minVal = val;
            minIdx = i;
        }
    }
    wave_argmin(minVal, minIdx);
    maxVal = wave_max(maxVal);
    if (minIdx == 0x7fffffff) minIdx = -1;  // nothing below FLT_MAX (serial loop leaves -1, :530)
}

// The scalar part of findMinimum (:543-633): sub-pixel quadratic fit.
template <typename Ptr>
__device__ __forceinline__ float2 subpixel_minimum(Ptr img, int maxShift, float minVal, int minIdx, float maxVal,
                                                   float threshold)
{
    const int R = 2 * maxShift + 1;
    float2 coord;
    coord.y = (float)(minIdx / R);
    coord.x = (float)minIdx - coord.y * (float)R;
    if (coord.x < 1 || coord.y < 1 || coord.x >= 2 * maxShift || coord.y >= 2 * maxShift) {
        coord.x = 0;
        coord.y = 0;
    } else {
        float A11 = 0, A22 = 0, A12 = 0, b1 = 0, b2 = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int r = i / 3, c = i % 3;
            const float v = img[minIdx + (c - 1) + (r - 1) * R];
            A11 += c_FA11[i] * v;
            A22 += c_FA22[i] * v;
            A12 += c_FA12[i] * v;
            b1 += c_Fb1[i] * v;
            b2 += c_Fb2[i] * v;
        }
        A11 = fmaxf(A11, 0.0f);
        A22 = fmaxf(A22, 0.0f);
        float detA = A11 * A22 - A12 * A12;
        if (detA < 0) {
            A12 = 0;
            detA = A11 * A22;
        }
        if (detA != 0) {
            float muX = (A22 * b1 - A12 * b2) / detA;
            float muY = (A11 * b2 - A12 * b1) / detA;
            if (fabsf(muX) > 1) muX = 0;
            if (fabsf(muY) > 1) muY = 0;
            coord.x -= muX;
            coord.y -= muY;
        }
        coord.x -= (float)maxShift;
        coord.y -= (float)maxShift;
    }
    if (threshold + minVal > maxVal) {
        coord.x = 0;
        coord.y = 0;
    }
    return coord;
}

__global__ void __launch_bounds__(256)
    k_findMinimum(const float* __restrict__ shiftImage, float2* __restrict__ coordinates, int coordinatesPitch,
                  int maxShift, int tileCount, int tileCountX, float threshold)
{
    const int lane = threadIdx.x & 63;
    const int tileIdx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tileIdx >= tileCount) return;
    const int R = 2 * maxShift + 1;
    const float* img = shiftImage + (size_t)tileIdx * R * R;
    float minVal, maxVal;
    int minIdx;
    wave_min_argmin_max(img, R * R, lane, minVal, minIdx, maxVal);
    if (lane == 0) {
        const float2 coord = subpixel_minimum(img, maxShift, minVal, minIdx, maxVal, threshold);
        const int tileIdxY = tileIdx / tileCountX;
        const int tileIdxX = tileIdx - tileIdxY * tileCountX;
        row_ptr(coordinates, coordinatesPitch, tileIdxY)[tileIdxX] = coord;
    }
}

extern "C" int mfsr_findMinimum(const float* shiftImage, mfsr_float2* coordinates, int coordinatesPitch, int maxShift,
                                int tileCount, int tileCountX, float threshold, mfsr_stream_t stream)
{
    MFSR_REQUIRE(shiftImage && coordinates && maxShift >= 0 && tileCount > 0 && tileCountX > 0);
    MFSR_REQUIRE((long long)coordinatesPitch >= 8LL * tileCountX && (coordinatesPitch & 7) == 0 &&
                 ((uintptr_t)coordinates & 7) == 0);
    hipLaunchKernelGGL(k_findMinimum, dim3(mfsr_cdiv(tileCount, 4)), dim3(256), 0, mfsr_s(stream), shiftImage,
                       (float2*)coordinates, coordinatesPitch, maxShift, tileCount, tileCountX, threshold);
    return mfsr_launch_status("findMinimum");
}

// ---- B8: UpSampleShifts (kernel.cu:642-688) ------------------------------------
__global__ void __launch_bounds__(256)
    k_UpSampleShifts(const float2* __restrict__ inShift, float2* __restrict__ outShift, int inPitch, int outPitch,
                     int oldLevel, int newLevel, int oldCountX, int oldCountY, int newCountX, int newCountY,
                     int oldTileSize, int newTileSize);

// B8 for one tile of the new level (kernel.cu:642-688): shared by k_UpSampleShifts and the fused tracker, which takes
// the pre-shift of its tiles straight from the previous level's shifts (no separate launch, no pre-shift buffer)
__device__ __forceinline__ float2 upsample_shift_at(const float2* __restrict__ inShift, int inPitch, int oldLevel, int newLevel,
                                                    int oldCountX, int oldCountY, int oldTileSize, int newTileSize, int newBlockX,
                                                    int newBlockY)
{
    const float factor = (float)oldLevel * (float)oldTileSize / (float)(newLevel * newTileSize);
    const float oldX = (float)newBlockX / factor;
    const float oldY = (float)newBlockY / factor;
    int oldXMin = f2i(floorf(oldX)), oldXMax = f2i(ceilf(oldX));
    int oldYMin = f2i(floorf(oldY)), oldYMax = f2i(ceilf(oldY));
    oldXMin = min(oldXMin, oldCountX - 1);
    oldXMax = min(oldXMax, oldCountX - 1);
    oldYMin = min(oldYMin, oldCountY - 1);
    oldYMax = min(oldYMax, oldCountY - 1);
    const float2 oldMinMin = row_ptr(inShift, inPitch, oldYMin)[oldXMin];
    const float2 oldMaxMin = row_ptr(inShift, inPitch, oldYMin)[oldXMax];
    const float2 oldMinMax = row_ptr(inShift, inPitch, oldYMax)[oldXMin];
    const float2 oldMaxMax = row_ptr(inShift, inPitch, oldYMax)[oldXMax];
    const float wx = 1.0f - ((float)oldXMax - oldX);
    const float wy = 1.0f - ((float)oldYMax - oldY);
    float temp1 = oldMinMin.x + (oldMaxMin.x - oldMinMin.x) * wx;
    float temp2 = oldMinMax.x + (oldMaxMax.x - oldMinMax.x) * wx;
    float2 old;
    old.x = temp1 + (temp2 - temp1) * wy;
    temp1 = oldMinMin.y + (oldMaxMin.y - oldMinMin.y) * wx;
    temp2 = oldMinMax.y + (oldMaxMax.y - oldMinMax.y) * wx;
    old.y = temp1 + (temp2 - temp1) * wy;
    old.x *= (float)oldLevel / (float)newLevel;
    old.y *= (float)oldLevel / (float)newLevel;
    return old;
}

__global__ void __launch_bounds__(256)
    k_UpSampleShifts(const float2* __restrict__ inShift, float2* __restrict__ outShift, int inPitch, int outPitch,
                     int oldLevel, int newLevel, int oldCountX, int oldCountY, int newCountX, int newCountY,
                     int oldTileSize, int newTileSize)
{
    const int newBlockX = blockIdx.x * blockDim.x + threadIdx.x;
    const int newBlockY = blockIdx.y * blockDim.y + threadIdx.y;
    if (newBlockX >= newCountX || newBlockY >= newCountY) return;
    row_ptr(outShift, outPitch, newBlockY)[newBlockX] =
        upsample_shift_at(inShift, inPitch, oldLevel, newLevel, oldCountX, oldCountY, oldTileSize, newTileSize, newBlockX, newBlockY);
}

extern "C" int mfsr_UpSampleShifts(const mfsr_float2* inShift, mfsr_float2* outShift, int inPitch, int outPitch,
                                   int oldLevel, int newLevel, int oldCountX, int oldCountY, int newCountX,
                                   int newCountY, int oldTileSize, int newTileSize, mfsr_stream_t stream)
{
    MFSR_REQUIRE(inShift && outShift && oldLevel > 0 && newLevel > 0 && oldCountX > 0 && oldCountY > 0);
    MFSR_REQUIRE(newCountX > 0 && newCountY > 0 && oldTileSize > 0 && newTileSize > 0);
    MFSR_REQUIRE((long long)inPitch >= 8LL * oldCountX && (long long)outPitch >= 8LL * newCountX);
    MFSR_REQUIRE((inPitch & 7) == 0 && (outPitch & 7) == 0 && ((uintptr_t)inShift & 7) == 0 && ((uintptr_t)outShift & 7) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(newCountX, 64), mfsr_cdiv(newCountY, 4));
    hipLaunchKernelGGL(k_UpSampleShifts, grid, block, 0, mfsr_s(stream), (const float2*)inShift, (float2*)outShift,
                       inPitch, outPitch, oldLevel, newLevel, oldCountX, oldCountY, newCountX, newCountY, oldTileSize,
                       newTileSize);
    return mfsr_launch_status("UpSampleShifts");
}

// ---- E2: ComputeStructureTensor (kernel.cu:691-715) ----------------------------
__global__ void __launch_bounds__(256) k_ComputeStructureTensor(const float* __restrict__ imgDx,
                                                               const float* __restrict__ imgDy, pix3* __restrict__ outImg,
                                                               int imgWidth, int imgHeight, int imgDxDyPitch,
                                                               int imgOutPitch)
{
    const int pxX = blockIdx.x * blockDim.x + threadIdx.x;
    const int pxY = blockIdx.y * blockDim.y + threadIdx.y;
    if (pxX >= imgWidth || pxY >= imgHeight) return;
    const float dx = row_ptr(imgDx, imgDxDyPitch, pxY)[pxX];
    const float dy = row_ptr(imgDy, imgDxDyPitch, pxY)[pxX];
    pix3 val = {dx * dx, dy * dy, dx * dy};
    row_ptr(outImg, imgOutPitch, pxY)[pxX] = val;
}

extern "C" int mfsr_ComputeStructureTensor(const float* imgDx, const float* imgDy, mfsr_float3* outImg, int imgWidth,
                                           int imgHeight, int imgDxDyPitch, int imgOutPitch, mfsr_stream_t stream)
{
    MFSR_REQUIRE(imgDx && imgDy && outImg && imgWidth > 0 && imgHeight > 0);
    MFSR_REQUIRE((long long)imgDxDyPitch >= 4LL * imgWidth && (long long)imgOutPitch >= 12LL * imgWidth);
    MFSR_REQUIRE((imgDxDyPitch & 3) == 0 && (imgOutPitch & 3) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(imgWidth, 64), mfsr_cdiv(imgHeight, 4));
    hipLaunchKernelGGL(k_ComputeStructureTensor, grid, block, 0, mfsr_s(stream), imgDx, imgDy, (pix3*)outImg, imgWidth,
                       imgHeight, imgDxDyPitch, imgOutPitch);
    return mfsr_launch_status("ComputeStructureTensor");
}

// ---- E3: ComputeKernelParam (kernel.cu:718-790) --------------------------------
__device__ __forceinline__ pix3 kernel_param(pix3 grad, float Dth, float Dtr, float kDetail, float kDenoise,
                                             float kStretch, float kShrink)
{
    const float a11 = grad.x, a22 = grad.y, a12 = grad.z;
    const float help = sqrtf((a22 - a11) * (a22 - a11) + 4.0f * a12 * a12);
    float c = 2.0f * a12;
    float s = a22 - a11 + help;
    const float norm = sqrtf(c * c + s * s);
    if (norm > 0) {
        c /= norm;
        s /= norm;
    } else {
        c = 1;
        s = 0;
    }
    const float lam1 = (a11 + a22 + help) / 2.0f;
    const float lam2 = (a11 + a22 - help) / 2.0f;
    const float A = 1 + sqrtf((lam1 - lam2) * (lam1 - lam2) / ((lam1 + lam2) * (lam1 + lam2)));
    float D = 1 - sqrtf(lam1) / Dtr + Dth;
    D = fmaxf(fminf(1.0f, D), 0.0f);
    const float k1h = kDetail * kStretch * A;
    const float k2h = kDetail / kShrink * A;
    float k1 = ((1.0f - D) * k1h + D * kDetail * kDenoise);
    float k2 = ((1.0f - D) * k2h + D * kDetail * kDenoise);
    k1 *= k1;
    k2 *= k2;
    const float x2 = c, y2 = s, x1 = s, y1 = -c;
    const float b11 = k1 * x1 * x1 + x2 * x2 * k2;
    const float b12 = k1 * x1 * y1 + x2 * y2 * k2;
    const float b22 = k1 * y1 * y1 + y2 * y2 * k2;
    const float det = b11 * b22 - b12 * b12 + 0.0000000001f;
    pix3 kernel = {b22 / det, b11 / det, -b12 / det};
    return kernel;
}

__global__ void __launch_bounds__(256) k_ComputeKernelParam(pix3* __restrict__ kernelImg, int imgWidth, int imgHeight,
                                                           int imgOutPitch, float Dth, float Dtr, float kDetail,
                                                           float kDenoise, float kStretch, float kShrink)
{
    const int pxX = blockIdx.x * blockDim.x + threadIdx.x;
    const int pxY = blockIdx.y * blockDim.y + threadIdx.y;
    if (pxX >= imgWidth || pxY >= imgHeight) return;
    pix3* p = row_ptr(kernelImg, imgOutPitch, pxY) + pxX;
    *p = kernel_param(*p, Dth, Dtr, kDetail, kDenoise, kStretch, kShrink);
}

extern "C" int mfsr_ComputeKernelParam(mfsr_float3* kernelImg, int imgWidth, int imgHeight, int imgOutPitch, float Dth,
                                       float Dtr, float kDetail, float kDenoise, float kStretch, float kShrink,
                                       mfsr_stream_t stream)
{
    MFSR_REQUIRE(kernelImg && imgWidth > 0 && imgHeight > 0 && (long long)imgOutPitch >= 12LL * imgWidth);
    MFSR_REQUIRE((imgOutPitch & 3) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(imgWidth, 64), mfsr_cdiv(imgHeight, 4));
    hipLaunchKernelGGL(k_ComputeKernelParam, grid, block, 0, mfsr_s(stream), (pix3*)kernelImg, imgWidth, imgHeight,
                       imgOutPitch, Dth, Dtr, kDetail, kDenoise, kStretch, kShrink);
    return mfsr_launch_status("ComputeKernelParam");
}

// ---- H2: GammasRGB (kernel.cu:380-422) ------------------------------------------
__device__ __forceinline__ float apply_srgb_gamma(float valIn)
{
    if (valIn <= 0.0031308f) return 12.92f * valIn;
    return (1.0f + 0.055f) * powf(valIn, 1.0f / 2.4f) - 0.055f;
}

__device__ __forceinline__ float gamma_channel(float v)
{
    if (isnan(v)) v = 0;
    v = fmaxf(fminf(v, 1.0f), 0.0f);
    return apply_srgb_gamma(v);
}

__global__ void __launch_bounds__(256) k_GammasRGB(pix3* __restrict__ inOutImg, int imgWidth, int imgHeight, int imgPitch)
{
    const int pxX = blockIdx.x * blockDim.x + threadIdx.x;
    const int pxY = blockIdx.y * blockDim.y + threadIdx.y;
    if (pxX >= imgWidth || pxY >= imgHeight) return;
    pix3* p = row_ptr(inOutImg, imgPitch, pxY) + pxX;
    pix3 val = *p;
    val.x = gamma_channel(val.x);
    val.y = gamma_channel(val.y);
    val.z = gamma_channel(val.z);
    *p = val;
}

extern "C" int mfsr_GammasRGB(mfsr_float3* inOutImg, int imgWidth, int imgHeight, int imgPitch, mfsr_stream_t stream)
{
    MFSR_REQUIRE(inOutImg && imgWidth > 0 && imgHeight > 0 && (long long)imgPitch >= 12LL * imgWidth && (imgPitch & 3) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(imgWidth, 64), mfsr_cdiv(imgHeight, 4));
    hipLaunchKernelGGL(k_GammasRGB, grid, block, 0, mfsr_s(stream), (pix3*)inOutImg, imgWidth, imgHeight, imgPitch);
    return mfsr_launch_status("GammasRGB");
}

// ---- H1: ApplyWeighting (kernel.cu:426-481) -------------------------------------
__device__ __forceinline__ float apply_weight(float inout, float val, float w, float threshold)
{
    if (w < threshold) {
        val += inout;
        w += 1;
    }
    inout = 0;
    if (w != 0) inout = val / w;
    return inout;
}

__global__ void __launch_bounds__(256) k_ApplyWeighting(pix3* __restrict__ inOutImg, const pix3* __restrict__ finalImg,
                                                       const pix3* __restrict__ weight, int imgWidth, int imgHeight,
                                                       int imgPitch, float threshold)
{
    const int pxX = blockIdx.x * blockDim.x + threadIdx.x;
    const int pxY = blockIdx.y * blockDim.y + threadIdx.y;
    if (pxX >= imgWidth || pxY >= imgHeight) return;
    pix3* p = row_ptr(inOutImg, imgPitch, pxY) + pxX;
    pix3 inout = *p;
    const pix3 val = row_ptr(finalImg, imgPitch, pxY)[pxX];
    const pix3 w = row_ptr(weight, imgPitch, pxY)[pxX];
    inout.x = apply_weight(inout.x, val.x, w.x, threshold);
    inout.y = apply_weight(inout.y, val.y, w.y, threshold);
    inout.z = apply_weight(inout.z, val.z, w.z, threshold);
    *p = inout;
}

extern "C" int mfsr_ApplyWeighting(mfsr_float3* inOutImg, const mfsr_float3* finalImg, const mfsr_float3* weight,
                                   int imgWidth, int imgHeight, int imgPitch, float threshold, mfsr_stream_t stream)
{
    MFSR_REQUIRE(inOutImg && finalImg && weight && imgWidth > 0 && imgHeight > 0);
    MFSR_REQUIRE((long long)imgPitch >= 12LL * imgWidth && (imgPitch & 3) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(imgWidth, 64), mfsr_cdiv(imgHeight, 4));
    hipLaunchKernelGGL(k_ApplyWeighting, grid, block, 0, mfsr_s(stream), (pix3*)inOutImg, (const pix3*)finalImg,
                       (const pix3*)weight, imgWidth, imgHeight, imgPitch, threshold);
    return mfsr_launch_status("ApplyWeighting");
}

// ---- I1: fourierFilter / fftshift (kernel.cu:794-891) ---------------------------
__global__ void __launch_bounds__(256) k_fourierFilter(float2* img, size_t stride, int width, int height, float lp,
                                                      float hp, float lps, float hps, int clearAxis)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= width / 2 + 1) return;
    if (y >= height) return;
    float mx = (float)x;
    float my = (float)y;
    if (my > (float)height * 0.5f) my = ((float)height - my) * -1.0f;
    mx /= (float)width;
    my /= (float)height;
    const float dist = sqrtf(mx * mx + my * my);
    float fil = 0;
    lp = lp - lps;
    hp = hp + hps;
    if (lp > 0) {
        if (dist <= lp) fil = 1;
    } else {
        if (dist <= 1.0f) fil = 1;
    }
    if (lps > 0) {
        const float fil2 = (-fil + 1.0f) * expf(-((dist - lp) * (dist - lp) / (2 * lps * lps)));
        if (fil2 > 0.001f) fil = fil2;
    }
    if (lps > 0 && lp == 0 && hp == 0 && hps == 0) fil = expf(-((dist - lp) * (dist - lp) / (2 * lps * lps)));
    if (hp > 0) {
        float fil2 = 0;
        if (dist >= hp) fil2 = 1;
        fil *= fil2;
        if (hps > 0) {
            const float fil3 = (-fil2 + 1.0f) * expf(-((dist - hp) * (dist - hp) / (2 * hps * hps)));
            if (fil3 > 0.001f) fil = fil3;
        }
    }
    float2* row = (float2*)((char*)img + stride * (size_t)y);
    float2 erg = row[x];
    erg.x *= fil;
    erg.y *= fil;
    if (x < clearAxis || fabsf(my) * (float)height < (float)clearAxis) {
        erg.x = 0;
        erg.y = 0;
    }
    row[x] = erg;
}

extern "C" int mfsr_fourierFilter(mfsr_float2* img, size_t stride, int width, int height, float lp, float hp, float lps,
                                  float hps, int clearAxis, mfsr_stream_t stream)
{
    MFSR_REQUIRE(img && width > 0 && height > 0 && stride >= 8ull * (size_t)(width / 2 + 1) && (stride & 7) == 0);
    MFSR_REQUIRE(((uintptr_t)img & 7) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(width / 2 + 1, 64), mfsr_cdiv(height, 4));
    hipLaunchKernelGGL(k_fourierFilter, grid, block, 0, mfsr_s(stream), (float2*)img, stride, width, height, lp, hp, lps,
                       hps, clearAxis);
    return mfsr_launch_status("fourierFilter");
}

__global__ void __launch_bounds__(256) k_fftshift(float2* fft, int width, int height)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= width) return;
    if (y >= height) return;
    const int mx = x - width / 2;
    const int my = y - height / 2;
    const float a = 1.0f - (float)(2 * (((mx + my) & 1)));
    float2 erg = fft[(size_t)y * width + x];
    erg.x *= a;
    erg.y *= a;
    fft[(size_t)y * width + x] = erg;
}

extern "C" int mfsr_fftshift(mfsr_float2* fft, int width, int height, mfsr_stream_t stream)
{
    MFSR_REQUIRE(fft && width > 0 && height > 0 && ((uintptr_t)fft & 7) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(width, 64), mfsr_cdiv(height, 4));
    hipLaunchKernelGGL(k_fftshift, grid, block, 0, mfsr_s(stream), (float2*)fft, width, height);
    return mfsr_launch_status("fftshift");
}

// ---- fused tracker: B1+B2+B3+B4+cc+B6+B7 (+ rounded pre-shift add) --------------
// Threads gather the T x T reference template and the (T+2S)^2 pre-shifted moved patch of a
// tile into LDS (same source-pixel rule as B1/B2, base shift/rotation = 0) and evaluate the
// L2 distance of every candidate shift (sx, sy) in [0, 2S]^2:
//     D = sum(ref^2) + sum_window(moved^2) - 2*sum(ref*moved)
// with every sum taken in the order the unfused chain uses (row-major serial for sum(ref^2);
// per-row sums then the sum of rows for the correlation and the box term), so D is
// bit-identical to squaredSum'(serial)/boxFilter/normalizedCC fed by the direct correlation.
//  * sum(ref^2) does not depend on the candidate or on the moved frame: it is taken once per
//    reference (mfsr_tileSquaredSums) and passed in; without it one lane per tile takes it here.
//  * the box term shares its row sums: rowsq[patch row][sx] is used by all 2S+1 candidates of a
//    column, so it is tabulated first (L*(2S+1) sums of T) instead of (2S+1)^2 * T per tile.
//  * a thread owns NSX neighbouring sx of one sy and (2S+1)*ceil((2S+1)/NSX) threads serve a tile; a
//    workgroup packs as many tiles as fit into its 128 threads.  NSX > 1 re-uses each LDS load for
//    NSX multiply-adds but leaves about one wavefront per SIMD at 4K (every candidate is one serial
//    chain of T*T multiply-adds, so tiles x candidates is all the parallelism there is): measured
//    36.6 us (NSX = 1), 59 us (2), 60 us (3) per launch, so NSX = 1 is the default (MFSR_TRK_NSX).
//    (Reading the template through the scalar cache instead of LDS -- its address is wave-uniform with
//    one tile per workgroup -- was also tried: 41 us, the SMEM and LDS waits share one counter.)
// One wavefront per tile then reduces D with shuffles and lane 0 runs the quadratic fit.
#define TRK_THREADS 128
template <int TRK_NSX>
__global__ void __launch_bounds__(TRK_THREADS)
    k_trackTilesFused(const float* __restrict__ refImg, const float* __restrict__ movedImg,
                      const float2* __restrict__ preShift, int preShiftPitch, float2* __restrict__ coordinates,
                      int coordinatesPitch, int imgWidth, int imgHeight, int imgPitch, int maxShift, int tileSize,
                      int tileCountX, int tileCountY, float threshold, const float* __restrict__ refSq, int tilesPerWg,
                      const mfsr_prealign* __restrict__ base, float baseInvScale, const float2* __restrict__ coarse, int coarsePitch,
                      int4 up, int upOldTile)
{
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    const int T = tileSize, S = maxShift, L = T + 2 * S, R = 2 * S + 1;
    const int Lp = L + TRK_NSX;                 // patch row stride: the last sx group may read past the row
    const int G = (R + TRK_NSX - 1) / TRK_NSX;  // sx groups per sy
    const int nRef = T * T, nMov = L * Lp + TRK_NSX, nRow = L * R, nDist = R * R;
    const int slotFloats = ((nRef + nMov + nRow + nDist + 1) + 3) & ~3;
    const int tid = threadIdx.x;
    const int tileCount = tileCountX * tileCountY;
    const int tile0 = blockIdx.x * tilesPerWg;
    const float2 zero2 = make_float2(0.0f, 0.0f);
    // global pre-alignment of the moved frame (B2's baseShift / baseRotation, kernel.cu:358-368), read from device
    // memory: base shift in pixels of THIS pyramid level, cos/sin of the base rotation from the host-built table
    float2 baseShift = zero2;
    float baseCos = 1.0f, baseSin = 0.0f;
    if (base) {
        baseShift = make_float2(base->shiftX * baseInvScale, base->shiftY * baseInvScale);
        baseCos = base->cosRotation;
        baseSin = base->sinRotation;
    }
    auto slot_ref = [&](int s) { return s_mem + s * slotFloats; };
    auto slot_mov = [&](int s) { return s_mem + s * slotFloats + nRef; };
    auto slot_row = [&](int s) { return s_mem + s * slotFloats + nRef + nMov; };
    auto slot_dist = [&](int s) { return s_mem + s * slotFloats + nRef + nMov + nRow; };
    auto slot_sq = [&](int s) { return s_mem + s * slotFloats + nRef + nMov + nRow + nDist; };

    // gather (tiles past the end of the grid re-read the last tile: harmless, never written back).
    // Four independent fetches per thread and round, so their global-load latencies overlap.
    for (int sl = 0; sl < tilesPerWg; sl++) {
        const int tileIdx = min(tile0 + sl, tileCount - 1);
        const int tileIdxY = tileIdx / tileCountX, tileIdxX = tileIdx - tileIdxY * tileCountX;
        // pre-shift of the tile: given (B8's output), or taken here from the previous level's shifts (B8 folded in:
        // up = {oldLevel, newLevel, oldCountX, oldCountY}), or none
        float2 pre = zero2;
        if (preShift) pre = row_ptr(preShift, preShiftPitch, tileIdxY)[tileIdxX];
        else if (coarse) pre = upsample_shift_at(coarse, coarsePitch, up.x, up.y, up.z, up.w, upOldTile, T, tileIdxX, tileIdxY);
        float* sr = slot_ref(sl);
        float* sm = slot_mov(sl);
        for (int j0 = tid; j0 < nRef; j0 += 4 * TRK_THREADS) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = min(j0 + u * TRK_THREADS, nRef - 1);
                const int y = j / T, x = j - y * T;
                v[u] = tile_fetch(refImg, imgWidth, imgHeight, imgPitch, T, tileIdxX, tileIdxY, x + S, y + S, zero2, zero2, 0.0f);
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (j0 + u * TRK_THREADS < nRef) sr[j0 + u * TRK_THREADS] = v[u];
        }
        for (int k0 = tid; k0 < nMov; k0 += 4 * TRK_THREADS) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int k = min(k0 + u * TRK_THREADS, nMov - 1);
                const int y = k / Lp, x = k - y * Lp;
                // the pad columns / tail read a valid pixel and are zeroed below
                const float f = tile_fetch_cs(movedImg, imgWidth, imgHeight, imgPitch, T, tileIdxX, tileIdxY, min(x, L - 1),
                                              min(y, L - 1), pre, baseShift, baseCos, baseSin);
                v[u] = (y < L && x < L) ? f : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (k0 + u * TRK_THREADS < nMov) sm[k0 + u * TRK_THREADS] = v[u];
        }
    }
    __syncthreads();

    // row sums of moved^2 (serial over x, as boxFilterWithBorderX) and sum(ref^2)
    for (int i = tid; i < tilesPerWg * (nRow + 1); i += TRK_THREADS) {
        const int sl = i / (nRow + 1), j = i - sl * (nRow + 1);
        if (tile0 + sl >= tileCount) continue;
        if (j < nRow) {
            const int ry = j / R, sx = j - ry * R;
            const float* m = slot_mov(sl) + ry * Lp + sx;
            float a = 0;
            for (int x = 0; x < T; x++) a += m[x] * m[x];
            slot_row(sl)[j] = a;
        } else if (refSq) {
            *slot_sq(sl) = refSq[tile0 + sl];
        } else {
            const float* r = slot_ref(sl);
            float a = 0;
            for (int x = 0; x < nRef; x++) a += r[x] * r[x];
            *slot_sq(sl) = a;
        }
    }
    __syncthreads();

    // correlation: thread = (tile slot, sy, group of three sx)
    const int perTile = R * G;
    for (int item = tid; item < tilesPerWg * perTile; item += TRK_THREADS) {
        const int sl = item / perTile, q = item - sl * perTile;
        const int sy = q / G, sx0 = (q - sy * G) * TRK_NSX;
        if (tile0 + sl < tileCount) {
            const float* ref = slot_ref(sl);
            const float* mov = slot_mov(sl) + sy * Lp + sx0;
            float cc[TRK_NSX];
#pragma unroll
            for (int j = 0; j < TRK_NSX; j++) cc[j] = 0;
            for (int y = 0; y < T; y++) {
                const float* rrow = ref + y * T;
                const float* mrow = mov + y * Lp;
                float rs[TRK_NSX];  // row sum, left to right; the rows add top to bottom
#pragma unroll
                for (int j = 0; j < TRK_NSX; j++) rs[j] = 0;
                int x = 0;
                for (; x + 8 <= T; x += 8) {
                    float r[8], m[8 + TRK_NSX - 1];
#pragma unroll
                    for (int d = 0; d < 8; d++) r[d] = rrow[x + d];
#pragma unroll
                    for (int d = 0; d < 8 + TRK_NSX - 1; d++) m[d] = mrow[x + d];
#pragma unroll
                    for (int d = 0; d < 8; d++)
#pragma unroll
                        for (int j = 0; j < TRK_NSX; j++) rs[j] += r[d] * m[d + j];
                }
                for (; x < T; x++) {
                    const float r = rrow[x];
#pragma unroll
                    for (int j = 0; j < TRK_NSX; j++) rs[j] += r * mrow[x + j];
                }
#pragma unroll
                for (int j = 0; j < TRK_NSX; j++) cc[j] += rs[j];
            }
            const float sq = *slot_sq(sl);
            const float* rows = slot_row(sl);
#pragma unroll
            for (int j = 0; j < TRK_NSX; j++) {
                const int sx = sx0 + j;
                if (sx < R) {
                    float box = 0;
                    for (int y = 0; y < T; y++) box += rows[(sy + y) * R + sx];
                    slot_dist(sl)[sy * R + sx] = sq + box - 2 * cc[j];
                }
            }
        }
    }
    __syncthreads();

    // one wavefront per tile: min / argmin / max and the sub-pixel fit
    const int wave = tid >> 6, lane = tid & 63;
    for (int sl = wave; sl < tilesPerWg; sl += TRK_THREADS / 64) {
        const int tileIdx = tile0 + sl;
        if (tileIdx >= tileCount) continue;
        float minVal, maxVal;
        int minIdx;
        wave_min_argmin_max(slot_dist(sl), R * R, lane, minVal, minIdx, maxVal);
        if (lane == 0) {
            const int tileIdxY = tileIdx / tileCountX, tileIdxX = tileIdx - tileIdxY * tileCountX;
            float2 pre = zero2;
            if (preShift) pre = row_ptr(preShift, preShiftPitch, tileIdxY)[tileIdxX];
            else if (coarse) pre = upsample_shift_at(coarse, coarsePitch, up.x, up.y, up.z, up.w, upOldTile, T, tileIdxX, tileIdxY);
            float2 coord = subpixel_minimum(slot_dist(sl), S, minVal, minIdx, maxVal, threshold);
            coord.x = roundf(pre.x) + coord.x;
            coord.y = roundf(pre.y) + coord.y;
            row_ptr(coordinates, coordinatesPitch, tileIdxY)[tileIdxX] = coord;
        }
    }
}

// ---- fused tracker, compile-time tile geometry (the pipeline's sizes) -----------------------------
// The kernel above gives every candidate shift to one thread: (2S+1)^2 chains of T*T dependent multiply-adds per tile,
// 81 of 128 lanes busy, about one wavefront per SIMD at 4K and two LDS reads per multiply-add -- 37 us per launch with
// the VALUs 45 % busy.  The correlation's summation order (row sums, then the rows; oracle/glue.c) leaves T*(2S+1)^2
// independent row sums per tile, which this kernel spreads differently:
//   * correlation item = (template row y, sy): the thread loads the template row (T floats) and the patch row y + sy
//     (T + 2S floats) ONCE with ds_read_b128 and forms the 2S+1 row sums of all sx from registers -- 2S+1 independent
//     chains per thread, (2T + 2S) / (T * (2S+1)) LDS floats per multiply-add instead of 2;
//   * TPW tiles share a workgroup so that the TPW*T*(2S+1) items fill whole wavefronts; one more wavefront takes the
//     box term's row sums (a patch row per lane, all sx from registers) at the same time;
//   * LDS row strides are 4 * odd, so the 16 lanes of a ds_read_b128 quarter, which read 16 different rows, touch 16
//     different bank groups; row sums land in rowcc[item * R + sx] (stride R, odd: conflict-free).
// Then (2S+1)^2 threads per tile add the T row sums top to bottom (+ the box rows), and one wavefront per tile takes
// min / argmin / max and the sub-pixel fit as above.  Same bits as the kernel above for every input.
constexpr int trk_odd4(int n)
{
    int q = (n + 3) / 4;
    if ((q & 1) == 0) q++;
    return 4 * q;
}

template <int T, int S, int TPW>
struct TrkFast {
    static constexpr int L = T + 2 * S, R = 2 * S + 1;
    static constexpr int Tp = trk_odd4(T), Lp = trk_odd4(L);
    static constexpr int nRef = T * Tp, nMov = L * Lp, nRowSq = L * R, nRowCc = T * R * R, nDist = R * R;
    static constexpr int slotFloats = (nRef + nMov + nRowSq + nRowCc + nDist + 4 + 3) & ~3;
    static constexpr int items = TPW * T * R;                   // correlation items per workgroup
    static constexpr int corrThreads = (items + 63) & ~63;
    static constexpr int threads = corrThreads + 64;            // + the box-term wavefront
    static constexpr size_t ldsBytes = sizeof(float) * (size_t)slotFloats * TPW;
    static_assert(threads <= 1024 && ldsBytes <= 64 * 1024 && TPW * R * R <= corrThreads && T % 4 == 0, "tile geometry");
};

template <int T, int S, int TPW>
__global__ void __launch_bounds__((TrkFast<T, S, TPW>::threads))
    k_trackTilesFast(const float* __restrict__ refImg, const float* __restrict__ movedImg, const float2* __restrict__ preShift,
                     int preShiftPitch, float2* __restrict__ coordinates, int coordinatesPitch, int imgWidth, int imgHeight,
                     int imgPitch, int tileCountX, int tileCountY, float threshold, const float* __restrict__ refSq,
                     const mfsr_prealign* __restrict__ base, float baseInvScale, const float2* __restrict__ coarse, int coarsePitch,
                     int4 up, int upOldTile, MfsrBatch bt)
{
    if (gridDim.z > 1) {
        movedImg = (const float*)bt.p[blockIdx.z][0];
        coarse = (const float2*)bt.p[blockIdx.z][1];
        coordinates = (float2*)bt.p[blockIdx.z][2];
        base = (const mfsr_prealign*)bt.p[blockIdx.z][3];
    }
    using G = TrkFast<T, S, TPW>;
    constexpr int L = G::L, R = G::R, Tp = G::Tp, Lp = G::Lp;
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    const int tid = threadIdx.x;
    const int tileCount = tileCountX * tileCountY;
    const int tile0 = blockIdx.x * TPW;
    const float2 zero2 = make_float2(0.0f, 0.0f);
    float2 baseShift = zero2;
    float baseCos = 1.0f, baseSin = 0.0f;
    if (base) {
        baseShift = make_float2(base->shiftX * baseInvScale, base->shiftY * baseInvScale);
        baseCos = base->cosRotation;
        baseSin = base->sinRotation;
    }
    auto slot_ref = [&](int s) { return s_mem + s * G::slotFloats; };
    auto slot_mov = [&](int s) { return s_mem + s * G::slotFloats + G::nRef; };
    auto slot_rowsq = [&](int s) { return s_mem + s * G::slotFloats + G::nRef + G::nMov; };
    auto slot_rowcc = [&](int s) { return s_mem + s * G::slotFloats + G::nRef + G::nMov + G::nRowSq; };
    auto slot_dist = [&](int s) { return s_mem + s * G::slotFloats + G::nRef + G::nMov + G::nRowSq + G::nRowCc; };
    auto pre_of = [&](int tileIdxX, int tileIdxY) {
        float2 pre = zero2;
        if (preShift) pre = row_ptr(preShift, preShiftPitch, tileIdxY)[tileIdxX];
        else if (coarse) pre = upsample_shift_at(coarse, coarsePitch, up.x, up.y, up.z, up.w, upOldTile, T, tileIdxX, tileIdxY);
        return pre;
    };

    // gather, with every load of a phase in flight at once (the kernel is bound by these latencies, not by arithmetic):
    //   1. template (B1: zero shift, the clamped tile itself) -> registers
    //   2. meanwhile one lane per tile slot takes the pre-shift (given, or B8 folded in) and the patch origin -> LDS
    //   3. pre-shifted patch (B2) -> registers; 4. both -> LDS
    constexpr int nRefAll = TPW * T * T, nMovAll = TPW * L * L;
    constexpr int NR = (nRefAll + G::threads - 1) / G::threads, NM = (nMovAll + G::threads - 1) / G::threads;
    __shared__ int2 s_origin[TPW];
    __shared__ float2 s_pre[TPW];
    float vr[NR], vm[NM];
#pragma unroll
    for (int u = 0; u < NR; u++) {
        const int j = min(tid + u * G::threads, nRefAll - 1);
        const int sl = j / (T * T), q = j - sl * (T * T), y = q / T, x = q - y * T;
        const int tileIdx = min(tile0 + sl, tileCount - 1);  // slots past the end re-read the last tile, never written back
        const int tileIdxY = tileIdx / tileCountX, tileIdxX = tileIdx - tileIdxY * tileCountX;
        vr[u] = tile_fetch_at(refImg, imgWidth, imgHeight, imgPitch, tileIdxX * T + S, tileIdxY * T + S, x, y);
    }
    if (tid < TPW) {
        const int tileIdx = min(tile0 + tid, tileCount - 1);
        const int tileIdxY = tileIdx / tileCountX, tileIdxX = tileIdx - tileIdxY * tileCountX;
        const float2 pre = pre_of(tileIdxX, tileIdxY);
        const int2 o = tile_shift_cs(imgWidth, imgHeight, T, tileIdxX, tileIdxY, pre, baseShift, baseCos, baseSin);
        s_origin[tid] = make_int2(tileIdxX * T + o.x, tileIdxY * T + o.y);
        s_pre[tid] = pre;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NM; u++) {
        const int k = min(tid + u * G::threads, nMovAll - 1);
        const int sl = k / (L * L), q = k - sl * (L * L), y = q / L, x = q - y * L;
        const int2 o = s_origin[sl];
        vm[u] = tile_fetch_at(movedImg, imgWidth, imgHeight, imgPitch, o.x, o.y, x, y);
    }
#pragma unroll
    for (int u = 0; u < NR; u++) {
        const int j = tid + u * G::threads;
        if (j < nRefAll) {
            const int sl = j / (T * T), q = j - sl * (T * T), y = q / T, x = q - y * T;
            slot_ref(sl)[y * Tp + x] = vr[u];
        }
    }
#pragma unroll
    for (int u = 0; u < NM; u++) {
        const int k = tid + u * G::threads;
        if (k < nMovAll) {
            const int sl = k / (L * L), q = k - sl * (L * L), y = q / L, x = q - y * L;
            slot_mov(sl)[y * Lp + x] = vm[u];
        }
    }
    __syncthreads();

    if (tid < G::items) {
        // correlation row sums of (tile slot, template row y, sy), all sx
        const int sl = tid / (T * R), q = tid - sl * (T * R), y = q / R, sy = q - y * R;
        const float4* rrow = (const float4*)(slot_ref(sl) + y * Tp);
        const float4* mrow = (const float4*)(slot_mov(sl) + (y + sy) * Lp);
        float m[(L + 3) & ~3];
#pragma unroll
        for (int d = 0; d < (L + 3) / 4; d++) {
            const float4 t = mrow[d];
            m[4 * d] = t.x, m[4 * d + 1] = t.y, m[4 * d + 2] = t.z, m[4 * d + 3] = t.w;
        }
        float acc[R];
#pragma unroll
        for (int j = 0; j < R; j++) acc[j] = 0;
#pragma unroll
        for (int x4 = 0; x4 < T; x4 += 4) {
            const float4 t = rrow[x4 / 4];
            const float r[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int d = 0; d < 4; d++)
#pragma unroll
                for (int j = 0; j < R; j++) acc[j] += r[d] * m[x4 + d + j];
        }
        float* out = slot_rowcc(sl) + q * R;
#pragma unroll
        for (int j = 0; j < R; j++) out[j] = acc[j];
    } else if (tid >= G::corrThreads) {
        // box term: row sums of moved^2 (serial over x, as boxFilterWithBorderX), a patch row per lane
        for (int i = tid - G::corrThreads; i < TPW * L; i += 64) {
            const int sl = i / L, ry = i - sl * L;
            const float4* mrow = (const float4*)(slot_mov(sl) + ry * Lp);
            float m2[(L + 3) & ~3];
#pragma unroll
            for (int d = 0; d < (L + 3) / 4; d++) {
                const float4 t = mrow[d];
                m2[4 * d] = t.x * t.x, m2[4 * d + 1] = t.y * t.y, m2[4 * d + 2] = t.z * t.z, m2[4 * d + 3] = t.w * t.w;
            }
            float* out = slot_rowsq(sl) + ry * R;
#pragma unroll
            for (int j = 0; j < R; j++) {
                float a = 0;
#pragma unroll
                for (int x = 0; x < T; x++) a += m2[j + x];
                out[j] = a;
            }
        }
    }
    __syncthreads();

    // candidate (tile slot, sy, sx): rows top to bottom, box term, distance
    if (tid < TPW * R * R) {
        const int sl = tid / (R * R), cand = tid - sl * (R * R), sy = cand / R, sx = cand - sy * R;
        const float* rc = slot_rowcc(sl) + cand;
        const float* rq = slot_rowsq(sl) + sy * R + sx;
        float cc = 0, box = 0;
#pragma unroll 8
        for (int y = 0; y < T; y++) cc += rc[y * R * R];
#pragma unroll 8
        for (int y = 0; y < T; y++) box += rq[y * R];
        const float sq = refSq[min(tile0 + sl, tileCount - 1)];
        slot_dist(sl)[cand] = sq + box - 2 * cc;
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    if (wave < TPW && tile0 + wave < tileCount) {
        const int tileIdx = tile0 + wave;
        float minVal, maxVal;
        int minIdx;
        wave_min_argmin_max(slot_dist(wave), R * R, lane, minVal, minIdx, maxVal);
        if (lane == 0) {
            const int tileIdxY = tileIdx / tileCountX, tileIdxX = tileIdx - tileIdxY * tileCountX;
            const float2 pre = s_pre[wave];
            float2 coord = subpixel_minimum(slot_dist(wave), S, minVal, minIdx, maxVal, threshold);
            coord.x = roundf(pre.x) + coord.x;
            coord.y = roundf(pre.y) + coord.y;
            row_ptr(coordinates, coordinatesPitch, tileIdxY)[tileIdxX] = coord;
        }
    }
}

// sum(ref^2) per tile in the serial row-major order of squaredSum' (B3): once per reference frame
__global__ void __launch_bounds__(64)
    k_tileSquaredSums(const float* __restrict__ refImg, float* __restrict__ out, int imgWidth, int imgHeight, int imgPitch,
                      int maxShift, int tileSize, int tileCountX, int tileCountY)
{
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    const int T = tileSize, S = maxShift;
    const int tileIdx = blockIdx.x;
    const int tileIdxY = tileIdx / tileCountX, tileIdxX = tileIdx - tileIdxY * tileCountX;
    const float2 zero2 = make_float2(0.0f, 0.0f);
    for (int i = threadIdx.x; i < T * T; i += 64) {
        const int y = i / T, x = i - y * T;
        s_mem[i] = tile_fetch(refImg, imgWidth, imgHeight, imgPitch, T, tileIdxX, tileIdxY, x + S, y + S, zero2, zero2, 0.0f);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0;
        for (int i = 0; i < T * T; i++) a += s_mem[i] * s_mem[i];
        out[tileIdx] = a;
    }
}

extern "C" int mfsr_tileSquaredSums(const float* refImg, float* outValues, int imgWidth, int imgHeight, int imgPitch,
                                    int maxShift, int tileSize, int tileCountX, int tileCountY, mfsr_stream_t stream)
{
    MFSR_REQUIRE(refImg && outValues && imgWidth > 0 && imgHeight > 0);
    MFSR_REQUIRE((long long)imgPitch >= 4LL * imgWidth && (imgPitch & 3) == 0);
    MFSR_REQUIRE(maxShift >= 1 && maxShift <= 15 && tileSize >= 4 && tileSize <= 128 && tileCountX > 0 && tileCountY > 0);
    hipLaunchKernelGGL(k_tileSquaredSums, dim3(tileCountX * tileCountY), dim3(64), sizeof(float) * tileSize * tileSize,
                       mfsr_s(stream), refImg, outValues, imgWidth, imgHeight, imgPitch, maxShift, tileSize, tileCountX,
                       tileCountY);
    return mfsr_launch_status("tileSquaredSums");
}

static int track_tiles_fused_impl(const float* refImg, const float* movedImg, const mfsr_float2* preShift, int preShiftPitch,
                                  mfsr_float2* coordinates, int coordinatesPitch, int imgWidth, int imgHeight, int imgPitch,
                                  int maxShift, int tileSize, int tileCountX, int tileCountY, float threshold,
                                  const float* refSquaredSums, const mfsr_prealign* base, float baseInvScale,
                                  const mfsr_float2* coarse, int coarsePitch, int4 up, int upOldTile, mfsr_stream_t stream,
                                  int nFrames = 1, const MfsrBatch* batch = nullptr)
{
    MfsrBatch bt;
    if (batch)
        bt = *batch;
    else
        memset(&bt, 0, sizeof(bt));
    MFSR_REQUIRE(refImg && movedImg && coordinates && imgWidth > 0 && imgHeight > 0);
    MFSR_REQUIRE((long long)imgPitch >= 4LL * imgWidth && (imgPitch & 3) == 0);
    MFSR_REQUIRE(maxShift >= 1 && maxShift <= 15 && tileSize >= 4 && tileSize <= 128 && tileCountX > 0 && tileCountY > 0);
    MFSR_REQUIRE((long long)coordinatesPitch >= 8LL * tileCountX && (coordinatesPitch & 7) == 0 &&
                 ((uintptr_t)coordinates & 7) == 0);
    if (preShift)
        MFSR_REQUIRE((long long)preShiftPitch >= 8LL * tileCountX && (preShiftPitch & 7) == 0 && ((uintptr_t)preShift & 7) == 0);
    // the pipeline's tile sizes run the compile-time kernel (sum(ref^2) given, as the pipeline does); MFSR_TRK_FAST=0: A/B
    static const bool fast = [] {
        const char* e = getenv("MFSR_TRK_FAST");
        return !(e && e[0] == '0');
    }();
#define TRK_FAST_CASE(TT, SS, TPW)                                                                                       \
    if (fast && refSquaredSums && tileSize == TT && maxShift == SS) {                                                    \
        using G = TrkFast<TT, SS, TPW>;                                                                                  \
        hipLaunchKernelGGL((k_trackTilesFast<TT, SS, TPW>), dim3(mfsr_cdiv(tileCountX * tileCountY, TPW), 1, nFrames),    \
                           dim3(G::threads), G::ldsBytes, mfsr_s(stream), refImg, movedImg, (const float2*)preShift,      \
                           preShiftPitch, (float2*)coordinates, coordinatesPitch, imgWidth, imgHeight, imgPitch, tileCountX, \
                           tileCountY, threshold, refSquaredSums, base, baseInvScale, (const float2*)coarse, coarsePitch, \
                           up, upOldTile, bt);                                                                            \
        return mfsr_launch_status("trackTilesFast");                                                                     \
    }
    TRK_FAST_CASE(32, 4, 2)
    TRK_FAST_CASE(16, 3, 4)
    TRK_FAST_CASE(16, 4, 4)
    TRK_FAST_CASE(32, 8, 1)
#undef TRK_FAST_CASE
    if (nFrames > 1) return MFSR_E_UNSUPPORTED;  // frame batches run the compile-time kernel only
    static const int nsx = [] {
        const char* e = getenv("MFSR_TRK_NSX");
        return (e && e[0] >= '1' && e[0] <= '3') ? e[0] - '0' : 1;
    }();
    const int TRK_NSX = nsx;
    const int T = tileSize, L = T + 2 * maxShift, R = 2 * maxShift + 1, Lp = L + TRK_NSX;
    const int G = (R + TRK_NSX - 1) / TRK_NSX;
    const int slotFloats = ((T * T + L * Lp + TRK_NSX + L * R + R * R + 1) + 3) & ~3;
    int tilesPerWg = TRK_THREADS / (R * G) > 0 ? TRK_THREADS / (R * G) : 1;  // large search ranges: several rounds per tile
    while (tilesPerWg > 1 && sizeof(float) * (size_t)slotFloats * tilesPerWg > 56 * 1024) tilesPerWg--;
    const size_t lds = sizeof(float) * (size_t)slotFloats * tilesPerWg;
    if (lds > 64 * 1024) return MFSR_E_UNSUPPORTED;
    const int tiles = tileCountX * tileCountY;
#define TRK_LAUNCH(N)                                                                                                  \
    hipLaunchKernelGGL(k_trackTilesFused<N>, dim3(mfsr_cdiv(tiles, tilesPerWg)), dim3(TRK_THREADS), lds, mfsr_s(stream),  \
                       refImg, movedImg, (const float2*)preShift, preShiftPitch, (float2*)coordinates, coordinatesPitch, \
                       imgWidth, imgHeight, imgPitch, maxShift, tileSize, tileCountX, tileCountY, threshold,           \
                       refSquaredSums, tilesPerWg, base, baseInvScale, (const float2*)coarse, coarsePitch, up, upOldTile)
    if (nsx == 1)
        TRK_LAUNCH(1);
    else if (nsx == 2)
        TRK_LAUNCH(2);
    else
        TRK_LAUNCH(3);
#undef TRK_LAUNCH
    return mfsr_launch_status("trackTilesFused");
}

extern "C" int mfsr_trackTilesFusedBase(const float* refImg, const float* movedImg, const mfsr_float2* preShift,
                                        int preShiftPitch, mfsr_float2* coordinates, int coordinatesPitch, int imgWidth,
                                        int imgHeight, int imgPitch, int maxShift, int tileSize, int tileCountX,
                                        int tileCountY, float threshold, const float* refSquaredSums,
                                        const mfsr_prealign* base, float baseInvScale, mfsr_stream_t stream)
{
    return track_tiles_fused_impl(refImg, movedImg, preShift, preShiftPitch, coordinates, coordinatesPitch, imgWidth, imgHeight,
                                  imgPitch, maxShift, tileSize, tileCountX, tileCountY, threshold, refSquaredSums, base, baseInvScale,
                                  nullptr, 0, make_int4(0, 0, 0, 0), 0, stream);
}

// the same with UpSampleShifts (B8, kernel.cu:642) folded in: the pre-shift of every tile is the bilinear up-sampling of the
// previous (coarser) level's shifts `coarseShifts` (oldCountX x oldCountY tiles of oldTileSize at factor oldLevel), taken
// inside the kernel with B8's own arithmetic -- one launch and one buffer less per level, same bits
extern "C" int mfsr_trackTilesFusedUp(const float* refImg, const float* movedImg, const mfsr_float2* coarseShifts, int coarsePitch,
                                      int oldLevel, int newLevel, int oldCountX, int oldCountY, int oldTileSize,
                                      mfsr_float2* coordinates, int coordinatesPitch, int imgWidth, int imgHeight, int imgPitch,
                                      int maxShift, int tileSize, int tileCountX, int tileCountY, float threshold,
                                      const float* refSquaredSums, const mfsr_prealign* base, float baseInvScale,
                                      mfsr_stream_t stream)
{
    MFSR_REQUIRE(coarseShifts && oldLevel > 0 && newLevel > 0 && oldCountX > 0 && oldCountY > 0 && oldTileSize > 0);
    MFSR_REQUIRE((long long)coarsePitch >= 8LL * oldCountX && (coarsePitch & 7) == 0 && ((uintptr_t)coarseShifts & 7) == 0);
    return track_tiles_fused_impl(refImg, movedImg, nullptr, 0, coordinates, coordinatesPitch, imgWidth, imgHeight, imgPitch, maxShift,
                                  tileSize, tileCountX, tileCountY, threshold, refSquaredSums, base, baseInvScale, coarseShifts,
                                  coarsePitch, make_int4(oldLevel, newLevel, oldCountX, oldCountY), oldTileSize, stream);
}

// 1 if (tileSize, maxShift) runs the compile-time tracker kernel (the only one that takes frame batches)
extern "C" int mfsr_trackTilesFastSupported(int tileSize, int maxShift)
{
    return (tileSize == 32 && maxShift == 4) || (tileSize == 16 && maxShift == 3) || (tileSize == 16 && maxShift == 4) ||
           (tileSize == 32 && maxShift == 8);
}

// the tracker of one pyramid level for 1 .. 4 moved frames against one reference in ONE launch.  Every frame takes its pre-shifts
// from its own coarser-level shifts (coarseShifts, as mfsr_trackTilesFusedUp) or, at the coarsest level, none (all NULL).
// MFSR_E_UNSUPPORTED (nothing launched) unless (tileSize, maxShift) is one of the compile-time kernels' pairs.
extern "C" int mfsr_trackTilesFusedBatch(int nFrames, const mfsr_track_frame* frames, const float* refImg, int coarsePitch, int oldLevel,
                                         int newLevel, int oldCountX, int oldCountY, int oldTileSize, int coordinatesPitch, int imgWidth,
                                         int imgHeight, int imgPitch, int maxShift, int tileSize, int tileCountX, int tileCountY,
                                         float threshold, const float* refSquaredSums, float baseInvScale, mfsr_stream_t stream)
{
    MFSR_REQUIRE(frames && nFrames >= 1 && nFrames <= MFSR_BATCH_MAX && refSquaredSums);
    MfsrBatch bt;
    memset(&bt, 0, sizeof(bt));
    const bool up = frames[0].coarseShifts != nullptr;
    for (int i = 0; i < nFrames; i++) {
        MFSR_REQUIRE(frames[i].movedImg && frames[i].coordinates && ((uintptr_t)frames[i].coordinates & 7) == 0);
        MFSR_REQUIRE((frames[i].coarseShifts != nullptr) == up && ((uintptr_t)frames[i].coarseShifts & 7) == 0);
        bt.p[i][0] = frames[i].movedImg;
        bt.p[i][1] = frames[i].coarseShifts;
        bt.p[i][2] = frames[i].coordinates;
        bt.p[i][3] = frames[i].base;
    }
    if (up) {
        MFSR_REQUIRE(oldLevel > 0 && newLevel > 0 && oldCountX > 0 && oldCountY > 0 && oldTileSize > 0);
        MFSR_REQUIRE((long long)coarsePitch >= 8LL * oldCountX && (coarsePitch & 7) == 0);
    }
    return track_tiles_fused_impl(refImg, frames[0].movedImg, nullptr, 0, frames[0].coordinates, coordinatesPitch, imgWidth, imgHeight,
                                  imgPitch, maxShift, tileSize, tileCountX, tileCountY, threshold, refSquaredSums, frames[0].base,
                                  baseInvScale, frames[0].coarseShifts, coarsePitch,
                                  up ? make_int4(oldLevel, newLevel, oldCountX, oldCountY) : make_int4(0, 0, 0, 0), up ? oldTileSize : 0,
                                  stream, nFrames, &bt);
}

extern "C" int mfsr_trackTilesFused(const float* refImg, const float* movedImg, const mfsr_float2* preShift,
                                    int preShiftPitch, mfsr_float2* coordinates, int coordinatesPitch, int imgWidth,
                                    int imgHeight, int imgPitch, int maxShift, int tileSize, int tileCountX,
                                    int tileCountY, float threshold, const float* refSquaredSums, mfsr_stream_t stream)
{
    return mfsr_trackTilesFusedBase(refImg, movedImg, preShift, preShiftPitch, coordinates, coordinatesPitch, imgWidth, imgHeight,
                                    imgPitch, maxShift, tileSize, tileCountX, tileCountY, threshold, refSquaredSums, nullptr, 1.0f,
                                    stream);
}
