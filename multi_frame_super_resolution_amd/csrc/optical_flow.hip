// optical_flow.hip -- flow field creation, warp, derivatives and the
// Lucas-Kanade update, one launch per
// reference kernel.  Behavioural spec: reference test_opencv/opticalFlow.cu.
// The fused single-launch LK iteration lives in lk_fused.hip.
#include "common.hpp"
#include "lk_math.hpp"

// ---- D2: WarpingKernel (opticalFlow.cu:28-44) ---------------------------------
__global__ void __launch_bounds__(256)
    k_Warping(int width, int height, int stride, mfsr_tex2d texUV, float* __restrict__ out, mfsr_tex2d texToWarp)
{
    const int ix = threadIdx.x + blockIdx.x * blockDim.x;
    const int iy = threadIdx.y + blockIdx.y * blockDim.y;
    if (ix >= width || iy >= height) return;
    const float2 shift = tex2<ADDR_CLAMP>(texUV, ((float)ix + 0.5f) / (float)width, ((float)iy + 0.5f) / (float)height);
    const float x = ((float)ix + 0.5f + shift.x) / (float)width;
    const float y = ((float)iy + 0.5f + shift.y) / (float)height;
    row_ptr(out, stride, iy)[ix] = tex1<ADDR_MIRROR>(texToWarp, x, y);
}

extern "C" int mfsr_WarpingKernel(int width, int height, int stride, mfsr_tex2d texUV, float* out, mfsr_tex2d texToWarp,
                                  mfsr_stream_t stream)
{
    MFSR_REQUIRE(out && width > 0 && height > 0 && (long long)stride >= 4LL * width && (stride & 3) == 0);
    MFSR_REQUIRE(mfsr_tex_ok(texUV, 8) && ((uintptr_t)texUV.ptr & 7) == 0 && (texUV.pitch & 7) == 0);
    MFSR_REQUIRE(mfsr_tex_ok(texToWarp, 4) && (texToWarp.pitch & 3) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(width, 64), mfsr_cdiv(height, 4));
    hipLaunchKernelGGL(k_Warping, grid, block, 0, mfsr_s(stream), width, height, stride, texUV, out, texToWarp);
    return mfsr_launch_status("WarpingKernel");
}

// ---- D1: CreateFlowFieldFromTiles (opticalFlow.cu:48-93) ----------------------
__global__ void __launch_bounds__(256)
    k_CreateFlowFieldFromTiles(float2* __restrict__ outImg, mfsr_tex2d texShift, int imgWidth, int imgHeight, int imgPitch,
                               float2 baseShift, float baseRotation)
{
    const int pxX = blockIdx.x * blockDim.x + threadIdx.x;
    const int pxY = blockIdx.y * blockDim.y + threadIdx.y;
    if (pxX >= imgWidth || pxY >= imgHeight) return;
    // cosf(0) = 1 and sinf(0) = 0 exactly: the (uniform) zero-rotation case skips the two libm expansions
    float cf = 1.0f, sf = 0.0f;
    if (baseRotation != 0.0f) {
        cf = cosf(baseRotation);
        sf = sinf(baseRotation);
    }
    float2 shift;
    shift.x = cf * -baseShift.x - sf * -baseShift.y;
    shift.y = sf * -baseShift.x + cf * -baseShift.y;
    const float patchCenterX = (float)(pxX - imgWidth / 2);
    const float patchCenterY = (float)(pxY - imgHeight / 2);
    shift.x += cf * patchCenterX - sf * patchCenterY - patchCenterX;
    shift.y += sf * patchCenterX + cf * patchCenterY - patchCenterY;
    const float2 shiftPatch =
        tex2<ADDR_CLAMP>(texShift, ((float)pxX + 0.5f) / (float)imgWidth, ((float)pxY + 0.5f) / (float)imgHeight);
    shift.x += shiftPatch.x;
    shift.y += shiftPatch.y;
    row_ptr(outImg, imgPitch, pxY)[pxX] = shift;
}

extern "C" int mfsr_CreateFlowFieldFromTiles(mfsr_float2* outImg, mfsr_tex2d texObjShiftXY, int tileSize, int tileCountX,
                                             int tileCountY, int imgWidth, int imgHeight, int imgPitch,
                                             mfsr_float2 baseShift, float baseRotation, mfsr_stream_t stream)
{
    (void)tileSize;
    (void)tileCountX;
    (void)tileCountY;  // tileIdx is computed but unused by the reference (:66-72)
    MFSR_REQUIRE(outImg && imgWidth > 0 && imgHeight > 0);
    MFSR_REQUIRE((long long)imgPitch >= 8LL * imgWidth && (imgPitch & 7) == 0 && ((uintptr_t)outImg & 7) == 0);
    MFSR_REQUIRE(mfsr_tex_ok(texObjShiftXY, 8) && ((uintptr_t)texObjShiftXY.ptr & 7) == 0 && (texObjShiftXY.pitch & 7) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(imgWidth, 64), mfsr_cdiv(imgHeight, 4));
    hipLaunchKernelGGL(k_CreateFlowFieldFromTiles, grid, block, 0, mfsr_s(stream), (float2*)outImg, texObjShiftXY,
                       imgWidth, imgHeight, imgPitch, make_float2(baseShift.x, baseShift.y), baseRotation);
    return mfsr_launch_status("CreateFlowFieldFromTiles");
}

// D1 with the base shift / rotation read from device memory (result of mfsr_preAlign): same arithmetic, cos/sin of the
// base rotation from the host-built table (= cosf/sinf(rotation) of the host's libm, what the CPU oracle evaluates)
__global__ void __launch_bounds__(256)
    k_CreateFlowFieldFromTilesBase(float2* __restrict__ outImg, mfsr_tex2d texShift, int imgWidth, int imgHeight, int imgPitch,
                                   const mfsr_prealign* __restrict__ base)
{
    const int pxX = blockIdx.x * blockDim.x + threadIdx.x;
    const int pxY = blockIdx.y * blockDim.y + threadIdx.y;
    if (pxX >= imgWidth || pxY >= imgHeight) return;
    const float cf = base->cosRotation, sf = base->sinRotation;
    const float bx = base->shiftX, by = base->shiftY;
    float2 shift;
    shift.x = cf * -bx - sf * -by;
    shift.y = sf * -bx + cf * -by;
    const float patchCenterX = (float)(pxX - imgWidth / 2);
    const float patchCenterY = (float)(pxY - imgHeight / 2);
    shift.x += cf * patchCenterX - sf * patchCenterY - patchCenterX;
    shift.y += sf * patchCenterX + cf * patchCenterY - patchCenterY;
    const float2 shiftPatch =
        tex2<ADDR_CLAMP>(texShift, ((float)pxX + 0.5f) / (float)imgWidth, ((float)pxY + 0.5f) / (float)imgHeight);
    shift.x += shiftPatch.x;
    shift.y += shiftPatch.y;
    row_ptr(outImg, imgPitch, pxY)[pxX] = shift;
}

extern "C" int mfsr_CreateFlowFieldFromTilesBase(mfsr_float2* outImg, mfsr_tex2d texObjShiftXY, int imgWidth, int imgHeight,
                                                 int imgPitch, const mfsr_prealign* base, mfsr_stream_t stream)
{
    MFSR_REQUIRE(outImg && base && imgWidth > 0 && imgHeight > 0);
    MFSR_REQUIRE((long long)imgPitch >= 8LL * imgWidth && (imgPitch & 7) == 0 && ((uintptr_t)outImg & 7) == 0);
    MFSR_REQUIRE(mfsr_tex_ok(texObjShiftXY, 8) && ((uintptr_t)texObjShiftXY.ptr & 7) == 0 && (texObjShiftXY.pitch & 7) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(imgWidth, 64), mfsr_cdiv(imgHeight, 4));
    hipLaunchKernelGGL(k_CreateFlowFieldFromTilesBase, grid, block, 0, mfsr_s(stream), (float2*)outImg, texObjShiftXY, imgWidth,
                       imgHeight, imgPitch, base);
    return mfsr_launch_status("CreateFlowFieldFromTilesBase");
}

// ---- D3/E1: 5-point derivatives (opticalFlow.cu:97-185) -----------------------
__device__ __forceinline__ float deriv5(const mfsr_tex2d& t, float x, float y, float dx, float dy)
{
    float t0 = tex1<ADDR_MIRROR>(t, x + 2.0f * dx, y + 2.0f * dy);
    t0 -= tex1<ADDR_MIRROR>(t, x + 1.0f * dx, y + 1.0f * dy) * 8.0f;
    t0 += tex1<ADDR_MIRROR>(t, x - 1.0f * dx, y - 1.0f * dy) * 8.0f;
    t0 -= tex1<ADDR_MIRROR>(t, x - 2.0f * dx, y - 2.0f * dy);
    t0 /= 12.0f;
    return t0;
}

__global__ void __launch_bounds__(256) k_ComputeDerivatives(int width, int height, int stride, float* __restrict__ Ix,
                                                           float* __restrict__ Iy, float* __restrict__ Iz,
                                                           mfsr_tex2d texSource, mfsr_tex2d texTarget)
{
    const int ix = threadIdx.x + blockIdx.x * blockDim.x;
    const int iy = threadIdx.y + blockIdx.y * blockDim.y;
    if (ix >= width || iy >= height) return;
    const float dx = 1.0f / (float)width;
    const float dy = 1.0f / (float)height;
    const float x = ((float)ix + 0.5f) * dx;
    const float y = ((float)iy + 0.5f) * dy;
    float t0 = deriv5(texSource, x, y, dx, 0.0f);
    float t1 = deriv5(texTarget, x, y, dx, 0.0f);
    row_ptr(Ix, stride, iy)[ix] = (t0 + t1) * 0.5f;
    row_ptr(Iz, stride, iy)[ix] = tex1<ADDR_MIRROR>(texSource, x, y) - tex1<ADDR_MIRROR>(texTarget, x, y);
    t0 = deriv5(texSource, x, y, 0.0f, dy);
    t1 = deriv5(texTarget, x, y, 0.0f, dy);
    row_ptr(Iy, stride, iy)[ix] = (t0 + t1) * 0.5f;
}

extern "C" int mfsr_ComputeDerivativesKernel(int width, int height, int stride, float* Ix, float* Iy, float* Iz,
                                             mfsr_tex2d texSource, mfsr_tex2d texTarget, mfsr_stream_t stream)
{
    MFSR_REQUIRE(Ix && Iy && Iz && width > 0 && height > 0 && (long long)stride >= 4LL * width && (stride & 3) == 0);
    MFSR_REQUIRE(mfsr_tex_ok(texSource, 4) && mfsr_tex_ok(texTarget, 4));
    MFSR_REQUIRE(texSource.width == width && texSource.height == height && texTarget.width == width &&
                 texTarget.height == height);
    dim3 block(64, 4), grid(mfsr_cdiv(width, 64), mfsr_cdiv(height, 4));
    hipLaunchKernelGGL(k_ComputeDerivatives, grid, block, 0, mfsr_s(stream), width, height, stride, Ix, Iy, Iz, texSource,
                       texTarget);
    return mfsr_launch_status("ComputeDerivativesKernel");
}

// rows [row0, row0 + rows) of the image only (row0 = 0, rows = height: the whole image)
__global__ void __launch_bounds__(256) k_ComputeDerivatives2(int width, int height, int stride, float* __restrict__ Ix,
                                                            float* __restrict__ Iy, mfsr_tex2d tex, int row0, int rows)
{
    const int ix = threadIdx.x + blockIdx.x * blockDim.x;
    const int iy = row0 + threadIdx.y + blockIdx.y * blockDim.y;
    if (ix >= width || iy >= row0 + rows || iy >= height) return;
    const float dx = 1.0f / (float)width;
    const float dy = 1.0f / (float)height;
    const float x = ((float)ix + 0.5f) * dx;
    const float y = ((float)iy + 0.5f) * dy;
    row_ptr(Ix, stride, iy)[ix] = deriv5(tex, x, y, dx, 0.0f);
    row_ptr(Iy, stride, iy)[ix] = deriv5(tex, x, y, 0.0f, dy);
}

extern "C" int mfsr_ComputeDerivatives2Kernel(int width, int height, int stride, float* Ix, float* Iy, mfsr_tex2d tex,
                                              mfsr_stream_t stream)
{
    MFSR_REQUIRE(Ix && Iy && width > 0 && height > 0 && (long long)stride >= 4LL * width && (stride & 3) == 0);
    MFSR_REQUIRE(mfsr_tex_ok(tex, 4) && tex.width == width && tex.height == height);
    dim3 block(64, 4), grid(mfsr_cdiv(width, 64), mfsr_cdiv(height, 4));
    hipLaunchKernelGGL(k_ComputeDerivatives2, grid, block, 0, mfsr_s(stream), width, height, stride, Ix, Iy, tex, 0, height);
    return mfsr_launch_status("ComputeDerivatives2Kernel");
}

// the same for image rows [row0, row0 + rows) only (Ix, Iy: the full-size images; other rows are left untouched)
extern "C" int mfsr_ComputeDerivatives2Rows(int width, int height, int stride, float* Ix, float* Iy, mfsr_tex2d tex, int row0, int rows,
                                            mfsr_stream_t stream)
{
    MFSR_REQUIRE(Ix && Iy && width > 0 && height > 0 && (long long)stride >= 4LL * width && (stride & 3) == 0);
    MFSR_REQUIRE(mfsr_tex_ok(tex, 4) && tex.width == width && tex.height == height);
    MFSR_REQUIRE(row0 >= 0 && rows > 0 && row0 + rows <= height);
    dim3 block(64, 4), grid(mfsr_cdiv(width, 64), mfsr_cdiv(rows, 4));
    hipLaunchKernelGGL(k_ComputeDerivatives2, grid, block, 0, mfsr_s(stream), width, height, stride, Ix, Iy, tex, row0, rows);
    return mfsr_launch_status("ComputeDerivatives2Rows");
}

// ---- D4: lucasKanadeOptim (opticalFlow.cu:190-325) ----------------------------
// (the 2x2 pseudo-inverse lives in lk_math.hpp)
__global__ void __launch_bounds__(256)
    k_lucasKanadeOptim(float2* __restrict__ shifts, const float* __restrict__ imFx, const float* __restrict__ imFy,
                       const float* __restrict__ imFt, int pitchShift, int pitchImg, int width, int height,
                       int halfWindowSize, float minDet)
{
    const int pxX = blockIdx.x * blockDim.x + threadIdx.x;
    const int pxY = blockIdx.y * blockDim.y + threadIdx.y;
    if (pxX < halfWindowSize || pxX >= width - halfWindowSize || pxY < halfWindowSize || pxY >= height - halfWindowSize)
        return;
    float m0 = 0, m1 = 0, m3 = 0;
    for (int y = -halfWindowSize; y <= halfWindowSize; y++) {
        const float* rx = row_ptr(imFx, pitchImg, pxY + y);
        const float* ry = row_ptr(imFy, pitchImg, pxY + y);
        for (int x = -halfWindowSize; x <= halfWindowSize; x++) {
            const float dx = rx[pxX + x];
            const float dy = ry[pxX + x];
            m0 += dx * dx;
            m1 += dx * dy;
            m3 += dy * dy;
        }
    }
    float inv[4];
    if (!lk_pinv(m0, m1, m3, minDet, inv)) return;
    float UV0 = 0, UV1 = 0;
    for (int y = -halfWindowSize; y <= halfWindowSize; y++) {
        const float* rx = row_ptr(imFx, pitchImg, pxY + y);
        const float* ry = row_ptr(imFy, pitchImg, pxY + y);
        const float* rt = row_ptr(imFt, pitchImg, pxY + y);
        for (int x = -halfWindowSize; x <= halfWindowSize; x++) {
            const float dx = rx[pxX + x];
            const float dy = ry[pxX + x];
            const float dt = rt[pxX + x];
            UV0 += (inv[0] * dx + inv[1] * dy) * dt;
            UV1 += (inv[2] * dx + inv[3] * dy) * dt;
        }
    }
    UV0 = isnan(UV0) ? 0 : UV0;
    UV1 = isnan(UV1) ? 0 : UV1;
    float2* sp = row_ptr(shifts, pitchShift, pxY) + pxX;
    float2 shift = *sp;
    shift.x += UV0;
    shift.y += UV1;
    *sp = shift;
}

extern "C" int mfsr_lucasKanadeOptim(mfsr_float2* shifts, const float* imFx, const float* imFy, const float* imFt,
                                     int pitchShift, int pitchImg, int width, int height, int halfWindowSize,
                                     float minDet, mfsr_stream_t stream)
{
    MFSR_REQUIRE(shifts && imFx && imFy && imFt && width > 0 && height > 0 && halfWindowSize >= 0);
    MFSR_REQUIRE((long long)pitchImg >= 4LL * width && (pitchImg & 3) == 0);
    MFSR_REQUIRE((long long)pitchShift >= 8LL * width && (pitchShift & 7) == 0 && ((uintptr_t)shifts & 7) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(width, 64), mfsr_cdiv(height, 4));
    hipLaunchKernelGGL(k_lucasKanadeOptim, grid, block, 0, mfsr_s(stream), (float2*)shifts, imFx, imFy, imFt, pitchShift,
                       pitchImg, width, height, halfWindowSize, minDet);
    return mfsr_launch_status("lucasKanadeOptim");
}
