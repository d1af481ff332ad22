// robustness.hip -- per-pixel robustness weight.
// Behavioural spec: reference test_opencv/RobustnessModell.cu:29-158.
//
// The reference keeps a 3x3 float3 patch per thread in dynamic shared memory
// purely as private scratch (:45-46, no barrier); here the 9 reference pixels
// stay in VGPRs.  Quirk kept: of the 25 flow fetches only the last one
// (x=2,y=2) can influence min/max (:62-72), so only that one is issued -- the
// other 24 are dead in the reference too.
#include "common.hpp"

// x / 9 and x / 3 as reciprocal multiply + one fma correction (3 instructions instead of the ~11 of the
// IEEE division expansion): q = x*r, q' = fma(fma(-d, q, x), r, q) with r = RN(1/d).  For d = 9 and
// d = 3 this equals the correctly rounded quotient for EVERY finite float x, checked exhaustively over
// all bit patterns on the host (it does not for 12 or sqrt(2), which keep the division).
__device__ __forceinline__ float div9(float x)
{
    const float q = x * 0.111111112f;
    return __builtin_fmaf(__builtin_fmaf(-9.0f, q, x), 0.111111112f, q);
}
__device__ __forceinline__ float div3(float x)
{
    const float q = x * 0.333333343f;
    return __builtin_fmaf(__builtin_fmaf(-3.0f, q, x), 0.333333343f, q);
}

__global__ void __launch_bounds__(256)
    k_ComputeRobustnessMask(const pix3* __restrict__ rawImgRef, const pix3* __restrict__ rawImgMoved,
                            float4* __restrict__ robustnessMask, mfsr_tex2d texUV, int imgWidth, int imgHeight,
                            int imgPitch, int maskPitch, float alpha, float beta, float thresholdM)
{
    const int pxX = blockIdx.x * blockDim.x + threadIdx.x;
    const int pxY = blockIdx.y * blockDim.y + threadIdx.y;
    if (pxX >= imgWidth - 1 || pxY >= imgHeight - 1 || pxX < 1 || pxY < 1) return;

    const float2 shiftf =
        tex2<ADDR_CLAMP>(texUV, ((float)pxX + 0.5f) / (float)imgWidth, ((float)pxY + 0.5f) / (float)imgHeight);
    // last sample of the 5x5 loop (:66 with x = y = 2)
    const float2 s = tex2<ADDR_CLAMP>(texUV, ((float)pxX + (float)2 + 0.5f) / (float)imgWidth,
                                      ((float)pxY + (float)2 + 0.5f) / (float)imgHeight);
    float2 maxShift, minShift;
    maxShift.x = fmaxf(s.x, shiftf.x);
    maxShift.y = fmaxf(s.y, shiftf.y);
    minShift.x = fminf(s.x, shiftf.x);
    minShift.y = fminf(s.y, shiftf.y);

    const int shx = round2i(shiftf.x * 0.5f);
    const int shy = round2i(shiftf.y * 0.5f);

    pix3 pixelsRef[9];
    float mrx = 0, mry = 0, mrz = 0, mmx = 0, mmy = 0, mmz = 0;
#pragma unroll
    for (int y = -1; y <= 1; y++) {
        const pix3* rr = row_ptr(rawImgRef, imgPitch, pxY + y);
        const int ppy = clampi(pxY + shy + y, 0, imgHeight - 1);
        const pix3* rm = row_ptr(rawImgMoved, imgPitch, ppy);
#pragma unroll
        for (int x = -1; x <= 1; x++) {
            pix3 p = rr[pxX + x];
            pixelsRef[(y + 1) * 3 + (x + 1)] = p;
            mrx += p.x;
            mry += p.y;
            mrz += p.z;
            const int ppx = clampi(pxX + shx + x, 0, imgWidth - 1);
            p = rm[ppx];
            mmx += p.x;
            mmy += p.y;
            mmz += p.z;
        }
    }
    mrx = div9(mrx);
    mry = div9(mry);
    mrz = div9(mrz);
    mmx = div9(mmx);
    mmy = div9(mmy);
    mmz = div9(mmz);

    float meandist = fabsf(mrx - mmx) + fabsf(mry - mmy) + fabsf(mrz - mmz);
    meandist = div3(meandist);
    maxShift.x *= 0.5f * meandist;
    maxShift.y *= 0.5f * meandist;
    minShift.x *= 0.5f * meandist;
    minShift.y *= 0.5f * meandist;
    const float M = sqrtf((maxShift.x - minShift.x) * (maxShift.x - minShift.x) +
                          (maxShift.y - minShift.y) * (maxShift.y - minShift.y));

    float sdx = 0, sdy = 0, sdz = 0;
#pragma unroll
    for (int p = 0; p < 9; p++) {
        sdx += (pixelsRef[p].x - mrx) * (pixelsRef[p].x - mrx);
        sdy += (pixelsRef[p].y - mry) * (pixelsRef[p].y - mry);
        sdz += (pixelsRef[p].z - mrz) * (pixelsRef[p].z - mrz);
    }
    sdx = sqrtf(div9(sdx));
    sdy = sqrtf(div9(sdy));
    sdz = sqrtf(div9(sdz));

    const float smx = sqrtf(alpha * mrx + beta);
    const float smy = sqrtf(alpha * mry + beta) / sqrtf(2.0f);
    const float smz = sqrtf(alpha * mrz + beta);

    float dx = fabsf(mrx - mmx), dy = fabsf(mry - mmy), dz = fabsf(mrz - mmz);
    const float sgx = fmaxf(smx, sdx), sgy = fmaxf(smy, sdy), sgz = fmaxf(smz, sdz);
    dx = dx * (sdx * sdx / (sdx * sdx + smx * smx));
    dy = dy * (sdy * sdy / (sdy * sdy + smy * smy));
    dz = dz * (sdz * sdz / (sdz * sdz + smz * smz));

    float sc = 1.5f;
    if (M > thresholdM) sc = 0;
    const float t = 0.12f;
    float4 mask;
    mask.x = fmaxf(fminf(sc * expf(-dx * dx / (sgx * sgx)) - t, 1.0f), 0.0f);
    mask.y = fmaxf(fminf(sc * expf(-dy * dy / (sgy * sgy)) - t, 1.0f), 0.0f);
    mask.z = fmaxf(fminf(sc * expf(-dz * dz / (sgz * sgz)) - t, 1.0f), 0.0f);
    mask.w = M;
    row_ptr(robustnessMask, maskPitch, pxY)[pxX] = mask;
}

extern "C" int mfsr_ComputeRobustnessMask(const mfsr_float3* rawImgRef, const mfsr_float3* rawImgMoved,
                                          mfsr_float4* robustnessMask, mfsr_tex2d texUV, int imgWidth, int imgHeight,
                                          int imgPitch, int maskPitch, float alpha, float beta, float thresholdM,
                                          mfsr_stream_t stream)
{
    MFSR_REQUIRE(rawImgRef && rawImgMoved && robustnessMask && imgWidth > 2 && imgHeight > 2);
    MFSR_REQUIRE((long long)imgPitch >= 12LL * imgWidth && (imgPitch & 3) == 0);
    MFSR_REQUIRE((long long)maskPitch >= 16LL * imgWidth && (maskPitch & 15) == 0 && ((uintptr_t)robustnessMask & 15) == 0);
    MFSR_REQUIRE(mfsr_tex_ok(texUV, 8) && ((uintptr_t)texUV.ptr & 7) == 0 && (texUV.pitch & 7) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(imgWidth, 64), mfsr_cdiv(imgHeight, 4));
    hipLaunchKernelGGL(k_ComputeRobustnessMask, grid, block, 0, mfsr_s(stream), (const pix3*)rawImgRef,
                       (const pix3*)rawImgMoved, (float4*)robustnessMask, texUV, imgWidth, imgHeight, imgPitch, maskPitch,
                       alpha, beta, thresholdM);
    return mfsr_launch_status("ComputeRobustnessMask");
}
