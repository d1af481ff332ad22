// robustness.hip -- per-pixel robustness weight.
// Behavioural spec: reference test_opencv/RobustnessModell.cu:29-158.
//
// The reference keeps a 3x3 float3 patch per thread in dynamic shared memory
// purely as private scratch (:45-46, no barrier); here the 9 reference pixels
// stay in VGPRs.  Quirk kept: of the 25 flow fetches only the last one
// (x=2,y=2) can influence min/max (:62-72), so only that one is issued -- the
// other 24 are dead in the reference too.
#include <cstring>
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

// ---- MI355X kernel ---------------------------------------------------------------------------------------------------
// The straight kernel above is VALU-bound, not memory-bound (706 VALU instructions per pixel: IEEE sqrt / division
// expansions, three ocml expf, two full bilinear fetches, 18 address computations; 38.9 us per 4K frame for 99.5 MB of
// algorithmic traffic, 164 MB counted).  This one
//  * stages the reference patch rows of a 64 x 8 tile (+1 halo) in LDS as three planes: the 27 reads of a pixel are
//    LDS reads at compile-time offsets instead of nine 12-byte global gathers with their address arithmetic
//    (HBM/L2 traffic of the reference image: once per tile instead of ~1.65x);
//  * takes every square root, reciprocal and exponential on the hardware units (v_sqrt_f32, v_rcp_f32, v_exp_f32: 1 ulp)
//    -- the mask is a continuous function of them, so the result moves by a few 1e-7 (test tolerance 2e-6); the two
//    discontinuities (round(0.5 * flow), M > thresholdM) keep their exact arithmetic: the flow fetch is the bit-exact
//    bilinear of the straight kernel and the means are summed in the reference's order and divided exactly;
//  * reads the one live sample of the 5 x 5 flow loop (x = y = 2, :66) as a plain texel when the flow field has the
//    image's resolution (the Bayer pipeline): the bilinear fetch lands on a texel centre up to 1 ulp of the coordinate;
//  * writes the zero ring the reference leaves to its caller (:48-49): no separate ring launch;
//  * (round 3, RB_FLOW_FIRST) fetches the pixel's flow BEFORE the reference patch is staged and gathers the moved patch
//    before the barrier: of the three dependent round trips (stage, flow, gather) two overlap -- the kernel waits twice as
//    long as it issues (SQ_WAIT_INST_ANY 1.05e8 vs SQ_ACTIVE_INST_VALU 5.2e7 per 4-frame launch); 115.5 -> 106 us.
#define RB_TX 64
#define RB_TY 8
#ifndef RB_FLOW_FIRST
#define RB_FLOW_FIRST 1
#endif
template <bool ALIGNED>
__global__ void __launch_bounds__(RB_TX* RB_TY)
    k_robustnessFused(const pix3* __restrict__ rawImgRef, const pix3* __restrict__ rawImgMoved, float4* __restrict__ robustnessMask,
                      mfsr_tex2d texUV, int imgWidth, int imgHeight, int imgPitch, int maskPitch, float alpha, float beta,
                      float thresholdM, MfsrBatch bt, MfsrExactDiv dW, MfsrExactDiv dH)
{
    if (gridDim.z > 1) {
        rawImgMoved = (const pix3*)bt.p[blockIdx.z][0];
        robustnessMask = (float4*)bt.p[blockIdx.z][1];
        texUV.ptr = bt.p[blockIdx.z][2];
    }
    __shared__ float sR[3][RB_TY + 2][RB_TX + 2];
    const int lx = threadIdx.x, ly = threadIdx.y;
    const int x0 = blockIdx.x * RB_TX, y0 = blockIdx.y * RB_TY;
#if RB_FLOW_FIRST
    // this pixel's flow first: its fetch and the moved-image gather that depends on it are then in flight while the
    // reference patch is staged (three dependent round trips become two)
    const int pxXe = min(x0 + lx, imgWidth - 1), pxYe = min(y0 + ly, imgHeight - 1);
    // (/ imgWidth, / imgHeight: mfsr_div, common.hpp)
    const float2 shiftfE = tex2<ADDR_CLAMP>(texUV, mfsr_div((float)pxXe + 0.5f, dW), mfsr_div((float)pxYe + 0.5f, dH));
    float2 sE;
    if (ALIGNED) {
        sE = row_ptr((const float2*)texUV.ptr, texUV.pitch, min(pxYe + 2, imgHeight - 1))[min(pxXe + 2, imgWidth - 1)];
    } else {
        sE = tex2<ADDR_CLAMP>(texUV, mfsr_div((float)pxXe + (float)2 + 0.5f, dW), mfsr_div((float)pxYe + (float)2 + 0.5f, dH));
    }
#endif
    for (int t = ly * RB_TX + lx; t < (RB_TY + 2) * (RB_TX + 2); t += RB_TX * RB_TY) {
        const int r = t / (RB_TX + 2), c = t - r * (RB_TX + 2);
        const int gy = clampi(y0 - 1 + r, 0, imgHeight - 1), gx = clampi(x0 - 1 + c, 0, imgWidth - 1);
        const pix3 p = row_ptr(rawImgRef, imgPitch, gy)[gx];
        sR[0][r][c] = p.x;
        sR[1][r][c] = p.y;
        sR[2][r][c] = p.z;
    }
#if RB_FLOW_FIRST
    // the moved patch under the rounded flow: gathered before the barrier, summed after it
    const int shxE = round2i(shiftfE.x * 0.5f), shyE = round2i(shiftfE.y * 0.5f);
    pix3 pmv[9];
#pragma unroll
    for (int y = 0; y < 3; y++) {
        const pix3* rm = row_ptr(rawImgMoved, imgPitch, clampi(pxYe + shyE + y - 1, 0, imgHeight - 1));
#pragma unroll
        for (int x = 0; x < 3; x++) pmv[y * 3 + x] = rm[clampi(pxXe + shxE + x - 1, 0, imgWidth - 1)];
    }
#endif
    __syncthreads();
    const int pxX = x0 + lx, pxY = y0 + ly;
    if (pxX >= imgWidth || pxY >= imgHeight) return;
    float4* out = row_ptr(robustnessMask, maskPitch, pxY) + pxX;
    if (pxX < 1 || pxY < 1 || pxX >= imgWidth - 1 || pxY >= imgHeight - 1) {
        *out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
#if RB_FLOW_FIRST
    const float2 shiftf = shiftfE, s = sE;   // (pxX == pxXe, pxY == pxYe for every pixel that gets here)
#else
    const float2 shiftf =
        tex2<ADDR_CLAMP>(texUV, ((float)pxX + 0.5f) / (float)imgWidth, ((float)pxY + 0.5f) / (float)imgHeight);
    float2 s;
    if (ALIGNED) {
        s = row_ptr((const float2*)texUV.ptr, texUV.pitch, min(pxY + 2, imgHeight - 1))[min(pxX + 2, imgWidth - 1)];
    } else {
        s = tex2<ADDR_CLAMP>(texUV, ((float)pxX + (float)2 + 0.5f) / (float)imgWidth, ((float)pxY + (float)2 + 0.5f) / (float)imgHeight);
    }
#endif
    float2 maxShift, minShift;
    maxShift.x = fmaxf(s.x, shiftf.x);
    maxShift.y = fmaxf(s.y, shiftf.y);
    minShift.x = fminf(s.x, shiftf.x);
    minShift.y = fminf(s.y, shiftf.y);
    const int shx = round2i(shiftf.x * 0.5f);
    const int shy = round2i(shiftf.y * 0.5f);
    (void)shx;

    // means in the reference's summation order (row-major from 0), exact division by 9
    float pr[3][9];
    float mr[3] = {0.0f, 0.0f, 0.0f}, mm[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int y = 0; y < 3; y++) {
        const pix3* rm = row_ptr(rawImgMoved, imgPitch, clampi(pxY + shy + y - 1, 0, imgHeight - 1));
#pragma unroll
        for (int x = 0; x < 3; x++) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                pr[c][y * 3 + x] = sR[c][ly + y][lx + x];
                mr[c] += pr[c][y * 3 + x];
            }
#if RB_FLOW_FIRST
            const pix3 p = pmv[y * 3 + x];
            (void)rm;
#else
            const pix3 p = rm[clampi(pxX + shx + x - 1, 0, imgWidth - 1)];
#endif
            mm[0] += p.x;
            mm[1] += p.y;
            mm[2] += p.z;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        mr[c] = div9(mr[c]);
        mm[c] = div9(mm[c]);
    }
    float meandist = fabsf(mr[0] - mm[0]) + fabsf(mr[1] - mm[1]) + fabsf(mr[2] - mm[2]);
    meandist = div3(meandist);
    const float hm = 0.5f * meandist;
    const float ddx = maxShift.x * hm - minShift.x * hm, ddy = maxShift.y * hm - minShift.y * hm;
    const float M = __builtin_amdgcn_sqrtf(ddx * ddx + ddy * ddy);
    const float sc = (M > thresholdM) ? 0.0f : 1.5f;
    float m3[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float sd2 = 0.0f;
#pragma unroll
        for (int p = 0; p < 9; p++) {
            const float d = pr[c][p] - mr[c];
            sd2 = __builtin_fmaf(d, d, sd2);
        }
        sd2 = div9(sd2);                                            // variance of the reference patch
        float smd2 = __builtin_fmaf(alpha, mr[c], beta);            // noise model variance (green: / 2, :131)
        if (c == 1) smd2 *= 0.5f;
        const float sg2 = fmaxf(smd2, sd2);                         // max(sigma_md, std)^2
        float d = fabsf(mr[c] - mm[c]);
        d = d * (sd2 * __builtin_amdgcn_rcpf(sd2 + smd2));          // Wiener shrink (:142-144); 0 * inf = NaN as in the reference
        const float e = __builtin_amdgcn_exp2f(-(d * d) * __builtin_amdgcn_rcpf(sg2) * 1.44269504088896340736f);
        m3[c] = fmaxf(fminf(__builtin_fmaf(sc, e, -0.12f), 1.0f), 0.0f);
    }
    *out = make_float4(m3[0], m3[1], m3[2], M);
}

// 0: the straight kernel (IEEE sqrt / division, ocml expf: tight parity tests); 1 (default): k_robustnessFused
static int g_robustness_fast = 1;
extern "C" int mfsr_set_robustness_fast(int enable)
{
    g_robustness_fast = enable ? 1 : 0;
    return MFSR_OK;
}

// F1 + the zero ring in one launch (what the burst pipeline calls); falls back to ring + straight kernel when the fast
// kernel is switched off
extern "C" int mfsr_robustnessMaskFused(const mfsr_float3* rawImgRef, const mfsr_float3* rawImgMoved, mfsr_float4* robustnessMask,
                                        mfsr_tex2d texUV, int imgWidth, int imgHeight, int imgPitch, int maskPitch, float alpha,
                                        float beta, float thresholdM, mfsr_stream_t stream)
{
    MFSR_REQUIRE(rawImgRef && rawImgMoved && robustnessMask && imgWidth > 2 && imgHeight > 2);
    MFSR_REQUIRE((long long)imgPitch >= 12LL * imgWidth && (imgPitch & 3) == 0);
    MFSR_REQUIRE((long long)maskPitch >= 16LL * imgWidth && (maskPitch & 15) == 0 && ((uintptr_t)robustnessMask & 15) == 0);
    MFSR_REQUIRE(mfsr_tex_ok(texUV, 8) && ((uintptr_t)texUV.ptr & 7) == 0 && (texUV.pitch & 7) == 0);
    if (!g_robustness_fast) {
        int rc = mfsr_zeroRing_f32x4(robustnessMask, maskPitch, imgWidth, imgHeight, stream);
        if (rc) return rc;
        return mfsr_ComputeRobustnessMask(rawImgRef, rawImgMoved, robustnessMask, texUV, imgWidth, imgHeight, imgPitch, maskPitch,
                                          alpha, beta, thresholdM, stream);
    }
    MfsrBatch bt;
    memset(&bt, 0, sizeof(bt));
    dim3 block(RB_TX, RB_TY), grid(mfsr_cdiv(imgWidth, RB_TX), mfsr_cdiv(imgHeight, RB_TY));
    const MfsrExactDiv dW = mfsr_exact_div((float)imgWidth), dH = mfsr_exact_div((float)imgHeight);
    if (texUV.width == imgWidth && texUV.height == imgHeight)
        hipLaunchKernelGGL(k_robustnessFused<true>, grid, block, 0, mfsr_s(stream), (const pix3*)rawImgRef, (const pix3*)rawImgMoved,
                           (float4*)robustnessMask, texUV, imgWidth, imgHeight, imgPitch, maskPitch, alpha, beta, thresholdM, bt, dW, dH);
    else
        hipLaunchKernelGGL(k_robustnessFused<false>, grid, block, 0, mfsr_s(stream), (const pix3*)rawImgRef, (const pix3*)rawImgMoved,
                           (float4*)robustnessMask, texUV, imgWidth, imgHeight, imgPitch, maskPitch, alpha, beta, thresholdM, bt, dW, dH);
    return mfsr_launch_status("robustnessMaskFused");
}

// mfsr_robustnessMaskFused for 1 .. 4 moved frames against one reference in one launch (the fused kernel only)
extern "C" int mfsr_robustnessMaskFusedBatch(int nFrames, const mfsr_robustness_frame* frames, const mfsr_float3* rawImgRef, int flowPitch,
                                             int flowWidth, int flowHeight, int imgWidth, int imgHeight, int imgPitch, int maskPitch,
                                             float alpha, float beta, float thresholdM, mfsr_stream_t stream)
{
    MFSR_REQUIRE(frames && nFrames >= 1 && nFrames <= MFSR_BATCH_MAX && rawImgRef && imgWidth > 2 && imgHeight > 2);
    MFSR_REQUIRE((long long)imgPitch >= 12LL * imgWidth && (imgPitch & 3) == 0);
    MFSR_REQUIRE((long long)maskPitch >= 16LL * imgWidth && (maskPitch & 15) == 0);
    if (!g_robustness_fast) return MFSR_E_UNSUPPORTED;
    mfsr_tex2d texUV;
    texUV.ptr = frames[0].flow;
    texUV.pitch = flowPitch;
    texUV.width = flowWidth;
    texUV.height = flowHeight;
    MFSR_REQUIRE(mfsr_tex_ok(texUV, 8) && (flowPitch & 7) == 0);
    MfsrBatch bt;
    memset(&bt, 0, sizeof(bt));
    for (int i = 0; i < nFrames; i++) {
        MFSR_REQUIRE(frames[i].movedHalf && frames[i].mask && frames[i].flow);
        MFSR_REQUIRE(((uintptr_t)frames[i].mask & 15) == 0 && ((uintptr_t)frames[i].flow & 7) == 0);
        bt.p[i][0] = frames[i].movedHalf;
        bt.p[i][1] = frames[i].mask;
        bt.p[i][2] = frames[i].flow;
    }
    dim3 block(RB_TX, RB_TY), grid(mfsr_cdiv(imgWidth, RB_TX), mfsr_cdiv(imgHeight, RB_TY), nFrames);
    const MfsrExactDiv dW = mfsr_exact_div((float)imgWidth), dH = mfsr_exact_div((float)imgHeight);
    if (flowWidth == imgWidth && flowHeight == imgHeight)
        hipLaunchKernelGGL(k_robustnessFused<true>, grid, block, 0, mfsr_s(stream), (const pix3*)rawImgRef, (const pix3*)frames[0].movedHalf,
                           (float4*)frames[0].mask, texUV, imgWidth, imgHeight, imgPitch, maskPitch, alpha, beta, thresholdM, bt, dW, dH);
    else
        hipLaunchKernelGGL(k_robustnessFused<false>, grid, block, 0, mfsr_s(stream), (const pix3*)rawImgRef, (const pix3*)frames[0].movedHalf,
                           (float4*)frames[0].mask, texUV, imgWidth, imgHeight, imgPitch, maskPitch, alpha, beta, thresholdM, bt, dW, dH);
    return mfsr_launch_status("robustnessMaskFusedBatch");
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
