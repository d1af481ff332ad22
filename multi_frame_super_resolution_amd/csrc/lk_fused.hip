// lk_fused.hip -- one Lucas-Kanade refinement iteration in ONE launch
// (SURVEY.md section 8a rows D2+D3+D4).  Behavioural spec: reference
// test_opencv/opticalFlow.cu:28-44 (warp), :97-147 (derivatives), :190-325 (LK).
//
// The reference runs three kernels per iteration and moves warped/Ix/Iy/It
// through HBM (64 B/px/iter) and makes every pixel re-read its (2h+1)^2 window
// twice.  Here one 32x16 workgroup
//   1. warps the moved image for its tile + (h+2)-px halo straight into LDS
//      (bilinear gather, MIRROR addressing) next to the reference tile,
//   2. forms Ix, Iy, It and the five products IxIx, IxIy, IyIy, IxIt, IyIt for
//      the tile + h halo in LDS,
//   3. takes the window sums separably (row pass, then column pass),
//   4. solves the 2x2 system per pixel (closed-form SVD, lk_math.hpp) and
//      updates the flow.
// HBM traffic: 4 (ref) + 8 (flow in) + 8 (flow out) B/px plus the cached gather.
//
// Numerical note: the second window pass of the reference sums
// (M^-1 grad I) * It term by term (:311-312); the fused kernel uses the
// algebraically identical M^-1 * sum(grad I * It).  Together with the separable
// summation order the flow update agrees with the three-kernel chain to fp32
// rounding (tests: |d flow| <= 1e-4 px), not bit for bit.
//
// Argument order of the derivative stage: texSource = warped moved frame,
// texTarget = reference.  The reference's 5-point stencil (:116-119) is MINUS the
// usual derivative and Iz = source - target (:131); only with this order does
// `shift += UV` (:322-323) descend on |ref(p) - moved(p+u)|.
//
// Flow is double-buffered (shiftsIn -> shiftsOut): the halo of a tile reads
// flow values owned by other workgroups, so an in-place update would race.
#include "common.hpp"
#include "lk_math.hpp"

#define LK_TX 32
#define LK_TY 16
#define LK_THREADS (LK_TX * LK_TY)

// reflection about the image edges for -n <= i < 2n (the tile halo never reaches further):
// ... 1 0 | 0 1 ... n-1 | n-1 n-2 ...   (no integer division)
__device__ __forceinline__ int lk_mirror_index(int i, int n)
{
    i = i < 0 ? -1 - i : i;
    return i >= n ? 2 * n - 1 - i : i;
}

// HT > 0: half window size known at compile time (tile geometry becomes constant, so the
// index arithmetic of the staging loops needs no runtime integer division); HT == 0: runtime h.
template <int HT>
__global__ void __launch_bounds__(LK_THREADS)
    k_lkIterationFused(const float2* __restrict__ shiftsIn, float2* __restrict__ shiftsOut, int pitchShift,
                       const float* __restrict__ refImg, const float* __restrict__ movedImg, int pitchImg, int width,
                       int height, int hRuntime, float minDet)
{
    const int h = HT > 0 ? HT : hRuntime;
    extern __shared__ __attribute__((aligned(16))) float s_lk[];
    const int BW = LK_TX + 2 * h + 4, BH = LK_TY + 2 * h + 4;  // warped / ref region
    const int AW = LK_TX + 2 * h, AH = LK_TY + 2 * h;          // derivative region
    float* s_ref = s_lk;
    float* s_wrp = s_ref + BW * BH;
    float* s_p = s_wrp + BW * BH;      // 5 planes of AW*AH
    float* s_h = s_p + 5 * AW * AH;    // 5 planes of LK_TX*AH
    const int x0 = blockIdx.x * LK_TX, y0 = blockIdx.y * LK_TY;
    const int tid = threadIdx.y * LK_TX + threadIdx.x;

    mfsr_tex2d texMoved;
    texMoved.ptr = movedImg;
    texMoved.pitch = pitchImg;
    texMoved.width = width;
    texMoved.height = height;

    // 1. reference + warped moved image for the tile and its (h+2) halo
    for (int i = tid; i < BW * BH; i += LK_THREADS) {
        const int ly = i / BW, lx = i - ly * BW;
        const int gx = lk_mirror_index(x0 + lx - h - 2, width);
        const int gy = lk_mirror_index(y0 + ly - h - 2, height);
        const float2 f = row_ptr(shiftsIn, pitchShift, gy)[gx];
        const float u = ((float)gx + 0.5f + f.x) / (float)width;   // opticalFlow.cu:38-39
        const float v = ((float)gy + 0.5f + f.y) / (float)height;
        s_wrp[i] = tex1<ADDR_MIRROR>(texMoved, u, v);
        s_ref[i] = row_ptr(refImg, pitchImg, gy)[gx];
    }
    __syncthreads();

    // 2. derivatives and products on the tile + h halo
    const int planeA = AW * AH;
    for (int i = tid; i < planeA; i += LK_THREADS) {
        const int ay = i / AW, ax = i - ay * AW;
        const float* r = s_ref + (ay + 2) * BW + (ax + 2);
        const float* w = s_wrp + (ay + 2) * BW + (ax + 2);
        float t0 = r[2];
        t0 -= r[1] * 8.0f;
        t0 += r[-1] * 8.0f;
        t0 -= r[-2];
        t0 /= 12.0f;
        float t1 = w[2];
        t1 -= w[1] * 8.0f;
        t1 += w[-1] * 8.0f;
        t1 -= w[-2];
        t1 /= 12.0f;
        const float Ix = (t1 + t0) * 0.5f;  // source = warped, target = reference (see header note)
        const float It = w[0] - r[0];       // Iz = source - target (opticalFlow.cu:131)
        t0 = r[2 * BW];
        t0 -= r[BW] * 8.0f;
        t0 += r[-BW] * 8.0f;
        t0 -= r[-2 * BW];
        t0 /= 12.0f;
        t1 = w[2 * BW];
        t1 -= w[BW] * 8.0f;
        t1 += w[-BW] * 8.0f;
        t1 -= w[-2 * BW];
        t1 /= 12.0f;
        const float Iy = (t1 + t0) * 0.5f;
        s_p[i] = Ix * Ix;
        s_p[planeA + i] = Ix * Iy;
        s_p[2 * planeA + i] = Iy * Iy;
        s_p[3 * planeA + i] = Ix * It;
        s_p[4 * planeA + i] = Iy * It;
    }
    __syncthreads();

    // 3. row pass of the separable window sums
    const int planeH = LK_TX * AH;
    const int win = 2 * h + 1;
    for (int i = tid; i < 5 * planeH; i += LK_THREADS) {
        const int k = i / planeH;
        const int r = i - k * planeH;
        const int ay = r / LK_TX, x = r - ay * LK_TX;
        const float* p = s_p + k * planeA + ay * AW + x;
        float s = 0;
        for (int d = 0; d < win; d++) s += p[d];
        s_h[i] = s;
    }
    __syncthreads();

    // 4. column pass + solve + update
    const int pxX = x0 + threadIdx.x, pxY = y0 + threadIdx.y;
    if (pxX >= width || pxY >= height) return;
    float2 shift = row_ptr(shiftsIn, pitchShift, pxY)[pxX];
    if (!(pxX < h || pxX >= width - h || pxY < h || pxY >= height - h)) {
        float V[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const float* p = s_h + k * planeH + threadIdx.y * LK_TX + threadIdx.x;
            float s = 0;
            for (int d = 0; d < win; d++) s += p[d * LK_TX];
            V[k] = s;
        }
        float inv[4];
        if (lk_pinv(V[0], V[1], V[2], minDet, inv)) {
            float UV0 = inv[0] * V[3] + inv[1] * V[4];
            float UV1 = inv[2] * V[3] + inv[3] * V[4];
            UV0 = isnan(UV0) ? 0 : UV0;
            UV1 = isnan(UV1) ? 0 : UV1;
            shift.x += UV0;
            shift.y += UV1;
        }
    }
    row_ptr(shiftsOut, pitchShift, pxY)[pxX] = shift;
}

extern "C" int mfsr_lucasKanadeIterationFused(const mfsr_float2* shiftsIn, mfsr_float2* shiftsOut, int pitchShift,
                                              const float* refImg, const float* movedImg, int pitchImg, int width,
                                              int height, int halfWindowSize, float minDet, mfsr_stream_t stream)
{
    MFSR_REQUIRE(shiftsIn && shiftsOut && shiftsIn != shiftsOut && refImg && movedImg && width > 0 && height > 0);
    MFSR_REQUIRE(halfWindowSize >= 0 && halfWindowSize <= 15);
    MFSR_REQUIRE((long long)pitchImg >= 4LL * width && (pitchImg & 3) == 0);
    MFSR_REQUIRE((long long)pitchShift >= 8LL * width && (pitchShift & 7) == 0 && ((uintptr_t)shiftsIn & 7) == 0 &&
                 ((uintptr_t)shiftsOut & 7) == 0);
    const int h = halfWindowSize;
    const int BW = LK_TX + 2 * h + 4, BH = LK_TY + 2 * h + 4, AW = LK_TX + 2 * h, AH = LK_TY + 2 * h;
    const size_t lds = sizeof(float) * ((size_t)2 * BW * BH + (size_t)5 * AW * AH + (size_t)5 * LK_TX * AH);
    if (lds > 64 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            MFSR_HIP_TRY(hipFuncSetAttribute((const void*)k_lkIterationFused<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             160 * 1024));
            MFSR_HIP_TRY(hipFuncSetAttribute((const void*)k_lkIterationFused<7>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             160 * 1024));
            attr_set = true;
        }
    }
    if (lds > 160 * 1024) return MFSR_E_UNSUPPORTED;
    MFSR_REQUIRE(width >= LK_TX + 2 * h + 4 && height >= LK_TY + 2 * h + 4);  // reflection range of the halo
    dim3 block(LK_TX, LK_TY), grid(mfsr_cdiv(width, LK_TX), mfsr_cdiv(height, LK_TY));
#define LK_LAUNCH(HT)                                                                                                  \
    hipLaunchKernelGGL(k_lkIterationFused<HT>, grid, block, lds, mfsr_s(stream), (const float2*)shiftsIn,              \
                       (float2*)shiftsOut, pitchShift, refImg, movedImg, pitchImg, width, height, h, minDet)
    switch (h) {
        case 1: LK_LAUNCH(1); break;
        case 2: LK_LAUNCH(2); break;
        case 3: LK_LAUNCH(3); break;
        case 4: LK_LAUNCH(4); break;
        case 5: LK_LAUNCH(5); break;
        case 7: LK_LAUNCH(7); break;
        default: LK_LAUNCH(0); break;
    }
#undef LK_LAUNCH
    return mfsr_launch_status("lucasKanadeIterationFused");
}
