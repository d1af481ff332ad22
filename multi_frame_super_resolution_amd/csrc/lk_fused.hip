// lk_fused.hip -- one Lucas-Kanade refinement iteration in ONE launch
// (SURVEY.md section 8a rows D2+D3+D4).  Behavioural spec: reference
// test_opencv/opticalFlow.cu:28-44 (warp), :97-147 (derivatives), :190-325 (LK).
//
// The reference runs three kernels per iteration and moves warped/Ix/Iy/It
// through HBM (64 B/px/iter) and makes every pixel re-read its (2h+1)^2 window
// twice.  Here one 32x16 workgroup
//   1. warps the moved image for its tile + (h+2)-px halo (bilinear gather, MIRROR
//      addressing) and keeps warped + ref and warped - ref in LDS,
//   2. forms Ix, Iy, It and the five products IxIx, IxIy, IyIy, IxIt, IyIt for
//      the tile + h halo in LDS,
//   3. takes the window sums separably (row pass, then column pass),
//   4. solves the 2x2 system per pixel (closed-form SVD, lk_math.hpp) and
//      updates the flow.
// HBM traffic: 4 (ref) + 8 (flow in) + 8 (flow out) B/px plus the cached gather.
//
// Numerical note: Ix, Iy are taken on (warped + ref) with one *(1/24) instead of two /12 and a *0.5;
// the window passes add in a tree shared by four neighbours; the second window pass of the reference sums
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
#include <cstdlib>
#include <cstring>
#include "common.hpp"
#include "lk_math.hpp"

// Tile: LK_TY rows x TX columns (template), one thread per pixel.  TX = 48 is the default: with the
// 7x7 window the warped region (58 x 26 = 1508 samples) is 1.96 rounds of the 768 threads, where a
// 32-wide tile needs 3 rounds of 512 for 2.13 (the warp is half of the kernel's instructions); TX = 32
// serves images narrower than 48 + 2h + 4.
#ifndef LK_TY
#define LK_TY 16
#endif

// reflection about the image edges for -n <= i < 2n (the tile halo never reaches further):
// ... 1 0 | 0 1 ... n-1 | n-1 n-2 ...   (no integer division)
__device__ __forceinline__ int lk_mirror_index(int i, int n)
{
    i = i < 0 ? -1 - i : i;
    return i >= n ? 2 * n - 1 - i : i;
}

// base[y][x] for images of less than 4 GB: the byte offset y * pitch + 4 x in 32 bits, added to the (wave-uniform) base
// pointer by the load's own address mode -- instead of a 64-bit multiply-add per row pointer and a 64-bit add per element
// (8 to 9 VALU instructions per row step of the sweep).  The launcher checks pitch * height < 2^32.
template <typename T>
__device__ __forceinline__ T* lk_at(T* base, int pitch, int y, int x)
{
    return (T*)((char*)base + (uint32_t)((uint32_t)y * (uint32_t)pitch + (uint32_t)x * (uint32_t)sizeof(T)));
}

// warped moved image and reference at pixel (gx, gy) (inside the image) under flow f: opticalFlow.cu:28-44.
// Interior samples (0 <= u,v < 1, both texel pairs inside the image): mirror_coord is the identity and no index is
// clamped, so the fetch is four plain loads with the very same arithmetic as tex1<ADDR_MIRROR>; the few other samples
// are redone with the full addressing (wave-uniform branch: only waves that touch the image border pay for it).
__device__ __forceinline__ void lk_warp_sample(const float* __restrict__ refImg, const float* __restrict__ movedImg, int pitchImg,
                                               int width, int height, int gx, int gy, float2 f, float& wv, float& rv,
                                               const MfsrExactDiv* dW = nullptr, const MfsrExactDiv* dH = nullptr)
{
    rv = *lk_at(refImg, pitchImg, gy, gx);
    const float u = dW ? mfsr_div((float)gx + 0.5f + f.x, *dW) : ((float)gx + 0.5f + f.x) / (float)width;   // opticalFlow.cu:38-39
    const float v = dH ? mfsr_div((float)gy + 0.5f + f.y, *dH) : ((float)gy + 0.5f + f.y) / (float)height;
    const float xB = u * (float)width - 0.5f, yB = v * (float)height - 0.5f;
    const float fx = floorf(xB), fy = floorf(yB);
    const int ix = f2i(fx), iy = f2i(fy);
    const bool interior = u >= 0.0f && u < 1.0f && v >= 0.0f && v < 1.0f && (uint32_t)ix <= (uint32_t)(width - 2) &&
                          (uint32_t)iy <= (uint32_t)(height - 2);
    const int ixc = clampi(ix, 0, width - 2), iyc = clampi(iy, 0, height - 2);
    const float* r0 = lk_at(movedImg, pitchImg, iyc, ixc);
    const float* r1 = lk_at(movedImg, pitchImg, iyc + 1, ixc);
    wv = lerp4(r0[0], r0[1], r1[0], r1[1], xB - fx, yB - fy);
    if (__ballot(!interior) != 0) {
        mfsr_tex2d texMoved;
        texMoved.ptr = movedImg;
        texMoved.pitch = pitchImg;
        texMoved.width = width;
        texMoved.height = height;
        const float full = tex1<ADDR_MIRROR>(texMoved, u, v);
        wv = interior ? wv : full;
    }
}

// HT > 0: half window size known at compile time (tile geometry becomes constant, so the
// index arithmetic of the staging loops needs no runtime integer division); HT == 0: runtime h.
// PRE: the warped moved image enters as two images made by the PREVIOUS launch -- sumIn = warped + ref, diffIn = warped - ref
// at every pixel (mfsr_CreateFlowFieldWarped before the first iteration, this kernel's own epilogue afterwards) -- and the
// tile + halo is loaded instead of warped: every pixel is warped once per iteration, by the thread that has just updated
// its flow, instead of 1.96 times (tile + halo of a 48 x 16 tile with h = 3).  Same samples, same arithmetic, same bits.
template <int HT, int LK_TX, bool PRE = false>
__global__ void __launch_bounds__(LK_TX * LK_TY)
    k_lkIterationFused(const float2* __restrict__ shiftsIn, float2* __restrict__ shiftsOut, int pitchShift,
                       const float* __restrict__ refImg, const float* __restrict__ movedImg, int pitchImg, int width,
                       int height, int hRuntime, float minDet, float outScale, const float* __restrict__ sumIn = nullptr,
                       const float* __restrict__ diffIn = nullptr, float* __restrict__ sumOut = nullptr,
                       float* __restrict__ diffOut = nullptr, int pitchSD = 0)
{
    constexpr int LK_THREADS = LK_TX * LK_TY;
    const int h = HT > 0 ? HT : hRuntime;
    extern __shared__ __attribute__((aligned(16))) float s_lk[];
    const int BW = LK_TX + 2 * h + 4, BH = LK_TY + 2 * h + 4;  // warped / ref region
    const int AW = LK_TX + 2 * h, AH = LK_TY + 2 * h;          // derivative region
    // row stride of the product planes: a multiple of 4 floats when h is a compile-time constant, so that the row pass
    // reads its window with aligned ds_read_b128 (lanes 16 bytes apart: conflict-free) instead of dwords 16 bytes apart
    // (4-way bank conflicts)
    const int AWp = HT > 0 ? ((AW + 3) & ~3) : AW;
    float* s_ref = s_lk;
    float* s_wrp = s_ref + BW * BH;
    float* s_p = s_wrp + BW * BH;      // 5 planes of AWp*AH
    float* s_h = s_p + 5 * AWp * AH;   // 5 planes of LK_TX*AH
    const int x0 = blockIdx.x * LK_TX, y0 = blockIdx.y * LK_TY;
    const int tid = threadIdx.y * LK_TX + threadIdx.x;

    // 1. reference + warped moved image for the tile and its (h+2) halo.  With the trip count known (HT > 0) the rounds
    // are unrolled so that the loads of all rounds are in flight together.
    auto warp_one = [&](int i, bool active) {
        const int ly = i / BW, lx = i - ly * BW;
        const int gx = lk_mirror_index(x0 + lx - h - 2, width);
        const int gy = lk_mirror_index(y0 + ly - h - 2, height);
        if (PRE) {
            const float sv = row_ptr(sumIn, pitchSD, gy)[gx], dv = row_ptr(diffIn, pitchSD, gy)[gx];
            if (active) {
                s_ref[i] = sv;
                s_wrp[i] = dv;
            }
            return;
        }
        const float2 f = row_ptr(shiftsIn, pitchShift, gy)[gx];
        float wv, rv;
        lk_warp_sample(refImg, movedImg, pitchImg, width, height, gx, gy, f, wv, rv);
        if (active) {
            // the derivative stencil is linear: keep (warped + ref) for Ix, Iy and (warped - ref) = It
            s_ref[i] = wv + rv;
            s_wrp[i] = wv - rv;
        }
    };
    if (HT > 0) {
        constexpr int H1 = HT > 0 ? HT : 1;
        constexpr int N1 = (LK_TX + 2 * H1 + 4) * (LK_TY + 2 * H1 + 4), R1 = (N1 + LK_THREADS - 1) / LK_THREADS;
#pragma unroll
        for (int r = 0; r < R1; r++) {
            const int i = tid + r * LK_THREADS;
            warp_one(i < N1 ? i : N1 - 1, i < N1);
        }
    } else {
        for (int i = tid; i < BW * BH; i += LK_THREADS) warp_one(i, true);
    }
    __syncthreads();

    // 2. derivatives and products on the tile + h halo
    const int planeA = AWp * AH;
    for (int i0 = tid; i0 < AW * AH; i0 += LK_THREADS) {
        const int ay = i0 / AW, ax = i0 - ay * AW;
        const int i = ay * AWp + ax;
        const float* r = s_ref + (ay + 2) * BW + (ax + 2);  // warped + ref
        // opticalFlow.cu:116-131 with source = warped, target = reference (see header note):
        // Ix = (d(warped) + d(ref)) / 2 with d = (f[2] - 8 f[1] + 8 f[-1] - f[-2]) / 12, taken on the sum image
        float t0 = r[2];
        t0 -= r[1] * 8.0f;
        t0 += r[-1] * 8.0f;
        t0 -= r[-2];
        const float Ix = t0 * (1.0f / 24.0f);
        const float It = s_wrp[(ay + 2) * BW + (ax + 2)];  // Iz = source - target (opticalFlow.cu:131)
        float t1 = r[2 * BW];
        t1 -= r[BW] * 8.0f;
        t1 += r[-BW] * 8.0f;
        t1 -= r[-2 * BW];
        const float Iy = t1 * (1.0f / 24.0f);
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
    if (HT > 0) {
        // four adjacent outputs per item share their loads and the sum of the common window part
        constexpr int WIN = 2 * (HT > 0 ? HT : 1) + 1, G = LK_TX / 4;
        for (int i = tid; i < 5 * AH * G; i += LK_THREADS) {
            const int k = i / (AH * G);
            const int r = i - k * (AH * G);
            const int ay = r / G, x = (r - ay * G) * 4;
            const float4* p4 = (const float4*)(s_p + k * planeA + ay * AWp + x);  // 16-byte aligned: AWp, planeA, x % 4 == 0
            constexpr int NV = (WIN + 3 + 3) / 4;
            float v[4 * NV];
#pragma unroll
            for (int d = 0; d < NV; d++) {
                const float4 t = p4[d];
                v[4 * d] = t.x, v[4 * d + 1] = t.y, v[4 * d + 2] = t.z, v[4 * d + 3] = t.w;
            }
            float o[4];
            if (WIN >= 4) {
                float core = v[3];
#pragma unroll
                for (int d = 4; d < WIN; d++) core += v[d];
                const float l2 = v[2], l1 = v[1] + v[2], l0 = v[0] + l1;
                const float r1 = v[WIN], r2 = r1 + v[WIN + 1], r3 = r2 + v[WIN + 2];
                o[0] = l0 + core;
                o[1] = (l1 + core) + r1;
                o[2] = (l2 + core) + r2;
                o[3] = core + r3;
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float a = v[q];
#pragma unroll
                    for (int d = 1; d < WIN; d++) a += v[q + d];
                    o[q] = a;
                }
            }
            float* out = s_h + k * planeH + ay * LK_TX + x;
            *(float4*)out = make_float4(o[0], o[1], o[2], o[3]);
        }
    } else {
        for (int i = tid; i < 5 * planeH; i += LK_THREADS) {
            const int k = i / planeH;
            const int r = i - k * planeH;
            const int ay = r / LK_TX, x = r - ay * LK_TX;
            const float* p = s_p + k * planeA + ay * AWp + x;
            float s = 0;
            for (int d = 0; d < win; d++) s += p[d];
            s_h[i] = s;
        }
    }
    __syncthreads();

    // 4. column pass + solve + update
    auto solve_store = [&](int pxX, int pxY, const float (&V)[5]) {
        if (pxX >= width || pxY >= height) return;
        float2 shift = row_ptr(shiftsIn, pitchShift, pxY)[pxX];
        if (!(pxX < h || pxX >= width - h || pxY < h || pxY >= height - h)) {
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
        if (PRE && sumOut) {
            // the next iteration's input: this pixel warped under its new flow (outScale is 1 on every iteration but the last)
            float wv, rv;
            lk_warp_sample(refImg, movedImg, pitchImg, width, height, pxX, pxY, shift, wv, rv);
            row_ptr(sumOut, pitchSD, pxY)[pxX] = wv + rv;
            row_ptr(diffOut, pitchSD, pxY)[pxX] = wv - rv;
        }
        // outScale != 1 on the last iteration of a pipeline whose tracking image is smaller than the raw
        // frame: the flow leaves in raw-pixel units without a separate scaling pass (x1 is exact)
        shift.x *= outScale;
        shift.y *= outScale;
        row_ptr(shiftsOut, pitchShift, pxY)[pxX] = shift;
    };
    {
        float V[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const float* p = s_h + k * planeH + threadIdx.y * LK_TX + threadIdx.x;
            float sum = 0;
            if (HT > 0) {
#pragma unroll
                for (int d = 0; d < 2 * HT + 1; d++) sum += p[d * LK_TX];
            } else {
                for (int d = 0; d < win; d++) sum += p[d * LK_TX];
            }
            V[k] = sum;
        }
        solve_store(x0 + (int)threadIdx.x, y0 + (int)threadIdx.y, V);
    }
}

template <bool PRE>
static int lk_iteration_impl(const mfsr_float2* shiftsIn, mfsr_float2* shiftsOut, int pitchShift, const float* refImg,
                             const float* movedImg, int pitchImg, int width, int height, int halfWindowSize, float minDet,
                             float outScale, const float* sumIn, const float* diffIn, float* sumOut, float* diffOut, int pitchSD,
                             mfsr_stream_t stream)
{
    MFSR_REQUIRE(shiftsIn && shiftsOut && shiftsIn != shiftsOut && refImg && movedImg && width > 0 && height > 0);
    MFSR_REQUIRE(halfWindowSize >= 0 && halfWindowSize <= 15);
    MFSR_REQUIRE((long long)pitchImg >= 4LL * width && (pitchImg & 3) == 0);
    MFSR_REQUIRE((long long)pitchShift >= 8LL * width && (pitchShift & 7) == 0 && ((uintptr_t)shiftsIn & 7) == 0 &&
                 ((uintptr_t)shiftsOut & 7) == 0);
    if ((long long)pitchImg * height >= (1LL << 32)) return MFSR_E_UNSUPPORTED;  // (lk_warp_sample: 32-bit byte offsets)
    const int h = halfWindowSize;
    // TX is a template parameter of the kernel: 48 only for the half-window sizes that have a <h,48>
    // instantiation below (1..7); every other size runs the generic <0,32> kernel
    const bool hasWide = h >= 1 && h <= 7;
    static const int forceTx = [] {
        const char* e = getenv("MFSR_LK_TX");
        return e ? atoi(e) : 0;
    }();
    // tile width: 48 when the kernel warps its tile + halo itself (fewest warps per pixel); 32 when the warped image is
    // handed in (PRE): the halo is then only loads, and the smaller workgroup (40 KB of LDS, four per CU instead of two)
    // hides the two global round trips of a tile better -- 6.55 against 6.69 ms per 4K burst; 64 wide: no better.
    // MFSR_LK_TX=32|48|64 overrides (A/B).
    int TX = (hasWide && !PRE && width >= 48 + 2 * h + 4) ? 48 : 32;
    if (forceTx == 32) TX = 32;
    if (forceTx == 48 && hasWide && width >= 48 + 2 * h + 4) TX = 48;
    if (PRE && forceTx == 64 && hasWide && width >= 64 + 2 * h + 4) TX = 64;
    const int BW = TX + 2 * h + 4, BH = LK_TY + 2 * h + 4, AW = TX + 2 * h, AH = LK_TY + 2 * h;
    const int AWp = hasWide ? ((AW + 3) & ~3) : AW;  // as in the kernel: padded when h is a template parameter
    const size_t lds = sizeof(float) * ((size_t)2 * BW * BH + (size_t)5 * AWp * AH + (size_t)5 * TX * AH);
    if (lds > 160 * 1024) return MFSR_E_UNSUPPORTED;
    MFSR_REQUIRE(width >= TX + 2 * h + 4 && height >= LK_TY + 2 * h + 4);  // reflection range of the halo
    dim3 block(TX, LK_TY), grid(mfsr_cdiv(width, TX), mfsr_cdiv(height, LK_TY));
#define LK_LAUNCH(HT, TXV)                                                                                             \
    do {                                                                                                               \
        if (lds > 64 * 1024) {                                                                                         \
            static bool attr_set = false;                                                                              \
            if (!attr_set) {                                                                                           \
                MFSR_HIP_TRY(hipFuncSetAttribute((const void*)k_lkIterationFused<HT, TXV, PRE>,                        \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));             \
                attr_set = true;                                                                                       \
            }                                                                                                          \
        }                                                                                                              \
        hipLaunchKernelGGL((k_lkIterationFused<HT, TXV, PRE>), grid, block, lds, mfsr_s(stream), (const float2*)shiftsIn, \
                           (float2*)shiftsOut, pitchShift, refImg, movedImg, pitchImg, width, height, h, minDet,       \
                           outScale, sumIn, diffIn, sumOut, diffOut, pitchSD);                                         \
    } while (0)
#define LK_CASE(HT)                                                                                                    \
    case HT:                                                                                                           \
        if (PRE && TX == 64)                                                                                           \
            LK_LAUNCH(HT, 64);                                                                                         \
        else if (TX == 48)                                                                                             \
            LK_LAUNCH(HT, 48);                                                                                         \
        else                                                                                                           \
            LK_LAUNCH(HT, 32);                                                                                         \
        break;
    switch (h) {
        LK_CASE(1)
        LK_CASE(2)
        LK_CASE(3)
        LK_CASE(4)
        LK_CASE(5)
        LK_CASE(6)
        LK_CASE(7)
        default:
            if (TX != 32) return MFSR_E_UNSUPPORTED;  // unreachable: hasWide covers exactly the cases above
            LK_LAUNCH(0, 32);
            break;
    }
#undef LK_CASE
#undef LK_LAUNCH
    return mfsr_launch_status("lucasKanadeIterationFused");
}

extern "C" int mfsr_lucasKanadeIterationFused(const mfsr_float2* shiftsIn, mfsr_float2* shiftsOut, int pitchShift,
                                              const float* refImg, const float* movedImg, int pitchImg, int width,
                                              int height, int halfWindowSize, float minDet, float outScale, mfsr_stream_t stream)
{
    return lk_iteration_impl<false>(shiftsIn, shiftsOut, pitchShift, refImg, movedImg, pitchImg, width, height, halfWindowSize, minDet,
                                    outScale, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int mfsr_lucasKanadeIterationWarped(const mfsr_float2* shiftsIn, mfsr_float2* shiftsOut, int pitchShift,
                                               const float* refImg, const float* movedImg, int pitchImg, const float* sumIn,
                                               const float* diffIn, float* sumOut, float* diffOut, int pitchSD, int width, int height,
                                               int halfWindowSize, float minDet, float outScale, mfsr_stream_t stream)
{
    MFSR_REQUIRE(sumIn && diffIn && (sumOut != nullptr) == (diffOut != nullptr) && sumOut != sumIn && diffOut != diffIn);
    MFSR_REQUIRE((long long)pitchSD >= 4LL * width && (pitchSD & 3) == 0);
    MFSR_REQUIRE(!sumOut || outScale == 1.0f);  // the flow a next iteration reads is in tracking pixels
    return lk_iteration_impl<true>(shiftsIn, shiftsOut, pitchShift, refImg, movedImg, pitchImg, width, height, halfWindowSize, minDet,
                                   outScale, sumIn, diffIn, sumOut, diffOut, pitchSD, stream);
}

// D1 (CreateFlowFieldFromTiles, opticalFlow.cu:48) + the warp of every pixel under that flow: the input of the first
// mfsr_lucasKanadeIterationWarped.  Base shift / rotation by value (cosf / sinf on the device, as mfsr_CreateFlowFieldFromTiles)
// or from a device mfsr_prealign (its cos / sin table, as mfsr_CreateFlowFieldFromTilesBase): the same flow bits either way.
template <bool BASE_PTR>
__global__ void __launch_bounds__(256)
    k_flowFieldWarped(float2* __restrict__ outImg, mfsr_tex2d texShift, int imgWidth, int imgHeight, int imgPitch, float2 baseShift,
                      float baseRotation, const mfsr_prealign* __restrict__ base, const float* __restrict__ refImg,
                      const float* __restrict__ movedImg, int pitchImg, float* __restrict__ sumOut, float* __restrict__ diffOut,
                      int pitchSD, MfsrBatch bt, MfsrExactDiv dW, MfsrExactDiv dH)
{
    if (gridDim.z > 1) {
        outImg = (float2*)bt.p[blockIdx.z][0];
        texShift.ptr = bt.p[blockIdx.z][1];
        base = (const mfsr_prealign*)bt.p[blockIdx.z][2];
        movedImg = (const float*)bt.p[blockIdx.z][3];
        sumOut = (float*)bt.p[blockIdx.z][4];
        diffOut = (float*)bt.p[blockIdx.z][5];
    }
    const int pxX = blockIdx.x * blockDim.x + threadIdx.x;
    const int pxY = blockIdx.y * blockDim.y + threadIdx.y;
    if (pxX >= imgWidth || pxY >= imgHeight) return;
    float cf = 1.0f, sf = 0.0f, bx, by;
    if (BASE_PTR) {
        cf = base->cosRotation;
        sf = base->sinRotation;
        bx = base->shiftX;
        by = base->shiftY;
    } else {
        // cosf(0) = 1 and sinf(0) = 0 exactly: the (uniform) zero-rotation case skips the two libm expansions
        if (baseRotation != 0.0f) {
            cf = cosf(baseRotation);
            sf = sinf(baseRotation);
        }
        bx = baseShift.x;
        by = baseShift.y;
    }
    float2 shift;
    shift.x = cf * -bx - sf * -by;
    shift.y = sf * -bx + cf * -by;
    const float patchCenterX = (float)(pxX - imgWidth / 2);
    const float patchCenterY = (float)(pxY - imgHeight / 2);
    shift.x += cf * patchCenterX - sf * patchCenterY - patchCenterX;
    shift.y += sf * patchCenterX + cf * patchCenterY - patchCenterY;
    const float2 shiftPatch =
        tex2<ADDR_CLAMP>(texShift, mfsr_div((float)pxX + 0.5f, dW), mfsr_div((float)pxY + 0.5f, dH));   // / imgWidth, / imgHeight
    shift.x += shiftPatch.x;
    shift.y += shiftPatch.y;
    *lk_at(outImg, imgPitch, pxY, pxX) = shift;
    float wv, rv;
    lk_warp_sample(refImg, movedImg, pitchImg, imgWidth, imgHeight, pxX, pxY, shift, wv, rv, &dW, &dH);
    *lk_at(sumOut, pitchSD, pxY, pxX) = wv + rv;
    *lk_at(diffOut, pitchSD, pxY, pxX) = wv - rv;
}

static int flow_field_warped_impl(int n, const mfsr_flowfield_frame* f, int tilePitch, int tileW, int tileH, int imgWidth, int imgHeight,
                                  int imgPitch, mfsr_float2 baseShift, float baseRotation, const float* refImg, int pitchImg, int pitchSD,
                                  mfsr_stream_t stream)
{
    MFSR_REQUIRE(f && n >= 1 && n <= MFSR_BATCH_MAX && refImg && imgWidth >= 2 && imgHeight >= 2);
    MFSR_REQUIRE((long long)imgPitch >= 8LL * imgWidth && (imgPitch & 7) == 0);
    MFSR_REQUIRE((long long)pitchImg >= 4LL * imgWidth && (pitchImg & 3) == 0 && (long long)pitchSD >= 4LL * imgWidth && (pitchSD & 3) == 0);
    // (32-bit byte offsets inside every image, lk_at: images of 4 GB and more are not this kernel's)
    if ((long long)imgPitch * imgHeight >= (1LL << 32) || (long long)pitchImg * imgHeight >= (1LL << 32) ||
        (long long)pitchSD * imgHeight >= (1LL << 32))
        return MFSR_E_UNSUPPORTED;
    mfsr_tex2d tex;
    tex.ptr = f[0].tileShifts;
    tex.pitch = tilePitch;
    tex.width = tileW;
    tex.height = tileH;
    MFSR_REQUIRE(mfsr_tex_ok(tex, 8) && (tilePitch & 7) == 0);
    MfsrBatch bt;
    memset(&bt, 0, sizeof(bt));
    for (int i = 0; i < n; i++) {
        MFSR_REQUIRE(f[i].outImg && f[i].tileShifts && f[i].movedImg && f[i].sumOut && f[i].diffOut);
        MFSR_REQUIRE(((uintptr_t)f[i].outImg & 7) == 0 && ((uintptr_t)f[i].tileShifts & 7) == 0);
        MFSR_REQUIRE((f[i].base != nullptr) == (f[0].base != nullptr));
        bt.p[i][0] = f[i].outImg;
        bt.p[i][1] = f[i].tileShifts;
        bt.p[i][2] = f[i].base;
        bt.p[i][3] = f[i].movedImg;
        bt.p[i][4] = f[i].sumOut;
        bt.p[i][5] = f[i].diffOut;
    }
    dim3 block(64, 4), grid(mfsr_cdiv(imgWidth, 64), mfsr_cdiv(imgHeight, 4), n);
    const MfsrExactDiv dW = mfsr_exact_div((float)imgWidth), dH = mfsr_exact_div((float)imgHeight);
    if (f[0].base)
        hipLaunchKernelGGL(k_flowFieldWarped<true>, grid, block, 0, mfsr_s(stream), (float2*)f[0].outImg, tex, imgWidth, imgHeight, imgPitch,
                           make_float2(0.0f, 0.0f), 0.0f, f[0].base, refImg, f[0].movedImg, pitchImg, f[0].sumOut, f[0].diffOut, pitchSD, bt, dW, dH);
    else
        hipLaunchKernelGGL(k_flowFieldWarped<false>, grid, block, 0, mfsr_s(stream), (float2*)f[0].outImg, tex, imgWidth, imgHeight, imgPitch,
                           make_float2(baseShift.x, baseShift.y), baseRotation, f[0].base, refImg, f[0].movedImg, pitchImg, f[0].sumOut,
                           f[0].diffOut, pitchSD, bt, dW, dH);
    return mfsr_launch_status("CreateFlowFieldWarped");
}

extern "C" int mfsr_CreateFlowFieldWarped(mfsr_float2* outImg, mfsr_tex2d texObjShiftXY, int imgWidth, int imgHeight, int imgPitch,
                                          mfsr_float2 baseShift, float baseRotation, const mfsr_prealign* base, const float* refImg,
                                          const float* movedImg, int pitchImg, float* sumOut, float* diffOut, int pitchSD,
                                          mfsr_stream_t stream)
{
    const mfsr_flowfield_frame f = {outImg, (const mfsr_float2*)texObjShiftXY.ptr, base, movedImg, sumOut, diffOut};
    return flow_field_warped_impl(1, &f, texObjShiftXY.pitch, texObjShiftXY.width, texObjShiftXY.height, imgWidth, imgHeight, imgPitch,
                                  baseShift, baseRotation, refImg, pitchImg, pitchSD, stream);
}

extern "C" int mfsr_CreateFlowFieldWarpedBatch(int nFrames, const mfsr_flowfield_frame* frames, int tilePitch, int tileCountX,
                                               int tileCountY, int imgWidth, int imgHeight, int imgPitch, const float* refImg,
                                               int pitchImg, int pitchSD, mfsr_stream_t stream)
{
    const mfsr_float2 z = {0.0f, 0.0f};
    return flow_field_warped_impl(nFrames, frames, tilePitch, tileCountX, tileCountY, imgWidth, imgHeight, imgPitch, z, 0.0f, refImg,
                                  pitchImg, pitchSD, stream);
}

// ---- the same iteration as a register / DPP column sweep, for up to MFSR_LK_MAX_BATCH frames per launch -----------------------
// k_lkIterationFused spends most of its ~430 VALU instructions per pixel around the arithmetic: a 32 x 16 tile stages
// (32+10) x (16+10) samples (2.1x), forms the products on (32+6) x (16+6) (1.6x), moves everything through LDS with
// index arithmetic for every element and crosses three barriers.  Here ONE WAVEFRONT owns 64 image columns and sweeps
// down a band of rows, one image row per step, with no LDS and no barrier:
//   * vertical state lives in registers of the lane that owns the column: the last five rows of (warped + ref) for the
//     y derivative, the last 2h+1 rows of the five row-summed products (a ring the row loop is unrolled over, so the ring
//     index is a compile-time register name);
//   * horizontal neighbours come through the GFX9 whole-wave DPP shifts: Ix from wave_shl:1 / wave_shr:1 moves, the
//     2h+1-wide row sums Horner-style -- acc = v + wave_shr:1(acc), 2h times: ONE v_add_f32_dpp per step, so lane l ends
//     with the sum of lanes l-2h .. l, i.e. the window centred on column l - h, which is the pixel that lane then owns
//     for the solve, the flow update and the warp that makes the next iteration's input (tools/ubench/dpp_wave_shift.hip
//     checks the shifts on the hardware: 2.1 ns per wave-instruction, against 1.2 for a plain add);
//   * of 64 lanes 64 - 2(h+2) produce a pixel (84 % at h = 3); a band of R rows reads R + 2(h+2) rows.
// Same products, same column-sum order (top to bottom) and the same solve / update / warp code as k_lkIterationFused<PRE>;
// the row sums add right to left instead of in the four-neighbour tree, so the flow agrees with it to fp32 rounding
// (<= 1e-5 px on the tests), like every other summation order of this stage (header of this file).
#define MFSR_LK_SWEEP_MAX_BATCH 4
struct LkSweepFrame {
    const float2* flowIn;
    float2* flowOut;
    const float* moved;
    const float* sumIn;
    const float* diffIn;
    float* sumOut;   // nullptr on the last iteration
    float* diffOut;
};
struct LkSweepBatch {
    LkSweepFrame f[MFSR_LK_SWEEP_MAX_BATCH];
};

__device__ __forceinline__ float lk_wshr1(float v)  // lane l <- lane l-1 (lane 0 <- 0)
{
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lk_wshl1(float v)  // lane l <- lane l+1 (lane 63 <- 0)
{
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x130, 0xf, 0xf, true));
}

// the warp of lk_warp_sample in two phases, so that the gather a pixel issues after its flow update is consumed one row
// later (the wave works on the next row meanwhile): same address arithmetic, same blend, same bits
// Register diet of k_lkSweep (round 4; all bit-identical, profiles/r04_lk_regs_ab.txt; h = 3): 32-bit offsets (lk_at) 95 -> 85
// VGPRs and -3.6 % instructions: 110.3 -> 107 us per 4-frame launch; the border fetch at issue time (no coordinates / flag in
// the pending gather) and the difference image loaded two steps before its use instead of four (LK_DIFF_LATE): 79 VGPRs = six
// waves per SIMD: 106 us.  LK_RV_REREAD=1 (the reference pixel re-read at the finish instead of held in a register: 76 VGPRs)
// puts a load in front of the two stores of every row: 120 us -- off.
#ifndef LK_RV_REREAD
#define LK_RV_REREAD 0
#endif
#ifndef LK_DIFF_LATE
#define LK_DIFF_LATE 1
#endif
// What a pending warp keeps across a row step: the four texels on their way, the blend weights and the reference pixel -- seven
// registers (LkGather + rv).  A sample the plain fetch does not cover (outside the unit square or on the last texel row / column: waves at the
// image border only) is fetched with the full MIRROR addressing right here, under a wave-uniform branch, and parked as four
// equal texels with zero weights: lerp4 of those returns it bit for bit ((t + 0) + 0 + 0), so the finish needs neither the
// coordinates nor an "interior" flag.
struct LkGather {
    float t00, t10, t01, t11, a, b;
#if !LK_RV_REREAD
    float rv;
#endif
};
__device__ __forceinline__ void lk_warp_issue(const float* __restrict__ movedImg, int pitchImg, int width, int height, int gx, int gy,
                                              float2 f, bool wanted, LkGather& g, const MfsrExactDiv& dW, const MfsrExactDiv& dH)
{
    const float u = mfsr_div((float)gx + 0.5f + f.x, dW);   // opticalFlow.cu:38-39: / (float)width
    const float v = mfsr_div((float)gy + 0.5f + f.y, dH);   //                        / (float)height
    const float xB = u * (float)width - 0.5f, yB = v * (float)height - 0.5f;
    const float fx = floorf(xB), fy = floorf(yB);
    const int ix = f2i(fx), iy = f2i(fy);
    const bool interior = u >= 0.0f && u < 1.0f && v >= 0.0f && v < 1.0f && (uint32_t)ix <= (uint32_t)(width - 2) &&
                          (uint32_t)iy <= (uint32_t)(height - 2);
    const int ixc = clampi(ix, 0, width - 2), iyc = clampi(iy, 0, height - 2);
    const float* r0 = lk_at(movedImg, pitchImg, iyc, ixc);
    const float* r1 = lk_at(movedImg, pitchImg, iyc + 1, ixc);
    g.t00 = r0[0], g.t10 = r0[1], g.t01 = r1[0], g.t11 = r1[1];
    g.a = xB - fx, g.b = yB - fy;
    if (__ballot(!interior && wanted) != 0) {   // (lanes whose result is not stored do not send the wave here)
        mfsr_tex2d texMoved;
        texMoved.ptr = movedImg;
        texMoved.pitch = pitchImg;
        texMoved.width = width;
        texMoved.height = height;
        const float full = tex1<ADDR_MIRROR>(texMoved, u, v);
        if (!interior) {
            g.t00 = g.t10 = g.t01 = g.t11 = full;
            g.a = g.b = 0.0f;
        }
    }
}
__device__ __forceinline__ float lk_warp_finish(const LkGather& g) { return lerp4(g.t00, g.t10, g.t01, g.t11, g.a, g.b); }

template <int HT>
__global__ void __launch_bounds__(64)
    k_lkSweep(LkSweepBatch batch, const float* __restrict__ refImg, int pitchShift, int pitchImg, int pitchSD, int width, int height,
              float minDet, float outScale, int bandRows, int strips, int bands, int perXcd, int nFramesLaunch, MfsrExactDiv dW, MfsrExactDiv dH)
{
    constexpr int h = HT, HALO = HT + 2, WIN = 2 * HT + 1, VW = 64 - 2 * HALO;
    // XCD-aware order (perXcd > 0): workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  A band re-reads
    // 2 (h + 2) rows of its vertical neighbours and a strip 2 (h + 2) columns of its horizontal ones; in launch order the
    // band below sits `strips` workgroups later, on another XCD.  Here XCD k walks its own contiguous run of (frame, band,
    // strip) tiles, so those halos are hits in its L2 instead of trips to the fabric.
    int sIdx, bIdx, fIdx;
    if (perXcd > 0) {
        const int t = ((int)blockIdx.x & 7) * perXcd + ((int)blockIdx.x >> 3);
        if (t >= strips * bands * nFramesLaunch) return;
        sIdx = t % strips;
        bIdx = (t / strips) % bands;
        fIdx = t / (strips * bands);
    } else {
        sIdx = blockIdx.x, bIdx = blockIdx.y, fIdx = blockIdx.z;
    }
    const LkSweepFrame F = batch.f[fIdx];
    const int lane = threadIdx.x;
    const int cx0 = sIdx * VW, ry0 = bIdx * bandRows;
    const int colIn = cx0 - HALO + lane;              // the column this lane loads and differentiates
    const int gxIn = lk_mirror_index(colIn, width);   // (-width <= colIn < 2 width: width >= 64 is required)
    const int colOut = colIn - h;                     // the column whose window sum the Horner shifts leave in this lane
    const bool outLane = lane >= 2 * h + 2 && lane <= 61 && colOut < width;
    const int colOutC = min(max(colOut, 0), width - 1);
    const bool ringCol = colOut < h || colOut >= width - h;
    const int rows = min(bandRows, height - ry0);
    const int T = rows + 2 * HALO;                    // input rows ry0 - HALO .. ry0 + rows - 1 + HALO

    float s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;     // (warped + ref) of rows r-4 .. r
    float d0 = 0;                                     // (warped - ref) of row r-2: loaded two steps before it is used, not four
    float H[5][WIN];                                  // row sums of the five products, ring over rows
#pragma unroll
    for (int k = 0; k < 5; k++)
#pragma unroll
        for (int j = 0; j < WIN; j++) H[k][j] = 0.0f;

    auto load_sum = [&](int t, float& sv) { sv = *lk_at(F.sumIn, pitchSD, lk_mirror_index(ry0 - HALO + t, height), gxIn); };
    auto load_diff = [&](int t, float& dv) { dv = *lk_at(F.diffIn, pitchSD, lk_mirror_index(ry0 - HALO + t, height), gxIn); };
    // two rows in flight ahead of the one being worked on: the sum image's rows t + 1, t + 2 (the y derivative of row t - 2 needs
    // rows up to t), the difference image's rows t - 1, t (It of row t - 2 is consumed two steps after its load)
    float sA, sB, dA = 0, dB = 0;
    load_sum(0, sA);
    load_sum(min(1, T - 1), sB);
#if !LK_DIFF_LATE
    float d1 = 0, d2 = 0;
    load_diff(0, dA);
    load_diff(min(1, T - 1), dB);
#endif
    LkGather pend;            // the previous output row's warp, issued and not yet consumed
    int pendY = -1;
    auto finish_pending = [&]() {
        if (pendY < 0) return;   // wave-uniform
        const float wv = lk_warp_finish(pend);
#if LK_RV_REREAD
        const float rv = *lk_at(refImg, pitchImg, pendY, colOutC);   // (A/B: a load in front of the row's stores -- slower)
#else
        const float rv = pend.rv;
#endif
        if (outLane) {
            *lk_at(F.sumOut, pitchSD, pendY, colOutC) = wv + rv;
            *lk_at(F.diffOut, pitchSD, pendY, colOutC) = wv - rv;
        }
    };

    for (int t0 = 0; t0 < T; t0 += WIN) {
#pragma unroll
        for (int j = 0; j < WIN; j++) {
            const int t = t0 + j;
            if (t >= T) break;
            // rotate the raw-row rings; fetch row t + 2
            s0 = s1, s1 = s2, s2 = s3, s3 = s4, s4 = sA;
#if LK_DIFF_LATE
            d0 = dA, dA = dB;
            sA = sB;
            load_sum(min(t + 2, T - 1), sB);
            load_diff(t, dB);
#else
            d0 = d1, d1 = d2, d2 = dA, dA = dB;
            sA = sB;
            load_sum(min(t + 2, T - 1), sB);
            load_diff(min(t + 2, T - 1), dB);
#endif
            if (t < 4) continue;   // the derivative of row r - 2 needs rows r - 4 .. r
            // this row's own flow and reference pixel: on their way while the sums are formed (every lane loads: clamped column)
            const int y = ry0 + t - 2 * HALO;                  // the row whose window this step completes (if t >= 2 HALO)
            const int yc = min(max(y, 0), height - 1);
            float2 shift = *lk_at(F.flowIn, pitchShift, yc, colOutC);
#if !LK_RV_REREAD
            const float rvOwn = *lk_at(refImg, pitchImg, yc, colOutC);
#endif
            // products of image row r - 2 (opticalFlow.cu:116-131 on the sum image, as k_lkIterationFused step 2)
            const float xp1 = lk_wshl1(s2), xp2 = lk_wshl1(xp1), xm1 = lk_wshr1(s2), xm2 = lk_wshr1(xm1);
            float tx = xp2;
            tx -= xp1 * 8.0f;
            tx += xm1 * 8.0f;
            tx -= xm2;
            const float Ix = tx * (1.0f / 24.0f);
            float ty = s4;
            ty -= s3 * 8.0f;
            ty += s1 * 8.0f;
            ty -= s0;
            const float Iy = ty * (1.0f / 24.0f);
            const float It = d0;
            float P[5] = {Ix * Ix, Ix * Iy, Iy * Iy, Ix * It, Iy * It};
            // row sums: lane l <- lanes l - 2h .. l
#pragma unroll
            for (int k = 0; k < 5; k++) {
                float acc = P[k];
#pragma unroll
                for (int q = 0; q < 2 * HT; q++) acc = P[k] + lk_wshr1(acc);
                H[k][j] = acc;
            }
            if (t < 2 * HALO) continue;   // the ring holds fewer than 2h+1 rows of this band yet
            // column sums, top row first: the ring's oldest entry is slot j + 1
            float V[5];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                float sum = 0;
#pragma unroll
                for (int d = 1; d <= WIN; d++) sum += H[k][(j + d) % WIN];
                V[k] = sum;
            }
            if (!(ringCol || y < h || y >= height - h)) {
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
            if (F.sumOut) {   // wave-uniform
                finish_pending();   // the previous row's gather: issued one row ago
                lk_warp_issue(F.moved, pitchImg, width, height, colOutC, y, shift, outLane, pend, dW, dH);
#if !LK_RV_REREAD
                pend.rv = rvOwn;
#endif
                pendY = y;
            }
            shift.x *= outScale;
            shift.y *= outScale;
            if (outLane) *lk_at(F.flowOut, pitchShift, y, colOutC) = shift;
        }
    }
    if (F.sumOut) finish_pending();
}

extern "C" int mfsr_lucasKanadeSweepBatch(int nFrames, const mfsr_lk_frame* frames, const float* refImg, int pitchShift, int pitchImg,
                                          int pitchSD, int width, int height, int halfWindowSize, float minDet, float outScale,
                                          mfsr_stream_t stream)
{
    MFSR_REQUIRE(frames && refImg && nFrames >= 1 && nFrames <= MFSR_LK_SWEEP_MAX_BATCH);
    if (halfWindowSize < 1 || halfWindowSize > 7 || width < 64 || height < 2 * (halfWindowSize + 2)) return MFSR_E_UNSUPPORTED;
    MFSR_REQUIRE((long long)pitchShift >= 8LL * width && (pitchShift & 7) == 0);
    MFSR_REQUIRE((long long)pitchImg >= 4LL * width && (pitchImg & 3) == 0 && (long long)pitchSD >= 4LL * width && (pitchSD & 3) == 0);
    // (32-bit byte offsets inside every image, lk_at)
    if ((long long)pitchShift * height >= (1LL << 32) || (long long)pitchImg * height >= (1LL << 32) || (long long)pitchSD * height >= (1LL << 32))
        return MFSR_E_UNSUPPORTED;
    LkSweepBatch b;
    memset(&b, 0, sizeof(b));
    for (int i = 0; i < nFrames; i++) {
        const mfsr_lk_frame& f = frames[i];
        MFSR_REQUIRE(f.shiftsIn && f.shiftsOut && f.shiftsIn != f.shiftsOut && f.movedImg && f.sumIn && f.diffIn);
        MFSR_REQUIRE(((uintptr_t)f.shiftsIn & 7) == 0 && ((uintptr_t)f.shiftsOut & 7) == 0);
        MFSR_REQUIRE((f.sumOut != nullptr) == (f.diffOut != nullptr) && f.sumOut != f.sumIn && f.diffOut != f.diffIn);
        MFSR_REQUIRE(!f.sumOut || outScale == 1.0f);  // the flow a next iteration reads is in tracking pixels
        b.f[i] = LkSweepFrame{(const float2*)f.shiftsIn, (float2*)f.shiftsOut, f.movedImg, f.sumIn, f.diffIn, f.sumOut, f.diffOut};
    }
    const int h = halfWindowSize, VW = 64 - 2 * (h + 2);
    const int strips = mfsr_cdiv(width, VW);
    // band height: enough wavefronts to fill 1024 SIMDs a few times over without paying too many halo rows
    static const int forceBand = [] {
        const char* e = getenv("MFSR_LK_BAND");
        return e ? atoi(e) : 0;
    }();
    // (measured at 1920 x 1080, h = 3: one frame 27.0 / 32.5 / 48.8 us with bands of 8 / 16 / 32 rows, four frames 27.1 / 26.5 /
    // 27.6 us per frame: the wave count decides, the halo rows cost less than idle SIMDs)
    int band = (int)((long long)height * strips * nFrames / 8192);
    band = (band + 4) & ~7;
    band = band < 8 ? 8 : (band > 32 ? 32 : band);   // (launches of several rounds: 12 .. 52 rows measure alike at 8K, 64 is 1 % slower)
    {
        // Round 4: the kernel holds 49 + 10 h registers, i.e. 8 / 7 / 6 / 5 / 4 / 4 / 4 waves per SIMD for h = 1 .. 7.  When all
        // waves of the launch can be resident at once with bands of at most 64 rows, take the lowest such band: one round
        // instead of a full one and a partly filled one, and fewer halo rows (1080 rows x 36 strips x 4 frames, h = 3: 26-row
        // bands = 6048 waves for 6144 places: 104.5 against 107 us with 16-row bands; 27 .. 30: 105 .. 111 us,
        // profiles/r04_lk_band_rule_ab.txt).  Larger launches keep the rule above.
        static const int occ[8] = {0, 8, 7, 6, 5, 4, 4, 4};
        const long long places = 1024LL * occ[h];
        const long long maxBands = places / ((long long)strips * nFrames);
        if (maxBands >= 1) {
            int b = (int)((height + maxBands - 1) / maxBands);   // lowest band with ceil(height / band) <= maxBands
            b = b < 8 ? 8 : b;
            if (b <= 64) band = b;
        }
    }
    if (forceBand >= 8) band = forceBand;
    static const int xcdRemap = [] {
        const char* e = getenv("MFSR_LK_XCD");
        return e ? atoi(e) : 1;
    }();
    const int bands = mfsr_cdiv(height, band);
    const int total = strips * bands * nFrames;
    const int perXcd = xcdRemap ? mfsr_cdiv(total, 8) : 0;
    // (remapped: a 1-D grid of 8 * perXcd workgroups whose .z still carries the frame count for the kernel's bound check)
    dim3 grid = perXcd ? dim3(8 * perXcd, 1, 1) : dim3(strips, bands, nFrames), block(64);
    const MfsrExactDiv dW = mfsr_exact_div((float)width), dH = mfsr_exact_div((float)height);
#define LKS_CASE(HT)                                                                                                          \
    case HT:                                                                                                                  \
        hipLaunchKernelGGL(k_lkSweep<HT>, grid, block, 0, mfsr_s(stream), b, refImg, pitchShift, pitchImg, pitchSD, width,    \
                           height, minDet, outScale, band, strips, bands, perXcd, nFrames, dW, dH);                           \
        break;
    switch (h) {
        LKS_CASE(1)
        LKS_CASE(2)
        LKS_CASE(3)
        LKS_CASE(4)
        LKS_CASE(5)
        LKS_CASE(6)
        LKS_CASE(7)
    }
#undef LKS_CASE
    return mfsr_launch_status("lucasKanadeSweepBatch");
}
