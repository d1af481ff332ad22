// debayer.hip -- DeBayer stages of the hot path.
// Behavioural spec: reference test_opencv/DeBayerKernels.cu:28-283.
// These kernels use only + - * / fabs, so with -ffp-contract=off they are
// bit-identical to the CPU oracle.
#include <cstring>
#include "common.hpp"

// ---- A0: c_cfaPattern (DeBayerKernels.cu:40-41) -------------------------------
// Process-wide like the reference's module constant, but passed to every kernel
// as a packed launch argument so that concurrent streams never race on a
// __constant__ symbol.
static int g_cfa[4] = {MFSR_RED, MFSR_GREEN, MFSR_GREEN, MFSR_BLUE};

int mfsr_cfa_packed() { return (g_cfa[0] & 0xff) | ((g_cfa[1] & 0xff) << 8) | ((g_cfa[2] & 0xff) << 16) | ((g_cfa[3] & 0xff) << 24); }

extern "C" int mfsr_set_cfa_pattern(const int32_t pattern[4])
{
    MFSR_REQUIRE(pattern != nullptr);
    for (int i = 0; i < 4; i++) MFSR_REQUIRE(pattern[i] >= 0 && pattern[i] <= MFSR_WHITE);
    for (int i = 0; i < 4; i++) g_cfa[i] = pattern[i];
    return MFSR_OK;
}

extern "C" int mfsr_get_cfa_pattern(int32_t pattern[4])
{
    MFSR_REQUIRE(pattern != nullptr);
    for (int i = 0; i < 4; i++) pattern[i] = g_cfa[i];
    return MFSR_OK;
}

// ---- A1: deBayersSubSample3 (DeBayerKernels.cu:244-283) -----------------------
// One thread per half-res pixel; each reads its 2x2 quad as two 4-byte loads
// (consecutive lanes -> consecutive 4 B: fully coalesced rows) and writes 12 B.
__global__ void __launch_bounds__(256) k_deBayersSubSample3(const uint16_t* __restrict__ dataIn, pix3* __restrict__ imgOut,
                                                           float maxVal, int dimX, int dimY, int strideOut, int cfa)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= dimX || y >= dimY) return;
    const size_t rowElems = (size_t)dimX * 2;
    const uint32_t top = *(const uint32_t*)(dataIn + (size_t)(2 * y) * rowElems + 2 * x);
    const uint32_t bot = *(const uint32_t*)(dataIn + (size_t)(2 * y + 1) * rowElems + 2 * x);
    // raw[iy][ix]
    const float raw[2][2] = {{(float)(top & 0xffffu), (float)(top >> 16)}, {(float)(bot & 0xffffu), (float)(bot >> 16)}};
    const float factor = 1.0f / maxVal;
    pix3 pixel = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int ix = 0; ix < 2; ix++) {
#pragma unroll
        for (int iy = 0; iy < 2; iy++) {
            const int c = cfa_at(cfa, iy, ix);
            const float v = raw[iy][ix] * factor;
            if (c == MFSR_GREEN)
                pixel.y += v * 0.5f;
            else if (c == MFSR_RED)
                pixel.x = v;
            else if (c == MFSR_BLUE)
                pixel.z = v;
        }
    }
    row_ptr(imgOut, strideOut, y)[x] = pixel;
}

extern "C" int mfsr_deBayersSubSample3(const uint16_t* dataIn, mfsr_float3* imgOut, float maxVal, int dimX, int dimY,
                                       int strideOut, mfsr_stream_t stream)
{
    MFSR_REQUIRE(dataIn && imgOut && dimX > 0 && dimY > 0);
    MFSR_REQUIRE((long long)strideOut >= 12LL * dimX && (strideOut & 3) == 0);
    MFSR_REQUIRE(((uintptr_t)dataIn & 3) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(dimX, 64), mfsr_cdiv(dimY, 4));
    hipLaunchKernelGGL(k_deBayersSubSample3, grid, block, 0, mfsr_s(stream), dataIn, (pix3*)imgOut, maxVal, dimX, dimY,
                       strideOut, mfsr_cfa_packed());
    return mfsr_launch_status("deBayersSubSample3");
}

// ---- A1 + gray + separable prefilter + first pyramid level in one launch ------------------
// What the burst driver does to every frame before tracking: deBayersSubSample3 (A1), luma of the
// half-res RGB, the separable Gaussian prefilter (gaussin_filter_1D taps, clamped borders) and the
// 2x2 mean of the first pyramid level -- five launches and 95 MB of intermediate traffic per 4K frame.
// Here a 64 x 16 tile of the tracking image + a (taps/2)-pixel halo is built in LDS straight from the
// raw quads; every value is computed by the very expressions of the five kernels, in their order
// (halo samples are the samples at the clamped coordinates, which is what the filters read), so the
// three outputs are bit-identical to the chain.
#define PREP_TX 64
#define PREP_TY 16
#define PREP_MAXC0 8
struct PrepTaps {
    float t[2 * PREP_MAXC0 + 1];
    int n;
};

__device__ __forceinline__ pix3 a1_pixel(const uint16_t* __restrict__ dataIn, int dimX, int x, int y, float factor, int cfa)
{
    const size_t rowElems = (size_t)dimX * 2;
    const uint32_t top = *(const uint32_t*)(dataIn + (size_t)(2 * y) * rowElems + 2 * x);
    const uint32_t bot = *(const uint32_t*)(dataIn + (size_t)(2 * y + 1) * rowElems + 2 * x);
    const float raw[2][2] = {{(float)(top & 0xffffu), (float)(top >> 16)}, {(float)(bot & 0xffffu), (float)(bot >> 16)}};
    pix3 pixel = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int ix = 0; ix < 2; ix++) {
#pragma unroll
        for (int iy = 0; iy < 2; iy++) {
            const int c = cfa_at(cfa, iy, ix);
            const float v = raw[iy][ix] * factor;
            if (c == MFSR_GREEN)
                pixel.y += v * 0.5f;
            else if (c == MFSR_RED)
                pixel.x = v;
            else if (c == MFSR_BLUE)
                pixel.z = v;
        }
    }
    return pixel;
}

__global__ void __launch_bounds__(256)
    k_prepareFrameFused(const uint16_t* __restrict__ dataIn, pix3* __restrict__ halfOut, int halfPitch, float maxVal,
                        int dimX, int dimY, float* __restrict__ pyr0, int pyr0Pitch, float* __restrict__ pyr1,
                        int pyr1Pitch, PrepTaps taps, int cfa, MfsrBatch bt)
{
    if (gridDim.z > 1) {
        dataIn = (const uint16_t*)bt.p[blockIdx.z][0];
        halfOut = (pix3*)bt.p[blockIdx.z][1];
        pyr0 = (float*)bt.p[blockIdx.z][2];
        pyr1 = (float*)bt.p[blockIdx.z][3];
    }
    extern __shared__ __attribute__((aligned(16))) float s_prep[];
    const int c0 = taps.n / 2;
    const int GW = PREP_TX + 2 * c0, GH = PREP_TY + 2 * c0;
    float* sG = s_prep;              // GH x GW luma of tile + halo
    float* sH = sG + GW * GH;        // GH x PREP_TX after the row pass
    float* sO = sH + PREP_TX * GH;   // PREP_TY x PREP_TX filtered tile
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int x0 = blockIdx.x * PREP_TX, y0 = blockIdx.y * PREP_TY;
    const float factor = 1.0f / maxVal;

    for (int i = tid; i < GW * GH; i += 256) {
        const int ly = i / GW, lx = i - ly * GW;
        const int gx = x0 + lx - c0, gy = y0 + ly - c0;
        const int cx = clampi(gx, 0, dimX - 1), cy = clampi(gy, 0, dimY - 1);
        const pix3 p = a1_pixel(dataIn, dimX, cx, cy, factor, cfa);
        if (gx == cx && gy == cy && lx >= c0 && lx < c0 + PREP_TX && ly >= c0 && ly < c0 + PREP_TY) row_ptr(halfOut, halfPitch, gy)[gx] = p;
        sG[i] = 0.299f * p.x + 0.587f * p.y + 0.114f * p.z;  // k_rgbToGray
    }
    __syncthreads();
    for (int i = tid; i < PREP_TX * GH; i += 256) {
        const int ly = i / PREP_TX, lx = i - ly * PREP_TX;
        // k_filter1d<true> at column x0+lx of (clamped) row: taps read clamp(x + t - c0); the halo already
        // holds the clamped-coordinate samples, except that a column beyond the image must itself behave
        // like the clamped column -- those outputs are never used (masked at the store)
        float s = 0;
        for (int t = 0; t < taps.n; t++) s += taps.t[t] * sG[ly * GW + lx + t];
        sH[i] = s;
    }
    __syncthreads();
    {
        const int lx = threadIdx.x;
#pragma unroll
        for (int r = 0; r < PREP_TY / 4; r++) {
            const int ly = threadIdx.y + 4 * r;
            float s = 0;
            for (int t = 0; t < taps.n; t++) s += taps.t[t] * sH[(ly + t) * PREP_TX + lx];
            sO[ly * PREP_TX + lx] = s;
            const int gx = x0 + lx, gy = y0 + ly;
            if (gx < dimX && gy < dimY) row_ptr(pyr0, pyr0Pitch, gy)[gx] = s;
        }
    }
    if (pyr1 == nullptr) return;
    __syncthreads();
    {
        // k_downsample2x on the filtered tile: 32 x 8 outputs
        const int ox = tid & 31, oy = tid >> 5;  // 256 threads = 32 x 8
        const int gx = x0 / 2 + ox, gy = y0 / 2 + oy;
        if (gx < dimX / 2 && gy < dimY / 2) {
            const float ax = sO[(2 * oy) * PREP_TX + 2 * ox], ay = sO[(2 * oy) * PREP_TX + 2 * ox + 1];
            const float bx = sO[(2 * oy + 1) * PREP_TX + 2 * ox], by = sO[(2 * oy + 1) * PREP_TX + 2 * ox + 1];
            row_ptr(pyr1, pyr1Pitch, gy)[gx] = ((ax + ay) + (bx + by)) * 0.25f;
        }
    }
}

static int prepare_frames_impl(int n, const mfsr_prepare_frame* f, int halfPitch, float maxVal, int dimX, int dimY, int pyr0Pitch,
                               int pyr1Pitch, const float* taps, int ntaps, mfsr_stream_t stream)
{
    MFSR_REQUIRE(f && n >= 1 && n <= MFSR_BATCH_MAX && taps && dimX > 0 && dimY > 0);
    MFSR_REQUIRE((long long)halfPitch >= 12LL * dimX && (halfPitch & 3) == 0);
    MFSR_REQUIRE((long long)pyr0Pitch >= 4LL * dimX && (pyr0Pitch & 3) == 0);
    MFSR_REQUIRE(ntaps > 0 && (ntaps & 1) == 1);
    if (ntaps / 2 > PREP_MAXC0) return MFSR_E_UNSUPPORTED;
    MfsrBatch bt;
    memset(&bt, 0, sizeof(bt));
    for (int i = 0; i < n; i++) {
        MFSR_REQUIRE(f[i].dataIn && f[i].halfOut && f[i].pyr0 && ((uintptr_t)f[i].dataIn & 3) == 0);
        MFSR_REQUIRE((f[i].pyr1 != nullptr) == (f[0].pyr1 != nullptr));
        bt.p[i][0] = f[i].dataIn;
        bt.p[i][1] = f[i].halfOut;
        bt.p[i][2] = f[i].pyr0;
        bt.p[i][3] = f[i].pyr1;
    }
    if (f[0].pyr1) MFSR_REQUIRE((long long)pyr1Pitch >= 4LL * (dimX / 2) && (pyr1Pitch & 3) == 0 && dimX >= 2 && dimY >= 2);
    PrepTaps tp;
    tp.n = ntaps;
    for (int i = 0; i < ntaps; i++) tp.t[i] = taps[i];
    const int c0 = ntaps / 2, GW = PREP_TX + 2 * c0, GH = PREP_TY + 2 * c0;
    const size_t lds = sizeof(float) * ((size_t)GW * GH + (size_t)PREP_TX * GH + (size_t)PREP_TX * PREP_TY);
    dim3 block(64, 4), grid(mfsr_cdiv(dimX, PREP_TX), mfsr_cdiv(dimY, PREP_TY), n);
    hipLaunchKernelGGL(k_prepareFrameFused, grid, block, lds, mfsr_s(stream), f[0].dataIn, (pix3*)f[0].halfOut, halfPitch, maxVal, dimX,
                       dimY, f[0].pyr0, pyr0Pitch, f[0].pyr1, pyr1Pitch, tp, mfsr_cfa_packed(), bt);
    return mfsr_launch_status("prepareFrameFused");
}

extern "C" int mfsr_prepareFrameFused(const uint16_t* dataIn, mfsr_float3* halfOut, int halfPitch, float maxVal, int dimX,
                                      int dimY, float* pyr0, int pyr0Pitch, float* pyr1, int pyr1Pitch, const float* taps,
                                      int ntaps, mfsr_stream_t stream)
{
    const mfsr_prepare_frame f = {dataIn, halfOut, pyr0, pyr1};
    return prepare_frames_impl(1, &f, halfPitch, maxVal, dimX, dimY, pyr0Pitch, pyr1Pitch, taps, ntaps, stream);
}

extern "C" int mfsr_prepareFrameFusedBatch(int nFrames, const mfsr_prepare_frame* frames, int halfPitch, float maxVal, int dimX, int dimY,
                                           int pyr0Pitch, int pyr1Pitch, const float* taps, int ntaps, mfsr_stream_t stream)
{
    return prepare_frames_impl(nFrames, frames, halfPitch, maxVal, dimX, dimY, pyr0Pitch, pyr1Pitch, taps, ntaps, stream);
}

// ---- A2: deBayerGreenKernel (DeBayerKernels.cu:55-149) ------------------------
struct Lvl {
    float bp[3], sc[3];
};

__device__ __forceinline__ float green_at(const float* __restrict__ imgIn, int strideIn, int x, int y, int thisPixel,
                                          const Lvl& L)
{
#define RAWC(xx, yy, c) ((row_ptr(imgIn, strideIn, (yy))[(xx)] - L.bp[c]) * L.sc[c])
    if (thisPixel == MFSR_GREEN) return RAWC(x, y, 1);
    if (thisPixel != MFSR_RED && thisPixel != MFSR_BLUE) return 0.0f;
    const int c = (thisPixel == MFSR_RED) ? 0 : 2;
    const float p = RAWC(x, y, c);
    const float xMinus2 = RAWC(x - 2, y, c), xMinus1 = RAWC(x - 1, y, 1), xPlus1 = RAWC(x + 1, y, 1),
                xPlus2 = RAWC(x + 2, y, c);
    const float yMinus2 = RAWC(x, y - 2, c), yMinus1 = RAWC(x, y - 1, 1), yPlus1 = RAWC(x, y + 1, 1),
                yPlus2 = RAWC(x, y + 2, c);
#undef RAWC
    const float gradientX = 0.5f * fabsf(xPlus1 - xMinus1);
    const float gradientY = 0.5f * fabsf(yPlus1 - yMinus1);
    const float laplaceX = 0.25f * fabsf(2.0f * p - xMinus2 - xPlus2);
    const float laplaceY = 0.25f * fabsf(2.0f * p - yMinus2 - yPlus2);
    const float interpolX = 0.125f * (-xMinus2 + 4.0f * xMinus1 + 2.0f * p + 4.0f * xPlus1 - xPlus2);
    const float interpolY = 0.125f * (-yMinus2 + 4.0f * yMinus1 + 2.0f * p + 4.0f * yPlus1 - yPlus2);
    const float weight = (gradientY + laplaceY) / (gradientX + gradientY + laplaceX + laplaceY + 0.000000001f);
    return weight * interpolX + (1.0f - weight) * interpolY;
}

__global__ void __launch_bounds__(256) k_deBayerGreen(int width, int height, const float* __restrict__ imgIn, int strideIn,
                                                     pix3* outImage, int strideOut, Lvl L, int cfa)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= width - 2 || x < 2) return;
    if (y >= height - 2 || y < 2) return;
    row_ptr(outImage, strideOut, y)[x].y = green_at(imgIn, strideIn, x, y, cfa_at(cfa, y, x), L);
}

static inline Lvl make_lvl(mfsr_float3 bp, mfsr_float3 sc)
{
    Lvl L;
    L.bp[0] = bp.x;
    L.bp[1] = bp.y;
    L.bp[2] = bp.z;
    L.sc[0] = sc.x;
    L.sc[1] = sc.y;
    L.sc[2] = sc.z;
    return L;
}

extern "C" int mfsr_deBayerGreenKernel(int width, int height, const float* imgIn, int strideIn, mfsr_float3* outImage,
                                       int strideOut, mfsr_float3 blackPoint, mfsr_float3 scale, mfsr_stream_t stream)
{
    MFSR_REQUIRE(imgIn && outImage && width > 0 && height > 0);
    MFSR_REQUIRE((long long)strideIn >= 4LL * width && (long long)strideOut >= 12LL * width);
    MFSR_REQUIRE((strideIn & 3) == 0 && (strideOut & 3) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(width, 64), mfsr_cdiv(height, 4));
    hipLaunchKernelGGL(k_deBayerGreen, grid, block, 0, mfsr_s(stream), width, height, imgIn, strideIn, (pix3*)outImage,
                       strideOut, make_lvl(blackPoint, scale), mfsr_cfa_packed());
    return mfsr_launch_status("deBayerGreenKernel");
}

// ---- A3: deBayerRedBlueKernel (DeBayerKernels.cu:153-231) ---------------------
// GREENF(xx,yy) abstracts where the green plane comes from (global image for the
// straight kernel, LDS tile for the fused one).
template <typename GreenF>
__device__ __forceinline__ void redblue_at(const float* __restrict__ imgIn, int strideIn, int x, int y, int cfa,
                                           const Lvl& L, GreenF GREENF, float g, float& r, float& b)
{
#define RAWR(xx, yy) ((row_ptr(imgIn, strideIn, (yy))[(xx)] - L.bp[0]) * L.sc[0])
#define RAWB(xx, yy) ((row_ptr(imgIn, strideIn, (yy))[(xx)] - L.bp[2]) * L.sc[2])
    const int thisPixel = cfa_at(cfa, y, x);
    const int thisRow = cfa_at(cfa, y, x + 1);
    if (thisPixel == MFSR_GREEN) {
        if (thisRow == MFSR_RED) {
            r = g + 0.5f * ((RAWR(x - 1, y) - GREENF(x - 1, y)) + (RAWR(x + 1, y) - GREENF(x + 1, y)));
            b = g + 0.5f * ((RAWB(x, y - 1) - GREENF(x, y - 1)) + (RAWB(x, y + 1) - GREENF(x, y + 1)));
        } else {
            b = g + 0.5f * ((RAWB(x - 1, y) - GREENF(x - 1, y)) + (RAWB(x + 1, y) - GREENF(x + 1, y)));
            r = g + 0.5f * ((RAWR(x, y - 1) - GREENF(x, y - 1)) + (RAWR(x, y + 1) - GREENF(x, y + 1)));
        }
    } else if (thisPixel == MFSR_RED) {
        r = RAWR(x, y);
        b = g + 0.25f * ((((RAWB(x - 1, y - 1) - GREENF(x - 1, y - 1)) + (RAWB(x + 1, y - 1) - GREENF(x + 1, y - 1))) +
                          (RAWB(x + 1, y + 1) - GREENF(x + 1, y + 1))) +
                         (RAWB(x - 1, y + 1) - GREENF(x - 1, y + 1)));
    } else if (thisPixel == MFSR_BLUE) {
        b = RAWB(x, y);
        r = g + 0.25f * ((((RAWR(x - 1, y - 1) - GREENF(x - 1, y - 1)) + (RAWR(x + 1, y - 1) - GREENF(x + 1, y - 1))) +
                          (RAWR(x + 1, y + 1) - GREENF(x + 1, y + 1))) +
                         (RAWR(x - 1, y + 1) - GREENF(x - 1, y + 1)));
    }
#undef RAWR
#undef RAWB
}

__global__ void __launch_bounds__(256) k_deBayerRedBlue(int width, int height, const float* __restrict__ imgIn,
                                                       int strideIn, pix3* outImage, int strideOut, Lvl L, int cfa)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= width - 2 || x < 2) return;
    if (y >= height - 2 || y < 2) return;
    pix3* rowc = row_ptr(outImage, strideOut, y);
    auto GREENF = [&](int xx, int yy) { return row_ptr(outImage, strideOut, yy)[xx].y; };
    float r = rowc[x].x, b = rowc[x].z;
    redblue_at(imgIn, strideIn, x, y, cfa, L, GREENF, rowc[x].y, r, b);
    rowc[x].x = r;
    rowc[x].z = b;
}

extern "C" int mfsr_deBayerRedBlueKernel(int width, int height, const float* imgIn, int strideIn, mfsr_float3* outImage,
                                         int strideOut, mfsr_float3 blackPoint, mfsr_float3 scale, mfsr_stream_t stream)
{
    MFSR_REQUIRE(imgIn && outImage && width > 0 && height > 0);
    MFSR_REQUIRE((long long)strideIn >= 4LL * width && (long long)strideOut >= 12LL * width);
    MFSR_REQUIRE((strideIn & 3) == 0 && (strideOut & 3) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(width, 64), mfsr_cdiv(height, 4));
    hipLaunchKernelGGL(k_deBayerRedBlue, grid, block, 0, mfsr_s(stream), width, height, imgIn, strideIn, (pix3*)outImage,
                       strideOut, make_lvl(blackPoint, scale), mfsr_cfa_packed());
    return mfsr_launch_status("deBayerRedBlueKernel");
}

// ---- A2+A3 fused (u16 in, one launch) -----------------------------------------
// Workgroup = 64 x 8 output pixels.  The u16 raw tile with a 3-px halo is staged
// in LDS as float, the green plane is computed for the tile + 1-px halo into a
// second LDS tile, then red/blue are interpolated from LDS.  HBM traffic: 2 B in
// + 12 B out per pixel instead of (4 + 4 + 12 + 4 + 12 + 12) for u16->f32 + A2 +
// A3 as separate launches.  Arithmetic is the same sequence as A2/A3, so the
// result is bit-identical to the two-launch chain.
#define DBF_TX 64
#define DBF_TY 8
#define DBF_RW (DBF_TX + 6)
#define DBF_RH (DBF_TY + 6)
#define DBF_GW (DBF_TX + 2)
#define DBF_GH (DBF_TY + 2)

__global__ void __launch_bounds__(DBF_TX* DBF_TY) k_deBayerFused(const uint16_t* __restrict__ raw, pix3* __restrict__ outImage,
                                                                int strideOut, int width, int height, Lvl L, int cfa)
{
    __shared__ float s_raw[DBF_RH][DBF_RW + 1];
    __shared__ float s_g[DBF_GH][DBF_GW + 1];
    const int x0 = blockIdx.x * DBF_TX, y0 = blockIdx.y * DBF_TY;
    const int tid = threadIdx.y * DBF_TX + threadIdx.x;
    for (int i = tid; i < DBF_RW * DBF_RH; i += DBF_TX * DBF_TY) {
        const int ly = i / DBF_RW, lx = i - ly * DBF_RW;
        const int gx = clampi(x0 + lx - 3, 0, width - 1), gy = clampi(y0 + ly - 3, 0, height - 1);
        s_raw[ly][lx] = (float)raw[(size_t)gy * width + gx];
    }
    __syncthreads();
    // raw accessor in image coordinates served from LDS (valid for the tile +-3)
    auto RAWF = [&](int xx, int yy) { return s_raw[yy - y0 + 3][xx - x0 + 3]; };
    for (int i = tid; i < DBF_GW * DBF_GH; i += DBF_TX * DBF_TY) {
        const int ly = i / DBF_GW, lx = i - ly * DBF_GW;
        const int x = x0 + lx - 1, y = y0 + ly - 1;
        float g = 0.0f;
        if (x >= 2 && x < width - 2 && y >= 2 && y < height - 2) {
            const int thisPixel = cfa_at(cfa, y, x);
#define RAWC(xx, yy, c) ((RAWF(xx, yy) - L.bp[c]) * L.sc[c])
            if (thisPixel == MFSR_GREEN) {
                g = RAWC(x, y, 1);
            } else if (thisPixel == MFSR_RED || thisPixel == MFSR_BLUE) {
                const int c = (thisPixel == MFSR_RED) ? 0 : 2;
                const float p = RAWC(x, y, c);
                const float xMinus2 = RAWC(x - 2, y, c), xMinus1 = RAWC(x - 1, y, 1), xPlus1 = RAWC(x + 1, y, 1),
                            xPlus2 = RAWC(x + 2, y, c);
                const float yMinus2 = RAWC(x, y - 2, c), yMinus1 = RAWC(x, y - 1, 1), yPlus1 = RAWC(x, y + 1, 1),
                            yPlus2 = RAWC(x, y + 2, c);
                const float gradientX = 0.5f * fabsf(xPlus1 - xMinus1);
                const float gradientY = 0.5f * fabsf(yPlus1 - yMinus1);
                const float laplaceX = 0.25f * fabsf(2.0f * p - xMinus2 - xPlus2);
                const float laplaceY = 0.25f * fabsf(2.0f * p - yMinus2 - yPlus2);
                const float interpolX = 0.125f * (-xMinus2 + 4.0f * xMinus1 + 2.0f * p + 4.0f * xPlus1 - xPlus2);
                const float interpolY = 0.125f * (-yMinus2 + 4.0f * yMinus1 + 2.0f * p + 4.0f * yPlus1 - yPlus2);
                const float weight = (gradientY + laplaceY) / (gradientX + gradientY + laplaceX + laplaceY + 0.000000001f);
                g = weight * interpolX + (1.0f - weight) * interpolY;
            }
#undef RAWC
        }
        s_g[ly][lx] = g;
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= width - 2 || x < 2 || y >= height - 2 || y < 2) return;
    auto GREENF = [&](int xx, int yy) { return s_g[yy - y0 + 1][xx - x0 + 1]; };
#define RAWR(xx, yy) ((RAWF(xx, yy) - L.bp[0]) * L.sc[0])
#define RAWB(xx, yy) ((RAWF(xx, yy) - L.bp[2]) * L.sc[2])
    const int thisPixel = cfa_at(cfa, y, x);
    const int thisRow = cfa_at(cfa, y, x + 1);
    const float g = GREENF(x, y);
    // The straight A3 kernel leaves r,b untouched for non-RGB CFA colours; the
    // fused kernel defines them as 0 (outImage is write-only here).
    float r = 0.0f, b = 0.0f;
    if (thisPixel == MFSR_GREEN) {
        if (thisRow == MFSR_RED) {
            r = g + 0.5f * ((RAWR(x - 1, y) - GREENF(x - 1, y)) + (RAWR(x + 1, y) - GREENF(x + 1, y)));
            b = g + 0.5f * ((RAWB(x, y - 1) - GREENF(x, y - 1)) + (RAWB(x, y + 1) - GREENF(x, y + 1)));
        } else {
            b = g + 0.5f * ((RAWB(x - 1, y) - GREENF(x - 1, y)) + (RAWB(x + 1, y) - GREENF(x + 1, y)));
            r = g + 0.5f * ((RAWR(x, y - 1) - GREENF(x, y - 1)) + (RAWR(x, y + 1) - GREENF(x, y + 1)));
        }
    } else if (thisPixel == MFSR_RED) {
        r = RAWR(x, y);
        b = g + 0.25f * ((((RAWB(x - 1, y - 1) - GREENF(x - 1, y - 1)) + (RAWB(x + 1, y - 1) - GREENF(x + 1, y - 1))) +
                          (RAWB(x + 1, y + 1) - GREENF(x + 1, y + 1))) +
                         (RAWB(x - 1, y + 1) - GREENF(x - 1, y + 1)));
    } else if (thisPixel == MFSR_BLUE) {
        b = RAWB(x, y);
        r = g + 0.25f * ((((RAWR(x - 1, y - 1) - GREENF(x - 1, y - 1)) + (RAWR(x + 1, y - 1) - GREENF(x + 1, y - 1))) +
                          (RAWR(x + 1, y + 1) - GREENF(x + 1, y + 1))) +
                         (RAWR(x - 1, y + 1) - GREENF(x - 1, y + 1)));
    }
#undef RAWR
#undef RAWB
    pix3 o = {r, g, b};
    row_ptr(outImage, strideOut, y)[x] = o;
}

extern "C" int mfsr_deBayerFused(const uint16_t* raw, mfsr_float3* outImage, int strideOut, int width, int height,
                                 mfsr_float3 blackPoint, mfsr_float3 scale, mfsr_stream_t stream)
{
    MFSR_REQUIRE(raw && outImage && width > 4 && height > 4);
    MFSR_REQUIRE((long long)strideOut >= 12LL * width && (strideOut & 3) == 0);
    dim3 block(DBF_TX, DBF_TY), grid(mfsr_cdiv(width, DBF_TX), mfsr_cdiv(height, DBF_TY));
    hipLaunchKernelGGL(k_deBayerFused, grid, block, 0, mfsr_s(stream), raw, (pix3*)outImage, strideOut, width, height,
                       make_lvl(blackPoint, scale), mfsr_cfa_packed());
    return mfsr_launch_status("deBayerFused");
}
