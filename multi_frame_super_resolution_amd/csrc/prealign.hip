// prealign.hip -- global pre-alignment of a moved frame against the reference frame: one base shift and one base
// rotation per frame, feeding baseShift / baseRotation of convertToTilesOverlapPreShift (reference
// test_opencv/kernel.cu:324-378) and CreateFlowFieldFromTiles (opticalFlow.cu:48-93).
//
// The reference has the slot but not the arithmetic: `class PreAlignment` is a field list (boxFilterNPP.cpp:102-166)
// and the FFT log-polar registration of test_opencv/main.cpp:861-1194 never returns a result (:840-851).  What its
// kernels fix is the model the estimate must satisfy (kernel.cu:358-368, opticalFlow.cu:78-85): a reference pixel p
// maps to the moved pixel   q = c + R(theta) * (p - c - base),   c = (width/2, height/2).
//
// Estimator (the build's own; oracle/prealign.c states it on the CPU): exhaustive coarse-to-fine search over
// (theta, base) on a 2x2-mean pyramid of the two tracking images, scored in INTEGER arithmetic (8-bit samples,
// 4.4 fixed-point bilinear, 64-bit sums) so that the result does not depend on the order in which lanes, waves and
// workgroups add their parts -- the GPU result is bit-identical to the serial CPU loop.  Angles live on a 1/16
// degree grid and their cos/sin come from a table built on the HOST with libm, so the float coordinates of both
// implementations are the same bits (only + - * follow, compiled without contraction).
//
// MI355X mapping: a candidate (angle, ty, tx) is scored by `parts` wavefronts, each taking a band of rows of the
// central window; lanes stride along x, per-lane 64-bit partial sums are reduced with DPP/shuffle adds and one
// 64-bit atomic add per wave lands in scores[candidate].  One small workgroup then reduces the <= 11849 scores of a
// level with a packed (score, index) minimum and writes the next level's search centre to device memory: the whole
// search is a fixed launch sequence without host round trips (hipGraph-capturable), and the consumers
// (mfsr_trackTilesFusedBase, mfsr_CreateFlowFieldFromTilesBase) read the result from device memory.
#include "common.hpp"

#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

constexpr int kPreMaxLevels = 16;
constexpr int kPreT0 = 8;
constexpr float kPreAngleK = 0.00109083078f;  // pi / 2880: radians per 1/16 degree
constexpr int kPreMaxCand = 16384;

struct PreLayout {
    int n;                       // pyramid levels (0 = the tracking image itself)
    int jmin, jmax;              // searched levels jmax (coarsest) .. jmin
    int w[kPreMaxLevels], h[kPreMaxLevels];
    int fpitch[kPreMaxLevels];   // float level pitch (bytes), levels 1..n-1
    size_t foff[kPreMaxLevels];  // float level offset in the pyramid buffer (level 0 is the caller's image)
    size_t qoff[kPreMaxLevels];  // u8 level offset (dense, stride w)
    size_t bytes;
};

inline size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }

PreLayout pre_layout(int width, int height)
{
    PreLayout L;
    memset(&L, 0, sizeof(L));
    L.w[0] = width;
    L.h[0] = height;
    L.n = 1;
    while (std::max(L.w[L.n - 1], L.h[L.n - 1]) > 64 && L.n < kPreMaxLevels && (L.w[L.n - 1] >> 1) >= 8 &&
           (L.h[L.n - 1] >> 1) >= 8) {
        L.w[L.n] = L.w[L.n - 1] >> 1;
        L.h[L.n] = L.h[L.n - 1] >> 1;
        L.n++;
    }
    L.jmax = L.n - 1;
    L.jmin = 0;
    while (L.jmin < L.jmax && std::max(L.w[L.jmin], L.h[L.jmin]) > 1024) L.jmin++;
    size_t off = 0;
    for (int j = 1; j < L.n; j++) {
        L.fpitch[j] = (int)up((size_t)L.w[j] * 4, 64);
        L.foff[j] = off;
        off += up((size_t)L.fpitch[j] * L.h[j], 256);
    }
    for (int j = L.jmin; j < L.n; j++) {
        L.qoff[j] = off;
        off += up((size_t)L.w[j] * L.h[j], 256);
    }
    L.bytes = up(off, 256);
    return L;
}

// workspace of one search: trig table | scores | per-level state
struct PreWork {
    int amax;
    size_t tabOff, scoreOff, stateOff, bytes;
};

PreWork pre_work(float maxAngleDeg)
{
    PreWork W;
    int A = (int)(maxAngleDeg * 16.0f);
    if (A < 0) A = 0;
    A = (A / 16) * 16;
    W.amax = A + 32;
    W.tabOff = 0;
    size_t off = up(sizeof(float) * 2 * (2 * (size_t)W.amax + 1), 256);
    W.scoreOff = off;
    off += up(sizeof(unsigned long long) * (size_t)kPreMaxCand * kPreMaxLevels, 256);
    W.stateOff = off;
    off += up(sizeof(int) * 4 * (kPreMaxLevels + 1), 256);
    W.bytes = off;
    return W;
}

__device__ __forceinline__ int pre_quant(float v)
{
    const int q = f2i(v * 255.0f + 0.5f);
    return min(max(q, 0), 255);
}

// level j-1 (float) -> level j (float, 2x2 mean as mfsr_downsample2x) and, if wanted, its 8-bit quantisation
__global__ void __launch_bounds__(256) k_preDown(const float* __restrict__ in, int inPitch, float* __restrict__ out, int outPitch,
                                                uint8_t* __restrict__ outQ, int outW, int outH)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= outW || y >= outH) return;
    const float* r0 = row_ptr(in, inPitch, 2 * y);
    const float* r1 = row_ptr(in, inPitch, 2 * y + 1);
    const float v = ((r0[2 * x] + r0[2 * x + 1]) + (r1[2 * x] + r1[2 * x + 1])) * 0.25f;
    row_ptr(out, outPitch, y)[x] = v;
    if (outQ) outQ[(size_t)y * outW + x] = (uint8_t)pre_quant(v);
}

__global__ void __launch_bounds__(256) k_preQuant(const float* __restrict__ in, int inPitch, uint8_t* __restrict__ outQ, int w, int h)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    outQ[(size_t)y * w + x] = (uint8_t)pre_quant(row_ptr(in, inPitch, y)[x]);
}

struct PreGrid {
    int na, nt, a0, astep, tx0, ty0;
};

// candidate grid of a level given the centre found on the previous (coarser) level
__device__ __forceinline__ PreGrid pre_grid(int j, int jmax, int A, const int* __restrict__ prev)
{
    PreGrid g;
    if (j == jmax) {
        g.astep = 16;
        g.na = 2 * (A / 16) + 1;
        g.a0 = -A;
        g.nt = 2 * kPreT0 + 1;
        g.tx0 = -kPreT0;
        g.ty0 = -kPreT0;
    } else {
        g.astep = max(16 >> (jmax - j), 1);
        g.na = 5;
        g.a0 = prev[0] - 2 * g.astep;
        g.nt = 5;
        g.tx0 = 2 * prev[1] - 2;
        g.ty0 = 2 * prev[2] - 2;
    }
    return g;
}

// one wavefront = one (candidate, row band); 4 wavefronts per workgroup
__global__ void __launch_bounds__(256) k_preScore(const uint8_t* __restrict__ ref, const uint8_t* __restrict__ mov, int w, int h,
                                                 int j, int jmax, int A, int amax, const float* __restrict__ tab,
                                                 const int* __restrict__ prevState, unsigned long long* __restrict__ scores,
                                                 int ncand, int parts)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int c = wave / parts, part = wave - c * parts;
    if (c >= ncand) return;
    const PreGrid g = pre_grid(j, jmax, A, prevState);
    const int ia = c / (g.nt * g.nt), r = c - ia * g.nt * g.nt, iy = r / g.nt, ix = r - iy * g.nt;
    const int ca = min(max(g.a0 + ia * g.astep, -amax), amax);
    const float cosv = tab[2 * (ca + amax)], sinv = tab[2 * (ca + amax) + 1];
    const int tx = g.tx0 + ix, ty = g.ty0 + iy;
    const int cx = w / 2, cy = h / 2;
    const int x0 = w / 4, x1 = w - w / 4, y0 = h / 4, y1 = h - h / 4;
    const int rows = y1 - y0;
    const int rb = part * rows / parts + y0, re = (part + 1) * rows / parts + y0;
    const int xmax16 = (w - 1) * 16, ymax16 = (h - 1) * 16;
    unsigned long long sum = 0;
    for (int y = rb; y < re; y++) {
        const float dy = (float)(y - cy - ty);
        const float sdy = sinv * dy, cdy = cosv * dy;
        const uint8_t* rrow = ref + (size_t)y * w;
        for (int x = x0 + lane; x < x1; x += 64) {
            const float dx = (float)(x - cx - tx);
            const float qx = (cosv * dx - sdy) + (float)cx;
            const float qy = (sinv * dx + cdy) + (float)cy;
            int fx = f2i(floorf(qx * 16.0f + 0.5f));
            int fy = f2i(floorf(qy * 16.0f + 0.5f));
            fx = min(max(fx, 0), xmax16);
            fy = min(max(fy, 0), ymax16);
            const int px = fx >> 4, ax = fx & 15, py = fy >> 4, ay = fy & 15;
            const int px1 = min(px + 1, w - 1), py1 = min(py + 1, h - 1);
            const uint8_t* m0 = mov + (size_t)py * w;
            const uint8_t* m1 = mov + (size_t)py1 * w;
            const int m = (16 - ax) * (16 - ay) * (int)m0[px] + ax * (16 - ay) * (int)m0[px1] + (16 - ax) * ay * (int)m1[px] +
                          ax * ay * (int)m1[px1];
            const int d = m - 256 * (int)rrow[x];  // |d| <= 65280, d*d < 2^32
            sum += (unsigned long long)((unsigned)(d * d));
        }
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) atomicAdd(&scores[c], sum);
}

// argmin of a level's scores (ties: lowest candidate index) -> search centre of the next level; the last level also
// writes the result
__global__ void __launch_bounds__(256) k_preArgmin(const unsigned long long* __restrict__ scores, int j, int jmin, int jmax, int A,
                                                  int amax, const float* __restrict__ tab, const int* __restrict__ prevState,
                                                  int* __restrict__ outState, mfsr_prealign* __restrict__ result)
{
    __shared__ unsigned long long s_v[256];
    __shared__ int s_i[256];
    const PreGrid g = pre_grid(j, jmax, A, prevState);
    const int ncand = g.na * g.nt * g.nt;
    unsigned long long bv = ~0ull;
    int bi = 0x7fffffff;
    for (int c = threadIdx.x; c < ncand; c += 256) {
        const unsigned long long v = scores[c];
        if (v < bv || (v == bv && c < bi)) {
            bv = v;
            bi = c;
        }
    }
    s_v[threadIdx.x] = bv;
    s_i[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const unsigned long long v = s_v[threadIdx.x + o];
            const int i = s_i[threadIdx.x + o];
            if (v < s_v[threadIdx.x] || (v == s_v[threadIdx.x] && i < s_i[threadIdx.x])) {
                s_v[threadIdx.x] = v;
                s_i[threadIdx.x] = i;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int best = s_i[0];
        const int ia = best / (g.nt * g.nt), r = best - ia * g.nt * g.nt, iy = r / g.nt, ix = r - iy * g.nt;
        const int a = min(max(g.a0 + ia * g.astep, -amax), amax);
        const int tx = g.tx0 + ix, ty = g.ty0 + iy;
        outState[0] = a;
        outState[1] = tx;
        outState[2] = ty;
        outState[3] = j;
        if (j == jmin) {
            result->shiftX = (float)(tx * (1 << jmin));
            result->shiftY = (float)(ty * (1 << jmin));
            result->rotation = (float)a * kPreAngleK;
            result->cosRotation = tab[2 * (a + amax)];
            result->sinRotation = tab[2 * (a + amax) + 1];
            result->angleIndex = a;
            result->tx = tx;
            result->ty = ty;
            result->level = jmin;
        }
    }
}

__global__ void k_preIdentity(mfsr_prealign* result)
{
    result->shiftX = result->shiftY = result->rotation = result->sinRotation = 0.0f;
    result->cosRotation = 1.0f;
    result->angleIndex = result->tx = result->ty = result->level = 0;
}

// host trig tables, one per amax, alive for the life of the process (hipMemcpyAsync from pageable memory stages the
// source before returning, but a captured graph replays the copy: the source must stay valid)
const float* host_table(int amax)
{
    static std::mutex mu;
    static std::vector<std::vector<float>*> tabs;
    std::lock_guard<std::mutex> lk(mu);
    for (auto* t : tabs)
        if ((int)t->size() == 2 * (2 * amax + 1)) return t->data();
    auto* t = new std::vector<float>(2 * (2 * (size_t)amax + 1));
    for (int a = -amax; a <= amax; a++) {
        const float th = (float)a * kPreAngleK;
        (*t)[2 * (a + amax)] = cosf(th);
        (*t)[2 * (a + amax) + 1] = sinf(th);
    }
    tabs.push_back(t);
    return t->data();
}

}  // namespace

extern "C" size_t mfsr_preAlign_pyramid_bytes(int width, int height)
{
    if (width < 8 || height < 8) return 0;
    return pre_layout(width, height).bytes;
}

extern "C" size_t mfsr_preAlign_workspace_bytes(float maxAngleDeg)
{
    if (!(maxAngleDeg >= 0.0f && maxAngleDeg <= 90.0f)) return 0;
    return pre_work(maxAngleDeg).bytes;
}

extern "C" int mfsr_preAlign_init(void* workspace, float maxAngleDeg, mfsr_stream_t stream)
{
    MFSR_REQUIRE(workspace && ((uintptr_t)workspace & 255) == 0 && maxAngleDeg >= 0.0f && maxAngleDeg <= 90.0f);
    const PreWork W = pre_work(maxAngleDeg);
    const float* tab = host_table(W.amax);
    MFSR_HIP_TRY(hipMemcpyAsync((char*)workspace + W.tabOff, tab, sizeof(float) * 2 * (2 * (size_t)W.amax + 1),
                                hipMemcpyHostToDevice, mfsr_s(stream)));
    return MFSR_OK;
}

extern "C" int mfsr_preAlignPyramid(const float* img, int width, int height, int pitch, void* pyramid, mfsr_stream_t stream)
{
    MFSR_REQUIRE(img && pyramid && width >= 8 && height >= 8 && ((uintptr_t)pyramid & 255) == 0);
    MFSR_REQUIRE((long long)pitch >= 4LL * width && (pitch & 3) == 0);
    const PreLayout L = pre_layout(width, height);
    char* base = (char*)pyramid;
    const dim3 block(64, 4);
    if (L.jmin == 0) {
        hipLaunchKernelGGL(k_preQuant, dim3(mfsr_cdiv(width, 64), mfsr_cdiv(height, 4)), block, 0, mfsr_s(stream), img, pitch,
                           (uint8_t*)(base + L.qoff[0]), width, height);
    }
    const float* in = img;
    int inPitch = pitch;
    for (int j = 1; j < L.n; j++) {
        float* out = (float*)(base + L.foff[j]);
        hipLaunchKernelGGL(k_preDown, dim3(mfsr_cdiv(L.w[j], 64), mfsr_cdiv(L.h[j], 4)), block, 0, mfsr_s(stream), in, inPitch, out,
                           L.fpitch[j], j >= L.jmin ? (uint8_t*)(base + L.qoff[j]) : nullptr, L.w[j], L.h[j]);
        in = out;
        inPitch = L.fpitch[j];
    }
    return mfsr_launch_status("preAlignPyramid");
}

extern "C" int mfsr_preAlign(const void* refPyramid, const void* movedPyramid, int width, int height, float maxAngleDeg,
                             void* workspace, mfsr_prealign* result, mfsr_stream_t stream)
{
    MFSR_REQUIRE(refPyramid && movedPyramid && workspace && result && width >= 8 && height >= 8);
    MFSR_REQUIRE(maxAngleDeg >= 0.0f && maxAngleDeg <= 90.0f && ((uintptr_t)workspace & 255) == 0);
    const PreLayout L = pre_layout(width, height);
    const PreWork W = pre_work(maxAngleDeg);
    const int A = W.amax - 32;
    char* ws = (char*)workspace;
    const float* tab = (const float*)(ws + W.tabOff);
    unsigned long long* scores = (unsigned long long*)(ws + W.scoreOff);
    int* state = (int*)(ws + W.stateOff);
    const int levels = L.jmax - L.jmin + 1;
    MFSR_REQUIRE((2 * (A / 16) + 1) * (2 * kPreT0 + 1) * (2 * kPreT0 + 1) <= kPreMaxCand && levels <= kPreMaxLevels);
    MFSR_HIP_TRY(hipMemsetAsync(scores, 0, sizeof(unsigned long long) * (size_t)kPreMaxCand * levels, mfsr_s(stream)));
    for (int j = L.jmax, li = 0; j >= L.jmin; j--, li++) {
        const int ncand = (j == L.jmax) ? (2 * (A / 16) + 1) * (2 * kPreT0 + 1) * (2 * kPreT0 + 1) : 125;
        const int rows = (L.h[j] - L.h[j] / 4) - L.h[j] / 4;
        int parts = (4096 + ncand - 1) / ncand;  // >= 4 wavefronts per SIMD's worth of independent work
        parts = std::max(1, std::min(parts, rows));
        const long long waves = (long long)ncand * parts;
        const uint8_t* qr = (const uint8_t*)((const char*)refPyramid + L.qoff[j]);
        const uint8_t* qm = (const uint8_t*)((const char*)movedPyramid + L.qoff[j]);
        unsigned long long* sc = scores + (size_t)kPreMaxCand * li;
        const int* prev = state + 4 * li;        // slot li holds the centre found on the previous (coarser) level
        int* next = state + 4 * (li + 1);
        hipLaunchKernelGGL(k_preScore, dim3(mfsr_cdiv(waves, 4)), dim3(256), 0, mfsr_s(stream), qr, qm, L.w[j], L.h[j], j, L.jmax, A,
                           W.amax, tab, prev, sc, ncand, parts);
        hipLaunchKernelGGL(k_preArgmin, dim3(1), dim3(256), 0, mfsr_s(stream), sc, j, L.jmin, L.jmax, A, W.amax, tab, prev, next,
                           result);
    }
    return mfsr_launch_status("preAlign");
}

extern "C" int mfsr_preAlign_identity(mfsr_prealign* result, mfsr_stream_t stream)
{
    MFSR_REQUIRE(result != nullptr);
    hipLaunchKernelGGL(k_preIdentity, dim3(1), dim3(1), 0, mfsr_s(stream), result);
    return mfsr_launch_status("preAlign_identity");
}
