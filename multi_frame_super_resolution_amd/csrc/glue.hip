// glue.hip -- the stages BETWEEN the reference kernels plus library utilities.
// The reference has no host that launches its ImageStackAlignator kernels
// (SURVEY.md section 0), so apart from gaussin_filter_1D
// (test_opencv/main.cpp:370-391) and sharpenImg2
// (finalProject/Project/multi_frame_sr.cpp:90-119) these are the build's own
// glue stages, specified in DESIGN.md and mirrored by oracle/glue.c.
#include <cmath>
#include <cstring>

#include "common.hpp"

extern "C" const char* mfsr_error_string(int code)
{
    switch (code) {
        case MFSR_OK: return "success";
        case MFSR_E_INVALID: return "invalid argument";
        case MFSR_E_UNSUPPORTED: return "unsupported parameter";
        case MFSR_E_NODEVICE: return "no HIP device";
        case MFSR_E_WORKSPACE: return "workspace too small";
        case -5: return "multi-GPU transport failed (mfsr_dist: RCCL error, or a peer of a local group failed / timed out)";  // MFSR_E_COMM, include/mfsr_dist.h
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}

extern "C" int mfsr_version(void) { return MFSR_VERSION; }

extern "C" int mfsr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// ---- sharpenImg (test_opencv/main.cpp:525-534): unsharp mask on device u8 ----------------------------------------------
// sigma = 1, threshold = 5, amount = 1.  cv::GaussianBlur is third-party (its 8-bit fixed-point path is not reproduced):
// 7 taps exp(-x^2/2) normalised, BORDER_REFLECT_101, separable, each pass rounded to 8 bit -- see oracle/glue.c.
struct SharpTaps {
    float t[7];
};
__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}
__device__ __forceinline__ int sat_u8_rne(float v)
{
    const float r = rintf(v);  // cvRound
    return (int)fminf(fmaxf(r, 0.0f), 255.0f);
}
__global__ void __launch_bounds__(256) k_sharpenBlurH(const uint8_t* __restrict__ img, uint8_t* __restrict__ tmp, int rows, int cols,
                                                     int ch, int stepIn, SharpTaps taps)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (j >= cols * ch) return;
    const int x = j / ch, c = j - x * ch;
    float a = 0;
#pragma unroll
    for (int k = -3; k <= 3; k++) a += taps.t[k + 3] * (float)img[(size_t)y * stepIn + (size_t)reflect101(x + k, cols) * ch + c];
    tmp[(size_t)y * cols * ch + j] = (uint8_t)sat_u8_rne(a);
}
__global__ void __launch_bounds__(256) k_sharpenFinish(const uint8_t* __restrict__ img, const uint8_t* __restrict__ tmp,
                                                      uint8_t* __restrict__ result, int rows, int cols, int ch, int stepIn,
                                                      int stepOut, SharpTaps taps)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (j >= cols * ch) return;
    float a = 0;
#pragma unroll
    for (int k = -3; k <= 3; k++) a += taps.t[k + 3] * (float)tmp[(size_t)reflect101(y + k, rows) * cols * ch + j];
    const int blurred = sat_u8_rne(a);
    const int src = img[(size_t)y * stepIn + j];
    const int diff = max(src - blurred, 0);  // uchar subtraction saturates at 0 (:530)
    const int sharp = min(max(2 * src - blurred, 0), 255);
    result[(size_t)y * stepOut + j] = (uint8_t)(diff < 5 ? src : sharp);
}

extern "C" int mfsr_sharpenImg(const uint8_t* img, uint8_t* result, uint8_t* tmp, int rows, int cols, int ch, int stepIn, int stepOut,
                               mfsr_stream_t stream)
{
    MFSR_REQUIRE(img && result && tmp && rows > 0 && cols > 0 && ch > 0 && stepIn >= cols * ch && stepOut >= cols * ch);
    MFSR_REQUIRE(rows <= 65535);
    SharpTaps taps;
    float sum = 0;
    for (int i = 0; i < 7; i++) {
        taps.t[i] = expf(-(float)((i - 3) * (i - 3)) / 2.0f);
        sum += taps.t[i];
    }
    for (int i = 0; i < 7; i++) taps.t[i] /= sum;
    dim3 grid(mfsr_cdiv((long long)cols * ch, 256), rows);
    hipLaunchKernelGGL(k_sharpenBlurH, grid, dim3(256), 0, mfsr_s(stream), img, tmp, rows, cols, ch, stepIn, taps);
    hipLaunchKernelGGL(k_sharpenFinish, grid, dim3(256), 0, mfsr_s(stream), img, (const uint8_t*)tmp, result, rows, cols, ch, stepIn,
                       stepOut, taps);
    return mfsr_launch_status("sharpenImg");
}

// ---- J1: gaussin_filter_1D (test_opencv/main.cpp:370-391), host ------------------
extern "C" int mfsr_gaussin_filter_1D(float sigma, float* taps)
{
    if (!taps) return 0;
    if (sigma <= 0) {
        static const float delta[9] = {0, 0, 0, 0, 1, 0, 0, 0, 0};
        memcpy(taps, delta, sizeof(delta));
        return 9;
    }
    int size = (int)(sigma / 0.6f - 0.4f) * 2 + 1 + 2;
    if (size > 99) size = 99;
    const int center = size / 2;
    for (int i = 0; i < size; i++) {
        const int x = i - center;
        taps[i] = expf((float)(-(x * x)) / (2 * sigma * sigma));
    }
    float sum = 0;
    for (int i = 0; i < size; i++) sum += taps[i];
    for (int i = 0; i < size; i++) taps[i] /= sum;
    return size;
}

// ---- sharpenImg2 (multi_frame_sr.cpp:90-119) on device u8 -------------------------
// Output byte j of row `row` (j < (cols-2)*ch) is the sharpened source byte
// j + ch (the reference's output pointer starts at column 0, :103/:110); the
// tail and the outer ring are 0.
__global__ void __launch_bounds__(256) k_sharpenImg2(const uint8_t* __restrict__ img, uint8_t* __restrict__ result, int rows,
                                                    int cols, int ch, int stepIn, int stepOut)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (j >= cols * ch || row >= rows) return;
    uint8_t out = 0;
    const int pxl = j / ch;
    if (row >= 1 && row < rows - 1 && pxl >= 1 && pxl < cols - 1 && j < (cols - 2) * ch) {
        const int col = j + ch;
        const uint8_t* previous = img + (size_t)(row - 1) * stepIn;
        const uint8_t* current = img + (size_t)row * stepIn;
        const uint8_t* next = img + (size_t)(row + 1) * stepIn;
        const int v = 5 * current[col] - current[col - ch] - current[col + ch] - previous[col] - next[col];
        out = (uint8_t)min(max(v, 0), 255);
    }
    result[(size_t)row * stepOut + j] = out;
}

extern "C" int mfsr_sharpenImg2(const uint8_t* img, uint8_t* result, int rows, int cols, int ch, int stepIn, int stepOut,
                                mfsr_stream_t stream)
{
    MFSR_REQUIRE(img && result && rows > 0 && cols > 0 && ch > 0 && stepIn >= cols * ch && stepOut >= cols * ch);
    MFSR_REQUIRE(rows <= 65535);
    hipLaunchKernelGGL(k_sharpenImg2, dim3(mfsr_cdiv((long long)cols * ch, 256), rows), dim3(256), 0, mfsr_s(stream), img,
                       result, rows, cols, ch, stepIn, stepOut);
    return mfsr_launch_status("sharpenImg2");
}

// ---- small elementwise stages -------------------------------------------------------
__global__ void __launch_bounds__(256) k_rgbToGray(const pix3* __restrict__ in, int inPitch, float* __restrict__ out,
                                                  int outPitch, int width, int height)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= width || y >= height) return;
    const pix3 p = row_ptr(in, inPitch, y)[x];
    row_ptr(out, outPitch, y)[x] = 0.299f * p.x + 0.587f * p.y + 0.114f * p.z;
}

extern "C" int mfsr_rgbToGray(const mfsr_float3* in, int inPitch, float* out, int outPitch, int width, int height,
                              mfsr_stream_t stream)
{
    MFSR_REQUIRE(in && out && width > 0 && height > 0 && (long long)inPitch >= 12LL * width &&
                 (long long)outPitch >= 4LL * width && (inPitch & 3) == 0 && (outPitch & 3) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(width, 64), mfsr_cdiv(height, 4));
    hipLaunchKernelGGL(k_rgbToGray, grid, block, 0, mfsr_s(stream), (const pix3*)in, inPitch, out, outPitch, width, height);
    return mfsr_launch_status("rgbToGray");
}

__global__ void __launch_bounds__(256) k_u16ToFloat(const uint16_t* __restrict__ in, float* __restrict__ out, int outPitch,
                                                   int width, int height, float factor)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= width || y >= height) return;
    row_ptr(out, outPitch, y)[x] = (float)in[(size_t)y * width + x] * factor;
}

extern "C" int mfsr_u16ToFloat(const uint16_t* in, float* out, int outPitch, int width, int height, float factor,
                               mfsr_stream_t stream)
{
    MFSR_REQUIRE(in && out && width > 0 && height > 0 && (long long)outPitch >= 4LL * width && (outPitch & 3) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(width, 64), mfsr_cdiv(height, 4));
    hipLaunchKernelGGL(k_u16ToFloat, grid, block, 0, mfsr_s(stream), in, out, outPitch, width, height, factor);
    return mfsr_launch_status("u16ToFloat");
}

struct Taps {
    float t[99];
    int n;
};

template <bool ALONG_X>
__global__ void __launch_bounds__(256) k_filter1d(const float* __restrict__ in, int inPitch, float* __restrict__ out,
                                                 int outPitch, int width, int height, int chan, Taps taps)
{
    const int xe = blockIdx.x * blockDim.x + threadIdx.x;  // element index within the row (x*chan + c)
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (xe >= width * chan || y >= height) return;
    const int x = xe / chan, c = xe - x * chan;
    const int c0 = taps.n / 2;
    float s = 0;
    for (int t = 0; t < taps.n; t++) {
        if (ALONG_X) {
            const int xx = clampi(x + t - c0, 0, width - 1);
            s += taps.t[t] * row_ptr(in, inPitch, y)[xx * chan + c];
        } else {
            const int yy = clampi(y + t - c0, 0, height - 1);
            s += taps.t[t] * row_ptr(in, inPitch, yy)[xe];
        }
    }
    row_ptr(out, outPitch, y)[xe] = s;
}

extern "C" int mfsr_separableFilter(const float* in, int inPitch, float* tmp, float* out, int outPitch, int width,
                                    int height, int chan, const float* taps, int ntaps, mfsr_stream_t stream)
{
    MFSR_REQUIRE(in && tmp && out && taps && width > 0 && height > 0 && (chan == 1 || chan == 3));
    MFSR_REQUIRE(ntaps > 0 && ntaps <= 99);
    MFSR_REQUIRE((long long)inPitch >= 4LL * width * chan && (long long)outPitch >= 4LL * width * chan);
    MFSR_REQUIRE((inPitch & 3) == 0 && (outPitch & 3) == 0);
    Taps tp;
    memset(&tp, 0, sizeof(tp));
    memcpy(tp.t, taps, sizeof(float) * ntaps);
    tp.n = ntaps;
    dim3 block(64, 4), grid(mfsr_cdiv((long long)width * chan, 64), mfsr_cdiv(height, 4));
    hipLaunchKernelGGL(k_filter1d<true>, grid, block, 0, mfsr_s(stream), in, inPitch, tmp, outPitch, width, height, chan, tp);
    int rc = mfsr_launch_status("separableFilter(x)");
    if (rc) return rc;
    hipLaunchKernelGGL(k_filter1d<false>, grid, block, 0, mfsr_s(stream), (const float*)tmp, outPitch, out, outPitch, width,
                       height, chan, tp);
    return mfsr_launch_status("separableFilter(y)");
}

__global__ void __launch_bounds__(256) k_downsample2x(const float* __restrict__ in, int inPitch, float* __restrict__ out,
                                                     int outPitch, int outW, int outH)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= outW || y >= outH) return;
    const float2 a = ((const float2*)row_ptr(in, inPitch, 2 * y))[x];
    const float2 b = ((const float2*)row_ptr(in, inPitch, 2 * y + 1))[x];
    row_ptr(out, outPitch, y)[x] = ((a.x + a.y) + (b.x + b.y)) * 0.25f;
}

extern "C" int mfsr_downsample2x(const float* in, int inPitch, float* out, int outPitch, int outW, int outH,
                                 mfsr_stream_t stream)
{
    MFSR_REQUIRE(in && out && outW > 0 && outH > 0 && (long long)inPitch >= 8LL * outW && (long long)outPitch >= 4LL * outW);
    MFSR_REQUIRE((inPitch & 7) == 0 && (outPitch & 3) == 0 && ((uintptr_t)in & 7) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(outW, 64), mfsr_cdiv(outH, 4));
    hipLaunchKernelGGL(k_downsample2x, grid, block, 0, mfsr_s(stream), in, inPitch, out, outPitch, outW, outH);
    return mfsr_launch_status("downsample2x");
}

// direct correlation in the FFT's wrapped layout (kernel.cu:248-254 reads it)
__global__ void __launch_bounds__(256) k_crossCorrelateTiles(const float* __restrict__ refTiles,
                                                            const float* __restrict__ movedTiles,
                                                            float* __restrict__ ccImage, int maxShift, int tileSize,
                                                            int tileCount)
{
    const int L = tileSize + 2 * maxShift, S = maxShift, R = 2 * S + 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int tile = blockIdx.y;
    if (i >= L * L || tile >= tileCount) return;
    const int fy = i / L, fx = i - fy * L;
    // wrapped index -> signed shift (only |s| <= S is produced)
    const int sy = (fy <= S) ? fy : fy - L;
    const int sx = (fx <= S) ? fx : fx - L;
    (void)R;
    float s = 0;
    if (sy >= -S && sy <= S && sx >= -S && sx <= S) {
        const float* rt = refTiles + (size_t)tile * L * L;
        const float* mt = movedTiles + (size_t)tile * L * L;
        // row sums left to right, then the rows top to bottom (the order oracle/glue.c defines)
        for (int y = 0; y < tileSize; y++) {
            float row = 0;
            for (int x = 0; x < tileSize; x++) row += rt[(S + y) * L + (S + x)] * mt[(S + y + sy) * L + (S + x + sx)];
            s += row;
        }
    }
    ccImage[(size_t)tile * L * L + i] = s;
}

extern "C" int mfsr_crossCorrelateTiles(const float* refTiles, const float* movedTiles, float* ccImage, int maxShift,
                                        int tileSize, int tileCount, mfsr_stream_t stream)
{
    MFSR_REQUIRE(refTiles && movedTiles && ccImage && maxShift >= 0 && tileSize > 2 * maxShift && tileCount > 0 &&
                 tileCount <= 65535);
    const int L = tileSize + 2 * maxShift;
    hipLaunchKernelGGL(k_crossCorrelateTiles, dim3(mfsr_cdiv(L * L, 256), tileCount), dim3(256), 0, mfsr_s(stream),
                       refTiles, movedTiles, ccImage, maxShift, tileSize, tileCount);
    return mfsr_launch_status("crossCorrelateTiles");
}

__global__ void __launch_bounds__(256) k_addRoundedPreShift(const float2* __restrict__ preShift, int prePitch,
                                                           float2* __restrict__ found, int foundPitch, int countX,
                                                           int countY)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= countX || y >= countY) return;
    const float2 p = row_ptr(preShift, prePitch, y)[x];
    float2* f = row_ptr(found, foundPitch, y) + x;
    float2 v = *f;
    v.x = roundf(p.x) + v.x;
    v.y = roundf(p.y) + v.y;
    *f = v;
}

extern "C" int mfsr_addRoundedPreShift(const mfsr_float2* preShift, int prePitch, mfsr_float2* found, int foundPitch,
                                       int countX, int countY, mfsr_stream_t stream)
{
    MFSR_REQUIRE(preShift && found && countX > 0 && countY > 0 && (long long)prePitch >= 8LL * countX &&
                 (long long)foundPitch >= 8LL * countX);
    MFSR_REQUIRE((prePitch & 7) == 0 && (foundPitch & 7) == 0 && ((uintptr_t)preShift & 7) == 0 && ((uintptr_t)found & 7) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(countX, 64), mfsr_cdiv(countY, 4));
    hipLaunchKernelGGL(k_addRoundedPreShift, grid, block, 0, mfsr_s(stream), (const float2*)preShift, prePitch,
                       (float2*)found, foundPitch, countX, countY);
    return mfsr_launch_status("addRoundedPreShift");
}

__global__ void __launch_bounds__(256) k_scaleFlow(float2* __restrict__ flow, int pitch, int width, int height, float factor)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= width || y >= height) return;
    float2* p = row_ptr(flow, pitch, y) + x;
    float2 v = *p;
    v.x *= factor;
    v.y *= factor;
    *p = v;
}

extern "C" int mfsr_scaleFlow(mfsr_float2* flow, int pitch, int width, int height, float factor, mfsr_stream_t stream)
{
    MFSR_REQUIRE(flow && width > 0 && height > 0 && (long long)pitch >= 8LL * width && (pitch & 7) == 0 &&
                 ((uintptr_t)flow & 7) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(width, 64), mfsr_cdiv(height, 4));
    hipLaunchKernelGGL(k_scaleFlow, grid, block, 0, mfsr_s(stream), (float2*)flow, pitch, width, height, factor);
    return mfsr_launch_status("scaleFlow");
}

__global__ void __launch_bounds__(256) k_float3ToFloat4(const pix3* __restrict__ in, int inPitch, float4* __restrict__ out,
                                                       int outPitch, int width, int height)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= width || y >= height) return;
    const pix3 p = row_ptr(in, inPitch, y)[x];
    row_ptr(out, outPitch, y)[x] = make_float4(p.x, p.y, p.z, 0.0f);
}

extern "C" int mfsr_float3ToFloat4(const mfsr_float3* in, int inPitch, mfsr_float4* out, int outPitch, int width,
                                   int height, mfsr_stream_t stream)
{
    MFSR_REQUIRE(in && out && width > 0 && height > 0 && (long long)inPitch >= 12LL * width &&
                 (long long)outPitch >= 16LL * width);
    MFSR_REQUIRE((inPitch & 3) == 0 && (outPitch & 15) == 0 && ((uintptr_t)out & 15) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(width, 64), mfsr_cdiv(height, 4));
    hipLaunchKernelGGL(k_float3ToFloat4, grid, block, 0, mfsr_s(stream), (const pix3*)in, inPitch, (float4*)out, outPitch,
                       width, height);
    return mfsr_launch_status("float3ToFloat4");
}

// bilinear fetch of a float3 image (clamp) -- shared with finishFused
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

__global__ void __launch_bounds__(256) k_resampleFloat3(const pix3* __restrict__ in, int inPitch, int inW, int inH,
                                                       pix3* __restrict__ out, int outPitch, int outW, int outH, float u0,
                                                       float u1, float v0, float v1)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= outW || y >= outH) return;
    const float u = u0 + (u1 - u0) * (((float)x + 0.5f) / (float)outW);
    const float v = v0 + (v1 - v0) * (((float)y + 0.5f) / (float)outH);
    row_ptr(out, outPitch, y)[x] = sample_pix3(in, inPitch, inW, inH, u, v);
}

extern "C" int mfsr_resampleFloat3(const mfsr_float3* in, int inPitch, int inW, int inH, mfsr_float3* out, int outPitch,
                                   int outW, int outH, float u0, float u1, float v0, float v1, mfsr_stream_t stream)
{
    MFSR_REQUIRE(in && out && inW > 0 && inH > 0 && outW > 0 && outH > 0);
    MFSR_REQUIRE((long long)inPitch >= 12LL * inW && (long long)outPitch >= 12LL * outW && (inPitch & 3) == 0 &&
                 (outPitch & 3) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(outW, 64), mfsr_cdiv(outH, 4));
    hipLaunchKernelGGL(k_resampleFloat3, grid, block, 0, mfsr_s(stream), (const pix3*)in, inPitch, inW, inH, (pix3*)out,
                       outPitch, outW, outH, u0, u1, v0, v1);
    return mfsr_launch_status("resampleFloat3");
}

__device__ __forceinline__ int quantize1(float f, float maxOut)
{
    if (isnan(f)) f = 0;
    f = fmaxf(fminf(f, 1.0f), 0.0f);
    return (int)(f * maxOut + 0.5f);
}

__global__ void __launch_bounds__(256) k_quantize(const pix3* __restrict__ in, int inPitch, uint16_t* __restrict__ out16,
                                                 uint8_t* __restrict__ out8, int width, int height, float maxOut)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= width || y >= height) return;
    const pix3 p = row_ptr(in, inPitch, y)[x];
    const size_t o = ((size_t)y * width + x) * 3;
    const int q0 = quantize1(p.x, maxOut), q1 = quantize1(p.y, maxOut), q2 = quantize1(p.z, maxOut);
    if (out16) {
        out16[o] = (uint16_t)q0;
        out16[o + 1] = (uint16_t)q1;
        out16[o + 2] = (uint16_t)q2;
    }
    if (out8) {
        out8[o] = (uint8_t)q0;
        out8[o + 1] = (uint8_t)q1;
        out8[o + 2] = (uint8_t)q2;
    }
}

extern "C" int mfsr_quantize(const mfsr_float3* in, int inPitch, uint16_t* out16, uint8_t* out8, int width, int height,
                             float maxOut, mfsr_stream_t stream)
{
    MFSR_REQUIRE(in && (out16 || out8) && width > 0 && height > 0 && (long long)inPitch >= 12LL * width && (inPitch & 3) == 0);
    MFSR_REQUIRE(maxOut > 0 && maxOut <= 65535.0f);
    dim3 block(64, 4), grid(mfsr_cdiv(width, 64), mfsr_cdiv(height, 4));
    hipLaunchKernelGGL(k_quantize, grid, block, 0, mfsr_s(stream), (const pix3*)in, inPitch, out16, out8, width, height,
                       maxOut);
    return mfsr_launch_status("quantize");
}

__global__ void __launch_bounds__(256) k_fill(float* __restrict__ dst, size_t count, float value)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) dst[i] = value;
}

// one-pixel border ring of a float4 image := 0 (ComputeRobustnessMask never writes it)
__global__ void __launch_bounds__(256) k_zeroRing4(float4* __restrict__ img, int pitch, int width, int height)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = 2 * width + 2 * height;
    if (i >= n) return;
    int x, y;
    if (i < width) {
        x = i;
        y = 0;
    } else if (i < 2 * width) {
        x = i - width;
        y = height - 1;
    } else if (i < 2 * width + height) {
        x = 0;
        y = i - 2 * width;
    } else {
        x = width - 1;
        y = i - 2 * width - height;
    }
    row_ptr(img, pitch, y)[x] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

extern "C" int mfsr_zeroRing_f32x4(mfsr_float4* img, int pitch, int width, int height, mfsr_stream_t stream)
{
    MFSR_REQUIRE(img && width > 0 && height > 0 && (long long)pitch >= 16LL * width && (pitch & 15) == 0 &&
                 ((uintptr_t)img & 15) == 0);
    hipLaunchKernelGGL(k_zeroRing4, dim3(mfsr_cdiv(2 * width + 2 * height, 256)), dim3(256), 0, mfsr_s(stream), (float4*)img,
                       pitch, width, height);
    return mfsr_launch_status("zeroRing_f32x4");
}

extern "C" int mfsr_fill_f32(float* dst, size_t count, float value, mfsr_stream_t stream)
{
    MFSR_REQUIRE(dst != nullptr);
    if (count == 0) return MFSR_OK;
    const unsigned blocks = (unsigned)((count + 255) / 256 > 4096 ? 4096 : (count + 255) / 256);
    hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, mfsr_s(stream), dst, count, value);
    return mfsr_launch_status("fill_f32");
}

// ---- E1+E2 fused: derivatives + structure tensor in one launch ----------------------
// The 5-point stencils read the image through LDS (tile + 2-px halo, MIRROR
// addressing resolved while staging).  At pixel centres the bilinear fetch of
// the reference degenerates to a plain texel read, so the fused kernel uses
// direct loads; it differs from the two-launch chain only by the <= 1e-6
// relative blend error of (ix+0.5)/W*W-0.5 != ix.
#define ST_TX 64
#define ST_TY 8
__device__ __forceinline__ int mirror_index(int i, int n)
{
    // reflect about the edges (…2 1 0 | 0 1 2 … n-1 | n-1 n-2 …): CUDA mirror mode at texel centres;
    // valid for -n <= i < 2n (the 2-px halo), no integer division
    i = i < 0 ? -1 - i : i;
    return i >= n ? 2 * n - 1 - i : i;
}

__global__ void __launch_bounds__(ST_TX* ST_TY) k_structureTensorFused(const float* __restrict__ img, int imgPitch,
                                                                     pix3* __restrict__ outImg, int outPitch, int width,
                                                                     int height)
{
    __shared__ float s_t[ST_TY + 4][ST_TX + 4 + 1];
    const int x0 = blockIdx.x * ST_TX, y0 = blockIdx.y * ST_TY;
    const int tid = threadIdx.y * ST_TX + threadIdx.x;
    for (int i = tid; i < (ST_TY + 4) * (ST_TX + 4); i += ST_TX * ST_TY) {
        const int ly = i / (ST_TX + 4), lx = i - ly * (ST_TX + 4);
        const int gx = mirror_index(x0 + lx - 2, width), gy = mirror_index(y0 + ly - 2, height);
        s_t[ly][lx] = row_ptr(img, imgPitch, gy)[gx];
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= width || y >= height) return;
    const int lx = threadIdx.x + 2, ly = threadIdx.y + 2;
    float t0 = s_t[ly][lx + 2];
    t0 -= s_t[ly][lx + 1] * 8.0f;
    t0 += s_t[ly][lx - 1] * 8.0f;
    t0 -= s_t[ly][lx - 2];
    t0 /= 12.0f;
    const float dx = t0;
    t0 = s_t[ly + 2][lx];
    t0 -= s_t[ly + 1][lx] * 8.0f;
    t0 += s_t[ly - 1][lx] * 8.0f;
    t0 -= s_t[ly - 2][lx];
    t0 /= 12.0f;
    const float dy = t0;
    pix3 val = {dx * dx, dy * dy, dx * dy};
    row_ptr(outImg, outPitch, y)[x] = val;
}

extern "C" int mfsr_structureTensorFused(const float* img, int imgPitch, mfsr_float3* outImg, int outPitch, int width,
                                         int height, mfsr_stream_t stream)
{
    MFSR_REQUIRE(img && outImg && width >= 4 && height >= 4 && (long long)imgPitch >= 4LL * width &&
                 (long long)outPitch >= 12LL * width && (imgPitch & 3) == 0 && (outPitch & 3) == 0);
    dim3 block(ST_TX, ST_TY), grid(mfsr_cdiv(width, ST_TX), mfsr_cdiv(height, ST_TY));
    hipLaunchKernelGGL(k_structureTensorFused, grid, block, 0, mfsr_s(stream), img, imgPitch, (pix3*)outImg, outPitch,
                       width, height);
    return mfsr_launch_status("structureTensorFused");
}

// ---- H1 (+fallback resample) + H2 + quantise in one launch ---------------------------
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

__global__ void __launch_bounds__(256)
    k_finishFused(const pix3* __restrict__ finalImg, const pix3* __restrict__ weight, int imgPitch,
                  const pix3* __restrict__ fallback, int fbPitch, int fbW, int fbH, float u0, float u1, float v0, float v1,
                  pix3* __restrict__ outImg, int outPitch, uint16_t* __restrict__ out16, int width, int height,
                  float threshold, int applyGamma, float maxOut, int rowOffset, int fullHeight)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= width || y >= height) return;
    const pix3 val = row_ptr(finalImg, imgPitch, y)[x];
    const pix3 w = row_ptr(weight, imgPitch, y)[x];
    pix3 inout = {0.0f, 0.0f, 0.0f};
    // ApplyWeighting reads the fallback only where a weight is under the threshold (kernel.cu:444-462): the resample (12
    // loads, two divisions, the bilinear mix: 40 % of this kernel's instructions) is skipped by the waves that need none
    if (fallback && (w.x < threshold || w.y < threshold || w.z < threshold)) {
        const float u = u0 + (u1 - u0) * (((float)x + 0.5f) / (float)width);
        // row y of this launch is row y + rowOffset of a fullHeight-row image: the same float expression as the whole-image
        // launch evaluates for that row, so a stripe-wise finish is bit-identical to the whole one
        const float v = v0 + (v1 - v0) * (((float)(y + rowOffset) + 0.5f) / (float)fullHeight);
        inout = sample_pix3(fallback, fbPitch, fbW, fbH, u, v);
    }
    inout.x = apply_weight_f(inout.x, val.x, w.x, threshold);
    inout.y = apply_weight_f(inout.y, val.y, w.y, threshold);
    inout.z = apply_weight_f(inout.z, val.z, w.z, threshold);
    if (applyGamma) {
        inout.x = gamma_f(inout.x);
        inout.y = gamma_f(inout.y);
        inout.z = gamma_f(inout.z);
    }
    if (outImg) row_ptr(outImg, outPitch, y)[x] = inout;
    if (out16) {
        const size_t o = ((size_t)y * width + x) * 3;
        out16[o] = (uint16_t)quantize1(inout.x, maxOut);
        out16[o + 1] = (uint16_t)quantize1(inout.y, maxOut);
        out16[o + 2] = (uint16_t)quantize1(inout.z, maxOut);
    }
}

// rows [rowOffset, rowOffset + height) of a fullHeight-row image whose fallback window is (u0..u1, v0..v1); the image
// pointers are those of the first row of the stripe
extern "C" int mfsr_finishFusedRows(const mfsr_float3* finalImg, const mfsr_float3* weight, int imgPitch,
                                    const mfsr_float3* fallback, int fbPitch, int fbW, int fbH, float u0, float u1, float v0,
                                    float v1, mfsr_float3* outImg, int outPitch, uint16_t* out16, int width, int height,
                                    float threshold, int applyGamma, float maxOut, int rowOffset, int fullHeight,
                                    mfsr_stream_t stream)
{
    MFSR_REQUIRE(finalImg && weight && (outImg || out16) && width > 0 && height > 0);
    MFSR_REQUIRE((long long)imgPitch >= 12LL * width && (imgPitch & 3) == 0);
    if (outImg) MFSR_REQUIRE((long long)outPitch >= 12LL * width && (outPitch & 3) == 0);
    if (fallback) MFSR_REQUIRE(fbW > 0 && fbH > 0 && (long long)fbPitch >= 12LL * fbW && (fbPitch & 3) == 0);
    MFSR_REQUIRE(maxOut > 0 && maxOut <= 65535.0f);
    MFSR_REQUIRE(rowOffset >= 0 && fullHeight >= rowOffset + height);
    dim3 block(64, 4), grid(mfsr_cdiv(width, 64), mfsr_cdiv(height, 4));
    hipLaunchKernelGGL(k_finishFused, grid, block, 0, mfsr_s(stream), (const pix3*)finalImg, (const pix3*)weight, imgPitch,
                       (const pix3*)fallback, fbPitch, fbW, fbH, u0, u1, v0, v1, (pix3*)outImg, outPitch, out16, width,
                       height, threshold, applyGamma, maxOut, rowOffset, fullHeight);
    return mfsr_launch_status("finishFused");
}

extern "C" int mfsr_finishFused(const mfsr_float3* finalImg, const mfsr_float3* weight, int imgPitch,
                                const mfsr_float3* fallback, int fbPitch, int fbW, int fbH, float u0, float u1, float v0,
                                float v1, mfsr_float3* outImg, int outPitch, uint16_t* out16, int width, int height,
                                float threshold, int applyGamma, float maxOut, mfsr_stream_t stream)
{
    return mfsr_finishFusedRows(finalImg, weight, imgPitch, fallback, fbPitch, fbW, fbH, u0, u1, v0, v1, outImg, outPitch, out16,
                                width, height, threshold, applyGamma, maxOut, 0, height, stream);
}


// ---- stripe-sharded bursts: is the vertical flow of these rows within the raw halo that was exchanged? ----------------
__global__ void __launch_bounds__(256) k_checkFlowBound(const float2* __restrict__ flow, int pitch, int width, int rows, float bound,
                                                       int* __restrict__ flag)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    bool bad = false;
    if (x < width && y < rows) {
        const float v = row_ptr(flow, pitch, y)[x].y;
        bad = fabsf(v) > bound;  // NaN rounds to a zero shift in the fuse kernels (v_cvt_i32_f32): harmless
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// *maxBits = max(*maxBits, bits of |flow.y|) over the rows: non-negative floats order like their bit patterns, so one
// integer atomicMax per wavefront does; NaN is skipped (it rounds to a zero shift in the fuse kernels)
__global__ void __launch_bounds__(256) k_maxAbsFlowY(const float2* __restrict__ flow, int pitch, int width, int rows, int* __restrict__ maxBits)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    float m = 0.0f;
    for (int y = blockIdx.y * blockDim.y + threadIdx.y; y < rows; y += gridDim.y * blockDim.y)
        if (x < width) {
            const float v = fabsf(row_ptr(flow, pitch, y)[x].y);
            if (v == v && v > m) m = v;
        }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(maxBits, __float_as_int(m));
}

extern "C" int mfsr_maxAbsFlowY(const mfsr_float2* flow, int pitch, int width, int rows, int* maxBits, mfsr_stream_t stream)
{
    MFSR_REQUIRE(flow && maxBits && width > 0 && rows > 0 && (long long)pitch >= 8LL * width && (pitch & 7) == 0);
    const int gy = mfsr_cdiv(rows, 4) < 64 ? mfsr_cdiv(rows, 4) : 64;
    dim3 block(64, 4), grid(mfsr_cdiv(width, 64), gy);
    hipLaunchKernelGGL(k_maxAbsFlowY, grid, block, 0, mfsr_s(stream), (const float2*)flow, pitch, width, rows, maxBits);
    return mfsr_launch_status("maxAbsFlowY");
}

extern "C" int mfsr_checkFlowBound(const mfsr_float2* flow, int pitch, int width, int rows, float bound, int* flag,
                                   mfsr_stream_t stream)
{
    MFSR_REQUIRE(flow && flag && width > 0 && rows > 0 && (long long)pitch >= 8LL * width && (pitch & 7) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(width, 64), mfsr_cdiv(rows, 4));
    hipLaunchKernelGGL(k_checkFlowBound, grid, block, 0, mfsr_s(stream), (const float2*)flow, pitch, width, rows, bound, flag);
    return mfsr_launch_status("checkFlowBound");
}

// ---- mfsr_exact_div (common.hpp) -----------------------------------------------------------------------------------------
#include <mutex>
MfsrExactDiv mfsr_exact_div(float d)
{
    static std::mutex mu;
    static MfsrExactDiv cache[32];
    static int nCache = 0;
    std::lock_guard<std::mutex> lk(mu);
    for (int i = 0; i < nCache; i++)
        if (cache[i].d == d) return cache[i];
    MfsrExactDiv e{d, 1.0f / d, 0};
    static const bool off = [] {
        const char* v = getenv("MFSR_EXACT_DIV");
        return v && v[0] == '0';
    }();
    if (!off && d > 0.0f && d < 16777216.0f) {
        bool ok = true;
        for (uint32_t m = 0; m < (1u << 23) && ok; m++) {
            const uint32_t bits = (127u << 23) | m;
            float x;
            memcpy(&x, &bits, 4);
            const float q = x * e.r;
            // (host arithmetic: -ffp-contract=off, fmaf is the correctly rounded one)
            ok = fmaf(fmaf(-d, q, x), e.r, q) == x / d;
        }
        e.ok = ok ? 1 : 0;
    }
    if (nCache < 32) cache[nCache++] = e;
    return e;
}

extern "C" int mfsr_exactDivisionOk(float d) { return mfsr_exact_div(d).ok; }
