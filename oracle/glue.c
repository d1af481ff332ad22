/*
 * oracle/glue.c -- CPU statement of the stages BETWEEN the reference kernels.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle_common.h).
 *
 * The reference contains no host that launches its ImageStackAlignator
 * kernels, so everything here except
 * gaussin_filter_1D / sharpenImg2 is the build's own orchestration glue,
 * specified in DESIGN.md ("Pipeline glue") and mirrored 1:1 by the HIP path.
 */
#include "oracle_common.h"
#ifdef _OPENMP
#include <omp.h>
#endif

/* number of host threads the oracle loops use (cpu_baseline.cores in bench.py) */
int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* J1: gaussin_filter_1D, test_opencv/main.cpp:370-391.  Returns the tap count
 * (<= 99); taps must hold 99 floats. */
int orc_gaussin_filter_1D(float sigma, float* taps)
{
    if (sigma <= 0) { /* :371-373 */
        static const float delta[9] = {0, 0, 0, 0, 1, 0, 0, 0, 0};
        memcpy(taps, delta, sizeof(delta));
        return 9;
    }
    int size = (int)(sigma / 0.6f - 0.4f) * 2 + 1 + 2; /* :374 */
    if (size > 99) size = 99;
    int center = size / 2;
    for (int i = 0; i < size; i++) {
        int x = i - center;
        taps[i] = expf((float)(-(x * x)) / (2 * sigma * sigma)); /* :381 */
    }
    float sum = 0;
    for (int i = 0; i < size; i++) sum += taps[i];
    for (int i = 0; i < size; i++) taps[i] /= sum;
    return size;
}

/* sharpenImg2, finalProject/Project/multi_frame_sr.cpp:90-119 (dup
 * test_opencv/main.cpp:537-566).  Quirk kept: the output pointer starts at
 * column 0 while the source column starts at `ch`, so the sharpened row is
 * written one pixel to the left; the never-written tail is defined as 0 here
 * (uninitialised in the reference) and the outer ring is zeroed. */
void orc_sharpenImg2(const uint8_t* img, uint8_t* result, int rows, int cols, int ch, int stepIn, int stepOut)
{
    for (int row = 0; row < rows; row++) memset(result + (size_t)row * stepOut, 0, (size_t)cols * ch);
    for (int row = 1; row < rows - 1; row++) {
        const uint8_t* previous = img + (size_t)(row - 1) * stepIn;
        const uint8_t* current = img + (size_t)row * stepIn;
        const uint8_t* next = img + (size_t)(row + 1) * stepIn;
        uint8_t* output = result + (size_t)row * stepOut;
        int starts = ch;
        int ends = (cols - 1) * ch;
        for (int col = starts; col < ends; col++) {
            int v = 5 * current[col] - current[col - ch] - current[col + ch] - previous[col] - next[col];
            *output++ = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); /* saturate_cast<uchar> */
        }
    }
    if (rows > 0) {
        memset(result, 0, (size_t)cols * ch);
        memset(result + (size_t)(rows - 1) * stepOut, 0, (size_t)cols * ch);
    }
    for (int row = 0; row < rows; row++) {
        memset(result + (size_t)row * stepOut, 0, ch);
        memset(result + (size_t)row * stepOut + (size_t)(cols - 1) * ch, 0, ch);
    }
}

/* sharpenImg, test_opencv/main.cpp:525-534: "unsharp mask" on an 8-bit interleaved image, sigma = 1, threshold = 5,
 * amount = 1.  The Gaussian blur itself is third-party (cv::GaussianBlur, OpenCV 4.5.1, absent here; its 8-bit path uses
 * OpenCV's own fixed-point kernel): restated as the published algorithm -- kernel size cvRound(sigma*6 + 1) | 1 = 7, taps
 * exp(-x^2 / (2 sigma^2)) normalised, BORDER_REFLECT_101, separable, each pass rounded to 8 bit -- PARITY UNPINNED.
 * The rest is the reference's own arithmetic: lowContrast = |src - blurred| < threshold with the uchar subtraction
 * SATURATING at 0 (so src < blurred always counts as low contrast), sharpened = saturate(src*(1+amount) - blurred*amount). */
static inline int orc_reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}
static inline uint8_t orc_sat_u8(float v)
{
    float r = nearbyintf(v); /* cvRound: round half to even */
    return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
}
void orc_sharpenImg(const uint8_t* img, uint8_t* result, uint8_t* tmp, int rows, int cols, int ch, int stepIn, int stepOut)
{
    float taps[7], sum = 0;
    for (int i = 0; i < 7; i++) {
        taps[i] = expf(-(float)((i - 3) * (i - 3)) / 2.0f);
        sum += taps[i];
    }
    for (int i = 0; i < 7; i++) taps[i] /= sum;
    /* horizontal pass -> tmp (dense rows of cols*ch), vertical pass -> blurred value on the fly */
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++)
            for (int c = 0; c < ch; c++) {
                float a = 0;
                for (int k = -3; k <= 3; k++) a += taps[k + 3] * (float)img[(size_t)y * stepIn + (size_t)orc_reflect101(x + k, cols) * ch + c];
                tmp[((size_t)y * cols + x) * ch + c] = orc_sat_u8(a);
            }
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++)
            for (int c = 0; c < ch; c++) {
                float a = 0;
                for (int k = -3; k <= 3; k++) a += taps[k + 3] * (float)tmp[((size_t)orc_reflect101(y + k, rows) * cols + x) * ch + c];
                int blurred = orc_sat_u8(a);
                int src = img[(size_t)y * stepIn + (size_t)x * ch + c];
                int diff = src - blurred;
                if (diff < 0) diff = 0; /* uchar subtraction saturates; abs() of that */
                int sharp = 2 * src - blurred;
                sharp = sharp < 0 ? 0 : (sharp > 255 ? 255 : sharp);
                result[(size_t)y * stepOut + (size_t)x * ch + c] = (uint8_t)(diff < 5 ? src : sharp);
            }
}

/* gray = 0.299 R + 0.587 G + 0.114 B (the weights of the cv::COLOR_BGR2GRAY
 * call at test_opencv/main.cpp:866-867), float3 pitched -> float pitched. */
void orc_rgbToGray(const of3* in, int inPitch, float* out, int outPitch, int width, int height)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < height; y++) {
        const of3* r = ORC_CROW(of3, in, inPitch, y);
        float* o = ORC_ROW(float, out, outPitch, y);
        for (int x = 0; x < width; x++) o[x] = 0.299f * r[x].x + 0.587f * r[x].y + 0.114f * r[x].z;
    }
}

/* mono raw u16 -> float pitched, value * factor */
void orc_u16ToFloat(const uint16_t* in, float* out, int outPitch, int width, int height, float factor)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < height; y++) {
        float* o = ORC_ROW(float, out, outPitch, y);
        for (int x = 0; x < width; x++) o[x] = (float)in[(size_t)y * width + x] * factor;
    }
}

/* separable filter with clamped borders on a `chan`-channel float image
 * (chan = 1 or 3); taps applied ascending, x pass then y pass; tmp has the
 * geometry of out. */
void orc_separableFilter(const float* in, int inPitch, float* tmp, float* out, int outPitch, int width, int height,
                         int chan, const float* taps, int ntaps)
{
    const int c0 = ntaps / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < height; y++) {
        const float* r = ORC_CROW(float, in, inPitch, y);
        float* o = ORC_ROW(float, tmp, outPitch, y);
        for (int x = 0; x < width; x++)
            for (int c = 0; c < chan; c++) {
                float s = 0;
                for (int t = 0; t < ntaps; t++) {
                    int xx = orc_imin(orc_imax(x + t - c0, 0), width - 1);
                    s += taps[t] * r[xx * chan + c];
                }
                o[x * chan + c] = s;
            }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < height; y++) {
        float* o = ORC_ROW(float, out, outPitch, y);
        for (int x = 0; x < width; x++)
            for (int c = 0; c < chan; c++) {
                float s = 0;
                for (int t = 0; t < ntaps; t++) {
                    int yy = orc_imin(orc_imax(y + t - c0, 0), height - 1);
                    s += taps[t] * ORC_CROW(float, tmp, outPitch, yy)[x * chan + c];
                }
                o[x * chan + c] = s;
            }
    }
}

/* 2x2 box downsample: out(x,y) = ((a+b)+(c+d))*0.25, out dims = in dims / 2 */
void orc_downsample2x(const float* in, int inPitch, float* out, int outPitch, int outW, int outH)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < outH; y++) {
        const float* r0 = ORC_CROW(float, in, inPitch, 2 * y);
        const float* r1 = ORC_CROW(float, in, inPitch, 2 * y + 1);
        float* o = ORC_ROW(float, out, outPitch, y);
        for (int x = 0; x < outW; x++) o[x] = ((r0[2 * x] + r0[2 * x + 1]) + (r1[2 * x] + r1[2 * x + 1])) * 0.25f;
    }
}

/* direct cross-correlation of tile stacks, the build's replacement for the
 * upstream FFT -> conjugateComplexMulKernel -> inverse FFT chain
 * (kernel.cu:484-501 is the only part of it in the reference).  Output has
 * the FFT's wrapped layout that normalizedCC reads (kernel.cu:248-254):
 * cc[tile][sy mod L][sx mod L] = sum_{y,x in TxT} ref[S+y][S+x] *
 * moved[S+y+sy][S+x+sx] for sx,sy in [-S,S]; all other entries 0.
 * Summation order (the build's choice -- the FFT fixes none): per tile row the
 * products are added left to right, then the row sums top to bottom -- the
 * order boxFilterWithBorderX / Y (kernel.cu:150-227) give the box term of the
 * same distance, and one that leaves T x (2S+1)^2 independent row sums per tile. */
void orc_crossCorrelateTiles(const float* refTiles, const float* movedTiles, float* ccImage, int maxShift, int tileSize,
                             int tileCount)
{
    const int L = tileSize + 2 * maxShift;
    const int S = maxShift;
#pragma omp parallel for schedule(static)
    for (int tile = 0; tile < tileCount; tile++) {
        const float* rt = refTiles + (size_t)tile * L * L;
        const float* mt = movedTiles + (size_t)tile * L * L;
        float* cc = ccImage + (size_t)tile * L * L;
        for (int i = 0; i < L * L; i++) cc[i] = 0;
        for (int sy = -S; sy <= S; sy++)
            for (int sx = -S; sx <= S; sx++) {
                float s = 0;
                for (int y = 0; y < tileSize; y++) {
                    float row = 0;
                    for (int x = 0; x < tileSize; x++) row += rt[(S + y) * L + (S + x)] * mt[(S + y + sy) * L + (S + x + sx)];
                    s += row;
                }
                int fy = sy < 0 ? L + sy : sy;
                int fx = sx < 0 ? L + sx : sx;
                cc[fy * L + fx] = s;
            }
    }
}

/* total tile shift after one pyramid level: the moved tile was gathered at
 * round(preShift) (kernel.cu:369-370 with base shift/rotation = 0), so
 * total = roundf(preShift) + found. */
void orc_addRoundedPreShift(const of2* preShift, int prePitch, of2* found, int foundPitch, int countX, int countY)
{
    for (int y = 0; y < countY; y++)
        for (int x = 0; x < countX; x++) {
            of2 p = ORC_CROW(of2, preShift, prePitch, y)[x];
            of2* f = &ORC_ROW(of2, found, foundPitch, y)[x];
            f->x = roundf(p.x) + f->x;
            f->y = roundf(p.y) + f->y;
        }
}

/* flow *= factor (tracking-pixel units -> raw-pixel units) */
void orc_scaleFlow(of2* flow, int pitch, int width, int height, float factor)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < height; y++) {
        of2* r = ORC_ROW(of2, flow, pitch, y);
        for (int x = 0; x < width; x++) {
            r[x].x *= factor;
            r[x].y *= factor;
        }
    }
}

/* float3 pitched -> float4 pitched (w = 0): the kernel-parameter texture of
 * accumulateImagesSuperRes is float4 (DeBayerKernels.cu:401). */
void orc_float3ToFloat4(const of3* in, int inPitch, of4* out, int outPitch, int width, int height)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < height; y++) {
        const of3* r = ORC_CROW(of3, in, inPitch, y);
        of4* o = ORC_ROW(of4, out, outPitch, y);
        for (int x = 0; x < width; x++) {
            of4 v = {r[x].x, r[x].y, r[x].z, 0.0f};
            o[x] = v;
        }
    }
}

/* bilinear resample of a float3 image onto an outW x outH grid covering the
 * window [u0,u1] x [v0,v1] of the source in normalised coordinates (clamp
 * addressing).  Used to bring the debayered reference frame (the fallback of
 * ApplyWeighting, kernel.cu:442-451) onto the HR grid. */
void orc_resampleFloat3(const of3* in, int inPitch, int inW, int inH, of3* out, int outPitch, int outW, int outH,
                        float u0, float u1, float v0, float v1)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < outH; y++) {
        of3* o = ORC_ROW(of3, out, outPitch, y);
        for (int x = 0; x < outW; x++) {
            float u = u0 + (u1 - u0) * (((float)x + 0.5f) / (float)outW);
            float v = v0 + (v1 - v0) * (((float)y + 0.5f) / (float)outH);
            float xB = u * (float)inW - 0.5f, yB = v * (float)inH - 0.5f;
            float fx = floorf(xB), fy = floorf(yB);
            float a = xB - fx, b = yB - fy;
            int i0 = orc_imin(orc_imax(orc_f2i(fx), 0), inW - 1), i1 = orc_imin(orc_imax(orc_f2i(fx) + 1, 0), inW - 1);
            int j0 = orc_imin(orc_imax(orc_f2i(fy), 0), inH - 1), j1 = orc_imin(orc_imax(orc_f2i(fy) + 1, 0), inH - 1);
            const of3* r0 = ORC_CROW(of3, in, inPitch, j0);
            const of3* r1 = ORC_CROW(of3, in, inPitch, j1);
            o[x].x = ORC_LERP4(r0[i0].x, r0[i1].x, r1[i0].x, r1[i1].x, a, b);
            o[x].y = ORC_LERP4(r0[i0].y, r0[i1].y, r1[i0].y, r1[i1].y, a, b);
            o[x].z = ORC_LERP4(r0[i0].z, r0[i1].z, r1[i0].z, r1[i1].z, a, b);
        }
    }
}

/* float3 in [0,1] -> interleaved u16 (or u8 when maxOut = 255):
 * q = (int)(clamp(v,0,1)*maxOut + 0.5), NaN -> 0. */
void orc_quantize(const of3* in, int inPitch, uint16_t* out16, uint8_t* out8, int width, int height, float maxOut)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < height; y++) {
        const of3* r = ORC_CROW(of3, in, inPitch, y);
        for (int x = 0; x < width; x++) {
            const float* v = &r[x].x;
            for (int c = 0; c < 3; c++) {
                float f = v[c];
                if (isnan(f)) f = 0;
                f = fmaxf(fminf(f, 1.0f), 0.0f);
                int q = (int)(f * maxOut + 0.5f);
                size_t o = ((size_t)y * width + x) * 3 + c;
                if (out16) out16[o] = (uint16_t)q;
                if (out8) out8[o] = (uint8_t)q;
            }
        }
    }
}
