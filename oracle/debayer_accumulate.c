/*
 * oracle/debayer_accumulate.c -- CPU restatement of the reference's
 * test_opencv/DeBayerKernels.cu (rows A0-A3, G1, G2 of SURVEY.md section 8a).
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle_common.h).
 *
 * One C function per reference kernel, same argument order and units (byte
 * pitches); the CUDA grid is replaced by plain loops over the guarded index
 * range.  Texture objects become (ptr, pitch, w, h) descriptors.
 */
#include "oracle_common.h"

int orc_cfa[2][2] = {{ORC_RED, ORC_GREEN}, {ORC_GREEN, ORC_BLUE}};

/* A0: c_cfaPattern, DeBayerKernels.cu:40-41 */
void orc_set_cfa_pattern(const int32_t* p)
{
    orc_cfa[0][0] = p[0];
    orc_cfa[0][1] = p[1];
    orc_cfa[1][0] = p[2];
    orc_cfa[1][1] = p[3];
}

/* A1: deBayersSubSample3, DeBayerKernels.cu:242-283 */
void orc_deBayersSubSample3(const uint16_t* dataIn, of3* imgOut, float maxVal, int dimX, int dimY, int strideOut)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dimY; y++) {
        of3* lineOut = ORC_ROW(of3, imgOut, strideOut, y);
        float factor = 1.0f / maxVal; /* :257 */
        for (int x = 0; x < dimX; x++) {
            of3 pixel = {0, 0, 0};
            for (int ix = 0; ix < 2; ix++) {
                for (int iy = 0; iy < 2; iy++) {
                    int c = orc_cfa[iy][ix];
                    /* RAW2(x,y) = dataIn[y*dimX*2 + x], :242 */
                    float raw = (float)dataIn[(size_t)(2 * y + iy) * (size_t)dimX * 2 + (size_t)(2 * x + ix)];
                    if (c == ORC_GREEN)
                        pixel.y += raw * factor * 0.5f; /* :267 */
                    else if (c == ORC_RED)
                        pixel.x = raw * factor; /* :272 */
                    else if (c == ORC_BLUE)
                        pixel.z = raw * factor; /* :277 */
                }
            }
            lineOut[x] = pixel;
        }
    }
}

/* A2: deBayerGreenKernel, DeBayerKernels.cu:55-149.  The 2-px border ring of
 * outImage is left untouched (:61-62). */
void orc_deBayerGreenKernel(int width, int height, const float* imgIn, int strideIn, of3* outImage, int strideOut,
                            const float* blackPoint, const float* scale)
{
#define RAW(xx, yy) (ORC_CROW(float, imgIn, strideIn, (yy))[(xx)])
#define RAWC(xx, yy, c) ((RAW(xx, yy) - blackPoint[c]) * scale[c]) /* :44-46 */
#pragma omp parallel for schedule(static)
    for (int y = 2; y < height - 2; y++) {
        for (int x = 2; x < width - 2; x++) {
            int thisPixel = orc_cfa[y % 2][x % 2];
            float g = 0;
            if (thisPixel == ORC_GREEN) {
                g = RAWC(x, y, 1);
            } else if (thisPixel == ORC_RED || thisPixel == ORC_BLUE) {
                int c = (thisPixel == ORC_RED) ? 0 : 2;
                float p = RAWC(x, y, c);
                float xMinus2 = RAWC(x - 2, y, c);
                float xMinus1 = RAWC(x - 1, y, 1);
                float xPlus1 = RAWC(x + 1, y, 1);
                float xPlus2 = RAWC(x + 2, y, c);
                float yMinus2 = RAWC(x, y - 2, c);
                float yMinus1 = RAWC(x, y - 1, 1);
                float yPlus1 = RAWC(x, y + 1, 1);
                float yPlus2 = RAWC(x, y + 2, c);
                float gradientX = 0.5f * fabsf(xPlus1 - xMinus1); /* :106 */
                float gradientY = 0.5f * fabsf(yPlus1 - yMinus1);
                float laplaceX = 0.25f * fabsf(2.0f * p - xMinus2 - xPlus2); /* :109 */
                float laplaceY = 0.25f * fabsf(2.0f * p - yMinus2 - yPlus2);
                float interpolX = 0.125f * (-xMinus2 + 4.0f * xMinus1 + 2.0f * p + 4.0f * xPlus1 - xPlus2); /* :112 */
                float interpolY = 0.125f * (-yMinus2 + 4.0f * yMinus1 + 2.0f * p + 4.0f * yPlus1 - yPlus2);
                float weight =
                    (gradientY + laplaceY) / (gradientX + gradientY + laplaceX + laplaceY + 0.000000001f); /* :115 */
                g = weight * interpolX + (1.0f - weight) * interpolY; /* :117 */
            }
            ORC_ROW(of3, outImage, strideOut, y)[x].y = g; /* :148 */
        }
    }
#undef RAWC
#undef RAW
}

/* A3: deBayerRedBlueKernel, DeBayerKernels.cu:153-231 (needs A2 complete). */
void orc_deBayerRedBlueKernel(int width, int height, const float* imgIn, int strideIn, of3* outImage, int strideOut,
                              const float* blackPoint, const float* scale)
{
#define RAW(xx, yy) (ORC_CROW(float, imgIn, strideIn, (yy))[(xx)])
#define RAWR(xx, yy) ((RAW(xx, yy) - blackPoint[0]) * scale[0])
#define RAWB(xx, yy) ((RAW(xx, yy) - blackPoint[2]) * scale[2])
#define GREEN(xx, yy) (ORC_ROW(of3, outImage, strideOut, (yy))[(xx)].y)
#pragma omp parallel for schedule(static)
    for (int y = 2; y < height - 2; y++) {
        for (int x = 2; x < width - 2; x++) {
            int thisPixel = orc_cfa[y % 2][x % 2];
            int thisRow = orc_cfa[y % 2][(x + 1) % 2];
            /* the reference leaves r,b uninitialised for non-RGB CFA colours */
            of3 cur = ORC_ROW(of3, outImage, strideOut, y)[x];
            float r = cur.x, b = cur.z;
            float g = GREEN(x, y);
            if (thisPixel == ORC_GREEN) {
                if (thisRow == ORC_RED) { /* :170-183 */
                    r = g + 0.5f * ((RAWR(x - 1, y) - GREEN(x - 1, y)) + (RAWR(x + 1, y) - GREEN(x + 1, y)));
                    b = g + 0.5f * ((RAWB(x, y - 1) - GREEN(x, y - 1)) + (RAWB(x, y + 1) - GREEN(x, y + 1)));
                } else { /* :184-197 */
                    b = g + 0.5f * ((RAWB(x - 1, y) - GREEN(x - 1, y)) + (RAWB(x + 1, y) - GREEN(x + 1, y)));
                    r = g + 0.5f * ((RAWR(x, y - 1) - GREEN(x, y - 1)) + (RAWR(x, y + 1) - GREEN(x, y + 1)));
                }
            } else if (thisPixel == ORC_RED) { /* :199-212 */
                r = RAWR(x, y);
                b = g + 0.25f * ((((RAWB(x - 1, y - 1) - GREEN(x - 1, y - 1)) + (RAWB(x + 1, y - 1) - GREEN(x + 1, y - 1))) +
                                  (RAWB(x + 1, y + 1) - GREEN(x + 1, y + 1))) +
                                 (RAWB(x - 1, y + 1) - GREEN(x - 1, y + 1)));
            } else if (thisPixel == ORC_BLUE) { /* :213-226 */
                b = RAWB(x, y);
                r = g + 0.25f * ((((RAWR(x - 1, y - 1) - GREEN(x - 1, y - 1)) + (RAWR(x + 1, y - 1) - GREEN(x + 1, y - 1))) +
                                  (RAWR(x + 1, y + 1) - GREEN(x + 1, y + 1))) +
                                 (RAWR(x - 1, y + 1) - GREEN(x - 1, y + 1)));
            }
            ORC_ROW(of3, outImage, strideOut, y)[x].x = r; /* :229 */
            ORC_ROW(of3, outImage, strideOut, y)[x].z = b; /* :230 */
        }
    }
#undef GREEN
#undef RAWB
#undef RAWR
#undef RAW
}

/* shared tap body of G1/G2: DeBayerKernels.cu:335-370 and :427-462 */
static inline void orc_accum_tap(int px, int py, float kx, float ky, float kz, float raw, int color, of4 cert4,
                                 const float* whiteLevel, const float* blackLevel, of3* pixel, of3* totalWeight)
{
    float w = (float)(px * px) * kx + (float)(2 * px * py) * kz + (float)(py * py) * ky; /* :335 / :427 */
    w = expf(-0.5f * w);
    if (!isfinite(w)) w = (px * py == 0) ? 1.0f : 0.0f; /* :337-338 */
    if (color == ORC_GREEN) {
        raw = (raw - blackLevel[1]) / whiteLevel[1];
        float certainty = cert4.y;
        if (!isfinite(certainty)) certainty = 0;
        pixel->y += raw * w * certainty;
        totalWeight->y += w * certainty;
    } else if (color == ORC_RED) {
        raw = (raw - blackLevel[0]) / whiteLevel[0];
        float certainty = cert4.x;
        if (!isfinite(certainty)) certainty = 0;
        pixel->x += raw * w * certainty;
        totalWeight->x += w * certainty;
    } else if (color == ORC_BLUE) {
        raw = (raw - blackLevel[2]) / whiteLevel[2];
        float certainty = cert4.z;
        if (!isfinite(certainty)) certainty = 0;
        pixel->z += raw * w * certainty;
        totalWeight->z += w * certainty;
    }
}

/* G1: accumulateImages, DeBayerKernels.cu:288-376 (x1 merge).  Quirk kept:
 * kernelParam rows are addressed with strideOut (:308). */
void orc_accumulateImages(const uint16_t* dataIn, of3* imgOut, of3* totalWeights, const of4* certaintyMask,
                          const of3* kernelParam, const of2* shifts, const float* whiteLevel, const float* blackLevel,
                          int dimX, int dimY, int strideOut, int strideMask, int strideShift)
{
#pragma omp parallel for schedule(static)
    for (int y = 1; y < dimY - 1; y++) {
        for (int x = 1; x < dimX - 1; x++) {
            of3 pixel = ORC_ROW(of3, imgOut, strideOut, y)[x];
            of3 totalWeight = ORC_ROW(of3, totalWeights, strideOut, y)[x];
            of3 kernel = ORC_CROW(of3, kernelParam, strideOut, y)[x];
            of2 shift = ORC_CROW(of2, shifts, strideShift, y)[x];
            int sx = orc_f2i(roundf(shift.x)); /* :310-313 */
            int sy = orc_f2i(roundf(shift.y));
            for (int py = -2; py <= 2; py++) {
                for (int px = -2; px <= 2; px++) {
                    int ppsx = orc_imin(orc_imax(x + px + sx, 0), dimX - 1);
                    int ppsy = orc_imin(orc_imax(y + py + sy, 0), dimY - 1);
                    int ppx = orc_imin(orc_imax(x + px, 0), dimX - 1);
                    int ppy = orc_imin(orc_imax(y + py, 0), dimY - 1);
                    int color = orc_cfa[ppsy % 2][ppsx % 2];
                    float raw = (float)dataIn[(size_t)ppsy * (size_t)dimX + (size_t)ppsx]; /* RAW, :288 */
                    of4 cert4 = ORC_CROW(of4, certaintyMask, strideMask, ppy / 2)[ppx / 2];
                    orc_accum_tap(px, py, kernel.x, kernel.y, kernel.z, raw, color, cert4, whiteLevel, blackLevel, &pixel,
                                  &totalWeight);
                }
            }
            ORC_ROW(of3, imgOut, strideOut, y)[x] = pixel;
            ORC_ROW(of3, totalWeights, strideOut, y)[x] = totalWeight;
        }
    }
}

/* G2: accumulateImagesSuperRes, DeBayerKernels.cu:378-468.  x2 merge onto an
 * output grid of the raw frame's size that covers the central half of the
 * frame.  kernelParam (float4) and shifts (float2) are textures (CLAMP). */
void orc_accumulateImagesSuperRes(const uint16_t* dataIn, of3* imgOut, of3* totalWeights, const of4* certaintyMask,
                                  const void* kpPtr, int kpPitch, int kpW, int kpH, const void* shPtr, int shPitch,
                                  int shW, int shH, const float* whiteLevel, const float* blackLevel, int dimX, int dimY,
                                  int strideOut, int strideMask)
{
    orc_tex texK = {kpPtr, kpPitch, kpW, kpH, ORC_ADDR_CLAMP};
    orc_tex texS = {shPtr, shPitch, shW, shH, ORC_ADDR_CLAMP};
#pragma omp parallel for schedule(static)
    for (int y = 1; y < dimY - 1; y++) {
        for (int x = 1; x < dimX - 1; x++) {
            of3 pixel = ORC_ROW(of3, imgOut, strideOut, y)[x];
            of3 totalWeight = ORC_ROW(of3, totalWeights, strideOut, y)[x];
            float posX = ((float)x + 0.5f + (float)(dimX / 2)) / 2.0f / (float)dimX; /* :398 */
            float posY = ((float)y + 0.5f + (float)(dimY / 2)) / 2.0f / (float)dimY;
            of4 kernel = orc_tex4(&texK, posX, posY);
            of2 shift = orc_tex2(&texS, posX, posY);
            int sx = orc_f2i(roundf(shift.x * 2)); /* :403-406 */
            int sy = orc_f2i(roundf(shift.y * 2));
            for (int py = -2; py <= 2; py++) {
                for (int px = -2; px <= 2; px++) {
                    int ppsx = x + px + sx + dimX / 2; /* :414-417 */
                    int ppsy = y + py + sy + dimY / 2;
                    int ppx = x + px + dimX / 2;
                    int ppy = y + py + dimY / 2;
                    ppsx = orc_imin(orc_imax(ppsx / 2, 0 + dimX / 4), dimX / 2 - 1 + dimX / 4); /* :419-423 */
                    ppsy = orc_imin(orc_imax(ppsy / 2, 0 + dimY / 4), dimY / 2 - 1 + dimY / 4);
                    ppx = orc_imin(orc_imax(ppx / 2, 0 + dimX / 4), dimX / 2 - 1 + dimX / 4);
                    ppy = orc_imin(orc_imax(ppy / 2, 0 + dimY / 4), dimY / 2 - 1 + dimY / 4);
                    int color = orc_cfa[ppsy % 2][ppsx % 2];
                    float raw = (float)dataIn[(size_t)ppsy * (size_t)dimX + (size_t)ppsx];
                    of4 cert4 = ORC_CROW(of4, certaintyMask, strideMask, ppy / 2)[ppx / 2];
                    orc_accum_tap(px, py, kernel.x, kernel.y, kernel.z, raw, color, cert4, whiteLevel, blackLevel, &pixel,
                                  &totalWeight);
                }
            }
            ORC_ROW(of3, imgOut, strideOut, y)[x] = pixel;
            ORC_ROW(of3, totalWeights, strideOut, y)[x] = totalWeight;
        }
    }
}

/* floor division for the generalised geometry (negative numerators occur) */
static inline int orc_floordiv(int a, int b)
{
    int q = a / b, r = a % b;
    return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}

/* G2 generalised (SURVEY.md Appendix A "Generalisation to scale s, full
 * frame"; no reference line -- it is the build's extension of :378-468):
 * output grid = (s*dimX) x (s*dimY) covering the whole frame.  HR pixel (X,Y),
 * tap (px,py): raw site floor((X+px+round(s*u))/s) clamped to the frame,
 * certainty site floor((X+px)/s)/2, fields sampled at ((X+.5)/(s*dimX), ..).
 * The 1-px border ring of the HR grid is skipped like the reference's. */
void orc_accumulateSuperResFull(const uint16_t* dataIn, of3* imgOut, of3* totalWeights, const of4* certaintyMask,
                                const void* kpPtr, int kpPitch, int kpW, int kpH, const void* shPtr, int shPitch,
                                int shW, int shH, const float* whiteLevel, const float* blackLevel, int dimX, int dimY,
                                int scale, int strideOut, int strideMask)
{
    orc_tex texK = {kpPtr, kpPitch, kpW, kpH, ORC_ADDR_CLAMP};
    orc_tex texS = {shPtr, shPitch, shW, shH, ORC_ADDR_CLAMP};
    const int hrW = dimX * scale, hrH = dimY * scale;
#pragma omp parallel for schedule(static)
    for (int y = 1; y < hrH - 1; y++) {
        for (int x = 1; x < hrW - 1; x++) {
            of3 pixel = ORC_ROW(of3, imgOut, strideOut, y)[x];
            of3 totalWeight = ORC_ROW(of3, totalWeights, strideOut, y)[x];
            float posX = ((float)x + 0.5f) / (float)hrW;
            float posY = ((float)y + 0.5f) / (float)hrH;
            of4 kernel = orc_tex4(&texK, posX, posY);
            of2 shift = orc_tex2(&texS, posX, posY);
            int sx = orc_f2i(roundf(shift.x * (float)scale));
            int sy = orc_f2i(roundf(shift.y * (float)scale));
            for (int py = -2; py <= 2; py++) {
                for (int px = -2; px <= 2; px++) {
                    int ppsx = orc_imin(orc_imax(orc_floordiv(x + px + sx, scale), 0), dimX - 1);
                    int ppsy = orc_imin(orc_imax(orc_floordiv(y + py + sy, scale), 0), dimY - 1);
                    int ppx = orc_imin(orc_imax(orc_floordiv(x + px, scale), 0), dimX - 1);
                    int ppy = orc_imin(orc_imax(orc_floordiv(y + py, scale), 0), dimY - 1);
                    int color = orc_cfa[ppsy % 2][ppsx % 2];
                    float raw = (float)dataIn[(size_t)ppsy * (size_t)dimX + (size_t)ppsx];
                    of4 cert4 = ORC_CROW(of4, certaintyMask, strideMask, ppy / 2)[ppx / 2];
                    orc_accum_tap(px, py, kernel.x, kernel.y, kernel.z, raw, color, cert4, whiteLevel, blackLevel, &pixel,
                                  &totalWeight);
                }
            }
            ORC_ROW(of3, imgOut, strideOut, y)[x] = pixel;
            ORC_ROW(of3, totalWeights, strideOut, y)[x] = totalWeight;
        }
    }
}
