/*
 * oracle/tile_tracker.c -- CPU restatement of the on-path kernels of the
 * reference's test_opencv/kernel.cu:116-891 (rows B1-B8, E2, E3, H1, H2, I1 of
 * SURVEY.md section 8a).
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle_common.h).
 */
#include <float.h>

#include "oracle_common.h"

/* B3: squaredSum, kernel.cu:117-143 */
void orc_squaredSum(const float* inTiles, float* outValues, int maxShift, int tileSize, int tileCount)
{
    const int L = tileSize + maxShift * 2;
#pragma omp parallel for schedule(static)
    for (int tileIdx = 0; tileIdx < tileCount; tileIdx++) {
        float sum = 0;
        size_t tileArray = (size_t)tileIdx * L * L;
        for (int y = 0; y < tileSize; y++) {
            int yShift = (y + maxShift) * L;
            for (int x = 0; x < tileSize; x++) {
                float pixel = inTiles[tileArray + yShift + x + maxShift];
                sum += pixel * pixel;
            }
        }
        outValues[tileIdx] = sum;
    }
}

/* B4a: boxFilterWithBorderX, kernel.cu:145-181 -- sliding sum of SQUARES over
 * [p - T/2, p + T/2 - 1] along x; zero outside columns [T/2, 2S + T/2]. */
void orc_boxFilterWithBorderX(const float* inTiles, float* outTiles, int maxShift, int tileSize, int tileCount)
{
    const int L = tileSize + maxShift * 2;
#pragma omp parallel for schedule(static)
    for (int tileIdx = 0; tileIdx < tileCount; tileIdx++) {
        for (int pxY = 0; pxY < L; pxY++) {
            const float* row = inTiles + (size_t)tileIdx * L * L + (size_t)pxY * L;
            float* orow = outTiles + (size_t)tileIdx * L * L + (size_t)pxY * L;
            for (int pxX = 0; pxX < L; pxX++) {
                float outVal = 0;
                if (pxX >= tileSize / 2 && pxX <= maxShift * 2 + tileSize / 2) {
                    for (int shift = -tileSize / 2; shift < tileSize / 2; shift++)
                        outVal += row[pxX + shift] * row[pxX + shift]; /* :177 */
                }
                orow[pxX] = outVal;
            }
        }
    }
}

/* B4b: boxFilterWithBorderY, kernel.cu:182-218 -- plain sliding sum along y. */
void orc_boxFilterWithBorderY(const float* inTiles, float* outTiles, int maxShift, int tileSize, int tileCount)
{
    const int L = tileSize + maxShift * 2;
#pragma omp parallel for schedule(static)
    for (int tileIdx = 0; tileIdx < tileCount; tileIdx++) {
        const float* tin = inTiles + (size_t)tileIdx * L * L;
        float* tout = outTiles + (size_t)tileIdx * L * L;
        for (int pxY = 0; pxY < L; pxY++) {
            for (int pxX = 0; pxX < L; pxX++) {
                float outVal = 0;
                if (pxY >= tileSize / 2 && pxY <= maxShift * 2 + tileSize / 2) {
                    for (int shift = -tileSize / 2; shift < tileSize / 2; shift++)
                        outVal += tin[(size_t)(pxY + shift) * L + pxX]; /* :214 */
                }
                tout[(size_t)pxY * L + pxX] = outVal;
            }
        }
    }
}

/* B6: normalizedCC, kernel.cu:221-259.  Guard uses '>' (:240), so the output
 * image is (2S+1) x (2S+1). */
void orc_normalizedCC(const float* ccImage, const float* squaredTemplate, const float* boxFilteredImage,
                      float* shiftImage, int maxShift, int tileSize, int tileCount)
{
    const int L = tileSize + maxShift * 2;
    const int R = maxShift * 2 + 1;
#pragma omp parallel for schedule(static)
    for (int tileIdx = 0; tileIdx < tileCount; tileIdx++) {
        for (int pxY = 0; pxY <= 2 * maxShift; pxY++) {
            for (int pxX = 0; pxX <= 2 * maxShift; pxX++) {
                int shiftX = pxX - maxShift;
                int shiftY = pxY - maxShift;
                int fftShiftX = shiftX, fftShiftY = shiftY;
                if (fftShiftX < 0) fftShiftX = L + shiftX; /* :248-251 */
                if (fftShiftY < 0) fftShiftY = L + shiftY;
                size_t pxInCCArray = (size_t)tileIdx * L * L + (size_t)fftShiftY * L + fftShiftX;
                size_t pxInBoxFilter = (size_t)tileIdx * L * L + (size_t)(L / 2 + shiftY) * L + (L / 2 + shiftX);
                size_t pxOut = (size_t)tileIdx * R * R + (size_t)pxY * R + pxX;
                shiftImage[pxOut] =
                    squaredTemplate[tileIdx] + boxFilteredImage[pxInBoxFilter] - 2 * ccImage[pxInCCArray]; /* :258 */
            }
        }
    }
}

/* common source-pixel computation of B1/B2: kernel.cu:299-313 / :358-372 */
static inline float orc_tile_fetch(const float* inImg, int imgWidth, int imgHeight, int imgPitch, int tileSize,
                                   int tileIdxX, int tileIdxY, int pxX, int pxY, of2 shift, float baseShiftX,
                                   float baseShiftY, float baseRotation)
{
    float sf = sinf(baseRotation);
    float cf = cosf(baseRotation);
    shift.x += cf * -baseShiftX - sf * -baseShiftY;
    shift.y += sf * -baseShiftX + cf * -baseShiftY;
    float patchCenterX = (float)(tileIdxX * tileSize + tileSize / 2 - imgWidth / 2);
    float patchCenterY = (float)(tileIdxY * tileSize + tileSize / 2 - imgHeight / 2);
    shift.x += cf * patchCenterX - sf * patchCenterY - patchCenterX;
    shift.y += sf * patchCenterX + cf * patchCenterY - patchCenterY;
    int pxInImgX = tileIdxX * tileSize + pxX + orc_f2i(roundf(shift.x));
    int pxInImgY = tileIdxY * tileSize + pxY + orc_f2i(roundf(shift.y));
    pxInImgX = orc_f2i(fminf(fmaxf((float)pxInImgX, 0), (float)(imgWidth - 1))); /* :312 (float clamp) */
    pxInImgY = orc_f2i(fminf(fmaxf((float)pxInImgY, 0), (float)(imgHeight - 1)));
    return ORC_CROW(float, inImg, imgPitch, pxInImgY)[pxInImgX];
}

/* B1: convertToTilesOverlapBorder, kernel.cu:261-318.  The reference starts
 * from shift = R(theta)*(-base) (no pre-shift); zero border of S. */
void orc_convertToTilesOverlapBorder(const float* inImg, float* outTiles, int imgWidth, int imgHeight, int imgPitch,
                                     int maxShift, int tileSize, int tileCountX, int tileCountY, float baseShiftX,
                                     float baseShiftY, float baseRotation)
{
    const int L = tileSize + maxShift * 2;
    const int tileCount = tileCountX * tileCountY;
#pragma omp parallel for schedule(static)
    for (int tileIdx = 0; tileIdx < tileCount; tileIdx++) {
        int tileIdxY = tileIdx / tileCountX;
        int tileIdxX = tileIdx - tileIdxY * tileCountX;
        for (int pxY = 0; pxY < L; pxY++) {
            for (int pxX = 0; pxX < L; pxX++) {
                size_t o = (size_t)tileIdx * L * L + (size_t)pxY * L + pxX;
                if (pxX < maxShift || pxY < maxShift || pxX >= tileSize + maxShift || pxY >= tileSize + maxShift) {
                    outTiles[o] = 0; /* :286-290 */
                    continue;
                }
                of2 z = {0.0f, 0.0f};
                outTiles[o] = orc_tile_fetch(inImg, imgWidth, imgHeight, imgPitch, tileSize, tileIdxX, tileIdxY, pxX, pxY,
                                             z, baseShiftX, baseShiftY, baseRotation);
            }
        }
    }
}

/* B2: convertToTilesOverlapPreShift, kernel.cu:320-378 */
void orc_convertToTilesOverlapPreShift(const float* inImg, float* outTiles, const of2* preShift, int preShiftPitch,
                                       int imgWidth, int imgHeight, int imgPitch, int maxShift, int tileSize,
                                       int tileCountX, int tileCountY, float baseShiftX, float baseShiftY,
                                       float baseRotation)
{
    const int L = tileSize + maxShift * 2;
    const int tileCount = tileCountX * tileCountY;
#pragma omp parallel for schedule(static)
    for (int tileIdx = 0; tileIdx < tileCount; tileIdx++) {
        int tileIdxY = tileIdx / tileCountX;
        int tileIdxX = tileIdx - tileIdxY * tileCountX;
        of2 shift = ORC_CROW(of2, preShift, preShiftPitch, tileIdxY)[tileIdxX]; /* :350-351 */
        for (int pxY = 0; pxY < L; pxY++) {
            for (int pxX = 0; pxX < L; pxX++) {
                size_t o = (size_t)tileIdx * L * L + (size_t)pxY * L + pxX;
                outTiles[o] = orc_tile_fetch(inImg, imgWidth, imgHeight, imgPitch, tileSize, tileIdxX, tileIdxY, pxX, pxY,
                                             shift, baseShiftX, baseShiftY, baseRotation);
            }
        }
    }
}

/* B5: conjugateComplexMulKernel, kernel.cu:484-501 */
void orc_conjugateComplexMulKernel(const of2* aIn, of2* bInOut, int maxElem)
{
    for (int idx = 0; idx < maxElem; idx++) {
        of2 valA = aIn[idx];
        valA.y = -valA.y;
        of2 valB = bInOut[idx];
        of2 res;
        res.x = valA.x * valB.x - valA.y * valB.y;
        res.y = valA.x * valB.y + valA.y * valB.x;
        bInOut[idx] = res;
    }
}

/* stencils, kernel.cu:503-507 */
static const float FA11[9] = {1.0f / 4.0f, -2.0f / 4.0f, 1.0f / 4.0f, 2.0f / 4.0f, -4.0f / 4.0f,
                              2.0f / 4.0f, 1.0f / 4.0f,  -2.0f / 4.0f, 1.0f / 4.0f};
static const float FA22[9] = {1.0f / 4.0f,  2.0f / 4.0f, 1.0f / 4.0f, -2.0f / 4.0f, -4.0f / 4.0f,
                              -2.0f / 4.0f, 1.0f / 4.0f, 2.0f / 4.0f, 1.0f / 4.0f};
static const float FA12[9] = {1.0f / 4.0f, 0.0f / 4.0f,  -1.0f / 4.0f, 0.0f / 4.0f, 0.0f / 4.0f,
                              0.0f / 4.0f, -1.0f / 4.0f, 0.0f / 4.0f,  1.0f / 4.0f};
static const float Fb1[9] = {-1.0f / 8.0f, 0.0f / 8.0f,  1.0f / 8.0f, -2.0f / 8.0f, 0.0f / 8.0f,
                             2.0f / 8.0f,  -1.0f / 8.0f, 0.0f / 8.0f, 1.0f / 8.0f};
static const float Fb2[9] = {-1.0f / 8.0f, -2.0f / 8.0f, -1.0f / 8.0f, 0.0f / 8.0f, 0.0f / 8.0f,
                             0.0f / 8.0f,  1.0f / 8.0f,  2.0f / 8.0f,  1.0f / 8.0f};

/* B7: findMinimum, kernel.cu:509-636 */
void orc_findMinimum(const float* shiftImage, of2* coordinates, int coordinatesPitch, int maxShift, int tileCount,
                     int tileCountX, float threshold)
{
    const int R = 2 * maxShift + 1;
#pragma omp parallel for schedule(static)
    for (int tileIdx = 0; tileIdx < tileCount; tileIdx++) {
        int pixelsInTile = R * R;
        size_t zOffset = (size_t)tileIdx * pixelsInTile;
        float minVal = FLT_MAX, maxVal = -FLT_MAX;
        int minIdx = -1;
        for (int i = 0; i < pixelsInTile; i++) {
            float val = shiftImage[zOffset + i];
            maxVal = fmaxf(maxVal, val);
            if (val < minVal) {
                minVal = val;
                minIdx = i;
            }
        }
        of2 coord;
        coord.y = (float)(minIdx / R); /* :544 (C integer division, minIdx may be -1 -> 0) */
        coord.x = (float)minIdx - coord.y * (float)R;

        if (coord.x < 1 || coord.y < 1 || coord.x >= 2 * maxShift || coord.y >= 2 * maxShift) {
            coord.x = 0;
            coord.y = 0;
        } else {
            float A11 = 0, A22 = 0, A12 = 0, b1 = 0, b2 = 0;
            for (int i = 0; i < 9; i++) {
                int r = i / 3, c = i % 3;
                float img = shiftImage[zOffset + minIdx + (c - 1) + (r - 1) * R]; /* :566,:575,:584 */
                A11 += FA11[i] * img;
                A22 += FA22[i] * img;
                A12 += FA12[i] * img;
                b1 += Fb1[i] * img;
                b2 += Fb2[i] * img;
            }
            A11 = fmaxf(A11, 0.0f);
            A22 = fmaxf(A22, 0.0f);
            float detA = A11 * A22 - A12 * A12;
            if (detA < 0) {
                A12 = 0;
                detA = A11 * A22;
            }
            if (detA != 0) {
                float muX = (A22 * b1 - A12 * b2) / detA;
                float muY = (A11 * b2 - A12 * b1) / detA;
                if (fabsf(muX) > 1) muX = 0;
                if (fabsf(muY) > 1) muY = 0;
                coord.x -= muX;
                coord.y -= muY;
            }
            coord.x -= (float)maxShift;
            coord.y -= (float)maxShift;
        }
        int tileIdxY = tileIdx / tileCountX;
        int tileIdxX = tileIdx - tileIdxY * tileCountX;
        if (threshold + minVal > maxVal) { /* :629-633 */
            coord.x = 0;
            coord.y = 0;
        }
        ORC_ROW(of2, coordinates, coordinatesPitch, tileIdxY)[tileIdxX] = coord;
    }
}

/* B8: UpSampleShifts, kernel.cu:641-688 */
void orc_UpSampleShifts(const of2* inShift, of2* outShift, int inPitch, int outPitch, int oldLevel, int newLevel,
                        int oldCountX, int oldCountY, int newCountX, int newCountY, int oldTileSize, int newTileSize)
{
    for (int newBlockY = 0; newBlockY < newCountY; newBlockY++) {
        for (int newBlockX = 0; newBlockX < newCountX; newBlockX++) {
            float factor = (float)oldLevel * (float)oldTileSize / (float)(newLevel * newTileSize); /* :652 */
            float oldX = (float)newBlockX / factor;
            float oldY = (float)newBlockY / factor;
            int oldXMin = orc_f2i(floorf(oldX));
            int oldXMax = orc_f2i(ceilf(oldX));
            int oldYMin = orc_f2i(floorf(oldY));
            int oldYMax = orc_f2i(ceilf(oldY));
            oldXMin = orc_imin(oldXMin, oldCountX - 1);
            oldXMax = orc_imin(oldXMax, oldCountX - 1);
            oldYMin = orc_imin(oldYMin, oldCountY - 1);
            oldYMax = orc_imin(oldYMax, oldCountY - 1);
            of2 oldMinMin = ORC_CROW(of2, inShift, inPitch, oldYMin)[oldXMin];
            of2 oldMaxMin = ORC_CROW(of2, inShift, inPitch, oldYMin)[oldXMax];
            of2 oldMinMax = ORC_CROW(of2, inShift, inPitch, oldYMax)[oldXMin];
            of2 oldMaxMax = ORC_CROW(of2, inShift, inPitch, oldYMax)[oldXMax];
            float wx = 1.0f - ((float)oldXMax - oldX); /* :676 */
            float wy = 1.0f - ((float)oldYMax - oldY);
            float temp1 = oldMinMin.x + (oldMaxMin.x - oldMinMin.x) * wx;
            float temp2 = oldMinMax.x + (oldMaxMax.x - oldMinMax.x) * wx;
            of2 old;
            old.x = temp1 + (temp2 - temp1) * wy;
            temp1 = oldMinMin.y + (oldMaxMin.y - oldMinMin.y) * wx;
            temp2 = oldMinMax.y + (oldMaxMax.y - oldMinMax.y) * wx;
            old.y = temp1 + (temp2 - temp1) * wy;
            old.x *= (float)oldLevel / (float)newLevel; /* :684 */
            old.y *= (float)oldLevel / (float)newLevel;
            ORC_ROW(of2, outShift, outPitch, newBlockY)[newBlockX] = old;
        }
    }
}

/* E2: ComputeStructureTensor, kernel.cu:690-715 */
void orc_ComputeStructureTensor(const float* imgDx, const float* imgDy, of3* outImg, int imgWidth, int imgHeight,
                                int imgDxDyPitch, int imgOutPitch)
{
#pragma omp parallel for schedule(static)
    for (int pxY = 0; pxY < imgHeight; pxY++) {
        for (int pxX = 0; pxX < imgWidth; pxX++) {
            float dx = ORC_CROW(float, imgDx, imgDxDyPitch, pxY)[pxX];
            float dy = ORC_CROW(float, imgDy, imgDxDyPitch, pxY)[pxX];
            of3 val = {dx * dx, dy * dy, dx * dy};
            ORC_ROW(of3, outImg, imgOutPitch, pxY)[pxX] = val;
        }
    }
}

/* E3: ComputeKernelParam, kernel.cu:717-790 */
void orc_ComputeKernelParam(of3* kernelImg, int imgWidth, int imgHeight, int imgOutPitch, float Dth, float Dtr,
                            float kDetail, float kDenoise, float kStretch, float kShrink)
{
#pragma omp parallel for schedule(static)
    for (int pxY = 0; pxY < imgHeight; pxY++) {
        for (int pxX = 0; pxX < imgWidth; pxX++) {
            of3 grad = ORC_ROW(of3, kernelImg, imgOutPitch, pxY)[pxX];
            float a11 = grad.x, a22 = grad.y, a12 = grad.z;
            float help = sqrtf((a22 - a11) * (a22 - a11) + 4.0f * a12 * a12); /* :741 */
            float c = 2.0f * a12;
            float s = a22 - a11 + help;
            float norm = sqrtf(c * c + s * s);
            if (norm > 0) {
                c /= norm;
                s /= norm;
            } else {
                c = 1;
                s = 0;
            }
            float lam1 = (a11 + a22 + help) / 2.0f;
            float lam2 = (a11 + a22 - help) / 2.0f;
            float A = 1 + sqrtf((lam1 - lam2) * (lam1 - lam2) / ((lam1 + lam2) * (lam1 + lam2))); /* :761 */
            float D = 1 - sqrtf(lam1) / Dtr + Dth;                                              /* :762 */
            D = fmaxf(fminf(1.0f, D), 0.0f);
            float k1h = kDetail * kStretch * A;
            float k2h = kDetail / kShrink * A;
            float k1 = ((1.0f - D) * k1h + D * kDetail * kDenoise);
            float k2 = ((1.0f - D) * k2h + D * kDetail * kDenoise);
            k1 *= k1;
            k2 *= k2;
            float x2 = c, y2 = s, x1 = s, y1 = -c;
            float b11 = k1 * x1 * x1 + x2 * x2 * k2; /* :779-781 */
            float b12 = k1 * x1 * y1 + x2 * y2 * k2;
            float b22 = k1 * y1 * y1 + y2 * y2 * k2;
            float det = b11 * b22 - b12 * b12 + 0.0000000001f;
            of3 kernel = {b22 / det, b11 / det, -b12 / det};
            ORC_ROW(of3, kernelImg, imgOutPitch, pxY)[pxX] = kernel;
        }
    }
}

/* H2: applysRGBGamma + GammasRGB, kernel.cu:380-422 */
static inline float orc_applysRGBGamma(float valIn)
{
    if (valIn <= 0.0031308f) return 12.92f * valIn;
    return (1.0f + 0.055f) * powf(valIn, 1.0f / 2.4f) - 0.055f;
}

void orc_GammasRGB(of3* inOutImg, int imgWidth, int imgHeight, int imgPitch)
{
#pragma omp parallel for schedule(static)
    for (int pxY = 0; pxY < imgHeight; pxY++) {
        for (int pxX = 0; pxX < imgWidth; pxX++) {
            of3 val = ORC_ROW(of3, inOutImg, imgPitch, pxY)[pxX];
            if (isnan(val.x)) val.x = 0;
            if (isnan(val.y)) val.y = 0;
            if (isnan(val.z)) val.z = 0;
            val.x = fmaxf(fminf(val.x, 1.0f), 0.0f);
            val.y = fmaxf(fminf(val.y, 1.0f), 0.0f);
            val.z = fmaxf(fminf(val.z, 1.0f), 0.0f);
            val.x = orc_applysRGBGamma(val.x);
            val.y = orc_applysRGBGamma(val.y);
            val.z = orc_applysRGBGamma(val.z);
            ORC_ROW(of3, inOutImg, imgPitch, pxY)[pxX] = val;
        }
    }
}

/* H1: ApplyWeighting, kernel.cu:425-481 */
void orc_ApplyWeighting(of3* inOutImg, const of3* finalImg, const of3* weight, int imgWidth, int imgHeight, int imgPitch,
                        float threshold)
{
#pragma omp parallel for schedule(static)
    for (int pxY = 0; pxY < imgHeight; pxY++) {
        for (int pxX = 0; pxX < imgWidth; pxX++) {
            of3 inout = ORC_ROW(of3, inOutImg, imgPitch, pxY)[pxX];
            of3 val = ORC_CROW(of3, finalImg, imgPitch, pxY)[pxX];
            of3 w = ORC_CROW(of3, weight, imgPitch, pxY)[pxX];
            float* io = &inout.x;
            float* v = &val.x;
            float* ww = &w.x;
            for (int c = 0; c < 3; c++) {
                if (ww[c] < threshold) {
                    v[c] += io[c];
                    ww[c] += 1;
                }
                io[c] = 0;
                if (ww[c] != 0) io[c] = v[c] / ww[c];
            }
            ORC_ROW(of3, inOutImg, imgPitch, pxY)[pxX] = inout;
        }
    }
}

/* I1a: fourierFilter, kernel.cu:792-869 */
void orc_fourierFilter(of2* img, size_t stride, int width, int height, float lp, float hp, float lps, float hps,
                       int clearAxis)
{
    lp = lp - lps; /* :815-816 (per thread in the reference; same value for all) */
    hp = hp + hps;
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width / 2 + 1; x++) {
            float mx = (float)x;
            float my = (float)y;
            if (my > (float)height * 0.5f) my = ((float)height - my) * -1.0f;
            mx /= (float)width;
            my /= (float)height;
            float dist = sqrtf(mx * mx + my * my);
            float fil = 0;
            if (lp > 0) {
                if (dist <= lp) fil = 1;
            } else {
                if (dist <= 1.0f) fil = 1;
            }
            if (lps > 0) {
                float fil2 = (-fil + 1.0f) * expf(-((dist - lp) * (dist - lp) / (2 * lps * lps))); /* :834 */
                if (fil2 > 0.001f) fil = fil2;
            }
            if (lps > 0 && lp == 0 && hp == 0 && hps == 0) fil = expf(-((dist - lp) * (dist - lp) / (2 * lps * lps)));
            if (hp > 0) {
                float fil2 = 0;
                if (dist >= hp) fil2 = 1;
                fil *= fil2;
                if (hps > 0) {
                    float fil3 = (-fil2 + 1.0f) * expf(-((dist - hp) * (dist - hp) / (2 * hps * hps))); /* :853 */
                    if (fil3 > 0.001f) fil = fil3;
                }
            }
            of2* row = (of2*)((char*)img + stride * (size_t)y);
            of2 erg = row[x];
            erg.x *= fil;
            erg.y *= fil;
            if (x < clearAxis || fabsf(my) * (float)height < (float)clearAxis) { /* :863 */
                erg.x = 0;
                erg.y = 0;
            }
            row[x] = erg;
        }
    }
}

/* I1b: fftshift, kernel.cu:871-891 */
void orc_fftshift(of2* fft, int width, int height)
{
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            int mx = x - width / 2;
            int my = y - height / 2;
            float a = 1.0f - (float)(2 * (((mx + my) & 1)));
            of2 erg = fft[(size_t)y * width + x];
            erg.x *= a;
            erg.y *= a;
            fft[(size_t)y * width + x] = erg;
        }
    }
}
