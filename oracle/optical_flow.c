/*
 * oracle/optical_flow.c -- CPU restatement of the reference's
 * test_opencv/opticalFlow.cu (rows D1-D4, E1 of SURVEY.md section 8a).
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle_common.h).
 *
 * Texture conventions (the reference's host code is absent, so these are the
 * build's canonical choices): images are sampled MIRROR + linear,
 * flow / tile-shift fields CLAMP + linear, normalised coordinates.
 */
#include "oracle_common.h"

/* D2: WarpingKernel, opticalFlow.cu:27-44 */
void orc_WarpingKernel(int width, int height, int stride, const void* uvPtr, int uvPitch, int uvW, int uvH, float* out,
                       const void* imgPtr, int imgPitch, int imgW, int imgH)
{
    orc_tex texUV = {uvPtr, uvPitch, uvW, uvH, ORC_ADDR_CLAMP};
    orc_tex texToWarp = {imgPtr, imgPitch, imgW, imgH, ORC_ADDR_MIRROR};
#pragma omp parallel for schedule(static)
    for (int iy = 0; iy < height; iy++) {
        for (int ix = 0; ix < width; ix++) {
            of2 shift = orc_tex2(&texUV, ((float)ix + 0.5f) / (float)width, ((float)iy + 0.5f) / (float)height); /* :36 */
            float x = ((float)ix + 0.5f + shift.x) / (float)width; /* :38 */
            float y = ((float)iy + 0.5f + shift.y) / (float)height;
            ORC_ROW(float, out, stride, iy)[ix] = orc_tex1(&texToWarp, x, y); /* :41-43 */
        }
    }
}

/* D1: CreateFlowFieldFromTiles, opticalFlow.cu:47-93 */
void orc_CreateFlowFieldFromTiles(of2* outImg, const void* tsPtr, int tsPitch, int tsW, int tsH, int tileSize,
                                  int tileCountX, int tileCountY, int imgWidth, int imgHeight, int imgPitch,
                                  float baseShiftX, float baseShiftY, float baseRotation)
{
    orc_tex texShift = {tsPtr, tsPitch, tsW, tsH, ORC_ADDR_CLAMP};
    (void)tileSize;
    (void)tileCountX;
    (void)tileCountY; /* tileIdx (:66-72) is computed but unused by the reference */
#pragma omp parallel for schedule(static)
    for (int pxY = 0; pxY < imgHeight; pxY++) {
        for (int pxX = 0; pxX < imgWidth; pxX++) {
            of2 shift;
            shift.x = cosf(baseRotation) * -baseShiftX - sinf(baseRotation) * -baseShiftY; /* :78 */
            shift.y = sinf(baseRotation) * -baseShiftX + cosf(baseRotation) * -baseShiftY; /* :79 */
            float patchCenterX = (float)(pxX - imgWidth / 2); /* :81 */
            float patchCenterY = (float)(pxY - imgHeight / 2);
            shift.x += cosf(baseRotation) * patchCenterX - sinf(baseRotation) * patchCenterY - patchCenterX; /* :84 */
            shift.y += sinf(baseRotation) * patchCenterX + cosf(baseRotation) * patchCenterY - patchCenterY; /* :85 */
            of2 shiftPatch =
                orc_tex2(&texShift, ((float)pxX + 0.5f) / (float)imgWidth, ((float)pxY + 0.5f) / (float)imgHeight); /* :88 */
            shift.x += shiftPatch.x;
            shift.y += shiftPatch.y;
            ORC_ROW(of2, outImg, imgPitch, pxY)[pxX] = shift;
        }
    }
}

/* 5-point derivative used by D3/E1: opticalFlow.cu:116-120 (x) / :134-138 (y) */
static inline float orc_deriv5(const orc_tex* t, float x, float y, float dx, float dy)
{
    float t0 = orc_tex1(t, x + 2.0f * dx, y + 2.0f * dy);
    t0 -= orc_tex1(t, x + 1.0f * dx, y + 1.0f * dy) * 8.0f;
    t0 += orc_tex1(t, x - 1.0f * dx, y - 1.0f * dy) * 8.0f;
    t0 -= orc_tex1(t, x - 2.0f * dx, y - 2.0f * dy);
    t0 /= 12.0f;
    return t0;
}

/* D3: ComputeDerivativesKernel, opticalFlow.cu:96-147 */
void orc_ComputeDerivativesKernel(int width, int height, int stride, float* Ix, float* Iy, float* Iz, const void* srcPtr,
                                  int srcPitch, int srcW, int srcH, const void* tgtPtr, int tgtPitch, int tgtW, int tgtH)
{
    orc_tex texSource = {srcPtr, srcPitch, srcW, srcH, ORC_ADDR_MIRROR};
    orc_tex texTarget = {tgtPtr, tgtPitch, tgtW, tgtH, ORC_ADDR_MIRROR};
#pragma omp parallel for schedule(static)
    for (int iy = 0; iy < height; iy++) {
        for (int ix = 0; ix < width; ix++) {
            float dx = 1.0f / (float)width;
            float dy = 1.0f / (float)height;
            float x = ((float)ix + 0.5f) * dx;
            float y = ((float)iy + 0.5f) * dy;
            float t0 = orc_deriv5(&texSource, x, y, dx, 0.0f);
            float t1 = orc_deriv5(&texTarget, x, y, dx, 0.0f);
            ORC_ROW(float, Ix, stride, iy)[ix] = (t0 + t1) * 0.5f;                                       /* :128 */
            ORC_ROW(float, Iz, stride, iy)[ix] = orc_tex1(&texSource, x, y) - orc_tex1(&texTarget, x, y); /* :131 */
            t0 = orc_deriv5(&texSource, x, y, 0.0f, dy);
            t1 = orc_deriv5(&texTarget, x, y, 0.0f, dy);
            ORC_ROW(float, Iy, stride, iy)[ix] = (t0 + t1) * 0.5f; /* :146 */
        }
    }
}

/* E1: ComputeDerivatives2Kernel, opticalFlow.cu:150-185 */
void orc_ComputeDerivatives2Kernel(int width, int height, int stride, float* Ix, float* Iy, const void* texPtr,
                                   int texPitch, int texW, int texH)
{
    orc_tex tex = {texPtr, texPitch, texW, texH, ORC_ADDR_MIRROR};
#pragma omp parallel for schedule(static)
    for (int iy = 0; iy < height; iy++) {
        for (int ix = 0; ix < width; ix++) {
            float dx = 1.0f / (float)width;
            float dy = 1.0f / (float)height;
            float x = ((float)ix + 0.5f) * dx;
            float y = ((float)iy + 0.5f) * dy;
            ORC_ROW(float, Ix, stride, iy)[ix] = orc_deriv5(&tex, x, y, dx, 0.0f);
            ORC_ROW(float, Iy, stride, iy)[ix] = orc_deriv5(&tex, x, y, 0.0f, dy);
        }
    }
}

/* D4: lucasKanadeOptim, opticalFlow.cu:189-325.  Quirk kept: smin =
 * fminf(sigma1, sigma1) (:255).  Only interior pixels are updated (:205-207). */
void orc_lucasKanadeOptim(of2* shifts, const float* imFx, const float* imFy, const float* imFt, int pitchShift,
                          int pitchImg, int width, int height, int halfWindowSize, float minDet)
{
#pragma omp parallel for schedule(static)
    for (int pxY = halfWindowSize; pxY < height - halfWindowSize; pxY++) {
        for (int pxX = halfWindowSize; pxX < width - halfWindowSize; pxX++) {
            int windowSize = halfWindowSize * 2 + 1;
            float matMul[4], matMulInv[4], UT[4], S[4], V[4], UV[2];
            matMul[0] = matMul[1] = matMul[2] = matMul[3] = 0;
            for (int y = -halfWindowSize; y <= halfWindowSize; y++) {
                for (int x = -halfWindowSize; x <= halfWindowSize; x++) {
                    float dx = ORC_CROW(float, imFx, pitchImg, pxY + y)[pxX + x];
                    float dy = ORC_CROW(float, imFy, pitchImg, pxY + y)[pxX + x];
                    matMul[0] += dx * dx; /* :229-231 */
                    matMul[1] += dx * dy;
                    matMul[3] += dy * dy;
                }
            }
            matMul[2] = matMul[1];
            float a = matMul[0], b = matMul[1], c = matMul[2], d = matMul[3];

            float theta = 0.5f * atan2f(2.0f * a * c + 2.0f * b * d, a * a + b * b - c * c - d * d); /* :242 */
            float ct = cosf(theta);
            float st = sinf(theta);
            UT[0] = ct;
            UT[2] = -st;
            UT[1] = st;
            UT[3] = ct;

            float S1 = a * a + b * b + c * c + d * d; /* :250 */
            float S2 = sqrtf((a * a + b * b - c * c - d * d) * (a * a + b * b - c * c - d * d) +
                             4 * (a * c + b * d) * (a * c + b * d)); /* :251 */
            float sigma1 = sqrtf((S1 + S2) / 2);
            float sigma2 = sqrtf((S1 - S2) / 2);

            float smin = fminf(sigma1, sigma1); /* :255 (sic) */
            if (smin < minDet) continue;

            sigma1 = sigma1 != 0 ? 1.0f / sigma1 : 0;
            sigma2 = sigma2 != 0 ? 1.0f / sigma2 : 0;
            S[0] = sigma1;
            S[1] = 0;
            S[2] = 0;
            S[3] = sigma2;

            float epsilon = 0.5f * atan2f(2.0f * a * b + 2.0f * c * d, a * a - b * b + c * c - d * d); /* :268 */
            float ce = cosf(epsilon);
            float se = sinf(epsilon);

            float s11 = (a * ct + c * st) * ce + (b * ct + d * st) * se; /* :273 */
            float s22 = (a * st - c * ct) * se + (-b * st + d * ct) * ce;
            s11 = s11 > 0.0f ? 1.0f : s11 < 0 ? -1.0f : 0.0f;
            s22 = s22 > 0.0f ? 1.0f : s22 < 0 ? -1.0f : 0.0f;

            V[0] = s11 * ce;
            V[1] = -s22 * se;
            V[2] = s11 * se;
            V[3] = s22 * ce;

            matMul[0] = S[0] * UT[0] + S[1] * UT[2]; /* :284-287 */
            matMul[1] = S[0] * UT[1] + S[1] * UT[3];
            matMul[2] = S[2] * UT[0] + S[3] * UT[2];
            matMul[3] = S[2] * UT[1] + S[3] * UT[3];

            matMulInv[0] = V[0] * matMul[0] + V[1] * matMul[2]; /* :289-292 */
            matMulInv[1] = V[0] * matMul[1] + V[1] * matMul[3];
            matMulInv[2] = V[2] * matMul[0] + V[3] * matMul[2];
            matMulInv[3] = V[2] * matMul[1] + V[3] * matMul[3];

            int ws2 = windowSize * windowSize;
            UV[0] = 0;
            UV[1] = 0;
            for (int i = 0; i < ws2; i++) { /* :298-313 */
                int y = i / windowSize;
                int x = i - (y * windowSize);
                int globalX = pxX + x - halfWindowSize;
                int globalY = pxY + y - halfWindowSize;
                float dx = ORC_CROW(float, imFx, pitchImg, globalY)[globalX];
                float dy = ORC_CROW(float, imFy, pitchImg, globalY)[globalX];
                float dt = ORC_CROW(float, imFt, pitchImg, globalY)[globalX];
                UV[0] += (matMulInv[0] * dx + matMulInv[1] * dy) * dt;
                UV[1] += (matMulInv[2] * dx + matMulInv[3] * dy) * dt;
            }
            UV[0] = isnan(UV[0]) ? 0 : UV[0];
            UV[1] = isnan(UV[1]) ? 0 : UV[1];

            of2* sp = &ORC_ROW(of2, shifts, pitchShift, pxY)[pxX];
            sp->x += UV[0];
            sp->y += UV[1];
        }
    }
}
