/*
 * oracle/diagnostics.c -- the two roundings through which a flow field reaches the output,
 * taken out of the oracle's own kernels so that tests can classify HIP-vs-oracle differences.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle_common.h).
 *
 * A per-pixel flow (u,v) influences the fused image only through
 *   (1) shift = round(s * tex(flow))            in accumulateImagesSuperRes (DeBayerKernels.cu:403-406;
 *                                               generalised to scale s in orc_accumulateSuperResFull),
 *   (2) shift = round(0.5 * tex(flow))          in ComputeRobustnessMask (RobustnessModell.cu:76-77),
 *   (3) the decision M > thresholdM             in ComputeRobustnessMask (:147-148), M = mask.w.
 * Two flows that differ by 1e-5 px give bit-different images only where one of these flips.
 */
#include "oracle_common.h"

/* (1): for every HR pixel the value s*tex(flow) (outValue, float2) and its rounding (outShift, int2);
 * same arithmetic, statement for statement, as orc_accumulateSuperResFull. */
void orc_dbgFuseShifts(const void* shPtr, int shPitch, int shW, int shH, int dimX, int dimY, int scale, float* outValue,
                       int* outShift)
{
    orc_tex texS = {shPtr, shPitch, shW, shH, ORC_ADDR_CLAMP};
    const int hrW = dimX * scale, hrH = dimY * scale;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < hrH; y++) {
        for (int x = 0; x < hrW; x++) {
            float posX = ((float)x + 0.5f) / (float)hrW;
            float posY = ((float)y + 0.5f) / (float)hrH;
            of2 shift = orc_tex2(&texS, posX, posY);
            float vx = shift.x * (float)scale, vy = shift.y * (float)scale;
            size_t o = ((size_t)y * hrW + x) * 2;
            outValue[o] = vx;
            outValue[o + 1] = vy;
            outShift[o] = orc_f2i(roundf(vx));
            outShift[o + 1] = orc_f2i(roundf(vy));
        }
    }
}

/* (2): for every half-res pixel 0.5*tex(flow) and its rounding; as orc_ComputeRobustnessMask. */
void orc_dbgRobustnessShifts(const void* uvPtr, int uvPitch, int uvW, int uvH, int imgWidth, int imgHeight, float* outValue,
                             int* outShift)
{
    orc_tex texUV = {uvPtr, uvPitch, uvW, uvH, ORC_ADDR_CLAMP};
#pragma omp parallel for schedule(static)
    for (int pxY = 0; pxY < imgHeight; pxY++) {
        for (int pxX = 0; pxX < imgWidth; pxX++) {
            of2 shiftf = orc_tex2(&texUV, ((float)pxX + 0.5f) / (float)imgWidth, ((float)pxY + 0.5f) / (float)imgHeight);
            float vx = shiftf.x * 0.5f, vy = shiftf.y * 0.5f;
            size_t o = ((size_t)pxY * imgWidth + pxX) * 2;
            outValue[o] = vx;
            outValue[o + 1] = vy;
            outShift[o] = orc_f2i(roundf(vx));
            outShift[o + 1] = orc_f2i(roundf(vy));
        }
    }
}

/* (1) for TWO flows at once, classified in place (what tests/flipset.py did with four HR-sized numpy arrays per frame):
 *   actual[p] |= the roundings of the two flows differ at HR pixel p;
 *   flips[p]  |= actual, or the values differ and one of them sits within tieEps * max(1, |v|) of a rounding tie
 *                (the guard against a last-ulp difference of the tex blend between implementations);
 * counts[0] += pixels newly or again flagged in `flips` by this frame, counts[1] += the same for `actual`.
 * All arithmetic in float, as the numpy float32 expressions it replaces. */
static int orc_near_tie(float v, float eps)
{
    float fr = fabsf(v - floorf(v) - 0.5f);
    return fr < eps * fmaxf(1.0f, fabsf(v));
}
void orc_dbgFuseFlips(const void* flowA, const void* flowB, int shPitch, int shW, int shH, int dimX, int dimY, int scale, float tieEps,
                      unsigned char* flips, unsigned char* actual, long long* counts)
{
    orc_tex texA = {flowA, shPitch, shW, shH, ORC_ADDR_CLAMP};
    orc_tex texB = {flowB, shPitch, shW, shH, ORC_ADDR_CLAMP};
    const int hrW = dimX * scale, hrH = dimY * scale;
    long long nF = 0, nA = 0;
#pragma omp parallel for schedule(static) reduction(+ : nF, nA)
    for (int y = 0; y < hrH; y++) {
        for (int x = 0; x < hrW; x++) {
            float posX = ((float)x + 0.5f) / (float)hrW;
            float posY = ((float)y + 0.5f) / (float)hrH;
            of2 a = orc_tex2(&texA, posX, posY), b = orc_tex2(&texB, posX, posY);
            float ax = a.x * (float)scale, ay = a.y * (float)scale, bx = b.x * (float)scale, by = b.y * (float)scale;
            int act = orc_f2i(roundf(ax)) != orc_f2i(roundf(bx)) || orc_f2i(roundf(ay)) != orc_f2i(roundf(by));
            int fl = act;
            if (!fl && (ax != bx || ay != by))
                fl = orc_near_tie(ax, tieEps) || orc_near_tie(ay, tieEps) || orc_near_tie(bx, tieEps) || orc_near_tie(by, tieEps);
            size_t o = (size_t)y * hrW + x;
            if (fl) {
                flips[o] = 1;
                nF++;
            }
            if (act) {
                actual[o] = 1;
                nA++;
            }
        }
    }
    counts[0] += nF;
    counts[1] += nA;
}
