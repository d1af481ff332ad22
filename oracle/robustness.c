/*
 * oracle/robustness.c -- CPU restatement of the reference's
 * test_opencv/RobustnessModell.cu (row F1 of SURVEY.md section 8a).
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle_common.h).
 */
#include "oracle_common.h"

/* F1: ComputeRobustnessMask, RobustnessModell.cu:28-158.  The 1-px border
 * ring of robustnessMask is never written (:48-49).  Quirk kept: the local
 * flow min/max compare each sample against the CENTRE value and overwrite, so
 * only the last sample (x=2,y=2) survives (:62-72). */
void orc_ComputeRobustnessMask(const of3* rawImgRef, const of3* rawImgMoved, of4* robustnessMask, const void* uvPtr,
                               int uvPitch, int uvW, int uvH, int imgWidth, int imgHeight, int imgPitch, int maskPitch,
                               float alpha, float beta, float thresholdM)
{
    orc_tex texUV = {uvPtr, uvPitch, uvW, uvH, ORC_ADDR_CLAMP};
#pragma omp parallel for schedule(static)
    for (int pxY = 1; pxY < imgHeight - 1; pxY++) {
        for (int pxX = 1; pxX < imgWidth - 1; pxX++) {
            of3 pixelsRef[9];
            of3 meanRef = {0, 0, 0}, meanMoved = {0, 0, 0}, stdRef = {0, 0, 0};
            of3 dist, sigma;

            of2 shiftf = orc_tex2(&texUV, ((float)pxX + 0.5f) / (float)imgWidth, ((float)pxY + 0.5f) / (float)imgHeight);
            of2 maxShift = shiftf, minShift = shiftf;
            for (int y = -2; y <= 2; y++) {
                for (int x = -2; x <= 2; x++) {
                    of2 s = orc_tex2(&texUV, ((float)pxX + (float)x + 0.5f) / (float)imgWidth,
                                     ((float)pxY + (float)y + 0.5f) / (float)imgHeight); /* :66 */
                    maxShift.x = fmaxf(s.x, shiftf.x);
                    maxShift.y = fmaxf(s.y, shiftf.y);
                    minShift.x = fminf(s.x, shiftf.x);
                    minShift.y = fminf(s.y, shiftf.y);
                }
            }

            int shx = orc_f2i(roundf(shiftf.x * 0.5f)); /* :76-77 */
            int shy = orc_f2i(roundf(shiftf.y * 0.5f));

            for (int y = -1; y <= 1; y++) {
                for (int x = -1; x <= 1; x++) {
                    of3 p = ORC_CROW(of3, rawImgRef, imgPitch, pxY + y)[pxX + x];
                    pixelsRef[(y + 1) * 3 + (x + 1)] = p;
                    meanRef.x += p.x;
                    meanRef.y += p.y;
                    meanRef.z += p.z;
                    int ppy = orc_imin(orc_imax(pxY + shy + y, 0), imgHeight - 1);
                    int ppx = orc_imin(orc_imax(pxX + shx + x, 0), imgWidth - 1);
                    p = ORC_CROW(of3, rawImgMoved, imgPitch, ppy)[ppx];
                    meanMoved.x += p.x;
                    meanMoved.y += p.y;
                    meanMoved.z += p.z;
                }
            }
            meanRef.x /= 9.0f;
            meanRef.y /= 9.0f;
            meanRef.z /= 9.0f;
            meanMoved.x /= 9.0f;
            meanMoved.y /= 9.0f;
            meanMoved.z /= 9.0f;

            float meandist =
                fabsf(meanRef.x - meanMoved.x) + fabsf(meanRef.y - meanMoved.y) + fabsf(meanRef.z - meanMoved.z); /* :105 */
            meandist /= 3.0f;
            maxShift.x *= 0.5f * meandist;
            maxShift.y *= 0.5f * meandist;
            minShift.x *= 0.5f * meandist;
            minShift.y *= 0.5f * meandist;

            float M = sqrtf((maxShift.x - minShift.x) * (maxShift.x - minShift.x) +
                            (maxShift.y - minShift.y) * (maxShift.y - minShift.y)); /* :112 */

            for (int p = 0; p < 9; p++) { /* :114-123 */
                stdRef.x += (pixelsRef[p].x - meanRef.x) * (pixelsRef[p].x - meanRef.x);
                stdRef.y += (pixelsRef[p].y - meanRef.y) * (pixelsRef[p].y - meanRef.y);
                stdRef.z += (pixelsRef[p].z - meanRef.z) * (pixelsRef[p].z - meanRef.z);
            }
            stdRef.x = sqrtf(stdRef.x / 9.0f);
            stdRef.y = sqrtf(stdRef.y / 9.0f);
            stdRef.z = sqrtf(stdRef.z / 9.0f);

            of3 sigmaMD;
            sigmaMD.x = sqrtf(alpha * meanRef.x + beta);
            sigmaMD.y = sqrtf(alpha * meanRef.y + beta) / sqrtf(2.0f); /* :131 */
            sigmaMD.z = sqrtf(alpha * meanRef.z + beta);

            dist.x = fabsf(meanRef.x - meanMoved.x);
            dist.y = fabsf(meanRef.y - meanMoved.y);
            dist.z = fabsf(meanRef.z - meanMoved.z);

            sigma.x = fmaxf(sigmaMD.x, stdRef.x);
            sigma.y = fmaxf(sigmaMD.y, stdRef.y);
            sigma.z = fmaxf(sigmaMD.z, stdRef.z);

            dist.x = dist.x * (stdRef.x * stdRef.x / (stdRef.x * stdRef.x + sigmaMD.x * sigmaMD.x)); /* :142-144 */
            dist.y = dist.y * (stdRef.y * stdRef.y / (stdRef.y * stdRef.y + sigmaMD.y * sigmaMD.y));
            dist.z = dist.z * (stdRef.z * stdRef.z / (stdRef.z * stdRef.z + sigmaMD.z * sigmaMD.z));

            of4 mask;
            float s = 1.5f;
            if (M > thresholdM) s = 0;
            const float t = 0.12f;
            mask.x = fmaxf(fminf(s * expf(-dist.x * dist.x / (sigma.x * sigma.x)) - t, 1.0f), 0.0f); /* :152-154 */
            mask.y = fmaxf(fminf(s * expf(-dist.y * dist.y / (sigma.y * sigma.y)) - t, 1.0f), 0.0f);
            mask.z = fmaxf(fminf(s * expf(-dist.z * dist.z / (sigma.z * sigma.z)) - t, 1.0f), 0.0f);
            mask.w = M;
            ORC_ROW(of4, robustnessMask, maskPitch, pxY)[pxX] = mask;
        }
    }
}
