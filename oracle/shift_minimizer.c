/*
 * oracle/shift_minimizer.c -- CPU restatement of the reference's
 * test_opencv/ShiftMinimizerKernels.cu (rows C1-C6 of SURVEY.md section 8a)
 * plus the batched least-squares solve (row C4) that the reference delegates
 * to a host that is not in the repository.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle_common.h).
 */
#include "oracle_common.h"

/* C3a: copyShiftMatrix, ShiftMinimizerKernels.cu:28-48 */
void orc_copyShiftMatrix(float* matrices, int tileCount, int imageCount, int shiftCount)
{
    int matrixSize = (imageCount - 1) * shiftCount;
    for (int tile = 1; tile < tileCount; tile++) {
        size_t offset = (size_t)matrixSize * tile;
        for (int elem = 0; elem < matrixSize; elem++) matrices[offset + elem] = matrices[elem];
    }
}

/* C3b: setPointers, ShiftMinimizerKernels.cu:50-76 */
void orc_setPointers(float** shiftMatrixArray, float** shiftMatrixSafeArray, float** matrixSquareArray,
                     float** matrixInvertedArray, float** solvedMatrixArray, of2** shiftOneToOneArray,
                     of2** shiftMeasuredArray, of2** shiftOptimArray, float* shiftMatrices, float* shiftSafeMatrices,
                     float* matricesSquared, float* matricesInverted, float* solvedMatrices, of2* shiftsOneToOne,
                     of2* shiftsMeasured, of2* shiftsOptim, int tileCount, int imageCount, int shiftCount)
{
    int n1 = imageCount - 1;
    int m = shiftCount;
    size_t sizeShiftMatrix = (size_t)n1 * m;
    size_t sizeSquared = (size_t)n1 * n1;
    for (int tile = 0; tile < tileCount; tile++) {
        shiftMatrixArray[tile] = shiftMatrices + tile * sizeShiftMatrix;
        shiftMatrixSafeArray[tile] = shiftSafeMatrices + tile * sizeShiftMatrix;
        matrixSquareArray[tile] = matricesSquared + tile * sizeSquared;
        matrixInvertedArray[tile] = matricesInverted + tile * sizeSquared;
        solvedMatrixArray[tile] = solvedMatrices + tile * sizeShiftMatrix;
        shiftOneToOneArray[tile] = shiftsOneToOne + (size_t)tile * n1;
        shiftOptimArray[tile] = shiftsOptim + (size_t)tile * m;
        shiftMeasuredArray[tile] = shiftsMeasured + (size_t)tile * m;
    }
}

/* C5: checkForOutliers, ShiftMinimizerKernels.cu:80-139 */
void orc_checkForOutliers(of2* measuredShifts, const float* optimShiftsT, float* shiftMatrix, int* status,
                          const int* inversionInfo, int tileCount, int imageCount, int shiftCount)
{
    for (int tile = 0; tile < tileCount; tile++) {
        if (status[tile] < 0) continue;
        if (inversionInfo[tile] != 0) {
            status[tile] = -1;
            continue;
        }
        int n1 = imageCount - 1;
        int m = shiftCount;
        size_t offsetMatrix = (size_t)(n1 * m) * tile;
        size_t offsetAllVec = (size_t)m * tile;
        float max = 1;
        int idxMax = -1;
        for (int i = 0; i < m; i++) {
            float distx = measuredShifts[offsetAllVec + i].x - optimShiftsT[2 * offsetAllVec + i];
            float disty = measuredShifts[offsetAllVec + i].y - optimShiftsT[2 * offsetAllVec + i + m];
            float dist = distx * distx + disty * disty;
            if (dist > max) {
                idxMax = i;
                max = dist;
            }
        }
        status[tile] = idxMax;
        if (idxMax == -1) continue;
        measuredShifts[offsetAllVec + idxMax].x = 0;
        measuredShifts[offsetAllVec + idxMax].y = 0;
        for (int col = 0; col < n1; col++) shiftMatrix[offsetMatrix + idxMax + (size_t)col * m] = 0; /* :137 */
    }
}

/* C2: transposeShifts, ShiftMinimizerKernels.cu:142-176 */
void orc_transposeShifts(of2* measuredShifts, const float* measuredShiftsT, const float* shiftsOneToOneT,
                         of2* shiftsOneToOne, int tileCount, int imageCount, int shiftCount)
{
    int n1 = imageCount - 1;
    int m = shiftCount;
    for (int tile = 0; tile < tileCount; tile++) {
        for (int i = 0; i < m; i++) {
            size_t offsetAllVec = (size_t)m * tile;
            of2 shift;
            shift.x = measuredShiftsT[2 * offsetAllVec + i];
            shift.y = measuredShiftsT[2 * offsetAllVec + i + m];
            measuredShifts[offsetAllVec + i] = shift;
            if (i >= n1) continue;
            size_t offsetOneToOne = (size_t)n1 * tile;
            of2 temp;
            temp.x = shiftsOneToOneT[2 * offsetOneToOne + i];
            temp.y = shiftsOneToOneT[2 * offsetOneToOne + i + n1];
            shiftsOneToOne[offsetOneToOne + i] = temp;
        }
    }
}

/* C6: getOptimalShifts, ShiftMinimizerKernels.cu:178-218 */
void orc_getOptimalShifts(of2* optimalShifts, const of2* bestShifts, int imageCount, int tileCountX, int tileCountY,
                          int optimalShiftsPitch, int referenceImage, int imageToTrack)
{
    int n1 = imageCount - 1;
    for (int tileIdxY = 0; tileIdxY < tileCountY; tileIdxY++) {
        for (int tileIdxX = 0; tileIdxX < tileCountX; tileIdxX++) {
            const of2* r = &bestShifts[(size_t)(tileIdxX + tileIdxY * tileCountX) * n1];
            of2 totalShift = {0, 0};
            if (referenceImage < imageToTrack) {
                for (int i = referenceImage; i < imageToTrack; i++) {
                    totalShift.x += r[i].x;
                    totalShift.y += r[i].y;
                }
            } else if (imageToTrack < referenceImage) {
                for (int i = imageToTrack; i < referenceImage; i++) {
                    totalShift.x -= r[i].x;
                    totalShift.y -= r[i].y;
                }
            }
            ORC_ROW(of2, optimalShifts, optimalShiftsPitch, tileIdxY)[tileIdxX] = totalShift;
        }
    }
}

/* C1a: concatenateShifts, ShiftMinimizerKernels.cu:222-239 */
void orc_concatenateShifts(const of2* const* shiftIn, const int* shiftInPitch, of2* shiftOut, int shiftCount,
                           int tileCountX, int tileCountY)
{
    for (int tileY = 0; tileY < tileCountY; tileY++)
        for (int tileX = 0; tileX < tileCountX; tileX++)
            for (int shift = 0; shift < shiftCount; shift++) {
                const of2* line = ORC_CROW(of2, shiftIn[shift], shiftInPitch[shift], tileY);
                shiftOut[(size_t)(tileX + tileY * tileCountX) * shiftCount + shift] = line[tileX];
            }
}

/* C1b: separateShifts, ShiftMinimizerKernels.cu:241-258 */
void orc_separateShifts(const of2* shiftIn, of2* const* shiftOut, const int* shiftOutPitch, int shiftCount,
                        int tileCountX, int tileCountY)
{
    for (int tileY = 0; tileY < tileCountY; tileY++)
        for (int tileX = 0; tileX < tileCountX; tileX++)
            for (int shift = 0; shift < shiftCount; shift++) {
                of2* line = ORC_ROW(of2, shiftOut[shift], shiftOutPitch[shift], tileY);
                line[tileX] = shiftIn[(size_t)(tileX + tileY * tileCountX) * shiftCount + shift];
            }
}

/* C4 (NOT in the reference -- upstream delegates it to batched cuBLAS calls
 * made by a host that is absent).  Per tile:
 *   N = A^T A (n1 x n1), N^-1 by Gauss-Jordan with partial pivoting,
 *   d = N^-1 (A^T b) for the two right-hand sides (x and y),
 *   o = A d  (written planar [x(m) | y(m)], the layout checkForOutliers reads,
 *             ShiftMinimizerKernels.cu:114-115).
 * A is column-major m x n1 (shiftMatrix[row + col*m], :137).  inversionInfo =
 * 0 on success, k+1 when the k-th pivot is exactly zero (LAPACK getrf style).
 * Summation order is ascending in every dot product. */
#define ORC_MAX_N1 64
void orc_solveShiftsBatched(const float* shiftMatrix, const of2* measuredShifts, of2* shiftsOneToOne,
                            float* optimShiftsT, int* inversionInfo, int tileCount, int imageCount, int shiftCount)
{
    const int n1 = imageCount - 1;
    const int m = shiftCount;
#pragma omp parallel for schedule(static)
    for (int tile = 0; tile < tileCount; tile++) {
        const float* A = shiftMatrix + (size_t)tile * n1 * m;
        const of2* b = measuredShifts + (size_t)tile * m;
        float N[ORC_MAX_N1][ORC_MAX_N1], Inv[ORC_MAX_N1][ORC_MAX_N1];
        float rx[ORC_MAX_N1], ry[ORC_MAX_N1];
        for (int i = 0; i < n1; i++) {
            for (int j = 0; j < n1; j++) {
                float s = 0;
                for (int r = 0; r < m; r++) s += A[r + (size_t)i * m] * A[r + (size_t)j * m];
                N[i][j] = s;
                Inv[i][j] = (i == j) ? 1.0f : 0.0f;
            }
            float sx = 0, sy = 0;
            for (int r = 0; r < m; r++) {
                sx += A[r + (size_t)i * m] * b[r].x;
                sy += A[r + (size_t)i * m] * b[r].y;
            }
            rx[i] = sx;
            ry[i] = sy;
        }
        int info = 0;
        for (int k = 0; k < n1 && info == 0; k++) {
            int piv = k;
            float best = fabsf(N[k][k]);
            for (int r = k + 1; r < n1; r++) {
                if (fabsf(N[r][k]) > best) {
                    best = fabsf(N[r][k]);
                    piv = r;
                }
            }
            if (best == 0.0f) {
                info = k + 1;
                break;
            }
            if (piv != k) {
                for (int j = 0; j < n1; j++) {
                    float t = N[k][j];
                    N[k][j] = N[piv][j];
                    N[piv][j] = t;
                    t = Inv[k][j];
                    Inv[k][j] = Inv[piv][j];
                    Inv[piv][j] = t;
                }
            }
            float p = N[k][k];
            for (int j = 0; j < n1; j++) {
                N[k][j] = N[k][j] / p;
                Inv[k][j] = Inv[k][j] / p;
            }
            for (int r = 0; r < n1; r++) {
                if (r == k) continue;
                float f = N[r][k];
                for (int j = 0; j < n1; j++) {
                    N[r][j] = N[r][j] - f * N[k][j];
                    Inv[r][j] = Inv[r][j] - f * Inv[k][j];
                }
            }
        }
        inversionInfo[tile] = info;
        of2* d = shiftsOneToOne + (size_t)tile * n1;
        float* o = optimShiftsT + (size_t)tile * 2 * m;
        if (info != 0) {
            for (int i = 0; i < n1; i++) d[i].x = d[i].y = 0;
            for (int r = 0; r < 2 * m; r++) o[r] = 0;
            continue;
        }
        for (int i = 0; i < n1; i++) {
            float sx = 0, sy = 0;
            for (int j = 0; j < n1; j++) {
                sx += Inv[i][j] * rx[j];
                sy += Inv[i][j] * ry[j];
            }
            d[i].x = sx;
            d[i].y = sy;
        }
        for (int r = 0; r < m; r++) {
            float sx = 0, sy = 0;
            for (int c = 0; c < n1; c++) {
                sx += A[r + (size_t)c * m] * d[c].x;
                sy += A[r + (size_t)c * m] * d[c].y;
            }
            o[r] = sx;
            o[r + m] = sy;
        }
    }
}
