// shift_minimizer.hip -- joint shift minimiser glue + batched least squares
// (SURVEY.md section 8a rows C1-C6).
// Behavioural spec: reference test_opencv/ShiftMinimizerKernels.cu.
//
// MI355X notes: checkForOutliers ("one thread per tile, serial over m",
// ShiftMinimizerKernels.cu:89-123) and the missing batched solve run as one
// 64-lane wavefront per tile: residual arg-max and pivot search are shuffle
// reductions whose tie-breaks reproduce the serial scans exactly, and every dot
// product is evaluated by one lane in ascending order, so results are
// bit-identical to the serial statement.
#include "common.hpp"

// (value, index) maximum; ties -> lower index (== first strict maximum of a serial scan)
__device__ __forceinline__ void wave_argmax(float& v, int& idx)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        const bool take = (ov > v) || (ov == v && oi < idx);
        v = take ? ov : v;
        idx = take ? oi : idx;
    }
}

// ---- C3a: copyShiftMatrix (:29-48) ---------------------------------------------
__global__ void __launch_bounds__(256) k_copyShiftMatrix(float* matrices, int tileCount, int matrixSize)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)matrixSize * tileCount;
    if (i >= total || i < (size_t)matrixSize) return;  // tile 0 is the source
    matrices[i] = matrices[i % matrixSize];
}

extern "C" int mfsr_copyShiftMatrix(float* matrices, int tileCount, int imageCount, int shiftCount, mfsr_stream_t stream)
{
    MFSR_REQUIRE(matrices && tileCount > 0 && imageCount > 1 && shiftCount > 0);
    const int matrixSize = (imageCount - 1) * shiftCount;
    const long long total = (long long)matrixSize * tileCount;
    hipLaunchKernelGGL(k_copyShiftMatrix, dim3(mfsr_cdiv(total, 256)), dim3(256), 0, mfsr_s(stream), matrices, tileCount,
                       matrixSize);
    return mfsr_launch_status("copyShiftMatrix");
}

// ---- C3b: setPointers (:51-76) --------------------------------------------------
__global__ void __launch_bounds__(256)
    k_setPointers(float** shiftMatrixArray, float** shiftMatrixSafeArray, float** matrixSquareArray,
                  float** matrixInvertedArray, float** solvedMatrixArray, float2** shiftOneToOneArray,
                  float2** shiftMeasuredArray, float2** shiftOptimArray, float* shiftMatrices, float* shiftSafeMatrices,
                  float* matricesSquared, float* matricesInverted, float* solvedMatrices, float2* shiftsOneToOne,
                  float2* shiftsMeasured, float2* shiftsOptim, int tileCount, int imageCount, int shiftCount)
{
    const int tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= tileCount) return;
    const int n1 = imageCount - 1;
    const int m = shiftCount;
    const size_t sizeShiftMatrix = (size_t)n1 * m;
    const size_t sizeSquared = (size_t)n1 * n1;
    shiftMatrixArray[tile] = shiftMatrices + tile * sizeShiftMatrix;
    shiftMatrixSafeArray[tile] = shiftSafeMatrices + tile * sizeShiftMatrix;
    matrixSquareArray[tile] = matricesSquared + tile * sizeSquared;
    matrixInvertedArray[tile] = matricesInverted + tile * sizeSquared;
    solvedMatrixArray[tile] = solvedMatrices + tile * sizeShiftMatrix;
    shiftOneToOneArray[tile] = shiftsOneToOne + (size_t)tile * n1;
    shiftOptimArray[tile] = shiftsOptim + (size_t)tile * m;
    shiftMeasuredArray[tile] = shiftsMeasured + (size_t)tile * m;
}

extern "C" int mfsr_setPointers(float** shiftMatrixArray, float** shiftMatrixSafeArray, float** matrixSquareArray,
                                float** matrixInvertedArray, float** solvedMatrixArray,
                                mfsr_float2** shiftOneToOneArray, mfsr_float2** shiftMeasuredArray,
                                mfsr_float2** shiftOptimArray, float* shiftMatrices, float* shiftSafeMatrices,
                                float* matricesSquared, float* matricesInverted, float* solvedMatrices,
                                mfsr_float2* shiftsOneToOne, mfsr_float2* shiftsMeasured, mfsr_float2* shiftsOptim,
                                int tileCount, int imageCount, int shiftCount, mfsr_stream_t stream)
{
    MFSR_REQUIRE(shiftMatrixArray && shiftMatrixSafeArray && matrixSquareArray && matrixInvertedArray &&
                 solvedMatrixArray && shiftOneToOneArray && shiftMeasuredArray && shiftOptimArray);
    MFSR_REQUIRE(tileCount > 0 && imageCount > 1 && shiftCount > 0);
    hipLaunchKernelGGL(k_setPointers, dim3(mfsr_cdiv(tileCount, 256)), dim3(256), 0, mfsr_s(stream), shiftMatrixArray,
                       shiftMatrixSafeArray, matrixSquareArray, matrixInvertedArray, solvedMatrixArray,
                       (float2**)shiftOneToOneArray, (float2**)shiftMeasuredArray, (float2**)shiftOptimArray,
                       shiftMatrices, shiftSafeMatrices, matricesSquared, matricesInverted, solvedMatrices,
                       (float2*)shiftsOneToOne, (float2*)shiftsMeasured, (float2*)shiftsOptim, tileCount, imageCount,
                       shiftCount);
    return mfsr_launch_status("setPointers");
}

// ---- C5: checkForOutliers (:81-139), one wavefront per tile ----------------------
// one wavefront; returns the dropped row or -1 (what the reference writes to status[tile])
__device__ __forceinline__ int check_tile(float2* measuredShifts, const float* optimShiftsT, float* shiftMatrix, int tile, int lane,
                                          int n1, int m)
{
    const size_t offsetMatrix = (size_t)(n1 * m) * tile;
    const size_t offsetAllVec = (size_t)m * tile;
    float mx = 1;  // threshold 1 px^2 (:109)
    int idxMax = 0x7fffffff;
    for (int i = lane; i < m; i += 64) {
        const float2 ms = measuredShifts[offsetAllVec + i];
        const float distx = ms.x - optimShiftsT[2 * offsetAllVec + i];
        const float disty = ms.y - optimShiftsT[2 * offsetAllVec + i + m];
        const float dist = distx * distx + disty * disty;
        if (dist > mx) {
            idxMax = i;
            mx = dist;
        }
    }
    wave_argmax(mx, idxMax);
    if (idxMax == 0x7fffffff) idxMax = -1;
    if (idxMax == -1) return -1;
    if (lane == 0) measuredShifts[offsetAllVec + idxMax] = make_float2(0.0f, 0.0f);
    for (int col = lane; col < n1; col += 64) shiftMatrix[offsetMatrix + idxMax + (size_t)col * m] = 0;
    return idxMax;
}

__global__ void __launch_bounds__(256)
    k_checkForOutliers(float2* __restrict__ measuredShifts, const float* __restrict__ optimShiftsT,
                       float* __restrict__ shiftMatrix, int* __restrict__ status, const int* __restrict__ inversionInfo,
                       int tileCount, int imageCount, int shiftCount)
{
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tile >= tileCount) return;
    if (status[tile] < 0) return;
    if (inversionInfo[tile] != 0) {
        if (lane == 0) status[tile] = -1;
        return;
    }
    const int idxMax = check_tile(measuredShifts, optimShiftsT, shiftMatrix, tile, lane, imageCount - 1, shiftCount);
    if (lane == 0) status[tile] = idxMax;
}

extern "C" int mfsr_checkForOutliers(mfsr_float2* measuredShifts, const float* optimShiftsT, float* shiftMatrix,
                                     int* status, int* inversionInfo, int tileCount, int imageCount, int shiftCount,
                                     mfsr_stream_t stream)
{
    MFSR_REQUIRE(measuredShifts && optimShiftsT && shiftMatrix && status && inversionInfo);
    MFSR_REQUIRE(tileCount > 0 && imageCount > 1 && shiftCount > 0 && ((uintptr_t)measuredShifts & 7) == 0);
    hipLaunchKernelGGL(k_checkForOutliers, dim3(mfsr_cdiv(tileCount, 4)), dim3(256), 0, mfsr_s(stream),
                       (float2*)measuredShifts, optimShiftsT, shiftMatrix, status, (const int*)inversionInfo, tileCount,
                       imageCount, shiftCount);
    return mfsr_launch_status("checkForOutliers");
}

// ---- C2: transposeShifts (:143-176) ----------------------------------------------
__global__ void __launch_bounds__(256)
    k_transposeShifts(float2* __restrict__ measuredShifts, const float* __restrict__ measuredShiftsT,
                      const float* __restrict__ shiftsOneToOneT, float2* __restrict__ shiftsOneToOne, int tileCount,
                      int imageCount, int shiftCount)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int tile = blockIdx.y;
    if (tile >= tileCount) return;
    const int n1 = imageCount - 1;
    const int m = shiftCount;
    if (i >= m) return;
    const size_t offsetAllVec = (size_t)m * tile;
    float2 shift;
    shift.x = measuredShiftsT[2 * offsetAllVec + i];
    shift.y = measuredShiftsT[2 * offsetAllVec + i + m];
    measuredShifts[offsetAllVec + i] = shift;
    if (i >= n1) return;
    const size_t offsetOneToOne = (size_t)n1 * tile;
    float2 temp;
    temp.x = shiftsOneToOneT[2 * offsetOneToOne + i];
    temp.y = shiftsOneToOneT[2 * offsetOneToOne + i + n1];
    shiftsOneToOne[offsetOneToOne + i] = temp;
}

extern "C" int mfsr_transposeShifts(mfsr_float2* measuredShifts, const float* measuredShiftsT,
                                    const float* shiftsOneToOneT, mfsr_float2* shiftsOneToOne, int tileCount,
                                    int imageCount, int shiftCount, mfsr_stream_t stream)
{
    MFSR_REQUIRE(measuredShifts && measuredShiftsT && shiftsOneToOneT && shiftsOneToOne);
    MFSR_REQUIRE(tileCount > 0 && tileCount <= 65535 && imageCount > 1 && shiftCount > 0);
    hipLaunchKernelGGL(k_transposeShifts, dim3(mfsr_cdiv(shiftCount, 64), tileCount), dim3(64), 0, mfsr_s(stream),
                       (float2*)measuredShifts, measuredShiftsT, shiftsOneToOneT, (float2*)shiftsOneToOne, tileCount,
                       imageCount, shiftCount);
    return mfsr_launch_status("transposeShifts");
}

// ---- C6: getOptimalShifts (:179-218) ----------------------------------------------
__global__ void __launch_bounds__(256)
    k_getOptimalShifts(float2* __restrict__ optimalShifts, const float2* __restrict__ bestShifts, int imageCount,
                       int tileCountX, int tileCountY, int optimalShiftsPitch, int referenceImage, int imageToTrack)
{
    const int tileIdxX = blockIdx.x * blockDim.x + threadIdx.x;
    const int tileIdxY = blockIdx.y * blockDim.y + threadIdx.y;
    if (tileIdxX >= tileCountX || tileIdxY >= tileCountY) return;
    const int n1 = imageCount - 1;
    const float2* r = &bestShifts[(size_t)(tileIdxX + tileIdxY * tileCountX) * n1];
    float2 totalShift = make_float2(0, 0);
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
    row_ptr(optimalShifts, optimalShiftsPitch, tileIdxY)[tileIdxX] = totalShift;
}

extern "C" int mfsr_getOptimalShifts(mfsr_float2* optimalShifts, const mfsr_float2* bestShifts, int imageCount,
                                     int tileCountX, int tileCountY, int optimalShiftsPitch, int referenceImage,
                                     int imageToTrack, mfsr_stream_t stream)
{
    MFSR_REQUIRE(optimalShifts && bestShifts && imageCount > 1 && tileCountX > 0 && tileCountY > 0);
    MFSR_REQUIRE(referenceImage >= 0 && referenceImage < imageCount && imageToTrack >= 0 && imageToTrack < imageCount);
    MFSR_REQUIRE((long long)optimalShiftsPitch >= 8LL * tileCountX && (optimalShiftsPitch & 7) == 0 &&
                 ((uintptr_t)optimalShifts & 7) == 0 && ((uintptr_t)bestShifts & 7) == 0);
    dim3 block(64, 4), grid(mfsr_cdiv(tileCountX, 64), mfsr_cdiv(tileCountY, 4));
    hipLaunchKernelGGL(k_getOptimalShifts, grid, block, 0, mfsr_s(stream), (float2*)optimalShifts,
                       (const float2*)bestShifts, imageCount, tileCountX, tileCountY, optimalShiftsPitch, referenceImage,
                       imageToTrack);
    return mfsr_launch_status("getOptimalShifts");
}

// ---- C1: concatenateShifts / separateShifts (:223-258) -----------------------------
__global__ void __launch_bounds__(256)
    k_concatenateShifts(const float2* const* __restrict__ shiftIn, const int* __restrict__ shiftInPitch,
                        float2* __restrict__ shiftOut, int shiftCount, int tileCountX, int tileCountY)
{
    const int shift = blockIdx.x * blockDim.x + threadIdx.x;
    const int tileX = blockIdx.y * blockDim.y + threadIdx.y;
    const int tileY = blockIdx.z * blockDim.z + threadIdx.z;
    if (tileX >= tileCountX || tileY >= tileCountY || shift >= shiftCount) return;
    const float2* line = (const float2*)((const char*)(shiftIn[shift]) + (size_t)shiftInPitch[shift] * tileY);
    shiftOut[(size_t)(tileX + tileY * tileCountX) * shiftCount + shift] = line[tileX];
}

__global__ void __launch_bounds__(256)
    k_separateShifts(const float2* __restrict__ shiftIn, float2* const* __restrict__ shiftOut,
                     const int* __restrict__ shiftOutPitch, int shiftCount, int tileCountX, int tileCountY)
{
    const int shift = blockIdx.x * blockDim.x + threadIdx.x;
    const int tileX = blockIdx.y * blockDim.y + threadIdx.y;
    const int tileY = blockIdx.z * blockDim.z + threadIdx.z;
    if (tileX >= tileCountX || tileY >= tileCountY || shift >= shiftCount) return;
    float2* line = (float2*)((char*)(shiftOut[shift]) + (size_t)shiftOutPitch[shift] * tileY);
    line[tileX] = shiftIn[(size_t)(tileX + tileY * tileCountX) * shiftCount + shift];
}

extern "C" int mfsr_concatenateShifts(const mfsr_float2* const* shiftIn, int* shiftInPitch, mfsr_float2* shiftOut,
                                      int shiftCount, int tileCountX, int tileCountY, mfsr_stream_t stream)
{
    MFSR_REQUIRE(shiftIn && shiftInPitch && shiftOut && shiftCount > 0 && tileCountX > 0 && tileCountY > 0);
    MFSR_REQUIRE(tileCountY <= 65535);
    dim3 block(16, 16, 1), grid(mfsr_cdiv(shiftCount, 16), mfsr_cdiv(tileCountX, 16), tileCountY);
    hipLaunchKernelGGL(k_concatenateShifts, grid, block, 0, mfsr_s(stream), (const float2* const*)shiftIn,
                       (const int*)shiftInPitch, (float2*)shiftOut, shiftCount, tileCountX, tileCountY);
    return mfsr_launch_status("concatenateShifts");
}

extern "C" int mfsr_separateShifts(const mfsr_float2* shiftIn, mfsr_float2* const* shiftOut, int* shiftOutPitch,
                                   int shiftCount, int tileCountX, int tileCountY, mfsr_stream_t stream)
{
    MFSR_REQUIRE(shiftIn && shiftOut && shiftOutPitch && shiftCount > 0 && tileCountX > 0 && tileCountY > 0);
    MFSR_REQUIRE(tileCountY <= 65535);
    dim3 block(16, 16, 1), grid(mfsr_cdiv(shiftCount, 16), mfsr_cdiv(tileCountX, 16), tileCountY);
    hipLaunchKernelGGL(k_separateShifts, grid, block, 0, mfsr_s(stream), (const float2*)shiftIn,
                       (float2* const*)shiftOut, (const int*)shiftOutPitch, shiftCount, tileCountX, tileCountY);
    return mfsr_launch_status("separateShifts");
}

// ---- C4: batched least squares (not in the reference; cuBLAS-batched upstream) -----
// One wavefront per tile.  N = A^T A and N^-1 live in LDS; Gauss-Jordan with
// partial pivoting; the pivot search is a wavefront arg-max over |N[r][k]|,
// r >= k, ties -> lowest row (same as a serial "strictly greater" scan).
#define SOLVE_MAXN 63
// one 64-lane workgroup per tile; returns the inversion info (0 or index of the zero pivot + 1)
__device__ __forceinline__ int solve_tile(float* s_solve, const float* shiftMatrix, const float2* measuredShifts,
                                          float2* shiftsOneToOne, float* optimShiftsT, int tile, int n1, int m)
{
    float* N = s_solve;           // n1*n1
    float* Inv = N + n1 * n1;     // n1*n1
    float* rx = Inv + n1 * n1;    // n1
    float* ry = rx + n1;          // n1
    float* fcol = ry + n1;        // n1
    float* dx = fcol + n1;        // n1
    float* dy = dx + n1;          // n1
    const int lane = threadIdx.x;
    const float* A = shiftMatrix + (size_t)tile * n1 * m;
    const float2* b = measuredShifts + (size_t)tile * m;

    for (int idx = lane; idx < n1 * n1; idx += 64) {
        const int i = idx / n1, j = idx - i * n1;
        float s = 0;
        for (int r = 0; r < m; r++) s += A[r + (size_t)i * m] * A[r + (size_t)j * m];
        N[idx] = s;
        Inv[idx] = (i == j) ? 1.0f : 0.0f;
    }
    for (int i = lane; i < n1; i += 64) {
        float sx = 0, sy = 0;
        for (int r = 0; r < m; r++) {
            const float a = A[r + (size_t)i * m];
            const float2 bb = b[r];
            sx += a * bb.x;
            sy += a * bb.y;
        }
        rx[i] = sx;
        ry[i] = sy;
    }
    __syncthreads();

    int info = 0;
    for (int k = 0; k < n1; k++) {
        // pivot search: lane r owns row r (n1 <= 63)
        float best = (lane >= k && lane < n1) ? fabsf(N[lane * n1 + k]) : -1.0f;
        int piv = (lane >= k && lane < n1) ? lane : 0x7fffffff;
        wave_argmax(best, piv);
        if (best == 0.0f) {
            info = k + 1;
            break;
        }
        __syncthreads();
        if (piv != k) {
            for (int j = lane; j < n1; j += 64) {
                float t = N[k * n1 + j];
                N[k * n1 + j] = N[piv * n1 + j];
                N[piv * n1 + j] = t;
                t = Inv[k * n1 + j];
                Inv[k * n1 + j] = Inv[piv * n1 + j];
                Inv[piv * n1 + j] = t;
            }
        }
        __syncthreads();
        const float p = N[k * n1 + k];
        __syncthreads();
        for (int j = lane; j < n1; j += 64) {
            N[k * n1 + j] = N[k * n1 + j] / p;
            Inv[k * n1 + j] = Inv[k * n1 + j] / p;
        }
        for (int r = lane; r < n1; r += 64) fcol[r] = N[r * n1 + k];
        __syncthreads();
        // fcol[k] was read before/after normalisation of N[k][k]; row k is skipped below
        for (int idx = lane; idx < n1 * n1; idx += 64) {
            const int r = idx / n1, j = idx - r * n1;
            if (r == k) continue;
            const float f = fcol[r];
            N[idx] = N[idx] - f * N[k * n1 + j];
            Inv[idx] = Inv[idx] - f * Inv[k * n1 + j];
        }
        __syncthreads();
    }

    float2* d = shiftsOneToOne + (size_t)tile * n1;
    float* o = optimShiftsT + (size_t)tile * 2 * m;
    if (info != 0) {
        for (int i = lane; i < n1; i += 64) d[i] = make_float2(0.0f, 0.0f);
        for (int r = lane; r < 2 * m; r += 64) o[r] = 0;
        return info;
    }
    for (int i = lane; i < n1; i += 64) {
        float sx = 0, sy = 0;
        for (int j = 0; j < n1; j++) {
            sx += Inv[i * n1 + j] * rx[j];
            sy += Inv[i * n1 + j] * ry[j];
        }
        dx[i] = sx;
        dy[i] = sy;
        d[i] = make_float2(sx, sy);
    }
    __syncthreads();
    for (int r = lane; r < m; r += 64) {
        float sx = 0, sy = 0;
        for (int c = 0; c < n1; c++) {
            const float a = A[r + (size_t)c * m];
            sx += a * dx[c];
            sy += a * dy[c];
        }
        o[r] = sx;
        o[r + m] = sy;
    }
    return 0;
}

__global__ void __launch_bounds__(64)
    k_solveShiftsBatched(const float* __restrict__ shiftMatrix, const float2* __restrict__ measuredShifts,
                         float2* __restrict__ shiftsOneToOne, float* __restrict__ optimShiftsT,
                         int* __restrict__ inversionInfo, int tileCount, int n1, int m)
{
    extern __shared__ __attribute__((aligned(16))) float s_solve[];
    const int tile = blockIdx.x;
    if (tile >= tileCount) return;
    const int info = solve_tile(s_solve, shiftMatrix, measuredShifts, shiftsOneToOne, optimShiftsT, tile, n1, m);
    if (threadIdx.x == 0) inversionInfo[tile] = info;
}

// C driver on the device: the tiles are independent problems, so the "solve -> checkForOutliers until converged" loop of
// mfsr_minimizeShifts (one host round trip per round) runs inside ONE launch, one wavefront per tile, with no host
// synchronisation: graph-capturable, and bit-identical to the host-driven loop (same statements per tile).
__global__ void __launch_bounds__(64)
    k_minimizeShiftsFused(float* shiftMatrix, float2* measuredShifts, float2* shiftsOneToOne, float* optimShiftsT, int* status,
                          int* inversionInfo, int tileCount, int n1, int m)
{
    extern __shared__ __attribute__((aligned(16))) float s_solve[];
    const int tile = blockIdx.x;
    if (tile >= tileCount) return;
    int st = 0, info = 0;
    for (int round = 0; round < m + 1; round++) {
        info = solve_tile(s_solve, shiftMatrix, measuredShifts, shiftsOneToOne, optimShiftsT, tile, n1, m);
        __threadfence_block();
        __syncthreads();
        if (info != 0) {
            st = -1;
            break;
        }
        st = check_tile(measuredShifts, optimShiftsT, shiftMatrix, tile, threadIdx.x, n1, m);
        __threadfence_block();
        __syncthreads();
        if (st == -1) break;
    }
    if (threadIdx.x == 0) {
        status[tile] = st;
        inversionInfo[tile] = info;
    }
}

extern "C" int mfsr_solveShiftsBatched(const float* shiftMatrix, const mfsr_float2* measuredShifts,
                                       mfsr_float2* shiftsOneToOne, float* optimShiftsT, int* inversionInfo,
                                       int tileCount, int imageCount, int shiftCount, mfsr_stream_t stream)
{
    MFSR_REQUIRE(shiftMatrix && measuredShifts && shiftsOneToOne && optimShiftsT && inversionInfo);
    MFSR_REQUIRE(tileCount > 0 && imageCount > 1 && shiftCount > 0);
    MFSR_REQUIRE(((uintptr_t)measuredShifts & 7) == 0 && ((uintptr_t)shiftsOneToOne & 7) == 0);
    const int n1 = imageCount - 1;
    if (n1 > SOLVE_MAXN) return MFSR_E_UNSUPPORTED;
    const size_t lds = sizeof(float) * ((size_t)2 * n1 * n1 + 5 * n1);
    hipLaunchKernelGGL(k_solveShiftsBatched, dim3(tileCount), dim3(64), lds, mfsr_s(stream), shiftMatrix,
                       (const float2*)measuredShifts, (float2*)shiftsOneToOne, optimShiftsT, inversionInfo, tileCount, n1,
                       shiftCount);
    return mfsr_launch_status("solveShiftsBatched");
}


extern "C" int mfsr_minimizeShiftsFused(float* shiftMatrix, mfsr_float2* measuredShifts, mfsr_float2* shiftsOneToOne,
                                        float* optimShiftsT, int* status, int* inversionInfo, int tileCount, int imageCount,
                                        int shiftCount, mfsr_stream_t stream)
{
    MFSR_REQUIRE(shiftMatrix && measuredShifts && shiftsOneToOne && optimShiftsT && status && inversionInfo);
    MFSR_REQUIRE(tileCount > 0 && imageCount > 1 && shiftCount > 0);
    MFSR_REQUIRE(((uintptr_t)measuredShifts & 7) == 0 && ((uintptr_t)shiftsOneToOne & 7) == 0);
    const int n1 = imageCount - 1;
    if (n1 > SOLVE_MAXN) return MFSR_E_UNSUPPORTED;
    const size_t lds = sizeof(float) * ((size_t)2 * n1 * n1 + 5 * n1);
    hipLaunchKernelGGL(k_minimizeShiftsFused, dim3(tileCount), dim3(64), lds, mfsr_s(stream), shiftMatrix, (float2*)measuredShifts,
                       (float2*)shiftsOneToOne, optimShiftsT, status, inversionInfo, tileCount, n1, shiftCount);
    return mfsr_launch_status("minimizeShiftsFused");
}
