/*
 * mfsr.h -- C-ABI of the MI355X-native multi-frame super-resolution hot path.
 *
 * Drop-in boundary for the align -> fuse -> upsample path of
 * zhongzisha/multi_frame_super_resolution (the ImageStackAlignator CUDA
 * kernels in test_opencv, the five .cu files).  The reference exposes that path as
 * unmangled `extern "C" __global__` symbols loaded by name by a host that is
 * not in the repository, and its own host-side FFI convention is
 * `extern "C" void f(T* devPtr..., int width, int height...)`
 * (test_opencv/myKernels.cu:114-120,156-165; decls test_opencv/main.cpp:763-767).
 * Each entry point below replaces the launch of ONE reference kernel: same
 * name (prefixed mfsr_), same argument order, same units (pitches in BYTES,
 * dims in elements, raw device pointers), with
 *   - cudaTextureObject_t  ->  mfsr_tex2d {ptr, pitch, width, height} by value
 *     (linear filtering, normalised coordinates; MIRROR addressing for images,
 *     CLAMP for flow / parameter fields -- stated per function),
 *   - the module constant c_cfaPattern -> mfsr_set_cfa_pattern(),
 *   - grid/block shapes chosen by the library,
 *   - one trailing mfsr_stream_t (a hipStream_t; NULL = default stream).
 *
 * Conventions (reference error convention: cudaError_t return + fprintf(stderr),
 * test_opencv/kernel.cu:36-114):
 *   - every function returns int: 0 = success, >0 = hipError_t, <0 = MFSR_E_*;
 *     nothing throws; failures are logged to stderr;
 *   - all calls are asynchronous on `stream` unless named *_sync;
 *   - the caller allocates and frees every buffer; kernels never allocate;
 *     accumulators are zero-initialised by the caller and accumulated across
 *     calls (DeBayerKernels.cu:306-307,374-375);
 *   - untouched border rings (caller initialises): deBayer* 2 px,
 *     accumulate* 1 px, lucasKanadeOptim halfWindowSize px,
 *     ComputeRobustnessMask 1 px.
 *   - there is NO CPU fallback: without a HIP device every call fails.
 */
#ifndef MFSR_H
#define MFSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFSR_VERSION 100
#define MFSR_MAX_FUSE_GROUP 4 /* frames one warp+fuse call takes (mfsr_accumulateSuperResFullN, cfg.pairFrames) */

typedef void* mfsr_stream_t; /* hipStream_t */

typedef struct { float x, y; } mfsr_float2;
typedef struct { float x, y, z; } mfsr_float3;
typedef struct { float x, y, z, w; } mfsr_float4;

/* stand-in for cudaTextureObject_t: pitched 2-D array in device memory */
typedef struct {
    const void* ptr;
    int32_t pitch; /* bytes */
    int32_t width; /* texels */
    int32_t height;
} mfsr_tex2d;

enum {
    MFSR_OK = 0,
    MFSR_E_INVALID = -1,   /* bad argument (null pointer, non-positive size, pitch too small) */
    MFSR_E_UNSUPPORTED = -2, /* parameter outside what the kernels are built for */
    MFSR_E_NODEVICE = -3,  /* no usable HIP device */
    MFSR_E_WORKSPACE = -4  /* workspace too small */
};

/* enum BayerColor, DeBayerKernels.cu:28-37 */
enum { MFSR_RED = 0, MFSR_GREEN = 1, MFSR_BLUE = 2, MFSR_CYAN = 3, MFSR_MAGENTA = 4, MFSR_YELLOW = 5, MFSR_WHITE = 6 };

const char* mfsr_error_string(int code);
int mfsr_version(void);
/* number of visible HIP devices (0 if none); never fails */
int mfsr_device_count(void);

/* ---- A0: c_cfaPattern[2][2], DeBayerKernels.cu:40-41 ---------------------- */
/* pattern[0..3] = {[0][0],[0][1],[1][0],[1][1]}; RGGB = {0,1,1,2}.  Process-
 * wide state like the reference's module constant; read at launch time. */
int mfsr_set_cfa_pattern(const int32_t pattern[4]);
int mfsr_get_cfa_pattern(int32_t pattern[4]);

/* ---- A: DeBayerKernels.cu ------------------------------------------------- */
/* A1 deBayersSubSample3 :244 -- dimX,dimY = OUTPUT (half-res) dims; dataIn is
 * dense u16 with row stride 2*dimX elements. */
int mfsr_deBayersSubSample3(const uint16_t* dataIn, mfsr_float3* imgOut, float maxVal, int dimX, int dimY, int strideOut,
                            mfsr_stream_t stream);
/* A2 deBayerGreenKernel :55 */
int mfsr_deBayerGreenKernel(int width, int height, const float* imgIn, int strideIn, mfsr_float3* outImage,
                            int strideOut, mfsr_float3 blackPoint, mfsr_float3 scale, mfsr_stream_t stream);
/* A3 deBayerRedBlueKernel :153 (run after A2 has completed on the stream) */
int mfsr_deBayerRedBlueKernel(int width, int height, const float* imgIn, int strideIn, mfsr_float3* outImage,
                              int strideOut, mfsr_float3 blackPoint, mfsr_float3 scale, mfsr_stream_t stream);
/* G1 accumulateImages :290 (x1 merge; kernelParam rows use strideOut, :308) */
int mfsr_accumulateImages(const uint16_t* dataIn, mfsr_float3* imgOut, mfsr_float3* totalWeights,
                          const mfsr_float4* certaintyMask, const mfsr_float3* kernelParam, const mfsr_float2* shifts,
                          mfsr_float3 whiteLevel, mfsr_float3 blackLevel, int dimX, int dimY, int strideOut,
                          int strideMask, int strideShift, mfsr_stream_t stream);
/* G2 accumulateImagesSuperRes :379 (x2 merge, output grid dimX x dimY covering
 * the central half of the frame).  kernelParam: float4 texture, shifts: float2
 * texture, both CLAMP.  strideKernelParam/strideShift of the reference are
 * carried inside the descriptors. */
int mfsr_accumulateImagesSuperRes(const uint16_t* dataIn, mfsr_float3* imgOut, mfsr_float3* totalWeights,
                                  const mfsr_float4* certaintyMask, mfsr_tex2d kernelParam, mfsr_tex2d shifts,
                                  mfsr_float3 whiteLevel, mfsr_float3 blackLevel, int dimX, int dimY, int strideOut,
                                  int strideMask, mfsr_stream_t stream);
/* G2 generalised to integer scale s (1..8) on the FULL frame: output grid
 * (s*dimX) x (s*dimY); the build's extension of :379-468. */
int mfsr_accumulateSuperResFull(const uint16_t* dataIn, mfsr_float3* imgOut, mfsr_float3* totalWeights,
                                const mfsr_float4* certaintyMask, mfsr_tex2d kernelParam, mfsr_tex2d shifts,
                                mfsr_float3 whiteLevel, mfsr_float3 blackLevel, int dimX, int dimY, int scale,
                                int strideOut, int strideMask, mfsr_stream_t stream);

/* Two frames in one call (frame 0, then frame 1): what two mfsr_accumulateSuperResFull calls compute,
 * with the accumulators read and written once where the x2 tile kernel applies.  Equal to the
 * two-call sequence to fp32 rounding (the two per-pixel sums are added to each other first). */
int mfsr_accumulateSuperResFull2(const uint16_t* dataIn0, const uint16_t* dataIn1, mfsr_float3* imgOut,
                                 mfsr_float3* totalWeights, const mfsr_float4* certaintyMask0,
                                 const mfsr_float4* certaintyMask1, mfsr_tex2d kernelParam, mfsr_tex2d shifts0,
                                 mfsr_tex2d shifts1, mfsr_float3 whiteLevel, mfsr_float3 blackLevel, int dimX, int dimY,
                                 int scale, int strideOut, int strideMask, mfsr_stream_t stream);

/* nFrames (1 .. MFSR_MAX_FUSE_GROUP) frames in one call (dataIn / certaintyMask / shifts: host arrays of nFrames entries);
 * the frames add in call order.  At scale 2 / 4 with the fields at a quarter / an eighth of the HR size (the Bayer pipeline)
 * the whole group is ONE pass over the accumulators; other geometries take it two frames at a time.
 * accumulatorsUndefined != 0: imgOut / totalWeights are OVERWRITTEN as if they had been zeroed before
 * the call -- the first launch of a burst needs neither the memset nor the read of the two planes. */
int mfsr_accumulateSuperResFullN(int nFrames, const uint16_t* const* dataIn, mfsr_float3* imgOut, mfsr_float3* totalWeights,
                                 const mfsr_float4* const* certaintyMask, mfsr_tex2d kernelParam, const mfsr_tex2d* shifts,
                                 mfsr_float3 whiteLevel, mfsr_float3 blackLevel, int dimX, int dimY, int scale,
                                 int strideOut, int strideMask, int accumulatorsUndefined, mfsr_stream_t stream);

/* mfsr_accumulateSuperResFullN restricted to HR rows [rowBegin, rowEnd) (rowBegin % 16 == 0; rowEnd % 16 == 0 or
 * rowEnd == scale*dimY): a burst whose fuse stage is sharded over HR row stripes (multi-GPU, mfsr_dist_*) calls it once
 * per stripe; every pixel of the window gets exactly the whole-frame result.  Raw / certainty / shift buffers keep their
 * whole-frame addressing -- only the rows the window's taps reach are read. */
int mfsr_accumulateSuperResFullRows(int nFrames, const uint16_t* const* dataIn, mfsr_float3* imgOut, mfsr_float3* totalWeights,
                                    const mfsr_float4* const* certaintyMask, mfsr_tex2d kernelParam, const mfsr_tex2d* shifts,
                                    mfsr_float3 whiteLevel, mfsr_float3 blackLevel, int dimX, int dimY, int scale,
                                    int strideOut, int strideMask, int accumulatorsUndefined, int rowBegin, int rowEnd,
                                    mfsr_stream_t stream);

/* ---- B/E/H/I: kernel.cu --------------------------------------------------- */
int mfsr_squaredSum(const float* inTiles, float* outValues, int maxShift, int tileSize, int tileCount,
                    mfsr_stream_t stream); /* :119 */
int mfsr_boxFilterWithBorderX(const float* inTiles, float* outTiles, int maxShift, int tileSize, int tileCount,
                              mfsr_stream_t stream); /* :149 */
int mfsr_boxFilterWithBorderY(const float* inTiles, float* outTiles, int maxShift, int tileSize, int tileCount,
                              mfsr_stream_t stream); /* :186 */
int mfsr_normalizedCC(const float* ccImage, const float* squaredTemplate, const float* boxFilteredImage,
                      float* shiftImage, int maxShift, int tileSize, int tileCount, mfsr_stream_t stream); /* :227 */
int mfsr_convertToTilesOverlapBorder(const float* inImg, float* outTiles, int imgWidth, int imgHeight, int imgPitch,
                                     int maxShift, int tileSize, int tileCountX, int tileCountY, mfsr_float2 baseShift,
                                     float baseRotation, mfsr_stream_t stream); /* :265 */
int mfsr_convertToTilesOverlapPreShift(const float* inImg, float* outTiles, const mfsr_float2* preShift,
                                       int preShiftPitch, int imgWidth, int imgHeight, int imgPitch, int maxShift,
                                       int tileSize, int tileCountX, int tileCountY, mfsr_float2 baseShift,
                                       float baseRotation, mfsr_stream_t stream); /* :324 */
int mfsr_GammasRGB(mfsr_float3* inOutImg, int imgWidth, int imgHeight, int imgPitch, mfsr_stream_t stream); /* :393 */
int mfsr_ApplyWeighting(mfsr_float3* inOutImg, const mfsr_float3* finalImg, const mfsr_float3* weight, int imgWidth,
                        int imgHeight, int imgPitch, float threshold, mfsr_stream_t stream); /* :426 */
int mfsr_conjugateComplexMulKernel(const mfsr_float2* aIn, mfsr_float2* bInOut, int maxElem,
                                   mfsr_stream_t stream); /* :485 */
int mfsr_findMinimum(const float* shiftImage, mfsr_float2* coordinates, int coordinatesPitch, int maxShift,
                     int tileCount, int tileCountX, float threshold, mfsr_stream_t stream); /* :512 */
int mfsr_UpSampleShifts(const mfsr_float2* inShift, mfsr_float2* outShift, int inPitch, int outPitch, int oldLevel,
                        int newLevel, int oldCountX, int oldCountY, int newCountX, int newCountY, int oldTileSize,
                        int newTileSize, mfsr_stream_t stream); /* :642 */
int mfsr_ComputeStructureTensor(const float* imgDx, const float* imgDy, mfsr_float3* outImg, int imgWidth,
                                int imgHeight, int imgDxDyPitch, int imgOutPitch, mfsr_stream_t stream); /* :691 */
int mfsr_ComputeKernelParam(mfsr_float3* kernelImg, int imgWidth, int imgHeight, int imgOutPitch, float Dth, float Dtr,
                            float kDetail, float kDenoise, float kStretch, float kShrink,
                            mfsr_stream_t stream); /* :718 */
int mfsr_fourierFilter(mfsr_float2* img, size_t stride, int width, int height, float lp, float hp, float lps,
                       float hps, int clearAxis, mfsr_stream_t stream);                             /* :794 */
int mfsr_fftshift(mfsr_float2* fft, int width, int height, mfsr_stream_t stream); /* :873 */

/* ---- C: ShiftMinimizerKernels.cu ------------------------------------------ */
int mfsr_copyShiftMatrix(float* matrices, int tileCount, int imageCount, int shiftCount,
                         mfsr_stream_t stream); /* :29 */
int mfsr_setPointers(float** shiftMatrixArray, float** shiftMatrixSafeArray, float** matrixSquareArray,
                     float** matrixInvertedArray, float** solvedMatrixArray, mfsr_float2** shiftOneToOneArray,
                     mfsr_float2** shiftMeasuredArray, mfsr_float2** shiftOptimArray, float* shiftMatrices,
                     float* shiftSafeMatrices, float* matricesSquared, float* matricesInverted, float* solvedMatrices,
                     mfsr_float2* shiftsOneToOne, mfsr_float2* shiftsMeasured, mfsr_float2* shiftsOptim, int tileCount,
                     int imageCount, int shiftCount, mfsr_stream_t stream); /* :51 */
int mfsr_checkForOutliers(mfsr_float2* measuredShifts, const float* optimShiftsT, float* shiftMatrix, int* status,
                          int* inversionInfo, int tileCount, int imageCount, int shiftCount,
                          mfsr_stream_t stream); /* :81 */
int mfsr_transposeShifts(mfsr_float2* measuredShifts, const float* measuredShiftsT, const float* shiftsOneToOneT,
                         mfsr_float2* shiftsOneToOne, int tileCount, int imageCount, int shiftCount,
                         mfsr_stream_t stream); /* :143 */
int mfsr_getOptimalShifts(mfsr_float2* optimalShifts, const mfsr_float2* bestShifts, int imageCount, int tileCountX,
                          int tileCountY, int optimalShiftsPitch, int referenceImage, int imageToTrack,
                          mfsr_stream_t stream); /* :179 */
int mfsr_concatenateShifts(const mfsr_float2* const* shiftIn, int* shiftInPitch, mfsr_float2* shiftOut, int shiftCount,
                           int tileCountX, int tileCountY, mfsr_stream_t stream); /* :223 */
int mfsr_separateShifts(const mfsr_float2* shiftIn, mfsr_float2* const* shiftOut, int* shiftOutPitch, int shiftCount,
                        int tileCountX, int tileCountY, mfsr_stream_t stream); /* :242 */
/* C4: the batched least-squares solve the reference leaves to a missing host
 * (batched cuBLAS upstream): per tile d = (A^T A)^-1 A^T b, o = A d.  A is
 * column-major m x (imageCount-1) per tile; optimShiftsT is planar
 * [x(m) | y(m)] per tile (what checkForOutliers reads, :114-115);
 * inversionInfo = 0 or (index of the zero pivot)+1.  imageCount-1 <= 63. */
int mfsr_solveShiftsBatched(const float* shiftMatrix, const mfsr_float2* measuredShifts, mfsr_float2* shiftsOneToOne,
                            float* optimShiftsT, int* inversionInfo, int tileCount, int imageCount, int shiftCount,
                            mfsr_stream_t stream);

/* C driver: iterate solve -> checkForOutliers until every tile's status is -1
 * (at most shiftCount+1 rounds; synchronises the stream once per round to read
 * the status array back).  status / inversionInfo: device int[tileCount]. */
int mfsr_minimizeShifts(float* shiftMatrix, mfsr_float2* measuredShifts, mfsr_float2* shiftsOneToOne,
                        float* optimShiftsT, int* status, int* inversionInfo, int tileCount, int imageCount,
                        int shiftCount, int* roundsOut, mfsr_stream_t stream);

/* the same loop inside one launch (the tiles are independent: one wavefront per tile iterates solve -> checkForOutliers
 * until its tile converges): no host synchronisation, graph-capturable, bit-identical to mfsr_minimizeShifts */
int mfsr_minimizeShiftsFused(float* shiftMatrix, mfsr_float2* measuredShifts, mfsr_float2* shiftsOneToOne, float* optimShiftsT,
                             int* status, int* inversionInfo, int tileCount, int imageCount, int shiftCount,
                             mfsr_stream_t stream);

/* ---- D/E: opticalFlow.cu -------------------------------------------------- */
/* texUV: CLAMP; texToWarp: MIRROR */
int mfsr_WarpingKernel(int width, int height, int stride, mfsr_tex2d texUV, float* out, mfsr_tex2d texToWarp,
                       mfsr_stream_t stream); /* :28 */
/* texObjShiftXY: CLAMP, tileCountX x tileCountY texels */
int mfsr_CreateFlowFieldFromTiles(mfsr_float2* outImg, mfsr_tex2d texObjShiftXY, int tileSize, int tileCountX,
                                  int tileCountY, int imgWidth, int imgHeight, int imgPitch, mfsr_float2 baseShift,
                                  float baseRotation, mfsr_stream_t stream); /* :48 */
/* texSource/texTarget: MIRROR, width x height */
int mfsr_ComputeDerivativesKernel(int width, int height, int stride, float* Ix, float* Iy, float* Iz,
                                  mfsr_tex2d texSource, mfsr_tex2d texTarget, mfsr_stream_t stream); /* :97 */
int mfsr_ComputeDerivatives2Kernel(int width, int height, int stride, float* Ix, float* Iy, mfsr_tex2d tex,
                                   mfsr_stream_t stream); /* :151 */
/* the same for image rows [row0, row0 + rows) only (Ix, Iy are the full-size images) */
int mfsr_ComputeDerivatives2Rows(int width, int height, int stride, float* Ix, float* Iy, mfsr_tex2d tex, int row0, int rows,
                                 mfsr_stream_t stream);
int mfsr_lucasKanadeOptim(mfsr_float2* shifts, const float* imFx, const float* imFy, const float* imFt, int pitchShift,
                          int pitchImg, int width, int height, int halfWindowSize, float minDet,
                          mfsr_stream_t stream); /* :190 */

/* ---- F: RobustnessModell.cu ----------------------------------------------- */
/* texUV: CLAMP */
int mfsr_ComputeRobustnessMask(const mfsr_float3* rawImgRef, const mfsr_float3* rawImgMoved,
                               mfsr_float4* robustnessMask, mfsr_tex2d texUV, int imgWidth, int imgHeight, int imgPitch,
                               int maskPitch, float alpha, float beta, float thresholdM,
                               mfsr_stream_t stream); /* :29 */

/* ---- host helpers of the reference ---------------------------------------- */
/* gaussin_filter_1D, test_opencv/main.cpp:370-391; taps must hold 99 floats;
 * returns the tap count (host function, no device work). */
int mfsr_gaussin_filter_1D(float sigma, float* taps);
/* sharpenImg2, finalProject/Project/multi_frame_sr.cpp:90-119 on DEVICE u8
 * interleaved images (step in bytes); never-written pixels are 0. */
int mfsr_sharpenImg2(const uint8_t* img, uint8_t* result, int rows, int cols, int ch, int stepIn, int stepOut,
                     mfsr_stream_t stream);

/* sharpenImg, test_opencv/main.cpp:525-534: unsharp mask (sigma 1, threshold 5, amount 1) on DEVICE u8 interleaved
 * images; tmp: rows*cols*ch device bytes.  The Gaussian blur is third-party there (cv::GaussianBlur): restated from its
 * published definition, parity unpinned (oracle/glue.c). */
int mfsr_sharpenImg(const uint8_t* img, uint8_t* result, uint8_t* tmp, int rows, int cols, int ch, int stepIn, int stepOut,
                    mfsr_stream_t stream);

/* ---- glue stages between the reference kernels (the build's own; the
 *      reference has no host for this path -- DESIGN.md "Pipeline glue") ---- */
int mfsr_rgbToGray(const mfsr_float3* in, int inPitch, float* out, int outPitch, int width, int height,
                   mfsr_stream_t stream);
int mfsr_u16ToFloat(const uint16_t* in, float* out, int outPitch, int width, int height, float factor,
                    mfsr_stream_t stream);
/* separable filter, clamped borders, chan = 1 or 3, ntaps <= 99 (taps on HOST) */
int mfsr_separableFilter(const float* in, int inPitch, float* tmp, float* out, int outPitch, int width, int height,
                         int chan, const float* taps, int ntaps, mfsr_stream_t stream);
int mfsr_downsample2x(const float* in, int inPitch, float* out, int outPitch, int outW, int outH, mfsr_stream_t stream);
/* direct correlation replacing FFT -> conjugateComplexMulKernel -> IFFT; output
 * in the wrapped layout normalizedCC reads (kernel.cu:248-254) */
int mfsr_crossCorrelateTiles(const float* refTiles, const float* movedTiles, float* ccImage, int maxShift, int tileSize,
                             int tileCount, mfsr_stream_t stream);
int mfsr_addRoundedPreShift(const mfsr_float2* preShift, int prePitch, mfsr_float2* found, int foundPitch, int countX,
                            int countY, mfsr_stream_t stream);
int mfsr_scaleFlow(mfsr_float2* flow, int pitch, int width, int height, float factor, mfsr_stream_t stream);
int mfsr_float3ToFloat4(const mfsr_float3* in, int inPitch, mfsr_float4* out, int outPitch, int width, int height,
                        mfsr_stream_t stream);
int mfsr_resampleFloat3(const mfsr_float3* in, int inPitch, int inW, int inH, mfsr_float3* out, int outPitch, int outW,
                        int outH, float u0, float u1, float v0, float v1, mfsr_stream_t stream);
/* exactly one of out16/out8 non-NULL; dense interleaved RGB */
int mfsr_quantize(const mfsr_float3* in, int inPitch, uint16_t* out16, uint8_t* out8, int width, int height,
                  float maxOut, mfsr_stream_t stream);
int mfsr_fill_f32(float* dst, size_t count, float value, mfsr_stream_t stream);
/* deBayersSubSample3 (A1) + mfsr_rgbToGray + mfsr_separableFilter + the first mfsr_downsample2x in one
 * launch (what the burst driver does to every Bayer frame before tracking); bit-identical to the chain.
 * dimX x dimY is the half-resolution size; pyr1 may be NULL; ntaps odd, <= 17. */
int mfsr_prepareFrameFused(const uint16_t* dataIn, mfsr_float3* halfOut, int halfPitch, float maxVal, int dimX, int dimY,
                           float* pyr0, int pyr0Pitch, float* pyr1, int pyr1Pitch, const float* taps, int ntaps,
                           mfsr_stream_t stream);
/* one-pixel border ring of a float4 image := 0: what ComputeRobustnessMask (:37) leaves unwritten */
int mfsr_zeroRing_f32x4(mfsr_float4* img, int pitch, int width, int height, mfsr_stream_t stream);
/* variant selector of the accumulate kernels (tests, A/B benchmarks):
 * 0 = straight kernel, exp(-w/2) with the ocml expf (tight parity against the oracle);
 * 1 = straight kernel, v_exp_f32; 2 (default) = additionally the restructured x2
 * strip kernel where it applies (scale 2, Bayer/mono CFA, 16-byte aligned accumulators). */
int mfsr_set_accumulate_fast_exp(int enable);

/* ---- fused MI355X kernels (same results as the chains they replace, within
 *      the tolerances stated in DESIGN.md) --------------------------------- */
/* B1+B2+B3+B4+cc+B6+B7 in one launch, one workgroup per tile: gathers the
 * reference tile and the pre-shifted moved patch into LDS, evaluates the L2
 * distance image directly and reduces it with wavefront shuffles. */
int mfsr_trackTilesFused(const float* refImg, const float* movedImg, const mfsr_float2* preShift, int preShiftPitch,
                         mfsr_float2* coordinates, int coordinatesPitch, int imgWidth, int imgHeight, int imgPitch,
                         int maxShift, int tileSize, int tileCountX, int tileCountY, float threshold,
                         const float* refSquaredSums, mfsr_stream_t stream);
/* sum(ref^2) of every reference tile in the serial order of squaredSum (B3, kernel.cu:119): it does
 * not depend on the moved frame, so a burst takes it once per reference and hands it to
 * mfsr_trackTilesFused (refSquaredSums; NULL = taken inside the tracker). */
int mfsr_tileSquaredSums(const float* refImg, float* outValues, int imgWidth, int imgHeight, int imgPitch, int maxShift,
                         int tileSize, int tileCountX, int tileCountY, mfsr_stream_t stream);
/* D2+D3+D4 for one Lucas-Kanade iteration in one launch (LDS-tiled warp,
 * derivative and separable window sums).  Flow is double-buffered: shiftsOut
 * must not alias shiftsIn (tile halos read neighbouring tiles' flow).  outScale
 * multiplies the flow written (1 = plain iteration; the last iteration of a
 * pipeline passes its tracking->raw pixel factor instead of a mfsr_scaleFlow pass). */
int mfsr_lucasKanadeIterationFused(const mfsr_float2* shiftsIn, mfsr_float2* shiftsOut, int pitchShift,
                                   const float* refImg, const float* movedImg, int pitchImg, int width, int height,
                                   int halfWindowSize, float minDet, float outScale, mfsr_stream_t stream);
/* F1 (ComputeRobustnessMask, RobustnessModell.cu:29) + its zero ring (:48-49) in one launch: reference patch through an
 * LDS tile, hardware sqrt / rcp / exp (mask within 2e-6 of the straight kernel; the roundings and the M threshold keep
 * their exact arithmetic).  mfsr_set_robustness_fast(0) routes it to ring + mfsr_ComputeRobustnessMask. */
int mfsr_robustnessMaskFused(const mfsr_float3* rawImgRef, const mfsr_float3* rawImgMoved, mfsr_float4* robustnessMask,
                             mfsr_tex2d texUV, int imgWidth, int imgHeight, int imgPitch, int maskPitch, float alpha, float beta,
                             float thresholdM, mfsr_stream_t stream);
int mfsr_set_robustness_fast(int enable);
/* E1+E2 (derivatives + structure tensor) in one launch */
int mfsr_structureTensorFused(const float* img, int imgPitch, mfsr_float3* outImg, int outPitch, int width, int height,
                              mfsr_stream_t stream);
/* A2+A3 (+u16 -> float) in one launch through an LDS green tile */
int mfsr_deBayerFused(const uint16_t* raw, mfsr_float3* outImage, int strideOut, int width, int height,
                      mfsr_float3 blackPoint, mfsr_float3 scale, mfsr_stream_t stream);
/* H1 (+fallback resample) + H2 + quantise in one launch */
int mfsr_finishFused(const mfsr_float3* finalImg, const mfsr_float3* weight, int imgPitch, const mfsr_float3* fallback,
                     int fbPitch, int fbW, int fbH, float u0, float u1, float v0, float v1, mfsr_float3* outImg,
                     int outPitch, uint16_t* out16, int width, int height, float threshold, int applyGamma,
                     float maxOut, mfsr_stream_t stream);

/* ---- global pre-alignment.  The reference has the slot -- `class PreAlignment`,
 *      boxFilterNPP.cpp:102-166; baseShift / baseRotation of kernel.cu:265,324 and opticalFlow.cu:48 -- but no finished
 *      estimator (test_opencv/main.cpp:861-1194 returns nothing).  Model fixed by those kernels: reference pixel p maps
 *      to moved pixel q = c + R(rotation) * (p - c - shift), c = (width/2, height/2).  The estimator is the build's own
 *      (csrc/prealign.hip): exhaustive coarse-to-fine search on 2x2-mean pyramids, integer scores, angles on a 1/16
 *      degree grid; no host round trip, the result stays in device memory. ---------------------------------------- */
typedef struct {
    float shiftX, shiftY;   /* base shift in pixels of the image the pyramids were built from */
    float rotation;         /* base rotation in radians (= angleIndex * pi/2880) */
    float cosRotation, sinRotation; /* cosf/sinf(rotation) of the host's libm */
    int32_t angleIndex, tx, ty, level; /* raw search result: angle in 1/16 degree, shift at pyramid level `level` */
    int32_t reserved[3];
} mfsr_prealign;
/* device bytes of one image's search pyramid / of the search workspace (trig table, scores, per-level state) */
size_t mfsr_preAlign_pyramid_bytes(int width, int height);
size_t mfsr_preAlign_workspace_bytes(float maxAngleDeg);
/* uploads the trig table for |angle| <= maxAngleDeg into the workspace (once per workspace) */
int mfsr_preAlign_init(void* workspace, float maxAngleDeg, mfsr_stream_t stream);
/* 2x2-mean pyramid of img (float, pitched) down to a long side <= 64, quantised to 8 bit */
int mfsr_preAlignPyramid(const float* img, int width, int height, int pitch, void* pyramid, mfsr_stream_t stream);
/* search: *result (DEVICE memory) := base shift / rotation of the moved image against the reference image */
int mfsr_preAlign(const void* refPyramid, const void* movedPyramid, int width, int height, float maxAngleDeg,
                  void* workspace, mfsr_prealign* result, mfsr_stream_t stream);
int mfsr_preAlign_identity(mfsr_prealign* result, mfsr_stream_t stream);
/* mfsr_trackTilesFused with B2's baseShift / baseRotation taken from *base (device; NULL = none); the base shift is
 * multiplied by baseInvScale (1 / down-sampling factor of this pyramid level) */
int mfsr_trackTilesFusedBase(const float* refImg, const float* movedImg, const mfsr_float2* preShift, int preShiftPitch,
                             mfsr_float2* coordinates, int coordinatesPitch, int imgWidth, int imgHeight, int imgPitch,
                             int maxShift, int tileSize, int tileCountX, int tileCountY, float threshold,
                             const float* refSquaredSums, const mfsr_prealign* base, float baseInvScale,
                             mfsr_stream_t stream);
/* mfsr_trackTilesFusedBase with UpSampleShifts (B8, kernel.cu:642) folded in: every tile's pre-shift is taken inside the
 * kernel from the previous level's shifts (oldCountX x oldCountY tiles of oldTileSize at down-sampling factor oldLevel)
 * with B8's own arithmetic: same bits, one launch and one buffer less per pyramid level */
int mfsr_trackTilesFusedUp(const float* refImg, const float* movedImg, const mfsr_float2* coarseShifts, int coarsePitch,
                           int oldLevel, int newLevel, int oldCountX, int oldCountY, int oldTileSize, mfsr_float2* coordinates,
                           int coordinatesPitch, int imgWidth, int imgHeight, int imgPitch, int maxShift, int tileSize,
                           int tileCountX, int tileCountY, float threshold, const float* refSquaredSums,
                           const mfsr_prealign* base, float baseInvScale, mfsr_stream_t stream);
/* mfsr_CreateFlowFieldFromTiles (opticalFlow.cu:48) with baseShift / baseRotation taken from *base (device) */
int mfsr_CreateFlowFieldFromTilesBase(mfsr_float2* outImg, mfsr_tex2d texObjShiftXY, int imgWidth, int imgHeight,
                                      int imgPitch, const mfsr_prealign* base, mfsr_stream_t stream);
/* The same iteration with the warped moved image handed over between launches instead of re-gathered for every tile halo:
 * sumIn / diffIn = (warped + ref) / (warped - ref) of every pixel under shiftsIn (made by mfsr_CreateFlowFieldWarped or by
 * the previous call's sumOut / diffOut); sumOut / diffOut (both NULL on the last iteration) receive them under shiftsOut,
 * taken by the thread that has just updated the pixel.  Bit-identical to mfsr_lucasKanadeIterationFused. */
int mfsr_lucasKanadeIterationWarped(const mfsr_float2* shiftsIn, mfsr_float2* shiftsOut, int pitchShift, const float* refImg,
                                    const float* movedImg, int pitchImg, const float* sumIn, const float* diffIn, float* sumOut,
                                    float* diffOut, int pitchSD, int width, int height, int halfWindowSize, float minDet,
                                    float outScale, mfsr_stream_t stream);
/* ---- frame batches: the per-frame stages of the alignment for 1 .. 4 moved frames against one reference in ONE launch each
 * (gridDim.z = frame; the kernels are the single-frame entry points' own, so a frame's result does not depend on the batch it
 * is in).  mfsr_burst_add_frame aligns the frames of a fuse group this way. */
typedef struct {
    const uint16_t* dataIn;
    mfsr_float3* halfOut;
    float* pyr0;
    float* pyr1; /* NULL for all frames: no second pyramid level */
} mfsr_prepare_frame;
int mfsr_prepareFrameFusedBatch(int nFrames, const mfsr_prepare_frame* frames, int halfPitch, float maxVal, int dimX, int dimY,
                                int pyr0Pitch, int pyr1Pitch, const float* taps, int ntaps, mfsr_stream_t stream);
typedef struct {
    const float* movedImg;
    const mfsr_float2* coarseShifts; /* this frame's shifts of the coarser level (mfsr_trackTilesFusedUp); NULL for all: none */
    mfsr_float2* coordinates;
    const mfsr_prealign* base; /* or NULL */
} mfsr_track_frame;
int mfsr_trackTilesFastSupported(int tileSize, int maxShift); /* 1: mfsr_trackTilesFusedBatch takes this (tile size, search range) */
int mfsr_trackTilesFusedBatch(int nFrames, const mfsr_track_frame* frames, const float* refImg, int coarsePitch, int oldLevel,
                              int newLevel, int oldCountX, int oldCountY, int oldTileSize, int coordinatesPitch, int imgWidth,
                              int imgHeight, int imgPitch, int maxShift, int tileSize, int tileCountX, int tileCountY, float threshold,
                              const float* refSquaredSums, float baseInvScale, mfsr_stream_t stream);
typedef struct {
    mfsr_float2* outImg;
    const mfsr_float2* tileShifts;
    const mfsr_prealign* base; /* NULL for all frames, or set for all */
    const float* movedImg;
    float* sumOut;
    float* diffOut;
} mfsr_flowfield_frame;
int mfsr_CreateFlowFieldWarpedBatch(int nFrames, const mfsr_flowfield_frame* frames, int tilePitch, int tileCountX, int tileCountY,
                                    int imgWidth, int imgHeight, int imgPitch, const float* refImg, int pitchImg, int pitchSD,
                                    mfsr_stream_t stream);
typedef struct {
    const mfsr_float3* movedHalf;
    mfsr_float4* mask;
    const mfsr_float2* flow;
} mfsr_robustness_frame;
int mfsr_robustnessMaskFusedBatch(int nFrames, const mfsr_robustness_frame* frames, const mfsr_float3* rawImgRef, int flowPitch,
                                  int flowWidth, int flowHeight, int imgWidth, int imgHeight, int imgPitch, int maskPitch, float alpha,
                                  float beta, float thresholdM, mfsr_stream_t stream);
/* The iteration of mfsr_lucasKanadeIterationWarped for 1 .. 4 frames against one reference in ONE launch (csrc/lk_fused.hip,
 * k_lkSweep: one wavefront per 64 columns sweeping down a band of rows, vertical state in registers, horizontal window sums
 * through whole-wave DPP shifts, no LDS).  Agrees with mfsr_lucasKanadeIterationWarped to fp32 rounding (another
 * summation order of the row sums).  MFSR_E_UNSUPPORTED (nothing launched) for half windows outside 1..7 or width < 64:
 * call the per-frame entry point then. */
typedef struct {
    const mfsr_float2* shiftsIn;
    mfsr_float2* shiftsOut;
    const float* movedImg;
    const float* sumIn;
    const float* diffIn;
    float* sumOut;  /* NULL (with diffOut) on the last iteration */
    float* diffOut;
} mfsr_lk_frame;
int mfsr_lucasKanadeSweepBatch(int nFrames, const mfsr_lk_frame* frames, const float* refImg, int pitchShift, int pitchImg,
                               int pitchSD, int width, int height, int halfWindowSize, float minDet, float outScale,
                               mfsr_stream_t stream);
/* D1 (mfsr_CreateFlowFieldFromTiles; base != NULL: mfsr_CreateFlowFieldFromTilesBase) + the warp of every pixel under the flow
 * it writes: outImg and the first iteration's sumIn / diffIn in one launch (opticalFlow.cu:48 + :28) */
int mfsr_CreateFlowFieldWarped(mfsr_float2* outImg, mfsr_tex2d texObjShiftXY, int imgWidth, int imgHeight, int imgPitch,
                               mfsr_float2 baseShift, float baseRotation, const mfsr_prealign* base, const float* refImg,
                               const float* movedImg, int pitchImg, float* sumOut, float* diffOut, int pitchSD,
                               mfsr_stream_t stream);

/* mfsr_finishFused on rows [rowOffset, rowOffset + height) of a fullHeight-row image (pointers = first row of the stripe;
 * u/v window = that of the WHOLE image): bit-identical to the rows of the whole-image call */
int mfsr_finishFusedRows(const mfsr_float3* finalImg, const mfsr_float3* weight, int imgPitch, const mfsr_float3* fallback,
                         int fbPitch, int fbW, int fbH, float u0, float u1, float v0, float v1, mfsr_float3* outImg,
                         int outPitch, uint16_t* out16, int width, int height, float threshold, int applyGamma, float maxOut,
                         int rowOffset, int fullHeight, mfsr_stream_t stream);

/* ---- burst pipeline (the L3 driver the reference lacks; mirrors the CLI
 *      contract of finalProject/Project/multi_frame_sr.cpp:122-210) ---------- */
typedef struct {
    int32_t width, height;   /* raw / LR frame size (even) */
    int32_t frames;          /* N */
    int32_t reference;       /* index of the reference frame */
    int32_t scale;           /* output scale s (1..8) */
    int32_t mono;            /* 0: Bayer mosaic per cfa; 1: monochrome (config 1) */
    int32_t cfa[4];          /* CFA pattern (ignored when mono) */
    float black[3], white[3]; /* per-channel levels: norm = (raw - black)/white */
    float maxVal;            /* deBayersSubSample3's maxVal */
    /* tile tracker */
    int32_t levels;          /* pyramid levels (1..4), coarsest first */
    int32_t levelFactor[4];  /* down-sampling factor per level (power of two) */
    int32_t tileSize[4];
    int32_t maxShift[4];
    float minimumThreshold;  /* findMinimum threshold */
    float sigmaTracking;     /* gaussin_filter_1D sigma of the tracking prefilter (main.cpp:1868) */
    /* Lucas-Kanade refinement */
    int32_t lkIterations;
    int32_t lkHalfWindow;
    float lkMinDet;
    /* robustness */
    float alpha, beta, thresholdM;
    /* kernel shape */
    float sigmaTensor;
    float Dth, Dtr, kDetail, kDenoise, kStretch, kShrink;
    /* finish */
    float weightThreshold;
    int32_t applyGamma;
    int32_t fused;           /* 1: fused MI355X kernels; 0: one launch per reference kernel */
    int32_t pairFrames;      /* frames per warp+fuse launch of add_frame (see mfsr_burst_add_frame): 0 = one, like the
                                reference; 1 (default) = as many as one launch takes (MFSR_MAX_FUSE_GROUP at scale 2 and 4 Bayer, else 2);
                                2 .. MFSR_MAX_FUSE_GROUP = that many */
    int32_t asyncFuse;       /* 1: the warp+fuse launches run on a stream owned by the burst, beside the alignment of the
                                following frames on the caller's stream (see mfsr_burst_add_frame); 0 (default since round 4:
                                the four-workgroup fuse kernel leaves no room on a CU for the alignment to run beside it, so
                                one stream is 1 % faster): everything on the caller's stream.  Same bits either way. */
    int32_t preAlign;        /* 1: estimate a global base shift + rotation per moved frame (mfsr_preAlign) and feed it to
                                the tile tracker and the flow field (baseShift / baseRotation of kernel.cu:324, opticalFlow.cu:48) */
    float preAlignMaxAngle;  /* search range of the base rotation in degrees (default 20) */
    int32_t uploadRing;      /* > 0: the workspace holds this many device raw-frame slots (>= 3) + 2 reference slots and the
                                burst owns a copy stream: mfsr_burst_*_host take frames from (pinned) HOST memory and upload
                                them ahead of the compute (BASELINE configs[4]: double-buffered H2D) */
    int32_t reserved[2];
} mfsr_config;

typedef struct mfsr_burst mfsr_burst;

/* fill *cfg with the build's defaults for a width x height x frames burst */
int mfsr_config_default(mfsr_config* cfg, int width, int height, int frames, int scale, int mono);
/* device scratch the pipeline needs for cfg (bytes) */
size_t mfsr_burst_workspace_bytes(const mfsr_config* cfg);
/* bytes of ONE accumulator plane-set (imgOut or totalWeights): float3, pitch = 12*scale*width */
size_t mfsr_burst_accumulator_bytes(const mfsr_config* cfg);
int mfsr_burst_create(mfsr_burst** out, const mfsr_config* cfg, void* workspace, size_t workspaceBytes);
void mfsr_burst_destroy(mfsr_burst* b);
/* Prepare the reference frame (tracking pyramid, half-res RGB, kernel
 * parameters, fallback image).  rawRef: dense u16 width x height on device. */
int mfsr_burst_set_reference(mfsr_burst* b, const uint16_t* rawRef, mfsr_stream_t stream);
/* mfsr_burst_set_reference for a burst whose fuse and finish touch HR rows [hrRow0, hrRow1) only (a stripe of a multi-GPU
 * burst): the products the alignment reads (half-resolution RGB, tracking pyramid, tile sums) are complete, the kernel
 * parameters and the debayered fallback image are made for the rows that stripe's mfsr_burst_fuse_rows /
 * mfsr_burst_finish_rows read -- bit-identical there to mfsr_burst_set_reference's, undefined elsewhere. */
int mfsr_burst_set_reference_rows(mfsr_burst* b, const uint16_t* rawRef, int hrRow0, int hrRow1, mfsr_stream_t stream);
/* Align + robustness + accumulate ONE frame into the caller's accumulators
 * (float3 HR, pitch 12*scale*width; zeroed by the caller before the first
 * call).  isReference != 0: identity flow, certainty 1.
 * With cfg.pairFrames the warp+fuse of a frame is deferred until its group (2 .. MFSR_MAX_FUSE_GROUP frames,
 * mfsr_burst_group_size) is aligned and the whole group is fused in one pass over the accumulators (a fraction of the
 * accumulator traffic, and everything that depends on the reference only -- kernel parameters, tap weights -- once
 * per pixel): between groups up to group - 1 frames are still waiting -- their raw buffers must stay untouched and
 * the accumulators do not contain them -- until the next add_frame, mfsr_burst_flush, finish or finish_rows has been
 * issued on the stream.
 * With cfg.asyncFuse the warp+fuse launches go to a high-priority stream the burst owns (ordered by
 * events after the alignment on the caller's stream), so the fuse of frames k, k+1 overlaps the
 * alignment of k+2, k+3.  The caller's stream sees the accumulators complete only after
 * mfsr_burst_flush / finish / finish_rows / set_reference (they join the two streams), and every raw
 * buffer handed to add_frame must stay untouched until one of those has been issued. */
int mfsr_burst_add_frame(mfsr_burst* b, const uint16_t* raw, int isReference, mfsr_float3* imgOut,
                         mfsr_float3* totalWeights, mfsr_stream_t stream);
/* Start a new burst on these accumulators WITHOUT zeroing them: the first warp+fuse launch that follows
 * overwrites them (as if zeroed), which saves the memset and the first read of both planes.  If no frame
 * is added before flush / finish, they are zeroed then.  Without this call the caller zeroes them. */
int mfsr_burst_begin(mfsr_burst* b, mfsr_float3* imgOut, mfsr_float3* totalWeights, mfsr_stream_t stream);
/* fuse a frame that is still waiting for its partner and make the caller's stream wait for every fuse issued so far */
int mfsr_burst_flush(mfsr_burst* b, mfsr_stream_t stream);
/* ApplyWeighting (+fallback) + optional gamma; outImg float3 HR (may be NULL),
 * out16 dense interleaved u16 HR (may be NULL). */
int mfsr_burst_finish(mfsr_burst* b, const mfsr_float3* imgOut, const mfsr_float3* totalWeights, mfsr_float3* outImg,
                      uint16_t* out16, mfsr_stream_t stream);
/* finish restricted to HR rows [row0, row0+rows) (pointers are those of the
 * full images): each rank of a reduce-scattered burst normalises its stripe.
 * Fused kernels only. */
int mfsr_burst_finish_rows(mfsr_burst* b, const mfsr_float3* imgOut, const mfsr_float3* totalWeights,
                           mfsr_float3* outImg, uint16_t* out16, int row0, int rows, mfsr_stream_t stream);
/* ---- bursts whose frames live in HOST memory (cfg.uploadRing > 0).  Same semantics as set_reference / add_frame /
 * finish, with the H2D copy of every frame enqueued by the library on a copy stream it owns, into a ring of device
 * slots inside the workspace, so that the upload of frame k+1.. overlaps the align+fuse of frame k (the reference
 * uploads its frames one blocking copy at a time, multi_frame_sr.cpp:167-174).  hostRaw must stay valid and unchanged
 * until the stream has passed the call's work; pinned memory (hipHostMalloc) is what makes the copies asynchronous.
 * add_frame_host(isReference) with the pointer last given to set_reference_host re-uses the uploaded reference. */
int mfsr_burst_set_reference_host(mfsr_burst* b, const uint16_t* hostRaw, mfsr_stream_t stream);
int mfsr_burst_add_frame_host(mfsr_burst* b, const uint16_t* hostRaw, int isReference, mfsr_float3* imgOut,
                              mfsr_float3* totalWeights, mfsr_stream_t stream);
/* Optional: enqueue the uploads of the next nFrames frames (in the order they will be handed to add_frame_host) right
 * away, before any of their kernels.  add_frame_host enqueues a frame's ~10 alignment launches after its copy, so the NEXT
 * frame's copy reaches the copy engine only when the host is through with those -- 16 us between copies most of the time,
 * 70-300 us at group boundaries (one burst at 4K x 16: the uploads end at 5.7 instead of 5.0 ms).  With the copies queued up
 * front the engine runs them back to back and every add_frame_host finds its frame already on its way (matched by the
 * host pointer; a frame that was not announced is uploaded as before).  At most cfg.uploadRing frames are taken, the
 * pointer of the current host reference is skipped; call it after set_reference_host. */
int mfsr_burst_prefetch_host(mfsr_burst* b, const uint16_t* const* hostRaws, int nFrames, mfsr_stream_t stream);
/* mfsr_burst_finish into out16Dev (device), then its D2H copy into out16Host on a stream the burst owns, so that the next
 * burst's uploads and kernels overlap the download (full-duplex PCIe).  out16Host is complete after
 * mfsr_burst_host_sync(b) (blocks the HOST on the download); out16Dev must not be written by the caller before that. */
int mfsr_burst_finish_host(mfsr_burst* b, const mfsr_float3* imgOut, const mfsr_float3* totalWeights, uint16_t* out16Dev,
                           uint16_t* out16Host, mfsr_stream_t stream);
int mfsr_burst_host_sync(mfsr_burst* b);
/* ---- building blocks of stripe-sharded bursts (multi-GPU, include/mfsr_dist.h): a frame is ALIGNED on the rank that
 * holds it (flow field + certainty mask into caller buffers, no accumulation), the ranks exchange the rows of raw / flow /
 * mask their stripes need, and every rank FUSES all frames, in frame order, onto its own stripe of HR rows -- the
 * summation order of the single-GPU burst, so the result is bit-identical to it. */
/* flow field: float2 flowW x flowH (tracking resolution, raw-pixel units); certainty mask: float4 maskW x maskH */
int mfsr_burst_field_dims(const mfsr_burst* b, int* flowW, int* flowH, int* maskW, int* maskH);
/* stages A1, (I), B, D, F of mfsr_burst_add_frame without G; results copied to flowOut / maskOut (byte pitches) */
int mfsr_burst_align_frame(mfsr_burst* b, const uint16_t* raw, int isReference, mfsr_float2* flowOut, int flowPitch,
                           mfsr_float4* maskOut, int maskPitch, mfsr_stream_t stream);
/* frames per warp+fuse launch cfg.pairFrames stands for with this configuration (1 .. MFSR_MAX_FUSE_GROUP) */
/* mfsr_burst_align_frame for nFrames frames (isReference: per-frame flags or NULL): groups of up to mfsr_burst_group_size
 * frames share their launches (one per stage and Lucas-Kanade iteration); same results as frame by frame */
int mfsr_burst_align_frames(mfsr_burst* b, int nFrames, const uint16_t* const* raws, const int* isReference,
                            mfsr_float2* const* flowOut, int flowPitch, mfsr_float4* const* maskOut, int maskPitch,
                            mfsr_stream_t stream);
int mfsr_burst_group_size(const mfsr_config* cfg);
/* stage G for 1 .. MFSR_MAX_FUSE_GROUP aligned frames on HR rows [rowBegin, rowEnd) (see mfsr_accumulateSuperResFullRows), with the kernel
 * parameters of the burst's reference */
int mfsr_burst_fuse_rows(mfsr_burst* b, int nFrames, const uint16_t* const* raws, const mfsr_float2* const* flows, int flowPitch,
                         const mfsr_float4* const* masks, int maskPitch, mfsr_float3* imgOut, mfsr_float3* totalWeights,
                         int accumulatorsUndefined, int rowBegin, int rowEnd, mfsr_stream_t stream);

/* rows rank `rank` of `worldSize` owns and reads (host arithmetic only): HR rows [rowBegin, rowEnd) (multiples of 16), and
 * of every frame's products the flow rows, certainty rows and raw rows its fuse can touch as long as the vertical flow
 * stays within maxFlowY raw pixels (rawHalo - 3); mfsr_checkFlowBound verifies that on the device. */
typedef struct {
    int32_t rowBegin, rowEnd;
    int32_t flowRow0, flowRows;
    int32_t maskRow0, maskRows;
    int32_t rawRow0, rawRows;
    float maxFlowY;
    int32_t reserved[3];
} mfsr_stripe_plan;
int mfsr_dist_stripe_plan(const mfsr_config* cfg, int worldSize, int rank, int rawHalo, mfsr_stripe_plan* out);
/* *flag |= 1 (device int) if any |flow.y| of the `rows` flow rows starting at `flow` exceeds bound (NaN passes: it
 * rounds to a zero shift) */
int mfsr_checkFlowBound(const mfsr_float2* flow, int pitch, int width, int rows, float bound, int* flag, mfsr_stream_t stream);
/* *maxBits = max(*maxBits, bit pattern of |flow.y|) over the rows (device int, zero it first; non-negative floats order
 * like their bit patterns; NaN is skipped): the measured vertical flow a multi-GPU caller sizes the raw-row halo of the
 * next bursts from (mfsr_dist_measured_flow) */
int mfsr_maxAbsFlowY(const mfsr_float2* flow, int pitch, int width, int rows, int* maxBits, mfsr_stream_t stream);
/* Host-only diagnostic (no device call): 1 if the kernels divide by `d` -- an image dimension, the divisor of the reference's
 * normalised texture coordinates (opticalFlow.cu:38-39, :88; RobustnessModell.cu:59-77) -- with the reciprocal sequence
 * q = x (1/d), q' = fma(fma(-d, q, x), 1/d, q), which this library takes only after checking on the host, over all 2^23
 * significands, that it IS the correctly rounded x / d for every x; 0 if they keep the division (check failed, d outside
 * (0, 2^24), or MFSR_EXACT_DIV=0). */
int mfsr_exactDivisionOk(float d);

/* ---- frame streams (BASELINE configs[4]): a sliding window of 2*radius+1 frames around every
 * frame, the reference's setTemporalAreaRadius(1) (finalProject/Project/multi_frame_sr.cpp:182).  Output t fuses frames
 * [t-radius, t+radius] (clipped to the stream) with frame t as the reference -- exactly what one mfsr_burst_* burst per
 * window gives -- but every frame is uploaded and PREPARED once (A1 half-resolution RGB, tracking pyramid, pre-alignment
 * search pyramid), not once per window it takes part in.  cfg->frames is ignored.  framesInHostMemory: frames are
 * (pinned) host pointers, uploaded on a copy stream the context owns, ahead of the compute. */
typedef struct mfsr_stream mfsr_stream;
size_t mfsr_stream_workspace_bytes(const mfsr_config* cfg, int radius);
int mfsr_stream_create(mfsr_stream** out, const mfsr_config* cfg, int radius, int framesInHostMemory, void* workspace,
                       size_t workspaceBytes);
void mfsr_stream_destroy(mfsr_stream* s);
/* hand over frame t = 0, 1, 2, ... (copied: the caller may reuse its buffer once the stream has passed the call).  From
 * t = radius on each call also produces output t - radius into outImg (float3 HR, may be NULL) / out16 (u16 HR, may be
 * NULL) and sets *produced to its index; before that *produced = -1. */
int mfsr_stream_push(mfsr_stream* s, const uint16_t* frame, mfsr_float3* outImg, uint16_t* out16, long long* produced,
                     mfsr_stream_t stream);
/* end of the stream: each call produces the next outstanding output (windows clipped at the last frame); *produced = -1
 * when none is left */
int mfsr_stream_drain(mfsr_stream* s, mfsr_float3* outImg, uint16_t* out16, long long* produced, mfsr_stream_t stream);
int mfsr_stream_reset(mfsr_stream* s);

/* ---- whole burst with the JOINT SHIFT MINIMISER (stage C; ShiftMinimizerKernels.cu:81-258) in the loop.  Besides every
 * (reference, k) pair the tracker measures every neighbouring pair (k, k+1); per tile the frame-to-frame shifts are the
 * least-squares solution of all measurements with the worst outlier dropped until every residual is below 1 px^2, and
 * the reference->k shifts (getOptimalShifts) replace the tracker's in the per-frame chain.  frames: cfg.frames device
 * pointers; jointWorkspace: mfsr_burst_joint_workspace_bytes(cfg) device bytes; accumulators as for add_frame (a
 * mfsr_burst_begin is issued inside); finish with mfsr_burst_finish.  Needs cfg.fused = 1, cfg.preAlign = 0. */
size_t mfsr_burst_joint_workspace_bytes(const mfsr_config* cfg);
int mfsr_burst_process_joint(mfsr_burst* b, const uint16_t* const* frames, void* jointWorkspace, size_t jointBytes,
                             mfsr_float3* imgOut, mfsr_float3* totalWeights, mfsr_stream_t stream);

/* ---- frame-source plug-in: the pull model of the reference's cv::superres::FrameSource subclass
 * (MultiFrameSource_CUDA, finalProject/Project/multi_frame_sr.cpp:18-49: nextFrame copies the next device-resident frame
 * into the caller's buffer and leaves it empty when exhausted; reset rewinds).  Same ownership: the library owns the
 * destination (a slot of the burst's upload ring, cfg.uploadRing >= 3), the callee fills it. */
typedef struct {
    /* copy the next frame (dense u16, width x height) into dst (DEVICE memory) with work enqueued on `stream`;
     * return 1 = delivered, 0 = source exhausted, < 0 = error (returned to the caller of process_source) */
    int (*next_frame)(void* user, uint16_t* dst, mfsr_stream_t stream);
    void (*reset)(void* user);   /* may be NULL */
    void* user;
} mfsr_frame_source;
/* reset, then pull up to cfg.frames frames (the first one is the reference: cfg.reference must be 0), fuse them and
 * finish into outImg / out16 (either may be NULL); *framesUsed = frames delivered.  imgOut / totalWeights: accumulators. */
int mfsr_burst_process_source(mfsr_burst* b, const mfsr_frame_source* src, mfsr_float3* imgOut, mfsr_float3* totalWeights,
                              mfsr_float3* outImg, uint16_t* out16, int* framesUsed, mfsr_stream_t stream);

/* HIP-event timing of the warp+fuse (accumulate) launches made by add_frame on
 * the caller's stream: timing(b,1) starts a series, timing_read synchronises with
 * the events and returns the summed kernel milliseconds, the launch count and the
 * number of frames those launches fused (2 per launch with pairFrames). */
int mfsr_burst_timing(mfsr_burst* b, int enable);
int mfsr_burst_timing_read(mfsr_burst* b, double* totalMs, int* launches, int* frames);
/* last per-frame flow field (tracking resolution, raw-pixel units) and mask,
 * for tests: returns device pointers valid until the next add_frame.  With frame-batched alignment a frame is aligned
 * when its group is complete (or on mfsr_burst_flush / finish): while the last frame is still waiting, asking for its
 * flow / mask returns MFSR_E_INVALID (never the previous frame's buffers) -- flush first. */
int mfsr_burst_debug_views(mfsr_burst* b, mfsr_tex2d* flow, mfsr_tex2d* mask, mfsr_tex2d* kernelParam,
                           mfsr_tex2d* tracking);
/* products of the frame aligned `framesBack` frames before the last one (0 = the last: what mfsr_burst_debug_views gives).
 * Valid while framesBack < 2 * MFSR_MAX_FUSE_GROUP (the ring of per-frame slots) and the burst has aligned that many;
 * MFSR_E_INVALID for a frame that is still waiting for its group (see above). */
int mfsr_burst_debug_frame_views(mfsr_burst* b, int framesBack, mfsr_tex2d* flow, mfsr_tex2d* mask);
/* global pre-alignment of the last add_frame (cfg.preAlign), copied to HOST memory; aligns a frame that is still waiting
 * for its group first, then synchronises the stream */
int mfsr_burst_prealign_result(mfsr_burst* b, mfsr_prealign* hostOut, mfsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MFSR_H */
