/*
 * oracle/prealign.c -- CPU statement of the global pre-alignment (base shift + base rotation of a moved frame
 * against the reference frame) that feeds baseShift / baseRotation of convertToTilesOverlapPreShift
 * (kernel.cu:324-378), and CreateFlowFieldFromTiles (opticalFlow.cu:48-93).
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle_common.h).
 *
 * The reference has the slot but not the arithmetic: `class PreAlignment` is a field list
 * (boxFilterNPP.cpp:102-166) and the FFT log-polar registration of test_opencv/main.cpp:861-1194 never
 * returns a result (:840-851).  What the kernels fix is the MODEL the estimate must satisfy
 * (kernel.cu:358-368, opticalFlow.cu:78-85): a reference pixel p maps to the moved pixel
 *
 *      q = c + R(theta) * (p - c - base),        c = (width/2, height/2)
 *
 * The estimator (the build's own, identical here and in csrc/prealign.hip) is an exhaustive
 * coarse-to-fine search over (theta, base) on a 2x2-mean pyramid of the tracking images, scored
 * with INTEGER arithmetic so that the result does not depend on summation order:
 *
 *   - level j image = j times mfsr_downsample2x of the tracking image, quantised to 8 bit
 *     (q = (int)(v*255 + 0.5) clamped to [0,255]);
 *   - angles live on a grid of 1/16 degree; cos/sin come from a table the HOST builds with libm
 *     (cosf((float)a * K), K = pi/2880), so both implementations use the same bits;
 *   - score(a, t) = sum over the central window [w/4, w-w/4) x [h/4, h-h/4) of
 *     (bilinear_4.4(moved, q) - 256*ref(p))^2 with 4-bit fixed-point fractions, coordinates clamped
 *     to the image; exact in 64-bit integers;
 *   - coarsest level (long side <= 64): a in [-A, A] step 1 degree, t in [-8, 8]^2; every finer level
 *     (down to the first with long side <= 1024): a = a' + {-2..2} * step/2, t = 2 t' + {-2..2}^2;
 *     ties go to the lowest candidate index (a-major, then ty, then tx).
 */
#include "oracle_common.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PRE_MAX_LEVELS 16
#define PRE_T0 8
#define PRE_ANGLE_K 0.00109083078f /* pi / 2880: radians per 1/16 degree */

void orc_downsample2x(const float* in, int inPitch, float* out, int outPitch, int outW, int outH);

static inline int pre_quant(float v)
{
    int q = orc_f2i(v * 255.0f + 0.5f);
    return q < 0 ? 0 : (q > 255 ? 255 : q);
}

/* trig table for a in [-amax, amax]: tab[2*(a+amax)] = cos, [..+1] = sin */
void orc_preAlignTable(int amax, float* tab)
{
    for (int a = -amax; a <= amax; a++) {
        float th = (float)a * PRE_ANGLE_K;
        tab[2 * (a + amax)] = cosf(th);
        tab[2 * (a + amax) + 1] = sinf(th);
    }
}

static uint64_t pre_score(const uint8_t* ref, const uint8_t* mov, int w, int h, float cosv, float sinv, int tx, int ty)
{
    const int cx = w / 2, cy = h / 2;
    const int x0 = w / 4, x1 = w - w / 4, y0 = h / 4, y1 = h - h / 4;
    uint64_t sum = 0;
    for (int y = y0; y < y1; y++) {
        for (int x = x0; x < x1; x++) {
            float dx = (float)(x - cx - tx), dy = (float)(y - cy - ty);
            float qx = (cosv * dx - sinv * dy) + (float)cx;
            float qy = (sinv * dx + cosv * dy) + (float)cy;
            int fx = orc_f2i(floorf(qx * 16.0f + 0.5f));
            int fy = orc_f2i(floorf(qy * 16.0f + 0.5f));
            fx = orc_imin(orc_imax(fx, 0), (w - 1) * 16);
            fy = orc_imin(orc_imax(fy, 0), (h - 1) * 16);
            int ix = fx >> 4, ax = fx & 15, iy = fy >> 4, ay = fy & 15;
            int ix1 = orc_imin(ix + 1, w - 1), iy1 = orc_imin(iy + 1, h - 1);
            int m = (16 - ax) * (16 - ay) * mov[iy * w + ix] + ax * (16 - ay) * mov[iy * w + ix1] +
                    (16 - ax) * ay * mov[iy1 * w + ix] + ax * ay * mov[iy1 * w + ix1];
            int64_t d = (int64_t)m - 256 * (int64_t)ref[y * w + x];
            sum += (uint64_t)(d * d);
        }
    }
    return sum;
}

/* result[0..4] = shiftX, shiftY (tracking pixels), rotation (rad), cos, sin; state[0..2] = a (1/16 deg), tx, ty at
 * the finest search level, state[3] = that level's index.  Returns the number of levels searched (0 = image too small:
 * identity result). */
int orc_preAlign(const float* refImg, const float* movedImg, int width, int height, int pitch, float maxAngleDeg,
                 float* result, int* state)
{
    int A = orc_f2i(maxAngleDeg * 16.0f);
    if (A < 0) A = 0;
    A = (A / 16) * 16;
    const int amax = A + 32;
    float* tab = (float*)malloc(sizeof(float) * 2 * (2 * amax + 1));
    orc_preAlignTable(amax, tab);

    /* pyramid */
    int lw[PRE_MAX_LEVELS], lh[PRE_MAX_LEVELS];
    float* pr[PRE_MAX_LEVELS];
    float* pm[PRE_MAX_LEVELS];
    int n = 0;
    lw[0] = width;
    lh[0] = height;
    pr[0] = (float*)malloc(sizeof(float) * (size_t)width * height);
    pm[0] = (float*)malloc(sizeof(float) * (size_t)width * height);
    for (int y = 0; y < height; y++) {
        memcpy(pr[0] + (size_t)y * width, ORC_CROW(float, refImg, pitch, y), sizeof(float) * width);
        memcpy(pm[0] + (size_t)y * width, ORC_CROW(float, movedImg, pitch, y), sizeof(float) * width);
    }
    n = 1;
    while (orc_imax(lw[n - 1], lh[n - 1]) > 64 && n < PRE_MAX_LEVELS && (lw[n - 1] >> 1) >= 8 && (lh[n - 1] >> 1) >= 8) {
        lw[n] = lw[n - 1] >> 1;
        lh[n] = lh[n - 1] >> 1;
        pr[n] = (float*)malloc(sizeof(float) * (size_t)lw[n] * lh[n]);
        pm[n] = (float*)malloc(sizeof(float) * (size_t)lw[n] * lh[n]);
        /* odd widths: the input row pitch stays 4*lw[n-1]; the last column/row is dropped */
        orc_downsample2x(pr[n - 1], 4 * lw[n - 1], pr[n], 4 * lw[n], lw[n], lh[n]);
        orc_downsample2x(pm[n - 1], 4 * lw[n - 1], pm[n], 4 * lw[n], lw[n], lh[n]);
        n++;
    }
    const int jmax = n - 1;
    int jmin = 0;
    while (jmin < jmax && orc_imax(lw[jmin], lh[jmin]) > 1024) jmin++;

    int a = 0, tx = 0, ty = 0;
    for (int j = jmax; j >= jmin; j--) {
        const int w = lw[j], h = lh[j];
        uint8_t* qr = (uint8_t*)malloc((size_t)w * h);
        uint8_t* qm = (uint8_t*)malloc((size_t)w * h);
        for (size_t i = 0; i < (size_t)w * h; i++) {
            qr[i] = (uint8_t)pre_quant(pr[j][i]);
            qm[i] = (uint8_t)pre_quant(pm[j][i]);
        }
        int na, nt, a0, astep, tx0, ty0;
        if (j == jmax) {
            astep = 16;
            na = 2 * (A / 16) + 1;
            a0 = -A;
            nt = 2 * PRE_T0 + 1;
            tx0 = -PRE_T0;
            ty0 = -PRE_T0;
        } else {
            astep = 16 >> (jmax - j);
            if (astep < 1) astep = 1;
            na = 5;
            a0 = a - 2 * astep;
            nt = 5;
            tx0 = 2 * tx - 2;
            ty0 = 2 * ty - 2;
        }
        const int ncand = na * nt * nt;
        uint64_t* sc = (uint64_t*)malloc(sizeof(uint64_t) * ncand);
#pragma omp parallel for schedule(dynamic, 8)
        for (int c = 0; c < ncand; c++) {
            int ia = c / (nt * nt), r = c - ia * nt * nt, iy = r / nt, ix = r - iy * nt;
            int ca = orc_imin(orc_imax(a0 + ia * astep, -amax), amax);
            sc[c] = pre_score(qr, qm, w, h, tab[2 * (ca + amax)], tab[2 * (ca + amax) + 1], tx0 + ix, ty0 + iy);
        }
        int best = 0;
        for (int c = 1; c < ncand; c++)
            if (sc[c] < sc[best]) best = c;
        {
            int ia = best / (nt * nt), r = best - ia * nt * nt, iy = r / nt, ix = r - iy * nt;
            a = orc_imin(orc_imax(a0 + ia * astep, -amax), amax);
            tx = tx0 + ix;
            ty = ty0 + iy;
        }
        free(sc);
        free(qr);
        free(qm);
    }
    result[0] = (float)(tx * (1 << jmin));
    result[1] = (float)(ty * (1 << jmin));
    result[2] = (float)a * PRE_ANGLE_K;
    result[3] = tab[2 * (a + amax)];
    result[4] = tab[2 * (a + amax) + 1];
    if (state) {
        state[0] = a;
        state[1] = tx;
        state[2] = ty;
        state[3] = jmin;
    }
    for (int j = 0; j < n; j++) {
        free(pr[j]);
        free(pm[j]);
    }
    free(tab);
    return jmax - jmin + 1;
}
