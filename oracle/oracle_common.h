/*
 * oracle/oracle_common.h -- shared types for the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ may be imported, linked or
 * executed by the product path (multi_frame_super_resolution_amd/, apps/).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker.
 *
 * PARITY UNPINNED: the reference (zhongzisha/multi_frame_super_resolution)
 * ships no tests, golden vectors or expected outputs for this path, and its
 * .cu kernels cannot be built in this image without writing stand-ins for the
 * CUDA headers/toolchain (not allowed).  This oracle is therefore a
 * line-by-line restatement of the reference arithmetic (each function cites
 * the reference file:line it follows), pinned only by hand-derived
 * known-answer tests (tests/test_oracle_kat.py).
 *
 * Numerical conventions (see DESIGN.md "Canonical semantics"):
 *   - compiled with -ffp-contract=off: every * and + rounds separately, as the
 *     reference's -G (debug, no-contraction) builds do;
 *   - `exp/cos/sin/atan2/sqrt/fabs` on float operands are the float overloads
 *     (CUDA semantics), i.e. expf/cosf/...;
 *   - texture fetches: exact-float bilinear, unnormalised coord = u*W-0.5,
 *     index clamp for CLAMP, coordinate reflection + index clamp for MIRROR.
 */
#ifndef MFSR_ORACLE_COMMON_H
#define MFSR_ORACLE_COMMON_H

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y; } of2;
typedef struct { float x, y, z; } of3;
typedef struct { float x, y, z, w; } of4;

/* enum BayerColor, DeBayerKernels.cu:28-37 */
enum { ORC_RED = 0, ORC_GREEN = 1, ORC_BLUE = 2 };

/* address modes of the texture stand-in */
enum { ORC_ADDR_CLAMP = 0, ORC_ADDR_MIRROR = 1 };

/* c_cfaPattern[2][2], DeBayerKernels.cu:40-41 (module-global constant) */
extern int orc_cfa[2][2];

static inline int orc_imin(int a, int b) { return a < b ? a : b; }
static inline int orc_imax(int a, int b) { return a > b ? a : b; }

/* row pointer helpers: all pitches are in BYTES, as in the reference */
#define ORC_ROW(type, base, pitch, y) ((type*)((char*)(base) + (size_t)(pitch) * (size_t)(y)))
#define ORC_CROW(type, base, pitch, y) ((const type*)((const char*)(base) + (size_t)(pitch) * (size_t)(y)))

/* float -> int as the GPU does it (NaN -> 0, saturating) so UB never occurs */
static inline int orc_f2i(float f)
{
    if (!(f == f)) return 0;
    if (f >= 2147483520.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

/* CUDA mirror addressing on a normalised coordinate: frac(x) on even periods,
 * 1-frac(x) on odd ones. */
static inline float orc_mirror(float x)
{
    float f = floorf(x);
    float fr = x - f;
    int odd = orc_f2i(f) & 1;
    return odd ? 1.0f - fr : fr;
}

typedef struct {
    const void* ptr;
    int pitch;  /* bytes */
    int w, h;   /* texels */
    int mode;   /* ORC_ADDR_* */
} orc_tex;

static inline void orc_tex_coords(const orc_tex* t, float u, float v, int* i0, int* i1, int* j0, int* j1, float* a,
                                  float* b)
{
    if (t->mode == ORC_ADDR_MIRROR) {
        u = orc_mirror(u);
        v = orc_mirror(v);
    }
    float xB = u * (float)t->w - 0.5f;
    float yB = v * (float)t->h - 0.5f;
    if (!isfinite(xB)) xB = 0.0f;
    if (!isfinite(yB)) yB = 0.0f;
    float fx = floorf(xB), fy = floorf(yB);
    *a = xB - fx;
    *b = yB - fy;
    int ix = orc_f2i(fx), iy = orc_f2i(fy);
    *i0 = orc_imin(orc_imax(ix, 0), t->w - 1);
    *i1 = orc_imin(orc_imax(ix + 1, 0), t->w - 1);
    if (ix >= 2147483647) *i1 = t->w - 1;
    *j0 = orc_imin(orc_imax(iy, 0), t->h - 1);
    *j1 = orc_imin(orc_imax(iy + 1, 0), t->h - 1);
    if (iy >= 2147483647) *j1 = t->h - 1;
}

#define ORC_LERP4(t00, t10, t01, t11, a, b) \
    ((((1.0f - (a)) * (1.0f - (b)) * (t00) + (a) * (1.0f - (b)) * (t10)) + (1.0f - (a)) * (b) * (t01)) + (a) * (b) * (t11))

/* tex2D<float> */
static inline float orc_tex1(const orc_tex* t, float u, float v)
{
    int i0, i1, j0, j1;
    float a, b;
    orc_tex_coords(t, u, v, &i0, &i1, &j0, &j1, &a, &b);
    const float* r0 = ORC_CROW(float, t->ptr, t->pitch, j0);
    const float* r1 = ORC_CROW(float, t->ptr, t->pitch, j1);
    return ORC_LERP4(r0[i0], r0[i1], r1[i0], r1[i1], a, b);
}

/* tex2D<float2> */
static inline of2 orc_tex2(const orc_tex* t, float u, float v)
{
    int i0, i1, j0, j1;
    float a, b;
    orc_tex_coords(t, u, v, &i0, &i1, &j0, &j1, &a, &b);
    const of2* r0 = ORC_CROW(of2, t->ptr, t->pitch, j0);
    const of2* r1 = ORC_CROW(of2, t->ptr, t->pitch, j1);
    of2 o;
    o.x = ORC_LERP4(r0[i0].x, r0[i1].x, r1[i0].x, r1[i1].x, a, b);
    o.y = ORC_LERP4(r0[i0].y, r0[i1].y, r1[i0].y, r1[i1].y, a, b);
    return o;
}

/* tex2D<float4> */
static inline of4 orc_tex4(const orc_tex* t, float u, float v)
{
    int i0, i1, j0, j1;
    float a, b;
    orc_tex_coords(t, u, v, &i0, &i1, &j0, &j1, &a, &b);
    const of4* r0 = ORC_CROW(of4, t->ptr, t->pitch, j0);
    const of4* r1 = ORC_CROW(of4, t->ptr, t->pitch, j1);
    of4 o;
    o.x = ORC_LERP4(r0[i0].x, r0[i1].x, r1[i0].x, r1[i1].x, a, b);
    o.y = ORC_LERP4(r0[i0].y, r0[i1].y, r1[i0].y, r1[i1].y, a, b);
    o.z = ORC_LERP4(r0[i0].z, r0[i1].z, r1[i0].z, r1[i1].z, a, b);
    o.w = ORC_LERP4(r0[i0].w, r0[i1].w, r1[i0].w, r1[i1].w, a, b);
    return o;
}

#ifdef __cplusplus
}
#endif
#endif
