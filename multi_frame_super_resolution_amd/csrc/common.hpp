// common.hpp -- shared device/host helpers of the MI355X (gfx950) MFSR kernels.
//
// Canonical arithmetic: compiled with -ffp-contract=off so that the
// straight ports of +,-,*,/ kernels are bit-identical to the no-contraction
// CPU oracle; fused multiply-adds appear only where written explicitly.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../../include/mfsr.h"

#define MFSR_WAVE 64

// ---- error handling: int codes, never throw (kernel.cu:36-114 convention) ----
#define MFSR_HIP_TRY(expr)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            fprintf(stderr, "mfsr: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, \
                    __LINE__);                                                                      \
            return (int)e_;                                                                         \
        }                                                                                           \
    } while (0)

#define MFSR_REQUIRE(cond)                                                                  \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            fprintf(stderr, "mfsr: invalid argument: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
            return MFSR_E_INVALID;                                                          \
        }                                                                                   \
    } while (0)

static inline int mfsr_launch_status(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fprintf(stderr, "mfsr: launch of %s failed: %s\n", what, hipGetErrorString(e));
        return (int)e;
    }
    return MFSR_OK;
}

static inline hipStream_t mfsr_s(mfsr_stream_t s) { return (hipStream_t)s; }
static inline unsigned mfsr_cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

// ---- CFA pattern: packed 4 x 8 bit, [y%2][x%2] -> bits ((y&1)*2+(x&1))*8 -----
// Frame batches: the per-frame pointer arguments of a kernel for up to MFSR_BATCH_MAX frames, appended to its argument list; a
// launch with gridDim.z > 1 takes frame blockIdx.z's pointers from the table (a scalar load), a plain launch ignores it.
#define MFSR_BATCH_MAX 4
struct MfsrBatch {
    const void* p[MFSR_BATCH_MAX][6];
};
int mfsr_cfa_packed();  // defined in debayer.hip (process-wide state)
// x / d for a divisor that is the same for a whole launch (an image dimension): q = x r, q' = fma(fma(-d, q, x), r, q) with
// r = RN(1 / d) -- two multiply-adds after the product instead of the ~10 instructions of the IEEE division expansion.  For a
// given d the sequence either is the correctly rounded quotient for EVERY x or it is not (scaling x by a power of two
// commutes with every rounding involved), so the host checks it once per divisor over all 2^23 significands of one binade
// (mfsr_exact_div in glue.hip: ~15 ms, cached) and the kernels take the sequence only when that check passed (ok), else the
// division: bit-identical to `x / d` by construction.  (Every image dimension tried passes; the check is the proof, not the
// assumption.)  Infinite and NaN quotients (an infinite flow) are passed through as the division gives them.
// MFSR_EXACT_DIV=0: always the division (A/B).
struct MfsrExactDiv {
    float d, r;
    int ok;
};
MfsrExactDiv mfsr_exact_div(float d);  // host; glue.hip
__device__ __forceinline__ float mfsr_div(float x, const MfsrExactDiv& e)
{
    if (e.ok) {  // uniform
        const float q = x * e.r;
        const float q2 = __builtin_fmaf(__builtin_fmaf(-e.d, q, x), e.r, q);
        return fabsf(q) < __builtin_inff() ? q2 : q;
    }
    return x / e.d;
}

__device__ __forceinline__ int cfa_at(int packed, int y, int x) { return (packed >> ((((y & 1) << 1) | (x & 1)) << 3)) & 0xff; }

// ---- row addressing with byte pitches ----------------------------------------
template <typename T>
__device__ __forceinline__ T* row_ptr(T* base, int pitch, int y)
{
    return (T*)((char*)base + (size_t)pitch * (size_t)y);
}
template <typename T>
__device__ __forceinline__ const T* row_ptr(const T* base, int pitch, int y)
{
    return (const T*)((const char*)base + (size_t)pitch * (size_t)y);
}

// packed 12-byte pixel (the reference's float3); float3 in HIP is also 12 B
struct __attribute__((packed, aligned(4))) pix3 {
    float x, y, z;
};

// float -> int: v_cvt_i32_f32 (NaN -> 0, saturating), the CPU oracle converts the same way
__device__ __forceinline__ int f2i(float f) { return __float2int_rz(f); }

// f2i(roundf(f)) in three instructions: adding the largest float below 0.5 (with f's sign) and
// truncating is roundf for EVERY float (half away from zero; checked exhaustively over all 2^32
// bit patterns on the host), and v_cvt_i32_f32 is the truncation.
__device__ __forceinline__ int round2i(float f) { return __float2int_rz(f + copysignf(0.49999997f, f)); }

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

__device__ __forceinline__ bool finitef(float v) { return fabsf(v) < __builtin_inff(); }

// ---- texture stand-in: exact-float bilinear, normalised coordinates ----------
// unnormalised coord = u*W - 0.5, texel indices clamped; MIRROR reflects the
// normalised coordinate first (CUDA cudaAddressModeMirror).
enum { ADDR_CLAMP = 0, ADDR_MIRROR = 1 };

__device__ __forceinline__ float mirror_coord(float x)
{
    float f = floorf(x);
    float fr = x - f;
    return (f2i(f) & 1) ? 1.0f - fr : fr;
}

struct TexCoord {
    int i0, i1, j0, j1;
    float a, b;
};

template <int MODE>
__device__ __forceinline__ TexCoord tex_coord(int w, int h, float u, float v)
{
    if (MODE == ADDR_MIRROR) {
        u = mirror_coord(u);
        v = mirror_coord(v);
    }
    float xB = u * (float)w - 0.5f;
    float yB = v * (float)h - 0.5f;
    if (!finitef(xB)) xB = 0.0f;
    if (!finitef(yB)) yB = 0.0f;
    float fx = floorf(xB), fy = floorf(yB);
    TexCoord c;
    c.a = xB - fx;
    c.b = yB - fy;
    int ix = f2i(fx), iy = f2i(fy);
    c.i0 = clampi(ix, 0, w - 1);
    c.i1 = (ix >= 2147483647) ? w - 1 : clampi(ix + 1, 0, w - 1);
    c.j0 = clampi(iy, 0, h - 1);
    c.j1 = (iy >= 2147483647) ? h - 1 : clampi(iy + 1, 0, h - 1);
    return c;
}

__device__ __forceinline__ float lerp4(float t00, float t10, float t01, float t11, float a, float b)
{
    return (((1.0f - a) * (1.0f - b) * t00 + a * (1.0f - b) * t10) + (1.0f - a) * b * t01) + a * b * t11;
}

template <int MODE>
__device__ __forceinline__ float tex1(const mfsr_tex2d& t, float u, float v)
{
    TexCoord c = tex_coord<MODE>(t.width, t.height, u, v);
    const float* r0 = row_ptr((const float*)t.ptr, t.pitch, c.j0);
    const float* r1 = row_ptr((const float*)t.ptr, t.pitch, c.j1);
    return lerp4(r0[c.i0], r0[c.i1], r1[c.i0], r1[c.i1], c.a, c.b);
}

template <int MODE>
__device__ __forceinline__ float2 tex2(const mfsr_tex2d& t, float u, float v)
{
    TexCoord c = tex_coord<MODE>(t.width, t.height, u, v);
    const float2* r0 = row_ptr((const float2*)t.ptr, t.pitch, c.j0);
    const float2* r1 = row_ptr((const float2*)t.ptr, t.pitch, c.j1);
    float2 t00 = r0[c.i0], t10 = r0[c.i1], t01 = r1[c.i0], t11 = r1[c.i1];
    return make_float2(lerp4(t00.x, t10.x, t01.x, t11.x, c.a, c.b), lerp4(t00.y, t10.y, t01.y, t11.y, c.a, c.b));
}

template <int MODE>
__device__ __forceinline__ float4 tex4(const mfsr_tex2d& t, float u, float v)
{
    TexCoord c = tex_coord<MODE>(t.width, t.height, u, v);
    const float4* r0 = row_ptr((const float4*)t.ptr, t.pitch, c.j0);
    const float4* r1 = row_ptr((const float4*)t.ptr, t.pitch, c.j1);
    float4 t00 = r0[c.i0], t10 = r0[c.i1], t01 = r1[c.i0], t11 = r1[c.i1];
    return make_float4(lerp4(t00.x, t10.x, t01.x, t11.x, c.a, c.b), lerp4(t00.y, t10.y, t01.y, t11.y, c.a, c.b),
                       lerp4(t00.z, t10.z, t01.z, t11.z, c.a, c.b), lerp4(t00.w, t10.w, t01.w, t11.w, c.a, c.b));
}

// ---- argument checks shared by the entry points -------------------------------
static inline bool mfsr_tex_ok(const mfsr_tex2d& t, int texel_bytes)
{
    return t.ptr != nullptr && t.width > 0 && t.height > 0 && (long long)t.pitch >= (long long)t.width * texel_bytes;
}
