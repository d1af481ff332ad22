// accumulate_fast.hip -- the full-frame warp+fuse kernels of the headline benchmark (x2 and x4),
// restructured for CDNA4.  Same mathematics as accumulateImagesSuperRes (reference
// test_opencv/DeBayerKernels.cu:379-468, full-frame generalisation), accumulators in HBM as in the
// reference (48 B per HR pixel and frame in its one-launch-per-frame structure; these kernels also take
// up to four frames per launch and move the accumulators once for all of them).
//
// Why not the straight port (accumulate.hip): it spends its time in VALU/SALU work, not in HBM -- per tap
// an IEEE division, a 3-way colour branch and a 16-byte certainty load (1.67 ms per 4K frame = 13 % of
// HBM peak).  These kernels remove that work without changing what is computed:
//
//  * 4-pixel strips: one thread owns HR pixels X0..X0+3 of one row; the certainty cell a tap reads is a
//    compile-time function of the pixel's position in the strip (template parameter K).
//  * raw sites instead of taps: the 5x5 HR taps of a pixel fall on 3x3 (x2) or 2x2 (x4) raw pixels.
//    Which site a tap lands on depends on the low bits of (X + round(s*u)); that choice is applied to the
//    WEIGHTS with lane masks (v_and / v_bitop3, no v_cndmask), the raw values enter with a few fma.
//  * no per-tap division: sum raw*w*c and w*c per CFA position, normalise once per pixel:
//    sum((raw-b)/wl * w*c) = (sum(raw*w*c) - b*sum(w*c)) / wl.
//  * no colour branches: sites are summed per CFA-position class (row/column parity relative to the first
//    site); classes map to R/G/B once per pixel (the CFA pattern is a template parameter).
//  * 12 exponentials per pixel instead of 25: w(px,py) = w(-px,-py), v_exp_f32, exponents from sums -- and, with
//    several frames per launch, once per pixel for all of them (pixel-major order: the weights depend on the kernel
//    parameters of the reference only).
//  * LDS tile kernels (fields at the tracking resolution): field / certainty texels, interpolation
//    fractions and the accumulator rows (LDS-DMA) staged per workgroup; see k_accumulate2xTile.
//
// Numerics: the flow rounding is the same expression as in the straight kernel and the oracle; weights
// and per-channel sums are re-associated (pre-scaled exponents, class sums, fma, one normalisation), so
// results agree with the straight kernel to ~1e-6 relative (tests: 3e-5), not bit for bit.  Pixels whose
// taps would be clamped at the frame border, strips with wild flow and strips whose kernel parameters
// are not positive semi-definite take the straight per-pixel arithmetic (margin kernel / in-kernel path).
#include <cstdlib>

#include "accumulate_common.hpp"
#include <type_traits>

namespace {

// CFA packed 2 bits per CFA position a = (yparity << 1) | xparity
template <int CFA>
struct Cfa {
    static constexpr int col(int yp, int xp) { return (CFA >> (2 * ((yp << 1) | xp))) & 3; }
    // position of the (single) red / blue site; -1 if the pattern has none
    static constexpr int pos_of(int c)
    {
        int n = 0, p = -1;
        for (int a = 0; a < 4; a++)
            if (((CFA >> (2 * a)) & 3) == c) {
                n++;
                p = a;
            }
        return n == 1 ? p : -1;
    }
    static constexpr int count(int c)
    {
        int n = 0;
        for (int a = 0; a < 4; a++)
            if (((CFA >> (2 * a)) & 3) == c) n++;
        return n;
    }
};

struct StripLevels {
    float black[3];
    float invWhite[3];
};

__device__ __forceinline__ float sane(float c) { return finitef(c) ? c : 0.0f; }

// fast-path admission: kernel parameters form a positive semi-definite inverse covariance
// (so every tap exponent is >= 0 and every weight is in [0,1]); NaN fails the comparisons
__device__ __forceinline__ bool psd_ok(float kx, float ky, float kz)
{
    return kx >= 0.0f && ky >= 0.0f && kz * kz <= kx * ky && kx < 1e30f && ky < 1e30f;
}

// one HR pixel of a safe strip.  K = position in the strip (compile time).
// MaskF: float mval(int jt, int cell, int ch) -> sanitised certainty of channel ch in mask cell
// `cell` (0..2, relative to the strip) on the mask row that tap row jt reads.
// Per-lane two-way selects are written as bit-field inserts on lane masks (v_bitop3_b32, a plain
// 3-operand VALU op) instead of v_cndmask_b32 + v_cmp: on gfx950 v_cndmask/v_cmp issue at ~0.69x the
// rate of v_add/v_fma/v_bitop3 (tools/ubench/valu_ops.hip: 1.95 vs 1.3 ns per wave-instruction per
// SIMD) and selects are a third of this kernel's instructions.
__device__ __forceinline__ float selm(uint32_t m, float a, float b)  // m == ~0u ? a : b
{
    // one v_bitop3_b32 with truth table 0xCA = (m & a) | (~m & b)
    return __uint_as_float(__builtin_amdgcn_bitop3_b32(m, __float_as_uint(a), __float_as_uint(b), 0xCA));
}

// certainty of the colour under an even (Ee) / odd (Eo) site column of a tap row, from the three channel
// certainties m[] of a mask cell.  mya: lane mask "the site row has absolute y parity 1"; mP: "site
// column 0 has absolute x parity 1".
template <int CFA>
__device__ __forceinline__ void resolve_certainty(uint32_t mya, uint32_t mP, const float (&m)[3], float& Ee, float& Eo)
{
    constexpr bool mono = Cfa<CFA>::count(MFSR_GREEN) == 4;
    if (mono) {
        Ee = Eo = m[MFSR_GREEN];
    } else if (Cfa<CFA>::col(0, 1) == Cfa<CFA>::col(1, 0)) {
        // Bayer, G on the anti-diagonal (RGGB, BGGR): the colour on the diagonal is picked by the y
        // parity alone; even site columns see it when x parity == y parity, else they see G
        const float X = selm(mya, m[Cfa<CFA>::col(1, 1)], m[Cfa<CFA>::col(0, 0)]);
        const uint32_t d = mya ^ mP;  // x parity != y parity
        Ee = selm(d, m[Cfa<CFA>::col(0, 1)], X);
        Eo = selm(d, X, m[Cfa<CFA>::col(0, 1)]);
    } else if (Cfa<CFA>::col(0, 0) == Cfa<CFA>::col(1, 1)) {
        // Bayer, G on the diagonal (GRBG, GBRG)
        const float X = selm(mya, m[Cfa<CFA>::col(1, 0)], m[Cfa<CFA>::col(0, 1)]);
        const uint32_t d = mya ^ mP;
        Ee = selm(d, X, m[Cfa<CFA>::col(0, 0)]);
        Eo = selm(d, m[Cfa<CFA>::col(0, 0)], X);
    } else {
        const float cA = selm(mya, m[Cfa<CFA>::col(1, 0)], m[Cfa<CFA>::col(0, 0)]);  // absolute x parity 0
        const float cB = selm(mya, m[Cfa<CFA>::col(1, 1)], m[Cfa<CFA>::col(0, 1)]);  // absolute x parity 1
        Ee = selm(mP, cB, cA);
        Eo = selm(mP, cA, cB);
    }
}

// relative CFA-position class sums (class (yc, xc) sits at CFA position (yc ^ Q, xc ^ P)) -> R, G, B,
// normalised and added to pixel K of the strip
template <int K, int CFA>
__device__ __forceinline__ void classes_to_channels(const float (&S)[2][2], const float (&W)[2][2], uint32_t mP, uint32_t mQ,
                                                    const StripLevels& lv, float* accP, float* accW)
{
    constexpr bool mono = Cfa<CFA>::count(MFSR_GREEN) == 4;
    auto at_pos = [&](const float(&T)[2][2], int yp, int xp) {
        const float q0 = selm(mP, T[yp][xp ^ 1], T[yp][xp]);          // Q == 0
        const float q1 = selm(mP, T[yp ^ 1][xp ^ 1], T[yp ^ 1][xp]);  // Q == 1
        return selm(mQ, q1, q0);
    };
    float chS[3] = {0, 0, 0}, chW[3] = {0, 0, 0};
    const float totS = (S[0][0] + S[0][1]) + (S[1][0] + S[1][1]);
    const float totW = (W[0][0] + W[0][1]) + (W[1][0] + W[1][1]);
    if (mono) {
        chS[1] = totS;
        chW[1] = totW;
    } else {
        constexpr int pr = Cfa<CFA>::pos_of(MFSR_RED), pb = Cfa<CFA>::pos_of(MFSR_BLUE);
        if ((pr ^ pb) == 3) {
            // red and blue on opposite corners of the 2x2 cell (every Bayer pattern): they are the two
            // classes of one diagonal; which diagonal follows from P ^ Q, which end from Q
            const uint32_t dq = ((pr >> 1) ^ (pr & 1)) ? ~(mP ^ mQ) : (mP ^ mQ);  // all ones: the anti-diagonal classes
            const uint32_t mR = (pr >> 1) ? ~mQ : mQ;                            // all ones: red is the class in row 1
            const float s0 = selm(dq, S[0][1], S[0][0]), s1 = selm(dq, S[1][0], S[1][1]);
            const float w0 = selm(dq, W[0][1], W[0][0]), w1 = selm(dq, W[1][0], W[1][1]);
            chS[0] = selm(mR, s1, s0);
            chS[2] = selm(mR, s0, s1);
            chW[0] = selm(mR, w1, w0);
            chW[2] = selm(mR, w0, w1);
        } else {
            chS[0] = at_pos(S, pr >> 1, pr & 1);
            chW[0] = at_pos(W, pr >> 1, pr & 1);
            chS[2] = at_pos(S, pb >> 1, pb & 1);
            chW[2] = at_pos(W, pb >> 1, pb & 1);
        }
        // weights are in [0,1] on this path (PSD kernel parameters), so the two green positions can be
        // taken as total - red - blue without cancellation trouble
        chS[1] = (totS - chS[0]) - chS[2];
        chW[1] = (totW - chW[0]) - chW[2];
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        // (chS - black * chW) / white, added to the strip's sum: two fma
        accP[3 * K + c] = __builtin_fmaf(__builtin_fmaf(-lv.black[c], chW[c], chS[c]), lv.invWhite[c], accP[3 * K + c]);
        accW[3 * K + c] += chW[c];
    }
}

// the 13 unique tap weights of a pixel: n = jt*5+it, w[n] == w[24-n]; exponents pre-scaled by
// -0.5*log2(e) and built from sums (2 adds per weight).  (Building ten of the twelve from products of
// four exponentials -- six v_exp_f32 instead of twelve -- was measured: no gain, 0.642 vs 0.636 ms per
// pair, and it needs a bound on |kz|; the twelve exponentials stay.)
__device__ __forceinline__ void tap_weights13(float kx, float ky, float kz, float (&w)[13])
{
    const float a1 = kx * -0.72134752044448170368f, b1 = ky * -0.72134752044448170368f;
    const float c1 = kz * -0.72134752044448170368f;
    const float A[3] = {0.0f, a1, 4.0f * a1}, B[3] = {0.0f, b1, 4.0f * b1};
#pragma unroll
    for (int n = 0; n < 12; n++) {
        const int py = n / 5 - 2, px = n % 5 - 2;
        const int apx = px < 0 ? -px : px, apy = py < 0 ? -py : py;
        const float d = apx == 0 ? B[apy] : (apy == 0 ? A[apx] : A[apx] + B[apy]);
        // the caller admits only positive semi-definite kernel parameters, so every exponent is <= 0 and
        // every weight in [0, 1]: the non-finite rule of :429-430 cannot fire
        w[n] = __builtin_amdgcn_exp2f(px * py == 0 ? d : d + (float)(2 * px * py) * c1);
    }
    w[12] = 1.0f;
}

// tail of a strip pixel: C[jt][i] = sum over the taps of tap row jt on site column i of weight x certainty -> the nine
// site weights (tap rows 1 and 3 change site row with the y parity bit), class sums, channels
template <int K, int CFA>
__device__ __forceinline__ void strip_pixel_sites(const float (&C)[5][3], const float (&s)[3][3], uint32_t mby, uint32_t nby, uint32_t mP,
                                                   uint32_t mQ, const StripLevels& lv, float* accP, float* accW)
{
    auto andm = [](uint32_t m, float a) { return __uint_as_float(m & __float_as_uint(a)); };
    // site rows: tap row 0, row 1 if by == 0 | row 1 if by == 1, row 2, row 3 if by == 0 | row 3 if by == 1, row 4
    float Om[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        Om[0][i] = C[0][i] + andm(nby, C[1][i]);
        // (C1 & by) + C2 + (C3 & ~by) with one of the two masked terms zero: C2 + (by ? C1 : C3), the same sum (the terms
        // are non-negative, x + 0 is exact) in two instructions instead of four
        Om[1][i] = C[2][i] + selm(mby, C[1][i], C[3][i]);
        Om[2][i] = andm(mby, C[3][i]) + C[4][i];
    }
    // class (yc, xc) = (j & 1, i & 1), relative to (Q, P)
    float S[2][2], W[2][2];
    S[0][0] = __builtin_fmaf(s[2][2], Om[2][2], __builtin_fmaf(s[2][0], Om[2][0], __builtin_fmaf(s[0][2], Om[0][2], s[0][0] * Om[0][0])));
    W[0][0] = (Om[0][0] + Om[0][2]) + (Om[2][0] + Om[2][2]);
    S[0][1] = __builtin_fmaf(s[2][1], Om[2][1], s[0][1] * Om[0][1]);
    W[0][1] = Om[0][1] + Om[2][1];
    S[1][0] = __builtin_fmaf(s[1][2], Om[1][2], s[1][0] * Om[1][0]);
    W[1][0] = Om[1][0] + Om[1][2];
    S[1][1] = s[1][1] * Om[1][1];
    W[1][1] = Om[1][1];

    classes_to_channels<K, CFA>(S, W, mP, mQ, lv, accP, accW);
}

// Second formulation of the same pixel: the dynamic part (which raw site a tap lands on) is
// applied to the WEIGHTS, not to the values.
//  * site sums: out = sum_sites raw[j][i] * Om[j][i], Om[j][i] = sum of w*certainty over the taps on
//    site (j,i).  The CFA-position class of a site is static ((j&1, i&1) relative to (Q,P)), so the
//    value side is 9 fma + 5 add with no selects at all;
//  * tap columns 1 and 3 change site with the x parity bit: their weights are split once per
//    pixel into (w & ~mbx, w & mbx) (5 unique weights, plain v_and), tap rows 1 and 3 are split
//    the same way on the column sums;
//  * per tap row the certainty is resolved to (even site column, odd site column) x (cell) = at most
//    four values instead of one select per tap;
//  * exponents from pre-scaled sums: 2 adds per weight instead of mul+2 adds+mul.
// ~245 VALU instructions per pixel instead of ~330; sums re-associated again (fma), same tolerance.
// PARITY = true (the LDS tile kernel): mval(jt, cell, 4 * e) returns the certainty of the colour at CFA position e =
// (y parity << 1) | x parity of mask cell `cell` on tap row jt's mask row -- the certainty texels are staged in LDS in
// CFA-position order, so "which channel does this site see" is an LDS ADDRESS (a few integer ops per pixel) instead of
// three v_bitop3 selects per tap row and cell.
// strip_pixel_w: the pixel with its 13 tap weights given (they depend on the kernel parameters only, i.e. not on the
// frame: a kernel that takes several frames per launch computes them once per pixel).
// RawF: void rawf(int x0, int y0, float (&s)[3][3]) -> the 3x3 raw sites whose top left one is (x0, y0).
template <int K, int CFA, bool PARITY = false, typename RawF, typename MaskF>
__device__ __forceinline__ void strip_pixel_w(int X, int Y, int sx, int sy, const float (&w)[13], RawF rawf, MaskF mval,
                                               const StripLevels& lv, float* accP, float* accW)
{
    const int qx = X + sx - 2, qy = Y + sy - 2;
    const int x0 = qx >> 1, y0 = qy >> 1;
    uint32_t mbx = 0u - (uint32_t)(qx & 1), mby = 0u - (uint32_t)(qy & 1);
    uint32_t nbx = ~mbx, nby = ~mby;
    // opaque to the compiler: it would turn `weight & mask` into v_cndmask_b32 on the parity bit, which issues at 0.63x
    // the rate of v_and_b32 on gfx950 (profiles/r02_ubench_valu_ops.txt)
    asm volatile("" : "+v"(mbx), "+v"(nbx), "+v"(mby), "+v"(nby));
    const uint32_t mP = 0u - (uint32_t)(x0 & 1), mQ = 0u - (uint32_t)(y0 & 1);
    auto andm = [](uint32_t m, float a) { return __uint_as_float(m & __float_as_uint(a)); };

    float s[3][3];
    rawf(x0, y0, s);

    // tap columns 1 and 3: part that joins the lower / the upper site column
    float wl[13], wh[13];
#pragma unroll
    for (int n = 0; n < 13; n++) {
        const int it = n % 5;
        if (it == 1 || it == 3) {
            wl[n] = andm(nbx, w[n]);
            wh[n] = andm(mbx, w[n]);
        }
    }
    // n = 13 (row 2, column 3) mirrors n = 11
    auto W_ = [&](int jt, int it) { const int n = jt * 5 + it; return w[n <= 12 ? n : 24 - n]; };
    auto WL = [&](int jt, int it) { const int n = jt * 5 + it; return wl[n <= 12 ? n : 24 - n]; };
    auto WH = [&](int jt, int it) { const int n = jt * 5 + it; return wh[n <= 12 ? n : 24 - n]; };

    const uint32_t mY13 = mQ ^ mby;  // y parity of tap row 1's site row (complement for row 3)
    constexpr int cellLo = (K + 0 + 2) >> 2;
    auto cidx = [](int it) { return (((K + it + 2) >> 2) == ((K + 2) >> 2)) ? 0 : 1; };

    float C[5][3];  // per tap row: w * certainty summed per site column
#pragma unroll
    for (int jt = 0; jt < 5; jt++) {
        uint32_t mya;
        if (jt == 0 || jt == 4) mya = mQ;
        if (jt == 2) mya = ~mQ;
        if (jt == 1) mya = mY13;
        if (jt == 3) mya = ~mY13;
        float Ee[2], Eo[2];  // certainty of the colour on even / odd site columns (relative to x0), per cell
        if constexpr (PARITY) {
            // CFA position of an even site column on this tap row: y parity Q (rows 0, 4), ~Q (row 2), Q ^ by (row 1),
            // ~(Q ^ by) (row 3); x parity P.  Odd site columns: x parity flipped.
            // (as byte offsets into the texel: the address is then a plain add)
            const int eQ = ((qy << 2) & 8) | ((qx + qx) & 4);   // ((y0 & 1) << 1 | (x0 & 1)) * 4 with x0 = qx >> 1, y0 = qy >> 1
            const int by2 = (qy << 3) & 8;
            const int e = (jt == 0 || jt == 4) ? eQ : (jt == 2 ? (eQ ^ 8) : (jt == 1 ? (eQ ^ by2) : (eQ ^ 8 ^ by2)));
#pragma unroll
            for (int c = 0; c < 2; c++) {
                Ee[c] = mval(jt, cellLo + c, e);
                Eo[c] = mval(jt, cellLo + c, e ^ 4);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                float m[3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++) m[ch] = mval(jt, cellLo + c, ch);
                resolve_certainty<CFA>(mya, mP, m, Ee[c], Eo[c]);
            }
        }
        // site column 0: tap 0, tap 1 if bx == 0
        C[jt][0] = cidx(0) == cidx(1) ? (W_(jt, 0) + WL(jt, 1)) * Ee[cidx(0)]
                                      : __builtin_fmaf(WL(jt, 1), Ee[cidx(1)], W_(jt, 0) * Ee[cidx(0)]);
        // site column 1: tap 1 if bx == 1, tap 2, tap 3 if bx == 0
        {
            float acc;
            if (cidx(1) == cidx(2))
                acc = (WH(jt, 1) + W_(jt, 2)) * Eo[cidx(1)];
            else
                acc = __builtin_fmaf(W_(jt, 2), Eo[cidx(2)], WH(jt, 1) * Eo[cidx(1)]);
            if (cidx(3) == cidx(1) && cidx(1) == cidx(2))
                acc = ((WH(jt, 1) + W_(jt, 2)) + WL(jt, 3)) * Eo[cidx(1)];
            else
                acc = __builtin_fmaf(WL(jt, 3), Eo[cidx(3)], acc);
            C[jt][1] = acc;
        }
        // site column 2: tap 3 if bx == 1, tap 4
        C[jt][2] = cidx(3) == cidx(4) ? (WH(jt, 3) + W_(jt, 4)) * Ee[cidx(3)]
                                      : __builtin_fmaf(W_(jt, 4), Ee[cidx(4)], WH(jt, 3) * Ee[cidx(3)]);
    }
    strip_pixel_sites<K, CFA>(C, s, mby, nby, mP, mQ, lv, accP, accW);
}

// The same pixel where every certainty texel its wave touches is exactly 1 in every channel (the robustness mask
// saturates at 1 over well-aligned content: most of a typical frame).  strip_pixel_w with Ee = Eo = 1 folded: the
// products with the certainty are exact no-ops, so the column sums of a tap row are sums of WEIGHTS that depend on the
// frame through the x parity bit only -- one select per site column between two sums the caller makes once per pixel
// for all frames (P[r][0..3] = W0+W1, W1+W2, W2+W3, W3+W4 of tap row r = 0, 1, 2; rows 3 and 4 mirror rows 1 and 0).
// No certainty reads, no certainty addresses, 15 selects instead of 45 and/add/mul/fma: bit-identical to strip_pixel_w
// on such a pixel (same sums in the same association; x + 0 and x * 1 are exact).
template <int K, int CFA, typename RawF>
__device__ __forceinline__ void strip_pixel_sat(int X, int Y, int sx, int sy, const float (&w)[13], const float (&P)[3][4], RawF rawf,
                                                 const StripLevels& lv, float* accP, float* accW)
{
    const int qx = X + sx - 2, qy = Y + sy - 2;
    const int x0 = qx >> 1, y0 = qy >> 1;
    uint32_t mbx = 0u - (uint32_t)(qx & 1), mby = 0u - (uint32_t)(qy & 1);
    uint32_t nby = ~mby;
    asm volatile("" : "+v"(mbx), "+v"(mby), "+v"(nby));
    const uint32_t mP = 0u - (uint32_t)(x0 & 1), mQ = 0u - (uint32_t)(y0 & 1);
    float s[3][3];
    rawf(x0, y0, s);
    auto W_ = [&](int jt, int it) { const int n = jt * 5 + it; return w[n <= 12 ? n : 24 - n]; };
    // W(jt, i) + W(jt, i + 1): tap rows 3 and 4 are rows 1 and 0 read backwards (w[n] == w[24 - n])
    auto P_ = [&](int jt, int i) { return jt <= 2 ? P[jt][i] : P[4 - jt][3 - i]; };
    float C[5][3];
#pragma unroll
    for (int jt = 0; jt < 5; jt++) {
        C[jt][0] = selm(mbx, W_(jt, 0), P_(jt, 0));   // tap 0, tap 1 if bx == 0
        C[jt][1] = selm(mbx, P_(jt, 1), P_(jt, 2));   // tap 1 if bx == 1, tap 2, tap 3 if bx == 0
        C[jt][2] = selm(mbx, P_(jt, 3), W_(jt, 4));   // tap 3 if bx == 1, tap 4
    }
    strip_pixel_sites<K, CFA>(C, s, mby, nby, mP, mQ, lv, accP, accW);
}

// P[r][i] = W(r, i) + W(r, i + 1) of tap rows 0..2, in the association strip_pixel_w uses (see strip_pixel_sat)
__device__ __forceinline__ void tap_pair_sums(const float (&w)[13], float (&P)[3][4])
{
    auto W_ = [&](int jt, int it) { const int n = jt * 5 + it; return w[n <= 12 ? n : 24 - n]; };
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int i = 0; i < 4; i++) P[r][i] = W_(r, i) + W_(r, i + 1);
}

template <int K, int CFA, bool PARITY = false, typename MaskF>
__device__ __forceinline__ void strip_pixel(int X, int Y, int sx, int sy, float kx, float ky, float kz,
                                             const uint16_t* __restrict__ raw, int dimX, MaskF mval,
                                             const StripLevels& lv, float* accP, float* accW)
{
    float w[13];
    tap_weights13(kx, ky, kz, w);
    auto rawf = [&](int x0, int y0, float (&s)[3][3]) {
        const uint16_t* r = raw + (size_t)y0 * dimX + x0;
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int i = 0; i < 3; i++) s[j][i] = (float)r[j * dimX + i];
    };
    strip_pixel_w<K, CFA, PARITY>(X, Y, sx, sy, w, rawf, mval, lv, accP, accW);
}

// Frame margin (HR pixels) that the strip/tile kernels leave to k_accumulateMargin: taps of
// pixels this close to the frame border can be clamped, which the fast path does not handle.
// Handling them in a separate small launch keeps the fast kernels free of divergent border
// waves (a wave with one border strip used to execute both paths: +12 % VALU instructions).
#define STRIP_MARGIN 16
// margin kernels: branch-free taps with per-colour sums (tap_accumulate_sums: 81 -> 55 us per 4-frame margin launch at
// 4K); 0: colour branches (A/B).  (The tile kernels' in-kernel straight path keeps the branches: the unrolled branch-free
// taps push those kernels over their register budget.)
#ifndef MARGIN_TAP_SUMS
#define MARGIN_TAP_SUMS 1
#endif

// SCALE is a template parameter: the straight arithmetic takes floor((x + px + sx) / scale) twenty times per pixel, and an
// integer division by a run-time value is ~40 instructions -- half of this kernel's work when the scale was a kernel argument
template <int SCALE>
__global__ void __launch_bounds__(256)
    k_accumulateMargin(const uint16_t* __restrict__ raw, pix3* __restrict__ imgOut, pix3* __restrict__ totalWeights,
                       const float4* __restrict__ certaintyMask, mfsr_tex2d kernelParam, mfsr_tex2d shifts, Levels3 glv,
                       int dimX, int dimY, int strideOut, int strideMask, int cfaPacked, int rowBegin, int rowEnd)
{
    constexpr int scale = SCALE;
    const int hrW = scale * dimX, hrH = scale * dimY, M = STRIP_MARGIN;
    const int rowLen = hrW - 2;                    // x in [1, hrW-1)
    const int nTop = (M - 1) * rowLen;             // y in [1, M)
    const int nBot = (M - 1) * rowLen;             // y in [hrH-M, hrH-1)
    const int sideLen = 2 * (M - 1);               // x in [1, M) and [hrW-M, hrW-1)
    const int nSide = (hrH - 2 * M) * sideLen;     // y in [M, hrH-M)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    int x, y;
    if (idx < nTop) {
        y = 1 + idx / rowLen;
        x = 1 + idx % rowLen;
    } else if (idx < nTop + nBot) {
        const int r = idx - nTop;
        y = hrH - M + r / rowLen;
        x = 1 + r % rowLen;
    } else if (idx < nTop + nBot + nSide) {
        const int r = idx - nTop - nBot;
        y = M + r / sideLen;
        const int c = r % sideLen;
        x = c < M - 1 ? 1 + c : hrW - M + (c - (M - 1));
    } else {
        return;
    }
    if (y < rowBegin || y >= rowEnd) return;  // HR row window of this launch (stripe-sharded bursts)
    accumulate_pixel_generic<GEOM_FULL, true, MARGIN_TAP_SUMS>(x, y, raw, imgOut, totalWeights, certaintyMask, kernelParam, shifts, glv, dimX,
                                              dimY, scale, strideOut, strideMask, cfaPacked);
}

// FR = HR pixels per field texel along each axis (4: fields at LR/2, the Bayer
// pipeline; 2: fields at LR, the monochrome pipeline; 0: any size, per-pixel fetch).
// per-frame arguments of the multi-frame kernels
struct TileFrame {
    const uint16_t* raw;
    const float4* mask;
    mfsr_tex2d shifts;
};
template <int NF>
struct TileFrames {
    TileFrame f[NF];
};

// the margin pixels of NF frames in one launch.  The straight arithmetic is a chain of dependent loads (flow, kernel
// parameters, then raw and certainty per tap) on 7.6 K wavefronts at 4K: one thread per (pixel, frame) -- 64 pixels x NF
// frames per workgroup -- keeps the chain one frame long; the frames' sums (each taken from zero) meet in LDS and are
// added to the accumulators in call order by the pixel's first thread, which reads and writes them once.
template <int NF, int SCALE>
__global__ void __launch_bounds__(64 * NF)
    k_accumulateMarginN(TileFrames<NF> fr, pix3* __restrict__ imgOut, pix3* __restrict__ totalWeights, mfsr_tex2d kernelParam,
                        Levels3 glv, int dimX, int dimY, int strideOut, int strideMask, int cfaPacked, int rowBegin, int rowEnd)
{
    constexpr int scale = SCALE;
    __shared__ float sSum[NF][6][64];
    const int hrW = scale * dimX, hrH = scale * dimY, M = STRIP_MARGIN;
    const int rowLen = hrW - 2;
    const int nTop = (M - 1) * rowLen, nBot = (M - 1) * rowLen;
    const int sideLen = 2 * (M - 1);
    const int nSide = (hrH - 2 * M) * sideLen;
    const int lane = threadIdx.x, n = threadIdx.y;
    const int idx = blockIdx.x * 64 + lane;
    int x = 0, y = -1;
    if (idx < nTop) {
        y = 1 + idx / rowLen;
        x = 1 + idx % rowLen;
    } else if (idx < nTop + nBot) {
        const int r = idx - nTop;
        y = hrH - M + r / rowLen;
        x = 1 + r % rowLen;
    } else if (idx < nTop + nBot + nSide) {
        const int r = idx - nTop - nBot;
        y = M + r / sideLen;
        const int c = r % sideLen;
        x = c < M - 1 ? 1 + c : hrW - M + (c - (M - 1));
    }
    const bool live = y >= rowBegin && y < rowEnd && y >= 0;
    pix3 pixel = {0.0f, 0.0f, 0.0f}, totalWeight = {0.0f, 0.0f, 0.0f};
    if (live) {
        TileFrame F = fr.f[0];
#pragma unroll
        for (int m = 1; m < NF; m++)
            if (n == m) F = fr.f[m];
        accumulate_pixel_core<GEOM_FULL, true, MARGIN_TAP_SUMS>(x, y, F.raw, F.mask, kernelParam, F.shifts, glv, dimX, dimY, scale,
                                                                strideMask, cfaPacked, pixel, totalWeight);
    }
    sSum[n][0][lane] = pixel.x;
    sSum[n][1][lane] = pixel.y;
    sSum[n][2][lane] = pixel.z;
    sSum[n][3][lane] = totalWeight.x;
    sSum[n][4][lane] = totalWeight.y;
    sSum[n][5][lane] = totalWeight.z;
    __syncthreads();
    if (n == 0 && live) {
        pix3 a = row_ptr(imgOut, strideOut, y)[x];
        pix3 w = row_ptr(totalWeights, strideOut, y)[x];
#pragma unroll
        for (int m = 0; m < NF; m++) {
            a.x += sSum[m][0][lane];
            a.y += sSum[m][1][lane];
            a.z += sSum[m][2][lane];
            w.x += sSum[m][3][lane];
            w.y += sSum[m][4][lane];
            w.z += sSum[m][5][lane];
        }
        row_ptr(imgOut, strideOut, y)[x] = a;
        row_ptr(totalWeights, strideOut, y)[x] = w;
    }
}

// one frame of one strip on the register-resident accumulators (k_accumulate2xStrip)
template <int CFA, int FR>
__device__ __forceinline__ void strip_frame(int tx, int Y, int X0, const uint16_t* __restrict__ raw,
                                            const float4* __restrict__ certaintyMask, const mfsr_tex2d& kernelParam,
                                            const mfsr_tex2d& shifts, const Levels3& glv, const StripLevels& lv, int dimX,
                                            int dimY, int strideMask, int cfaPacked, float* accP, float* accW)
{
    const int hrW = 2 * dimX, hrH = 2 * dimY;
    const float posY = ((float)Y + 0.5f) / (float)hrH;
    int sx[4], sy[4];
    float kx[4], ky[4], kz[4];
    // a strip is "safe" when no tap of its four pixels is clamped at the frame border
    bool safe = true;  // (the frame margin is handled by k_accumulateMargin)
    if (FR == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float posX = ((float)(X0 + k) + 0.5f) / (float)hrW;
            const float4 kp = tex4<ADDR_CLAMP>(kernelParam, posX, posY);
            const float2 sh = tex2<ADDR_CLAMP>(shifts, posX, posY);
            kx[k] = kp.x;
            ky[k] = kp.y;
            kz[k] = kp.z;
            sx[k] = round2i(sh.x * 2.0f);
            sy[k] = round2i(sh.y * 2.0f);
        }
    } else {
        // The four pixels of a strip read the same few field texels: fetch them once
        // (FR=4: 3 columns x 2 rows, FR=2: 4 x 2) and interpolate with exactly the
        // arithmetic of tex_coord/lerp4, so roundf(2*flow) stays bit-identical.
        constexpr int NC = FR == 4 ? 3 : 4;
        const int fw = kernelParam.width, fh = kernelParam.height;  // == shifts' size (checked on the host)
        float yB = posY * (float)fh - 0.5f;
        if (!finitef(yB)) yB = 0.0f;
        const float fy = floorf(yB);
        const float b = yB - fy;
        const int j0 = clampi(f2i(fy), 0, fh - 1), j1 = clampi(f2i(fy) + 1, 0, fh - 1);
        const int cbase = clampi(FR == 4 ? tx - 1 : 2 * tx - 1, 0, fw - NC);
        float K0[NC][3], K1[NC][3], F0[NC][2], F1[NC][2];
        {
            const float4* k0 = row_ptr((const float4*)kernelParam.ptr, kernelParam.pitch, j0) + cbase;
            const float4* k1 = row_ptr((const float4*)kernelParam.ptr, kernelParam.pitch, j1) + cbase;
            const float2* f0 = row_ptr((const float2*)shifts.ptr, shifts.pitch, j0) + cbase;
            const float2* f1 = row_ptr((const float2*)shifts.ptr, shifts.pitch, j1) + cbase;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const float4 u = k0[c], v = k1[c];
                const float2 p = f0[c], q = f1[c];
                K0[c][0] = u.x; K0[c][1] = u.y; K0[c][2] = u.z;
                K1[c][0] = v.x; K1[c][1] = v.y; K1[c][2] = v.z;
                F0[c][0] = p.x; F0[c][1] = p.y;
                F1[c][0] = q.x; F1[c][1] = q.y;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float posX = ((float)(X0 + k) + 0.5f) / (float)hrW;
            float xB = posX * (float)fw - 0.5f;
            if (!finitef(xB)) xB = 0.0f;
            const float fx = floorf(xB);
            const float a = xB - fx;
            // texel column predicted from the strip geometry; verified against the float path
            const int ci = FR == 4 ? (k < 2 ? 0 : 1) : (k == 0 ? 0 : (k == 3 ? 2 : 1));
            safe = safe && (f2i(fx) == cbase + ci) && (cbase + ci + 1 <= fw - 1);
            kx[k] = lerp4(K0[ci][0], K0[ci + 1][0], K1[ci][0], K1[ci + 1][0], a, b);
            ky[k] = lerp4(K0[ci][1], K0[ci + 1][1], K1[ci][1], K1[ci + 1][1], a, b);
            kz[k] = lerp4(K0[ci][2], K0[ci + 1][2], K1[ci][2], K1[ci + 1][2], a, b);
            const float ux = lerp4(F0[ci][0], F0[ci + 1][0], F1[ci][0], F1[ci + 1][0], a, b);
            const float uy = lerp4(F0[ci][1], F0[ci + 1][1], F1[ci][1], F1[ci + 1][1], a, b);
            sx[k] = round2i(ux * 2.0f);
            sy[k] = round2i(uy * 2.0f);
        }
        safe = safe && (f2i(fy) >= 0) && (f2i(fy) + 1 <= fh - 1);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) safe = safe && psd_ok(kx[k], ky[k], kz[k]);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int qx = X0 + k + sx[k] - 2, qy = Y + sy[k] - 2;
        safe = safe && qx >= 0 && ((qx + 4) >> 1) <= dimX - 1 && qy >= 0 && ((qy + 4) >> 1) <= dimY - 1;
        // guard against saturated conversions of wild flow values
        safe = safe && sx[k] > -(1 << 20) && sx[k] < (1 << 20) && sy[k] > -(1 << 20) && sy[k] < (1 << 20);
    }
    if (!safe) {
        // border-free but wild-flow / non-PSD strips: the straight per-pixel arithmetic on the register values
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
            const int X = X0 + k;
            if (X >= 1 && X < hrW - 1) {
                pix3 px = {accP[3 * k], accP[3 * k + 1], accP[3 * k + 2]};
                pix3 tw = {accW[3 * k], accW[3 * k + 1], accW[3 * k + 2]};
                accumulate_pixel_core<GEOM_FULL, true>(X, Y, raw, certaintyMask, kernelParam, shifts, glv, dimX, dimY, 2,
                                                       strideMask, cfaPacked, px, tw);
                accP[3 * k] = px.x; accP[3 * k + 1] = px.y; accP[3 * k + 2] = px.z;
                accW[3 * k] = tw.x; accW[3 * k + 1] = tw.y; accW[3 * k + 2] = tw.z;
            }
        }
        return;
    }

    // certainty texels: rows (Y-2)>>2 and (Y+2)>>2, cells tx-1, tx, tx+1; non-finite -> 0 (:438-439)
    const int r0 = (Y - 2) >> 2;
    const int sw = 4 * (r0 + 1) - (Y - 2);  // first tap row that reads mask row r0+1 (1..4), wave-uniform
    float M0[3][3], M1[3][3];
    {
        const float4* m0 = row_ptr(certaintyMask, strideMask, r0) + (tx - 1);
        const float4* m1 = row_ptr(certaintyMask, strideMask, r0 + 1) + (tx - 1);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float4 u = m0[c], v = m1[c];
            M0[c][0] = sane(u.x); M0[c][1] = sane(u.y); M0[c][2] = sane(u.z);
            M1[c][0] = sane(v.x); M1[c][1] = sane(v.y); M1[c][2] = sane(v.z);
        }
    }

    auto mval = [&](int jt, int cell, int ch) {
        // mask row (Y-2+jt)>>2 is r0 for jt < sw and r0+1 otherwise (wave-uniform)
        return jt == 0 ? M0[cell][ch] : (jt == 4 ? M1[cell][ch] : ((jt >= sw) ? M1[cell][ch] : M0[cell][ch]));
    };
    strip_pixel<0, CFA>(X0 + 0, Y, sx[0], sy[0], kx[0], ky[0], kz[0], raw, dimX, mval, lv, accP, accW);
    strip_pixel<1, CFA>(X0 + 1, Y, sx[1], sy[1], kx[1], ky[1], kz[1], raw, dimX, mval, lv, accP, accW);
    strip_pixel<2, CFA>(X0 + 2, Y, sx[2], sy[2], kx[2], ky[2], kz[2], raw, dimX, mval, lv, accP, accW);
    strip_pixel<3, CFA>(X0 + 3, Y, sx[3], sy[3], kx[3], ky[3], kz[3], raw, dimX, mval, lv, accP, accW);

}

// Register-only strip kernel: serves every field resolution (FR) without LDS; NF frames add into the
// accumulators while they sit in registers (one read-modify-write of HBM for both).
template <int CFA, int FR, int NF>
__global__ void __launch_bounds__(256)
    k_accumulate2xStrip(TileFrames<NF> fr, pix3* __restrict__ imgOut, pix3* __restrict__ totalWeights, mfsr_tex2d kernelParam,
                        Levels3 glv, StripLevels lv, int dimX, int dimY, int strideOut, int strideMask, int cfaPacked, int rowBlock0)
{
    const int tx = blockIdx.x * 64 + threadIdx.x;
    const int Y = (blockIdx.y + rowBlock0) * 4 + threadIdx.y;
    const int hrW = 2 * dimX, hrH = 2 * dimY;
    const int X0 = 4 * tx;
    if (X0 < STRIP_MARGIN || X0 >= hrW - STRIP_MARGIN || Y < STRIP_MARGIN || Y >= hrH - STRIP_MARGIN) return;

    // accumulators: 4 pixels x 3 channels = 48 contiguous bytes per plane-set
    float4* pP = (float4*)((char*)imgOut + (size_t)Y * strideOut + (size_t)X0 * 12);
    float4* pW = (float4*)((char*)totalWeights + (size_t)Y * strideOut + (size_t)X0 * 12);
    float accP[12], accW[12];
    {
        const float4 a0 = pP[0], a1 = pP[1], a2 = pP[2];
        const float4 b0 = pW[0], b1 = pW[1], b2 = pW[2];
        accP[0] = a0.x; accP[1] = a0.y; accP[2] = a0.z; accP[3] = a0.w;
        accP[4] = a1.x; accP[5] = a1.y; accP[6] = a1.z; accP[7] = a1.w;
        accP[8] = a2.x; accP[9] = a2.y; accP[10] = a2.z; accP[11] = a2.w;
        accW[0] = b0.x; accW[1] = b0.y; accW[2] = b0.z; accW[3] = b0.w;
        accW[4] = b1.x; accW[5] = b1.y; accW[6] = b1.z; accW[7] = b1.w;
        accW[8] = b2.x; accW[9] = b2.y; accW[10] = b2.z; accW[11] = b2.w;
    }
#pragma unroll
    for (int n = 0; n < NF; n++)
        strip_frame<CFA, FR>(tx, Y, X0, fr.f[n].raw, fr.f[n].mask, kernelParam, fr.f[n].shifts, glv, lv, dimX, dimY, strideMask,
                             cfaPacked, accP, accW);
    pP[0] = make_float4(accP[0], accP[1], accP[2], accP[3]);
    pP[1] = make_float4(accP[4], accP[5], accP[6], accP[7]);
    pP[2] = make_float4(accP[8], accP[9], accP[10], accP[11]);
    pW[0] = make_float4(accW[0], accW[1], accW[2], accW[3]);
    pW[1] = make_float4(accW[4], accW[5], accW[6], accW[7]);
    pW[2] = make_float4(accW[8], accW[9], accW[10], accW[11]);
}

// ---- LDS tile kernel (fields at HR/4: the Bayer pipeline) --------------------------------------
// One 64x4 workgroup covers a 256 x 4 HR tile.
//  * The kernel-parameter / flow / certainty texels the tile touches (3 rows x 66 columns each) are
//    staged once in LDS (certainties sanitised, kernel parameters tagged "positive semi-definite"
//    while staging), and so are the per-column and per-row interpolation fractions (one IEEE
//    division per column of the tile instead of one per pixel).
//  * The two accumulator row segments of a wave (3 KiB each) go global -> LDS with
//    global_load_lds (no VGPRs, 1 KiB contiguous per instruction, in flight during the whole tap
//    arithmetic), are updated in LDS and written back in memory order: every accumulator access
//    on the HBM side is a fully used 128-byte line.
#define TILE_COLS 66
// waves per SIMD the register budget is sized for: 4 with one frame per launch (125 VGPRs), 3 with two
// (163 VGPRs, no spill; 128 would spill 57)
#ifndef TILE_WAVES
#define TILE_WAVES 4
#endif
#ifndef TILE_WAVES2
#define TILE_WAVES2 4
#endif
// Three and four frames per launch: both plane-sets staged (43 / 48 KiB of LDS: three workgroups per CU, so the register
// budget is 168) or one at a time (31 / 36 KiB: four workgroups, 128 registers).  Measured at 4K x 16, four frames per
// launch: 1.004 ms with both (no spills) against 1.044 - 1.068 ms with one (10 spilled registers and the second plane-set's
// load exposed) -- both is the default; 0 keeps the other form for A/B.
#ifndef TILE_GROUP_BOTH_PLANES
#define TILE_GROUP_BOTH_PLANES 1
#endif
#define TILE_WAVES_NF(NF) ((NF) == 1 ? TILE_WAVES : ((NF) > 2 && TILE_GROUP_BOTH_PLANES) ? (TILE_LDS_ALIAS_ON ? 4 : 3) : TILE_WAVES2)
// several frames per launch: pixel-major order with the tap weights shared by the frames (0: frame-major, weights per
// frame; two frames at most)
#ifndef TILE_PIXEL_MAJOR
#define TILE_PIXEL_MAJOR 1
#endif
#define TILE_LDS_ALIAS_ON (TILE_LDS_ALIAS && TILE_PIXEL_MAJOR && TILE_GROUP_BOTH_PLANES)
// Four workgroups per CU for the group kernels (three and four frames, fields at HR/4): the weight-sum plane-set is staged
// over the kernel-parameter and flow texels once every wave is past its last read of them (39 KiB of LDS instead of 48), and
// the rounded flow of a strip is kept in 8:8 bits per pixel relative to the tile's own rounded flow (8 registers instead of 16;
// strips more than 127 HR pixels away from it take the straight arithmetic), which fits the 128-register budget of four waves per SIMD.  0: three workgroups (A/B).
#ifndef TILE_LDS_ALIAS
#define TILE_LDS_ALIAS 1
#endif
// pixel-major loop: frames whose certainty is saturated over the wave's footprint take strip_pixel_sat (0: A/B)
#ifndef TILE_SAT_PATH
#define TILE_SAT_PATH 1
#endif
// (Measured and not kept: the nine raw sites of a body loaded one body ahead -- 161 VGPRs, the address arithmetic twice:
// 0.96 against 0.876 ms per isolated launch.  The round trip of the raw loads is already hidden.)
// NF frames per launch (1 to 4).  Everything that does not depend on the frame is done once for
// both: the accumulator staging and write-back (the 48 B/px/frame of HBM traffic become 24), the
// kernel-parameter mix, the column/row fractions.  The two frames add into the same registers.

// FR = HR pixels per kernel-parameter / flow texel: 4 (the Bayer pipeline: fields at LR/2) or 2 (the monochrome pipeline:
// fields at LR; 130 x 4 texels per tile, four texel columns per strip); the certainty mask is at LR/2 either way.
template <int CFA, int NF, int FR = 4>
__global__ void __launch_bounds__(256, (FR != 4 ? 3 : TILE_WAVES_NF(NF)))  // FR = 2: 41 / 49 KB of LDS, three workgroups per CU
    k_accumulate2xTile(TileFrames<NF> fr, pix3* __restrict__ imgOut, pix3* __restrict__ totalWeights, mfsr_tex2d kernelParam,
                       Levels3 glv, StripLevels lv, int dimX, int dimY, int strideOut, int strideMask, int cfaPacked,
                       int tilesX, int tilesY, int tilesPerXcd, int fresh, int tileY0)
{
    // XCD-aware tile order (tilesPerXcd > 0, the default since round 4).  Workgroups are dealt round-robin to the 8 XCDs
    // (workgroup i -> XCD i & 7), each with its own L2.  A 256 x 4 tile stages 66 x 3 field texels (kernel parameters, the
    // flows and certainties of its frames: 112 B per texel with four frames) of which it owns 64 x 1, and its raw rows
    // overlap those of the tiles above and below: in launch order vertically adjacent tiles sit on DIFFERENT XCDs, so every
    // one of those re-reads misses its L2 and goes to the fabric (served by the Infinity Cache, counted by the L2's
    // memory-side counters): 1.16 GB read per launch beyond the accumulators against 0.30 GB of inputs (PMC, request-size
    // counters, profiles/r04_pmc_fuse_xcd.txt).  With the remap XCD k walks its own contiguous band of tiles in row-major
    // order, the neighbours are L2 hits, and the launch reads 0.305 GB + the accumulators: exactly its inputs.  The launch
    // time does not move (0.86 ms either way: the kernel is not bound by bytes) -- what it buys is 0.87 GB less fabric
    // traffic per launch for the kernels running beside it, and a traffic figure that is the algorithmic one.
    const int pid = (int)blockIdx.x;
    const int tile = tilesPerXcd > 0 ? (pid & 7) * tilesPerXcd + (pid >> 3) : pid;  // tilesPerXcd == 0: launch order (A/B)
    if (tile >= tilesX * tilesY) return;  // whole workgroup (uniform), before any barrier
    const int bIdYrel = tile / tilesX, bIdX = tile - bIdYrel * tilesX;
    const int bIdY = bIdYrel + tileY0;  // tileY0: first tile row of this launch's HR row window
    static_assert(FR == 4 || FR == 2, "field resolution");
    constexpr int FC = 256 / FR + 2, FROWS = 4 / FR + 2;  // field texels a tile touches: 66 x 3 (FR = 4), 130 x 4 (FR = 2)
    constexpr int PL = (NF <= 2 || TILE_GROUP_BOTH_PLANES) ? 2 : 1;
    // ALIAS: the field texels live in the (not yet staged) weight-sum plane-set; see TILE_LDS_ALIAS
    constexpr bool ALIAS = TILE_LDS_ALIAS_ON && FR == 4 && NF > 2 && PL == 2;
    __shared__ __attribute__((aligned(16))) float4 sAcc[PL][4][192];
    __shared__ float4 sKown[ALIAS ? 1 : FROWS][ALIAS ? 1 : FC];
    __shared__ float2 sFown[ALIAS ? 1 : NF][ALIAS ? 1 : FROWS][ALIAS ? 1 : FC];
    static_assert(!ALIAS || sizeof(float4) * FROWS * FC + sizeof(float2) * NF * FROWS * FC <= sizeof(float4) * 4 * 192, "field texels fit a plane-set");
    float4(*sK)[FC] = ALIAS ? (float4(*)[FC]) & sAcc[PL - 1][0][0] : (float4(*)[FC]) & sKown[0][0];  // .w = 1 if the texel is PSD and finite, else 0
    float2(*sF)[FROWS][FC] = ALIAS ? (float2(*)[FROWS][FC])((char*)&sAcc[PL - 1][0][0] + sizeof(float4) * FROWS * FC)
                                   : (float2(*)[FROWS][FC]) & sFown[0][0][0];
    __shared__ float4 sM[NF][3][TILE_COLS];
    __shared__ uint32_t sUnsat[3][TILE_COLS];  // bit n: frame n's certainty texel is not (1, 1, 1)  (TILE_SAT_PATH)
    __shared__ __attribute__((aligned(16))) float sColA[256];  // x fraction per HR column of the tile, -1 = not on the predicted texel
    __shared__ float sRowB[4];                                 // y fraction per HR row of the tile, -1 likewise
    // accumulator staging: per wave and plane-set the wave's 3 KiB row segment, in memory order
    // (TILE_GROUP_BOTH_PLANES = 0, three and four frames per launch: one plane-set at a time through the same 3 KiB so that
    // four workgroups per CU fit the 160 KiB -- the weight sums wait in registers, the second plane-set's load is exposed)
    const int lx = threadIdx.x, ly = threadIdx.y;
    const int tx = bIdX * 64 + lx;
    const int Y = bIdY * 4 + ly;
    const int hrW = 2 * dimX, hrH = 2 * dimY;
    const int X0 = 4 * tx;
    const int fw = kernelParam.width, fh = kernelParam.height;  // == hrW/FR, hrH/FR (checked on the host)
    const int mw = dimX / 2, mh = dimY / 2;                       // certainty mask size
    {
        const int t = ly * 64 + lx;
        if constexpr (FR != 4) {
            // fields and certainty on different grids: two staging loops
            for (int i = t; i < FROWS * FC; i += 256) {
                const int r = i / FC, c = i - r * FC;
                const int gy = bIdY * (4 / FR) - 1 + r, gx = bIdX * (256 / FR) - 1 + c;
                const int fy = clampi(gy, 0, fh - 1), fx = clampi(gx, 0, fw - 1);
                float4 k = row_ptr((const float4*)kernelParam.ptr, kernelParam.pitch, fy)[fx];
                k.w = psd_ok(k.x, k.y, k.z) ? 1.0f : 0.0f;
                sK[r][c] = k;
#pragma unroll
                for (int n = 0; n < NF; n++) sF[n][r][c] = row_ptr((const float2*)fr.f[n].shifts.ptr, fr.f[n].shifts.pitch, fy)[fx];
            }
            if (t < 3 * TILE_COLS) {
                const int r = t / TILE_COLS, c = t - r * TILE_COLS;
                const int gy = bIdY - 1 + r, gx = bIdX * 64 - 1 + c;
                uint32_t unsat = 0;
#pragma unroll
                for (int n = 0; n < NF; n++) {
                    const float4 m = row_ptr(fr.f[n].mask, strideMask, clampi(gy, 0, mh - 1))[clampi(gx, 0, mw - 1)];
                    const float mc[3] = {sane(m.x), sane(m.y), sane(m.z)};
                    sM[n][r][c] = make_float4(mc[Cfa<CFA>::col(0, 0)], mc[Cfa<CFA>::col(0, 1)], mc[Cfa<CFA>::col(1, 0)],
                                              mc[Cfa<CFA>::col(1, 1)]);
                    if (!(mc[0] == 1.0f && mc[1] == 1.0f && mc[2] == 1.0f)) unsat |= 1u << n;
                }
                sUnsat[r][c] = unsat;
            }
        } else if (t < 3 * TILE_COLS) {
            const int r = t / TILE_COLS, c = t - r * TILE_COLS;
            const int gy = bIdY - 1 + r, gx = bIdX * 64 - 1 + c;
            const int fy = clampi(gy, 0, fh - 1), fx = clampi(gx, 0, fw - 1);
            float4 k = row_ptr((const float4*)kernelParam.ptr, kernelParam.pitch, fy)[fx];
            // a bilinear mix of PSD matrices is PSD, so admitting texels admits every pixel between them
            k.w = psd_ok(k.x, k.y, k.z) ? 1.0f : 0.0f;
            sK[r][c] = k;
            uint32_t unsat = 0;
#pragma unroll
            for (int n = 0; n < NF; n++) {
                sF[n][r][c] = row_ptr((const float2*)fr.f[n].shifts.ptr, fr.f[n].shifts.pitch, fy)[fx];
                const float4 m = row_ptr(fr.f[n].mask, strideMask, clampi(gy, 0, mh - 1))[clampi(gx, 0, mw - 1)];
                // stored by CFA position (y parity << 1 | x parity), not by channel: strip_pixel<PARITY> indexes it
                const float mc[3] = {sane(m.x), sane(m.y), sane(m.z)};
                sM[n][r][c] = make_float4(mc[Cfa<CFA>::col(0, 0)], mc[Cfa<CFA>::col(0, 1)], mc[Cfa<CFA>::col(1, 0)],
                                          mc[Cfa<CFA>::col(1, 1)]);
                if (!(mc[0] == 1.0f && mc[1] == 1.0f && mc[2] == 1.0f)) unsat |= 1u << n;
            }
            sUnsat[r][c] = unsat;
        }
        {
            // column t of the tile: the float path of tex_coord, texel column predicted and verified
            const int X = bIdX * 256 + t;
            const float posX = ((float)X + 0.5f) / (float)hrW;
            float xB = posX * (float)fw - 0.5f;
            if (!finitef(xB)) xB = 0.0f;
            const float fxf = floorf(xB);
            // texel column floor(xB) as predicted: (X / FR) - 1 for the first half of a texel's pixels, X / FR for the second
            const int txc = FR == 4 ? X >> 2 : X >> 1, ci = FR == 4 ? ((t & 3) < 2 ? 0 : 1) : (t & 1);
            const bool ok = (f2i(fxf) == txc - 1 + ci) && (txc + ci <= fw - 1);
            sColA[t] = ok ? xB - fxf : -1.0f;
        }
        if (t < 4) {
            const int Yr = bIdY * 4 + t;
            const float posY = ((float)Yr + 0.5f) / (float)hrH;
            float yB = posY * (float)fh - 0.5f;
            if (!finitef(yB)) yB = 0.0f;
            const float fyf = floorf(yB);
            const int frr = FR == 4 ? (t < 2 ? 0 : 1) : (t + 1) >> 1;  // LDS row of texel row floor(yB)
            const bool ok = (f2i(fyf) == bIdY * (4 / FR) - 1 + frr) && f2i(fyf) >= 0 && f2i(fyf) + 1 <= fh - 1;
            sRowB[t] = ok ? yB - fyf : -1.0f;
        }
    }
    // asynchronous global -> LDS copy of both accumulator segments; consumed only after the tap arithmetic
    const bool rowLive = Y >= STRIP_MARGIN && Y < hrH - STRIP_MARGIN;
    const size_t rowBytes = (size_t)hrW * 12;
    const size_t segByte = (size_t)bIdX * 3072;
    char* gP = (char*)imgOut + (size_t)Y * strideOut + segByte;
    char* gW = (char*)totalWeights + (size_t)Y * strideOut + segByte;
    // plane-set g (this wave's row segment) -> LDS plane pl: zeroes on the first launch of a burst (the accumulators are
    // defined to be zero and are not read at all), else the asynchronous copy
    auto stage_plane = [&](char* g, int pl) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            if (fresh) {
                sAcc[pl][ly][j * 64 + lx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            } else {
                const size_t off = (size_t)(j * 64 + lx) * 16;
                if (segByte + off + 16 <= rowBytes)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + off),
                                                     (__attribute__((address_space(3))) void*)&sAcc[pl][ly][j * 64], 16, 0, 0);
            }
        }
    };
    if (rowLive) {
        stage_plane(gP, 0);
        if (PL == 2 && !ALIAS) stage_plane(gW, 1);  // (ALIAS: after the last read of the field texels, below)
    }
    __syncthreads();
    if (!rowLive) return;
    const bool stripLive = X0 >= STRIP_MARGIN && X0 < hrW - STRIP_MARGIN;

    const int fr_ = FR == 4 ? (ly < 2 ? 0 : 1) : (ly + 1) >> 1;
    constexpr int SC = FR == 4 ? 3 : 4;                               // texel columns a strip touches
    const int cb = FR == 4 ? lx : 2 * lx;                             // LDS column of the first of them
    auto ciOf = [](int k) { return FR == 4 ? (k < 2 ? 0 : 1) : (k + 1) >> 1; };  // left texel of pixel k's pair
    const float b = sRowB[ly];
    const float4 av4 = ((const float4*)sColA)[lx];
    const float av[4] = {av4.x, av4.y, av4.z, av4.w};
    const bool geomOk = stripLive && b >= 0.0f && fminf(fminf(av[0], av[1]), fminf(av[2], av[3])) >= 0.0f;

    // kernel parameters of this strip (frame independent): 3 columns x 2 rows of texels
    float kxa[4], kya[4], kza[4];
    bool kOk;
    {
        float4 Kt[2][SC];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < SC; c++) Kt[r][c] = sK[fr_ + r][cb + c];
        float kAll = ((Kt[0][0].w * Kt[0][1].w) * (Kt[0][2].w * Kt[1][0].w)) * (Kt[1][1].w * Kt[1][2].w);
        if (SC == 4) kAll *= Kt[0][SC - 1].w * Kt[1][SC - 1].w;
        kOk = kAll > 0.0f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int ci = ciOf(k);
            // same four bilinear weights as the flow, summed with fma (no rounding step depends on them)
            const float w00 = (1.0f - av[k]) * (1.0f - b), w10 = av[k] * (1.0f - b), w01 = (1.0f - av[k]) * b, w11 = av[k] * b;
            auto mix = [&](float t00, float t10, float t01, float t11) {
                return __builtin_fmaf(w11, t11, __builtin_fmaf(w01, t01, __builtin_fmaf(w10, t10, w00 * t00)));
            };
            kxa[k] = mix(Kt[0][ci].x, Kt[0][ci + 1].x, Kt[1][ci].x, Kt[1][ci + 1].x);
            kya[k] = mix(Kt[0][ci].y, Kt[0][ci + 1].y, Kt[1][ci].y, Kt[1][ci + 1].y);
            kza[k] = mix(Kt[0][ci].z, Kt[0][ci + 1].z, Kt[1][ci].z, Kt[1][ci + 1].z);
        }
    }

    float accP[12], accW[12];
#pragma unroll
    for (int i = 0; i < 12; i++) accP[i] = accW[i] = 0.0f;
    const uint32_t xmax = (uint32_t)(2 * dimX - 5), ymax = (uint32_t)(2 * dimY - 5);
    uint32_t safeBits = 0;  // bit n: frame n took the fast path
    static_assert(NF <= 2 || TILE_PIXEL_MAJOR, "three and four frames per launch exist in pixel-major order only");
#if TILE_PIXEL_MAJOR
    if constexpr (NF > 1 || FR != 4) {
        // Two frames, pixel-major: the 13 tap weights of a pixel (12 v_exp_f32 and their exponents: a fifth of the
        // pixel's arithmetic) depend on the kernel parameters only, so each pixel takes them ONCE and applies them to
        // both frames before the next pixel starts -- 13 live registers instead of the 4 x 13 a frame-major loop would
        // have to keep (that was 214 VGPRs; recomputing them per frame was the faster frame-major form).
        // Pass 1, per frame: whole-pixel flow of the four pixels (packed 16:16) and the fast-path admission.
        // (ALIAS: 8:8 bits per pixel, sxy[n][0] = the four sx, sxy[n][1] = the four sy; else 16:16 per pixel)
        uint32_t sxy[NF][ALIAS ? 2 : 4];
        // (ALIAS: the 8 bits hold the flow RELATIVE to the rounded flow of the tile's centre texel, a per-frame scalar -- so
        // a large global shift of a frame costs nothing; a strip more than 127 HR pixels away from it takes the straight path)
        int bsx[NF], bsy[NF];
#pragma unroll
        for (int n = 0; n < NF; n++) {
            bsx[n] = bsy[n] = 0;
            if constexpr (ALIAS) {
                sxy[n][0] = sxy[n][1] = 0u;
                const float2 cf = sF[n][1][FC / 2];
                bsx[n] = __builtin_amdgcn_readfirstlane(min(max(round2i(cf.x * 2.0f), -32000), 32000));
                bsy[n] = __builtin_amdgcn_readfirstlane(min(max(round2i(cf.y * 2.0f), -32000), 32000));
            }
            float2 Ft[2][SC];
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int c = 0; c < SC; c++) Ft[r][c] = sF[n][fr_ + r][cb + c];
            bool safe = geomOk && kOk;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int ci = ciOf(k);
                const float ux = lerp4(Ft[0][ci].x, Ft[0][ci + 1].x, Ft[1][ci].x, Ft[1][ci + 1].x, av[k], b);
                const float uy = lerp4(Ft[0][ci].y, Ft[0][ci + 1].y, Ft[1][ci].y, Ft[1][ci + 1].y, av[k], b);
                const int sx = round2i(ux * 2.0f), sy = round2i(uy * 2.0f);
                const int qx = X0 + k + sx - 2, qy = Y + sy - 2;
                // as below, with the rounded flow inside 16 signed bits (wilder strips take the straight arithmetic)
                safe = safe && (uint32_t)(sx + (1 << 15)) < (2u << 15) && (uint32_t)(sy + (1 << 15)) < (2u << 15) &&
                       (uint32_t)qx <= xmax && (uint32_t)qy <= ymax;
                if constexpr (ALIAS) {
                    const int dx = sx - bsx[n], dy = sy - bsy[n];  // (no overflow: both inside 16 bits when it matters)
                    safe = safe && (uint32_t)(dx + 128) < 256u && (uint32_t)(dy + 128) < 256u;
                    sxy[n][0] |= ((uint32_t)dx & 0xffu) << (8 * k);
                    sxy[n][1] |= ((uint32_t)dy & 0xffu) << (8 * k);
                } else {
                    sxy[n][k] = ((uint32_t)sx & 0xffffu) | ((uint32_t)sy << 16);
                }
            }
            if (safe) safeBits |= 1u << n;
        }
        if constexpr (ALIAS) {
            // every wave of the workgroup is past its reads of sK / sF (the rows outside the frame left before the first
            // barrier as whole workgroups): the weight-sum plane-set may land on them
            __syncthreads();
            stage_plane(gW, 1);
        }
        // Pass 2, per pixel: weights once, then frame after frame.  What the eight pixel-frame bodies share is taken
        // here once (each body is its own divergent region, the compiler does not hoist across them): the LDS address of
        // the certainty row each tap row reads, and the raw row bases (uniform: SGPR pairs, 32-bit lane offsets).
        const float* mrow[5];
#pragma unroll
        for (int jt = 0; jt < 5; jt++) mrow[jt] = (const float*)&sM[0][((ly + jt - 2) >> 2) + 1][lx];
        const char* rawRow[NF][3];
#pragma unroll
        for (int n = 0; n < NF; n++)
#pragma unroll
            for (int j = 0; j < 3; j++) rawRow[n][j] = (const char*)(fr.f[n].raw + (size_t)j * dimX);
        // Frames whose certainty is exactly 1 on every texel this WAVE reads (its 66 texel columns on the two mask rows
        // its tap rows touch) take strip_pixel_sat: a wave-uniform (scalar) branch per frame, bit-identical results.
        uint32_t satW = 0;
        if constexpr (TILE_SAT_PATH) {
            const int r0 = ((ly - 2) >> 2) + 1;  // mask rows ((ly + jt - 2) >> 2) + 1, jt = 0..4: r0 and r0 + 1
            uint32_t u = sUnsat[r0][lx] | sUnsat[r0 + 1][lx];
            if (lx < 2) u |= sUnsat[r0][64 + lx] | sUnsat[r0 + 1][64 + lx];
#pragma unroll
            for (int n = 0; n < NF; n++)
                if (__builtin_amdgcn_ballot_w64((u >> n) & 1u) == 0) satW |= 1u << n;
            satW = (uint32_t)__builtin_amdgcn_readfirstlane((int)satW);
        }
        auto pixel = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if (safeBits == 0) return;
            // this pixel's value sums live here only while the pixel is worked on, then go into the staged plane-set
            // (12 registers fewer across the other pixels; the weight sums have no staged plane yet when NF > 2)
            float aP[12];
            aP[3 * k] = aP[3 * k + 1] = aP[3 * k + 2] = 0.0f;
            float w[13];
            tap_weights13(kxa[k], kya[k], kza[k], w);
            float P[3][4];
            if (TILE_SAT_PATH && satW) tap_pair_sums(w, P);
#pragma unroll
            for (int n = 0; n < NF; n++) {
                if ((safeBits >> n) & 1u) {
                    auto rawf = [&](int x0, int y0, float (&sv)[3][3]) {
                        const uint32_t boff = (uint32_t)(y0 * dimX + x0) * 2u;  // admitted strips: x0, y0 >= 0, inside the frame
#pragma unroll
                        for (int j = 0; j < 3; j++) {
                            const char* rb = rawRow[n][j] + (size_t)boff;
#pragma unroll
                            for (int i = 0; i < 3; i++) sv[j][i] = (float)*(const uint16_t*)(rb + 2 * i);
                        }
                    };
                    auto mval = [&](int jt, int cell, int e4) {
                        return *(const float*)((const char*)(mrow[jt] + n * (3 * TILE_COLS * 4) + cell * 4) + e4);
                    };
                    const int sx = ALIAS ? bsx[n] + (int)(int8_t)(sxy[n][0] >> (8 * k)) : (int)(int16_t)(sxy[n][ALIAS ? 0 : k] & 0xffffu);
                    const int sy = ALIAS ? bsy[n] + (int)(int8_t)(sxy[n][1] >> (8 * k)) : (int)sxy[n][ALIAS ? 0 : k] >> 16;
                    if (TILE_SAT_PATH && ((satW >> n) & 1u))
                        strip_pixel_sat<k, CFA>(X0 + k, Y, sx, sy, w, P, rawf, lv, aP, accW);
                    else
                        strip_pixel_w<k, CFA, true>(X0 + k, Y, sx, sy, w, rawf, mval, lv, aP, accW);
                }
            }
            if (k == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the staged plane-set has landed (long ago)
            float* myP = (float*)&sAcc[0][ly][0] + lx * 12 + 3 * k;
            myP[0] += aP[3 * k];
            myP[1] += aP[3 * k + 1];
            myP[2] += aP[3 * k + 2];
        };
        pixel(std::integral_constant<int, 0>{});
        pixel(std::integral_constant<int, 1>{});
        pixel(std::integral_constant<int, 2>{});
        pixel(std::integral_constant<int, 3>{});
    } else
#endif
    {
    // one frame after the other in a real loop (not unrolled: the two bodies would only compete for
    // registers); the frame's pointers are picked with scalar selects so the argument struct stays
    // in SGPRs
#pragma unroll 1
    for (int n = 0; n < NF; n++) {
        const uint16_t* raw = (NF > 1 && n) ? fr.f[NF - 1].raw : fr.f[0].raw;
        // Per-frame copies of the interpolation fractions that the compiler cannot see through: otherwise
        // it hoists the four bilinear weights of each pixel (and more) out of the frame loop and the
        // kernel needs 194 VGPRs; like this it fits 163 with no spill (profiles/r01_ab_pair_opaque.txt:
        // 0.667 -> 0.626 ms per pair).
        float avn[4] = {av[0], av[1], av[2], av[3]};
        float bn = b;
#pragma unroll
        for (int k = 0; k < 4; k++) asm volatile("" : "+v"(avn[k]));
        asm volatile("" : "+v"(bn));
#define av avn
#define b bn
        float2 Ft[2][3];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) Ft[r][c] = sF[n][fr_ + r][lx + c];
        int sx[4], sy[4];
        bool safe = geomOk && kOk;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int ci = k < 2 ? 0 : 1;
            const float ux = lerp4(Ft[0][ci].x, Ft[0][ci + 1].x, Ft[1][ci].x, Ft[1][ci + 1].x, av[k], b);
            const float uy = lerp4(Ft[0][ci].y, Ft[0][ci + 1].y, Ft[1][ci].y, Ft[1][ci + 1].y, av[k], b);
            sx[k] = round2i(ux * 2.0f);
            sy[k] = round2i(uy * 2.0f);
            // every tap inside the frame: 0 <= q and ((q + 4) >> 1) <= dim - 1  <=>  (unsigned)q <= 2*dim - 5;
            // the range test on the rounded flow keeps saturated conversions from wrapping back into range
            const int qx = X0 + k + sx[k] - 2, qy = Y + sy[k] - 2;
            safe = safe && (uint32_t)(sx[k] + (1 << 20)) < (2u << 20) && (uint32_t)(sy[k] + (1 << 20)) < (2u << 20) &&
                   (uint32_t)qx <= xmax && (uint32_t)qy <= ymax;
        }
#undef av
#undef b
        if (safe) {
            safeBits |= 1u << n;
            // The 4 x 13 tap weights depend on the kernel parameters only, so the compiler would keep
            // them in registers for the second frame (214 VGPRs, 2 waves per SIMD).  Recomputing them per
            // frame (+10 % instructions) fits 168 VGPRs = 3 waves per SIMD and is 9 % faster (0.724 ->
            // 0.660 ms per pair, profiles/r01_ab_pair_nohoist.txt): occupancy beats instruction count here.
#pragma unroll
            for (int k = 0; k < 4; k++) asm volatile("" : "+v"(kxa[k]), "+v"(kya[k]), "+v"(kza[k]));
            // certainty: LDS row of the mask row that tap row jt reads = ((ly + jt - 2) >> 2) + 1; column lx + cell
            auto mval = [&](int jt, int cell, int e4) {
                const int mr = ((ly + jt - 2) >> 2) + 1;
                return *(const float*)((const char*)&sM[n][mr][lx + cell] + e4);
            };
            strip_pixel<0, CFA, true>(X0 + 0, Y, sx[0], sy[0], kxa[0], kya[0], kza[0], raw, dimX, mval, lv, accP, accW);
            strip_pixel<1, CFA, true>(X0 + 1, Y, sx[1], sy[1], kxa[1], kya[1], kza[1], raw, dimX, mval, lv, accP, accW);
            strip_pixel<2, CFA, true>(X0 + 2, Y, sx[2], sy[2], kxa[2], kya[2], kza[2], raw, dimX, mval, lv, accP, accW);
            strip_pixel<3, CFA, true>(X0 + 3, Y, sx[3], sy[3], kxa[3], kya[3], kza[3], raw, dimX, mval, lv, accP, accW);
        }
    }
    }
    // staged plane pl += this lane's 4 pixels x 3 channels
    auto add_plane = [&](int pl, const float* acc) {
        float4* my = (float4*)((float*)&sAcc[pl][ly][0] + lx * 12);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            float4 a = my[j];
            a.x += acc[4 * j + 0]; a.y += acc[4 * j + 1]; a.z += acc[4 * j + 2]; a.w += acc[4 * j + 3];
            my[j] = a;
        }
    };
    // border / wild-flow / non-PSD strips of a frame: the straight per-pixel arithmetic on the staged values (after the
    // sums have left the registers: it needs most of them).  With one staged plane-set at a time it runs once per set
    // and keeps that set's result only -- these strips are a few per frame.
    auto slow_frames = [&](auto doP, auto doW, int plP, int plW) {
        constexpr bool DO_P = decltype(doP)::value, DO_W = decltype(doW)::value;
        float* myP = (float*)&sAcc[plP][ly][0] + lx * 12;
        float* myW = (float*)&sAcc[plW][ly][0] + lx * 12;
#pragma unroll 1
        for (int n = 0; n < NF; n++) {
            if (!((safeBits >> n) & 1u) && stripLive) {
                // the frame's arguments picked with scalar selects: the argument struct stays in SGPRs
                TileFrame F = fr.f[0];
#pragma unroll
                for (int m = 1; m < NF; m++)
                    if (n == m) F = fr.f[m];
#pragma unroll 1
                for (int k = 0; k < 4; k++) {
                    const int X = X0 + k;
                    if (X >= 1 && X < hrW - 1) {
                        pix3 px = {0.0f, 0.0f, 0.0f}, tw = {0.0f, 0.0f, 0.0f};
                        if (DO_P) px = {myP[3 * k], myP[3 * k + 1], myP[3 * k + 2]};
                        if (DO_W) tw = {myW[3 * k], myW[3 * k + 1], myW[3 * k + 2]};
                        accumulate_pixel_core<GEOM_FULL, true>(X, Y, F.raw, F.mask, kernelParam, F.shifts, glv, dimX, dimY, 2,
                                                               strideMask, cfaPacked, px, tw);
                        if (DO_P) { myP[3 * k] = px.x; myP[3 * k + 1] = px.y; myP[3 * k + 2] = px.z; }
                        if (DO_W) { myW[3 * k] = tw.x; myW[3 * k + 1] = tw.y; myW[3 * k + 2] = tw.z; }
                    }
                }
            }
        }
    };
    // staged plane pl back to g in memory order: contiguous 1 KiB per store instruction
    auto store_plane = [&](int pl, char* g) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const size_t off = (size_t)(j * 64 + lx) * 16;
            // the 16-pixel side margins of the row belong to k_accumulateMargin, which may run concurrently on the margin
            // stream: their (unchanged) chunks are not written back.  A fresh launch defines them (zero) instead.
            const size_t gb = segByte + off;
            const bool sideMargin = gb < (size_t)STRIP_MARGIN * 12 || gb + 16 > rowBytes - (size_t)STRIP_MARGIN * 12;
            if (gb + 16 <= rowBytes && (fresh || !sideMargin)) *(float4*)(g + off) = sAcc[pl][ly][j * 64 + lx];
        }
    };
    using Yes = std::true_type;
    using No = std::false_type;
    // staged accumulators must have landed before anyone reads them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    constexpr bool valueSumsStaged = TILE_PIXEL_MAJOR && (NF > 1 || FR != 4);  // the pixel-major loop has added them already
    if constexpr (PL == 2) {
        if (!valueSumsStaged) add_plane(0, accP);
        add_plane(1, accW);
        if (safeBits != (1u << NF) - 1u) slow_frames(Yes{}, Yes{}, 0, 1);
        store_plane(0, gP);
        store_plane(1, gW);
    } else {
        if (!valueSumsStaged) add_plane(0, accP);
        if (safeBits != (1u << NF) - 1u) slow_frames(Yes{}, No{}, 0, 0);
        store_plane(0, gP);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // plane 0 has been read out before the copy overwrites it
        stage_plane(gW, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        add_plane(0, accW);
        if (safeBits != (1u << NF) - 1u) slow_frames(No{}, Yes{}, 0, 0);
        store_plane(0, gW);
    }
}

// ---- x4 ------------------------------------------------------------------------------------------
// The same pixel at scale 4.  The 5x5 HR taps of a pixel fall on 2x2 raw sites: tap column `it` lands
// on site column (ph + it) >> 2 with ph = (X + sx - 2) & 3, i.e. on column 1 iff ph + it >= 4 -- a lane
// mask per tap column (none for it = 0, all for it = 4), applied to the weights as in strip_pixel.
// Each site is its own CFA-position class (relative to the parity of (x0, y0)).  K8 = X & 7 is the
// pixel's position in its certainty cell (8 HR pixels at scale 4), K8 & 3 its position in the strip.
// strip_pixel4_w: the tap weights given (frame independent, see strip_pixel_w); RawF: void rawf(x0, y0, float (&s)[2][2])
template <int K8, int CFA, bool PARITY = false, typename RawF, typename MaskF>
__device__ __forceinline__ void strip_pixel4_w(int X, int Y, int sx, int sy, const float (&w)[13], RawF rawf, MaskF mval,
                                                const StripLevels& lv, float* accP, float* accW)
{
    const int qx = X + sx - 2, qy = Y + sy - 2;
    const int x0 = qx >> 2, y0 = qy >> 2;
    const uint32_t bx0 = 0u - (uint32_t)(qx & 1), bx1 = 0u - (uint32_t)((qx >> 1) & 1);
    const uint32_t by0 = 0u - (uint32_t)(qy & 1), by1 = 0u - (uint32_t)((qy >> 1) & 1);
    // mx[it]: tap column it lands on site column 1 (ph + it >= 4); same for rows
    const uint32_t mx[5] = {0u, bx0 & bx1, bx1, bx0 | bx1, ~0u};
    const uint32_t my[5] = {0u, by0 & by1, by1, by0 | by1, ~0u};
    const uint32_t mP = 0u - (uint32_t)(x0 & 1), mQ = 0u - (uint32_t)(y0 & 1);
    auto andm = [](uint32_t m, float a) { return __uint_as_float(m & __float_as_uint(a)); };

    float s[2][2];
    rawf(x0, y0, s);
    auto W_ = [&](int jt, int it) { const int n = jt * 5 + it; return w[n <= 12 ? n : 24 - n]; };

    // certainty cell column of tap column it, relative to the cell left of the strip's own: 0..2
    auto cellOf = [](int it) { return (K8 + it - 2 + 8) >> 3; };
    constexpr int cellLo = (K8 - 2 + 8) >> 3;
    constexpr bool twoCells = ((K8 + 4 - 2 + 8) >> 3) != cellLo;

    float C[5][2];  // per tap row: w * certainty summed per site column
#pragma unroll
    for (int jt = 0; jt < 5; jt++) {
        const uint32_t mya = mQ ^ my[jt];  // absolute y parity of the site row this tap row lands on
        float Ee[2], Eo[2];
        if constexpr (PARITY) {
            // CFA position (y parity << 1 | x parity) of site column 0 on this tap row; tap row jt sits on site row
            // ((qy & 3) + jt) >> 2, whose parity is Q flipped when that is 1
            const int ph = qy & 3;
            const int flip = jt == 0 ? 0 : (jt == 4 ? 1 : (jt == 1 ? (ph == 3) : (jt == 2 ? (ph >> 1) : (ph != 0))));
            const int e = (((y0 & 1) ^ flip) << 1) | (x0 & 1);
#pragma unroll
            for (int c = 0; c < (twoCells ? 2 : 1); c++) {
                Ee[c] = mval(jt, cellLo + c, e);
                Eo[c] = mval(jt, cellLo + c, e ^ 1);
            }
        } else {
#pragma unroll
            for (int c = 0; c < (twoCells ? 2 : 1); c++) {
                float m[3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++) m[ch] = mval(jt, cellLo + c, ch);
                resolve_certainty<CFA>(mya, mP, m, Ee[c], Eo[c]);
            }
        }
        // weights per (site column, cell), then one multiply with the certainty of that pair
        float lo[2] = {0.0f, 0.0f}, hi[2] = {0.0f, 0.0f};
        bool loUsed[2] = {false, false}, hiUsed[2] = {false, false};
#pragma unroll
        for (int it = 0; it < 5; it++) {
            const int c = cellOf(it) - cellLo;
            const float wt = W_(jt, it);
            if (it < 4) {
                const float v = it == 0 ? wt : andm(~mx[it], wt);
                lo[c] = loUsed[c] ? lo[c] + v : v;
                loUsed[c] = true;
            }
            if (it > 0) {
                const float v = it == 4 ? wt : andm(mx[it], wt);
                hi[c] = hiUsed[c] ? hi[c] + v : v;
                hiUsed[c] = true;
            }
        }
        float c0 = loUsed[0] ? lo[0] * Ee[0] : 0.0f;
        if (twoCells && loUsed[1]) c0 = loUsed[0] ? __builtin_fmaf(lo[1], Ee[1], c0) : lo[1] * Ee[1];
        float c1 = hiUsed[0] ? hi[0] * Eo[0] : 0.0f;
        if (twoCells && hiUsed[1]) c1 = hiUsed[0] ? __builtin_fmaf(hi[1], Eo[1], c1) : hi[1] * Eo[1];
        C[jt][0] = c0;
        C[jt][1] = c1;
    }
    float S[2][2], W[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float o0 = ((C[0][i] + andm(~my[1], C[1][i])) + andm(~my[2], C[2][i])) + andm(~my[3], C[3][i]);
        const float o1 = ((andm(my[1], C[1][i]) + andm(my[2], C[2][i])) + andm(my[3], C[3][i])) + C[4][i];
        W[0][i] = o0;
        W[1][i] = o1;
        S[0][i] = s[0][i] * o0;
        S[1][i] = s[1][i] * o1;
    }
    classes_to_channels<(K8 & 3), CFA>(S, W, mP, mQ, lv, accP, accW);
}

template <int K8, int CFA, bool PARITY = false, typename MaskF>
__device__ __forceinline__ void strip_pixel4(int X, int Y, int sx, int sy, float kx, float ky, float kz,
                                              const uint16_t* __restrict__ raw, int dimX, MaskF mval,
                                              const StripLevels& lv, float* accP, float* accW)
{
    float w[13];
    tap_weights13(kx, ky, kz, w);
    auto rawf = [&](int x0, int y0, float (&s)[2][2]) {
        const uint16_t* r = raw + (size_t)y0 * dimX + x0;
        s[0][0] = (float)r[0];
        s[0][1] = (float)r[1];
        s[1][0] = (float)r[dimX];
        s[1][1] = (float)r[dimX + 1];
    };
    strip_pixel4_w<K8, CFA, PARITY>(X, Y, sx, sy, w, rawf, mval, lv, accP, accW);
}

// x4 LDS tile kernel (fields at HR/8: the Bayer pipeline at scale 4).  One 64x4 workgroup covers a
// 512 x 2 HR tile: wave (r, h) = (threadIdx.y >> 1, threadIdx.y & 1) owns row r and the strips whose
// position in the 8-pixel certainty cell is h (lane lx -> strip 2*lx + h), so K8 is a compile-time
// constant per wave.  Staging as in the x2 kernel (64 + 2 field texels x 3 rows, column/row fractions,
// accumulator rows through LDS-DMA); the two waves of a row share its staged segment, hence the
// workgroup barriers around the LDS update.
// waves per SIMD the x4 kernel's register budget is sized for with two frames per launch (one frame: 113 VGPRs = 4)
#ifndef TILE4_WAVES2
#define TILE4_WAVES2 3
#endif
// (three and four frames per launch: four workgroups per CU by the layout of TILE_LDS_ALIAS, see k_accumulate2xTile)
template <int CFA, int NF>
__global__ void __launch_bounds__(256, (NF) == 1 ? 4 : ((NF) > 2 && TILE_LDS_ALIAS_ON) ? 4 : TILE4_WAVES2)
    k_accumulate4xTile(TileFrames<NF> fr, pix3* __restrict__ imgOut, pix3* __restrict__ totalWeights, mfsr_tex2d kernelParam,
                       Levels3 glv, StripLevels lv, int dimX, int dimY, int strideOut, int strideMask, int cfaPacked,
                       int tilesX, int fresh, int tileY0)
{
    const int bIdYrel = (int)blockIdx.x / tilesX, bIdX = (int)blockIdx.x - bIdYrel * tilesX;
    const int bIdY = bIdYrel + tileY0;  // tileY0: first tile row of this launch's HR row window
    const int hrW = 4 * dimX, hrH = 4 * dimY;
    const int Y0 = 2 * bIdY;
    if (Y0 < STRIP_MARGIN || Y0 >= hrH - STRIP_MARGIN) return;  // whole workgroup (margin rows)
    constexpr bool ALIAS = TILE_LDS_ALIAS_ON && NF > 2;
    __shared__ __attribute__((aligned(16))) float4 sAcc[2][2][384];  // [plane-set][row][6 KiB row segment]
    __shared__ float4 sKown[ALIAS ? 1 : 3][ALIAS ? 1 : TILE_COLS];
    __shared__ float2 sFown[ALIAS ? 1 : NF][ALIAS ? 1 : 3][ALIAS ? 1 : TILE_COLS];
    static_assert(!ALIAS || sizeof(float4) * 3 * TILE_COLS + sizeof(float2) * NF * 3 * TILE_COLS <= sizeof(float4) * 2 * 384, "field texels fit a plane-set");
    // ALIAS: the field texels live in the weight-sum plane-set's staging area until every wave has read them (TILE_LDS_ALIAS)
    float4(*sK)[TILE_COLS] = ALIAS ? (float4(*)[TILE_COLS]) & sAcc[1][0][0] : (float4(*)[TILE_COLS]) & sKown[0][0];  // .w = 1 if PSD and finite
    float2(*sF)[3][TILE_COLS] = ALIAS ? (float2(*)[3][TILE_COLS])((char*)&sAcc[1][0][0] + sizeof(float4) * 3 * TILE_COLS)
                                      : (float2(*)[3][TILE_COLS]) & sFown[0][0][0];
    __shared__ float4 sM[NF][3][TILE_COLS];
    __shared__ __attribute__((aligned(16))) float sColA[512];
    __shared__ float sRowB[2];
    const int lx = threadIdx.x, ly = threadIdx.y;
    const int r = ly >> 1, h = ly & 1;
    const int X0 = bIdX * 512 + 8 * lx + 4 * h;
    const int Y = Y0 + r;
    const int fw = kernelParam.width, fh = kernelParam.height;  // == hrW/8, hrH/8 (checked on the host)
    const int mw = dimX / 2, mh = dimY / 2;
    const int band = Y0 >> 3;  // field / certainty row of the tile
    {
        const int t = ly * 64 + lx;
        if (t < 3 * TILE_COLS) {
            const int rr = t / TILE_COLS, c = t - rr * TILE_COLS;
            const int gy = band - 1 + rr, gx = bIdX * 64 - 1 + c;
            const int fy = clampi(gy, 0, fh - 1), fx = clampi(gx, 0, fw - 1);
            float4 k = row_ptr((const float4*)kernelParam.ptr, kernelParam.pitch, fy)[fx];
            k.w = psd_ok(k.x, k.y, k.z) ? 1.0f : 0.0f;
            sK[rr][c] = k;
#pragma unroll
            for (int n = 0; n < NF; n++) {
                sF[n][rr][c] = row_ptr((const float2*)fr.f[n].shifts.ptr, fr.f[n].shifts.pitch, fy)[fx];
                const float4 m = row_ptr(fr.f[n].mask, strideMask, clampi(gy, 0, mh - 1))[clampi(gx, 0, mw - 1)];
                const float mc[3] = {sane(m.x), sane(m.y), sane(m.z)};  // by CFA position, see k_accumulate2xTile
                sM[n][rr][c] = make_float4(mc[Cfa<CFA>::col(0, 0)], mc[Cfa<CFA>::col(0, 1)], mc[Cfa<CFA>::col(1, 0)],
                                          mc[Cfa<CFA>::col(1, 1)]);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            // column of the tile: the float path of tex_coord, texel column predicted and verified
            const int cl = t + 256 * u, X = bIdX * 512 + cl;
            const float posX = ((float)X + 0.5f) / (float)hrW;
            float xB = posX * (float)fw - 0.5f;
            if (!finitef(xB)) xB = 0.0f;
            const float fxf = floorf(xB);
            const int c8 = X >> 3, ci = (X & 7) < 4 ? 0 : 1;
            const bool ok = (f2i(fxf) == c8 - 1 + ci) && (c8 + ci <= fw - 1);
            sColA[cl] = ok ? xB - fxf : -1.0f;
        }
        if (t < 2) {
            const int Yr = Y0 + t;
            const float posY = ((float)Yr + 0.5f) / (float)hrH;
            float yB = posY * (float)fh - 0.5f;
            if (!finitef(yB)) yB = 0.0f;
            const float fyf = floorf(yB);
            const int frr = (Yr & 7) < 4 ? 0 : 1;
            const bool ok = (f2i(fyf) == band - 1 + frr) && f2i(fyf) >= 0 && f2i(fyf) + 1 <= fh - 1;
            sRowB[t] = ok ? yB - fyf : -1.0f;
        }
    }
    // accumulator rows of the tile -> LDS (wave (r, h) brings half h of row r), asynchronously
    const size_t rowBytes = (size_t)hrW * 12;
    const size_t segByte = (size_t)bIdX * 6144 + (size_t)h * 3072;
    char* gP = (char*)imgOut + (size_t)Y * strideOut + segByte;
    char* gW = (char*)totalWeights + (size_t)Y * strideOut + segByte;
    auto stage_plane = [&](char* g, int pl) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            if (fresh) {
                sAcc[pl][r][h * 192 + j * 64 + lx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            } else {
                const size_t off = (size_t)(j * 64 + lx) * 16;
                if (segByte + off + 16 <= rowBytes)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + off),
                                                     (__attribute__((address_space(3))) void*)&sAcc[pl][r][h * 192 + j * 64], 16, 0, 0);
            }
        }
    };
    stage_plane(gP, 0);
    if (!ALIAS) stage_plane(gW, 1);  // (ALIAS: after the last read of the field texels, below)
    __syncthreads();
    const bool stripLive = X0 >= STRIP_MARGIN && X0 < hrW - STRIP_MARGIN;

    const int fr_ = (Y0 & 7) < 4 ? 0 : 1;
    const float b = sRowB[r];
    const float4 av4 = ((const float4*)sColA)[2 * lx + h];
    const float av[4] = {av4.x, av4.y, av4.z, av4.w};
    const bool geomOk = stripLive && b >= 0.0f && fminf(fminf(av[0], av[1]), fminf(av[2], av[3])) >= 0.0f;
    const int cb = lx + h;  // LDS column of the left texel of this strip's texel pair

    float kxa[4], kya[4], kza[4];
    bool kOk;
    {
        float4 Kt[2][2];
#pragma unroll
        for (int r2 = 0; r2 < 2; r2++)
#pragma unroll
            for (int c = 0; c < 2; c++) Kt[r2][c] = sK[fr_ + r2][cb + c];
        kOk = (Kt[0][0].w * Kt[0][1].w) * (Kt[1][0].w * Kt[1][1].w) > 0.0f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float w00 = (1.0f - av[k]) * (1.0f - b), w10 = av[k] * (1.0f - b), w01 = (1.0f - av[k]) * b, w11 = av[k] * b;
            auto mix = [&](float t00, float t10, float t01, float t11) {
                return __builtin_fmaf(w11, t11, __builtin_fmaf(w01, t01, __builtin_fmaf(w10, t10, w00 * t00)));
            };
            kxa[k] = mix(Kt[0][0].x, Kt[0][1].x, Kt[1][0].x, Kt[1][1].x);
            kya[k] = mix(Kt[0][0].y, Kt[0][1].y, Kt[1][0].y, Kt[1][1].y);
            kza[k] = mix(Kt[0][0].z, Kt[0][1].z, Kt[1][0].z, Kt[1][1].z);
        }
    }

    float accP[12], accW[12];
#pragma unroll
    for (int i = 0; i < 12; i++) accP[i] = accW[i] = 0.0f;
    const uint32_t xmax = (uint32_t)(4 * dimX - 5), ymax = (uint32_t)(4 * dimY - 5);
    const int yq = Y & 7;
    uint32_t safeBits = 0;
    static_assert(NF <= 2 || TILE_PIXEL_MAJOR, "three and four frames per launch exist in pixel-major order only");
#if TILE_PIXEL_MAJOR
    if constexpr (NF > 1) {
        // pixel-major, as k_accumulate2xTile: per frame the whole-pixel flow of the strip's pixels and the admission, then
        // per pixel the tap weights once and frame after frame
        uint32_t sxy[NF][ALIAS ? 2 : 4];  // (ALIAS: 8:8 bits per pixel, [0] = the four sx, [1] = the four sy, relative to bsx / bsy)
        int bsx[NF], bsy[NF];             // rounded flow of the tile's centre texel (per-frame scalars), see k_accumulate2xTile
#pragma unroll
        for (int n = 0; n < NF; n++) {
            bsx[n] = bsy[n] = 0;
            if constexpr (ALIAS) {
                sxy[n][0] = sxy[n][1] = 0u;
                const float2 cf = sF[n][1][TILE_COLS / 2];
                bsx[n] = __builtin_amdgcn_readfirstlane(min(max(round2i(cf.x * 4.0f), -32000), 32000));
                bsy[n] = __builtin_amdgcn_readfirstlane(min(max(round2i(cf.y * 4.0f), -32000), 32000));
            }
            float2 Ft[2][2];
#pragma unroll
            for (int r2 = 0; r2 < 2; r2++)
#pragma unroll
                for (int c = 0; c < 2; c++) Ft[r2][c] = sF[n][fr_ + r2][cb + c];
            bool safe = geomOk && kOk;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float ux = lerp4(Ft[0][0].x, Ft[0][1].x, Ft[1][0].x, Ft[1][1].x, av[k], b);
                const float uy = lerp4(Ft[0][0].y, Ft[0][1].y, Ft[1][0].y, Ft[1][1].y, av[k], b);
                const int sx = round2i(ux * 4.0f), sy = round2i(uy * 4.0f);
                const int qx = X0 + k + sx - 2, qy = Y + sy - 2;
                safe = safe && (uint32_t)(sx + (1 << 15)) < (2u << 15) && (uint32_t)(sy + (1 << 15)) < (2u << 15) &&
                       (uint32_t)qx <= xmax && (uint32_t)qy <= ymax;
                if constexpr (ALIAS) {
                    const int dx = sx - bsx[n], dy = sy - bsy[n];
                    safe = safe && (uint32_t)(dx + 128) < 256u && (uint32_t)(dy + 128) < 256u;
                    sxy[n][0] |= ((uint32_t)dx & 0xffu) << (8 * k);
                    sxy[n][1] |= ((uint32_t)dy & 0xffu) << (8 * k);
                } else {
                    sxy[n][k] = ((uint32_t)sx & 0xffffu) | ((uint32_t)sy << 16);
                }
            }
            if (safe) safeBits |= 1u << n;
        }
        if constexpr (ALIAS) {
            __syncthreads();  // every wave is past its reads of sK / sF
            stage_plane(gW, 1);
        }
        const float* mrow[5];
#pragma unroll
        for (int jt = 0; jt < 5; jt++) mrow[jt] = (const float*)&sM[0][(yq + jt - 2 + 8) >> 3][lx];
        const char* rawRow[NF][2];
#pragma unroll
        for (int n = 0; n < NF; n++)
#pragma unroll
            for (int j = 0; j < 2; j++) rawRow[n][j] = (const char*)(fr.f[n].raw + (size_t)j * dimX);
        auto pixel = [&](auto k8c) {
            constexpr int K8 = decltype(k8c)::value, k = K8 & 3;
            if (safeBits == 0) return;
            float w[13];
            tap_weights13(kxa[k], kya[k], kza[k], w);
#pragma unroll
            for (int n = 0; n < NF; n++) {
                if ((safeBits >> n) & 1u) {
                    auto rawf = [&](int x0, int y0, float (&sv)[2][2]) {
                        const uint32_t boff = (uint32_t)(y0 * dimX + x0) * 2u;  // admitted strips: inside the frame
#pragma unroll
                        for (int j = 0; j < 2; j++) {
                            const char* rb = rawRow[n][j] + (size_t)boff;
                            sv[j][0] = (float)*(const uint16_t*)rb;
                            sv[j][1] = (float)*(const uint16_t*)(rb + 2);
                        }
                    };
                    auto mval = [&](int jt, int cell, int e) { return mrow[jt][n * (3 * TILE_COLS * 4) + cell * 4 + e]; };
                    const int sx = ALIAS ? bsx[n] + (int)(int8_t)(sxy[n][0] >> (8 * k)) : (int)(int16_t)(sxy[n][ALIAS ? 0 : k] & 0xffffu);
                    const int sy = ALIAS ? bsy[n] + (int)(int8_t)(sxy[n][1] >> (8 * k)) : (int)sxy[n][ALIAS ? 0 : k] >> 16;
                    strip_pixel4_w<K8, CFA, true>(X0 + k, Y, sx, sy, w, rawf, mval, lv, accP, accW);
                }
            }
        };
        if (h == 0) {
            pixel(std::integral_constant<int, 0>{});
            pixel(std::integral_constant<int, 1>{});
            pixel(std::integral_constant<int, 2>{});
            pixel(std::integral_constant<int, 3>{});
        } else {
            pixel(std::integral_constant<int, 4>{});
            pixel(std::integral_constant<int, 5>{});
            pixel(std::integral_constant<int, 6>{});
            pixel(std::integral_constant<int, 7>{});
        }
    } else
#endif
#pragma unroll 1
    for (int n = 0; n < NF; n++) {
        const uint16_t* raw = (NF > 1 && n) ? fr.f[NF - 1].raw : fr.f[0].raw;
        float avn[4] = {av[0], av[1], av[2], av[3]};
        float bn = b;
#pragma unroll
        for (int k = 0; k < 4; k++) asm volatile("" : "+v"(avn[k]));
        asm volatile("" : "+v"(bn));
        float2 Ft[2][2];
#pragma unroll
        for (int r2 = 0; r2 < 2; r2++)
#pragma unroll
            for (int c = 0; c < 2; c++) Ft[r2][c] = sF[n][fr_ + r2][cb + c];
        int sx[4], sy[4];
        bool safe = geomOk && kOk;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float ux = lerp4(Ft[0][0].x, Ft[0][1].x, Ft[1][0].x, Ft[1][1].x, avn[k], bn);
            const float uy = lerp4(Ft[0][0].y, Ft[0][1].y, Ft[1][0].y, Ft[1][1].y, avn[k], bn);
            sx[k] = round2i(ux * 4.0f);
            sy[k] = round2i(uy * 4.0f);
            const int qx = X0 + k + sx[k] - 2, qy = Y + sy[k] - 2;
            safe = safe && (uint32_t)(sx[k] + (1 << 20)) < (2u << 20) && (uint32_t)(sy[k] + (1 << 20)) < (2u << 20) &&
                   (uint32_t)qx <= xmax && (uint32_t)qy <= ymax;
        }
        if (safe) {
            safeBits |= 1u << n;
#pragma unroll
            for (int k = 0; k < 4; k++) asm volatile("" : "+v"(kxa[k]), "+v"(kya[k]), "+v"(kza[k]));
            // certainty: LDS row of the mask row that tap row jt reads, column lx + cell
            auto mval = [&](int jt, int cell, int e) {
                const int mr = ((yq + jt - 2 + 8) >> 3);
                const float* p = (const float*)&sM[n][mr][lx + cell];
                return p[e];
            };
            if (h == 0) {
                strip_pixel4<0, CFA, true>(X0 + 0, Y, sx[0], sy[0], kxa[0], kya[0], kza[0], raw, dimX, mval, lv, accP, accW);
                strip_pixel4<1, CFA, true>(X0 + 1, Y, sx[1], sy[1], kxa[1], kya[1], kza[1], raw, dimX, mval, lv, accP, accW);
                strip_pixel4<2, CFA, true>(X0 + 2, Y, sx[2], sy[2], kxa[2], kya[2], kza[2], raw, dimX, mval, lv, accP, accW);
                strip_pixel4<3, CFA, true>(X0 + 3, Y, sx[3], sy[3], kxa[3], kya[3], kza[3], raw, dimX, mval, lv, accP, accW);
            } else {
                strip_pixel4<4, CFA, true>(X0 + 0, Y, sx[0], sy[0], kxa[0], kya[0], kza[0], raw, dimX, mval, lv, accP, accW);
                strip_pixel4<5, CFA, true>(X0 + 1, Y, sx[1], sy[1], kxa[1], kya[1], kza[1], raw, dimX, mval, lv, accP, accW);
                strip_pixel4<6, CFA, true>(X0 + 2, Y, sx[2], sy[2], kxa[2], kya[2], kza[2], raw, dimX, mval, lv, accP, accW);
                strip_pixel4<7, CFA, true>(X0 + 3, Y, sx[3], sy[3], kxa[3], kya[3], kza[3], raw, dimX, mval, lv, accP, accW);
            }
        }
    }
    // every wave's part of the staged rows must have landed before any lane updates them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* myP = (float*)&sAcc[0][r][0] + (2 * lx + h) * 12;
    float* myW = (float*)&sAcc[1][r][0] + (2 * lx + h) * 12;
    if (safeBits) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            float4 a = ((float4*)myP)[j], c = ((float4*)myW)[j];
            a.x += accP[4 * j + 0]; a.y += accP[4 * j + 1]; a.z += accP[4 * j + 2]; a.w += accP[4 * j + 3];
            c.x += accW[4 * j + 0]; c.y += accW[4 * j + 1]; c.z += accW[4 * j + 2]; c.w += accW[4 * j + 3];
            ((float4*)myP)[j] = a;
            ((float4*)myW)[j] = c;
        }
    }
#pragma unroll 1
    for (int n = 0; n < NF; n++) {
        TileFrame F = fr.f[0];  // picked with scalar selects: the argument struct stays in SGPRs
#pragma unroll
        for (int m = 1; m < NF; m++)
            if (n == m) F = fr.f[m];
        if (!((safeBits >> n) & 1u) && stripLive) {
#pragma unroll 1
            for (int k = 0; k < 4; k++) {
                const int X = X0 + k;
                if (X >= 1 && X < hrW - 1) {
                    pix3 px = {myP[3 * k], myP[3 * k + 1], myP[3 * k + 2]};
                    pix3 tw = {myW[3 * k], myW[3 * k + 1], myW[3 * k + 2]};
                    accumulate_pixel_core<GEOM_FULL, true>(X, Y, F.raw, F.mask, kernelParam, F.shifts, glv, dimX, dimY, 4,
                                                           strideMask, cfaPacked, px, tw);
                    myP[3 * k] = px.x; myP[3 * k + 1] = px.y; myP[3 * k + 2] = px.z;
                    myW[3 * k] = tw.x; myW[3 * k + 1] = tw.y; myW[3 * k + 2] = tw.z;
                }
            }
        }
    }
    __syncthreads();
    // write the half-row segments back in memory order
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const size_t off = (size_t)(j * 64 + lx) * 16;
        const size_t g = segByte + off;  // side-margin chunks: see k_accumulate2xTile
        const bool sideMargin = g < (size_t)STRIP_MARGIN * 12 || g + 16 > rowBytes - (size_t)STRIP_MARGIN * 12;
        if (g + 16 <= rowBytes && (fresh || !sideMargin)) {
            *(float4*)(gP + off) = sAcc[0][r][h * 192 + j * 64 + lx];
            *(float4*)(gW + off) = sAcc[1][r][h * 192 + j * 64 + lx];
        }
    }
}

constexpr int pack_cfa(int c00, int c01, int c10, int c11) { return c00 | (c01 << 2) | (c10 << 4) | (c11 << 6); }

int g_strip_xcd_remap = 1;  // MFSR_XCD_REMAP=0: launch order (A/B); see k_accumulate2xTile
int g_strip_use_tile = 1;  // MFSR_STRIP_TILE: 0 register-only strip kernel, 1 LDS tile kernel (fields at HR/4)

// true when the LDS tile kernel serves this geometry (fields at HR/4, whole tiles)
bool tile_kernel_ok(mfsr_tex2d kp, mfsr_tex2d sh, int dimX, int dimY)
{
    const int hrW = 2 * dimX, hrH = 2 * dimY;
    return kp.width == sh.width && kp.height == sh.height && kp.width * 4 == hrW && kp.height * 4 == hrH && kp.width >= 4 &&
           (dimX % 4) == 0 && (dimY % 4) == 0 && g_strip_use_tile == 1;
}

// the same for fields at HR/2 (the monochrome pipeline: tracking at the raw resolution)
bool tile_kernel_ok_fr2(mfsr_tex2d kp, mfsr_tex2d sh, int dimX, int dimY)
{
    const int hrW = 2 * dimX, hrH = 2 * dimY;
    return kp.width == sh.width && kp.height == sh.height && kp.width * 2 == hrW && kp.height * 2 == hrH && kp.width >= 4 &&
           (dimX % 4) == 0 && (dimY % 4) == 0 && g_strip_use_tile == 1;
}

template <int CFA, int NF, int FR = 4>
void launch_tile(dim3 grid, dim3 block, hipStream_t st, const TileFrames<NF>& fr, pix3* imgOut, pix3* tw, mfsr_tex2d kp,
                 Levels3 glv, StripLevels lv, int dimX, int dimY, int strideOut, int strideMask, int cfaPacked, int fresh, int tileY0)
{
    const int tilesX = (int)grid.x, tilesY = (int)grid.y;
    const int tilesPerXcd = g_strip_xcd_remap ? mfsr_cdiv(tilesX * tilesY, 8) : 0;
    // occupancy probe (A/B only): unused dynamic LDS so that fewer workgroups fit a CU
    static const int ldsPad = [] {
        const char* e = getenv("MFSR_TILE_LDS_PAD");
        return e ? atoi(e) : 0;
    }();
    hipLaunchKernelGGL((k_accumulate2xTile<CFA, NF, FR>), dim3(tilesPerXcd ? 8 * tilesPerXcd : tilesX * tilesY), block, ldsPad, st, fr,
                       imgOut, tw, kp, glv, lv, dimX, dimY, strideOut, strideMask, cfaPacked, tilesX, tilesY, tilesPerXcd, fresh, tileY0);
}

template <int CFA, int NF>
void launch_strip_regs(dim3 grid, dim3 block, hipStream_t st, const TileFrames<NF>& fr, pix3* imgOut, pix3* tw, mfsr_tex2d kp,
                       Levels3 glv, StripLevels lv, int dimX, int dimY, int strideOut, int strideMask, int cfaPacked, int rowBlock0)
{
    const int hrW = 2 * dimX, hrH = 2 * dimY;
    bool same = true;
    for (int n = 0; n < NF; n++) same = same && kp.width == fr.f[n].shifts.width && kp.height == fr.f[n].shifts.height;
    if (same && kp.width * 4 == hrW && kp.height * 4 == hrH && kp.width >= 4)
        hipLaunchKernelGGL((k_accumulate2xStrip<CFA, 4, NF>), grid, block, 0, st, fr, imgOut, tw, kp, glv, lv, dimX, dimY,
                           strideOut, strideMask, cfaPacked, rowBlock0);
    else if (same && kp.width * 2 == hrW && kp.height * 2 == hrH && kp.width >= 4)
        hipLaunchKernelGGL((k_accumulate2xStrip<CFA, 2, NF>), grid, block, 0, st, fr, imgOut, tw, kp, glv, lv, dimX, dimY,
                           strideOut, strideMask, cfaPacked, rowBlock0);
    else
        hipLaunchKernelGGL((k_accumulate2xStrip<CFA, 0, NF>), grid, block, 0, st, fr, imgOut, tw, kp, glv, lv, dimX, dimY,
                           strideOut, strideMask, cfaPacked, rowBlock0);
}

// fresh accumulators: the tile kernels write every row of their window outside the top and bottom margin bands
// (rows [0, M) and [hrH - M, hrH)); zero the part of those bands that lies inside the window
int zero_margin_bands(mfsr_float3* imgOut, mfsr_float3* totalWeights, int hrH, int strideOut, int rowBegin, int rowEnd, hipStream_t st)
{
    const int bands[2][2] = {{0, STRIP_MARGIN}, {hrH - STRIP_MARGIN, hrH}};
    for (int i = 0; i < 2; i++) {
        const int r0 = bands[i][0] > rowBegin ? bands[i][0] : rowBegin, r1 = bands[i][1] < rowEnd ? bands[i][1] : rowEnd;
        if (r1 <= r0) continue;
        for (int p2 = 0; p2 < 2; p2++) {
            char* base = p2 ? (char*)totalWeights : (char*)imgOut;
            if (hipMemsetAsync(base + (size_t)r0 * strideOut, 0, (size_t)(r1 - r0) * strideOut, st) != hipSuccess) return -1;
        }
    }
    return 0;
}

// The margin kernel (one thread per pixel of the 16-pixel frame margin, the straight per-pixel arithmetic) is small and
// latency-bound (5.6 K waves, 25 us per 4K frame) while the tile kernel saturates the VALUs: the two touch disjoint
// accumulator bytes (the tile kernel does not write the side-margin chunks back), so the margin launches of a call run
// on a side stream, forked after the tile launch's predecessors and joined before the call returns -- they fill the
// tile kernel's stalls instead of adding 2 x 25 us per pair.  The first launch of a burst ("fresh": the tile kernel
// defines the margins' zeroes) keeps the serial order.  MFSR_MARGIN_OVERLAP=0 restores it everywhere (A/B).
int g_margin_overlap = 0;  // measured: no gain (8.79 vs 8.88 ms per burst, profiles/r02_ab_margin_overlap.txt): off by default
struct MarginStream {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
MarginStream* margin_stream()
{
    static thread_local MarginStream ms[16];  // per host thread: contexts on different threads never share the fork/join events
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    MarginStream& m = ms[dev];
    if (m.device != dev) {
        if (hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&m.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&m.join, hipEventDisableTiming) != hipSuccess) return nullptr;
        m.device = dev;
    }
    return &m;
}

void read_env_once()
{
    static const bool env_read = [] {
        const char* e = getenv("MFSR_STRIP_TILE");
        if (e && e[0] >= '0' && e[0] <= '1') g_strip_use_tile = e[0] - '0';
        const char* x = getenv("MFSR_XCD_REMAP");
        if (x && (x[0] == '0' || x[0] == '1')) g_strip_xcd_remap = x[0] - '0';
        const char* mo = getenv("MFSR_MARGIN_OVERLAP");
        if (mo && (mo[0] == '0' || mo[0] == '1')) g_margin_overlap = mo[0] - '0';
        return true;
    }();
    (void)env_read;
}

}  // namespace

// Returns 1 if the fast kernels were launched for all `nFrames` (1 to 4) frames, 0 if the
// configuration is not one they handle (caller falls back to the straight kernel, or splits a group of
// three or four), -1 if a memset of the fresh-accumulator mode failed.
// With two to four frames the LDS tile kernel fuses all of them in one pass over the accumulators; the other
// geometries take at most two, frame after frame.
int mfsr_try_launch_accumulate2x_strip(int nFrames, const uint16_t* const* dataIn, mfsr_float3* imgOut,
                                       mfsr_float3* totalWeights, const mfsr_float4* const* certaintyMask,
                                       mfsr_tex2d kernelParam, const mfsr_tex2d* shifts, mfsr_float3 whiteLevel,
                                       mfsr_float3 blackLevel, int dimX, int dimY, int strideOut, int strideMask,
                                       int fresh, int rowBegin, int rowEnd, mfsr_stream_t stream)
{
    read_env_once();
    if (nFrames < 1 || nFrames > 4) return 0;
    int cfa[4];
    mfsr_get_cfa_pattern(cfa);
    for (int i = 0; i < 4; i++)
        if (cfa[i] > MFSR_BLUE) return 0;
    const int packed2 = pack_cfa(cfa[0], cfa[1], cfa[2], cfa[3]);
    // layout requirements of the vectorised accumulator access
    if ((dimX & 1) || ((uintptr_t)imgOut & 15) || ((uintptr_t)totalWeights & 15) || (strideOut & 15)) return 0;
    if (dimX < 2 * STRIP_MARGIN || dimY < 2 * STRIP_MARGIN) return 0;
    Levels3 glv;
    StripLevels lv;
    const float wl[3] = {whiteLevel.x, whiteLevel.y, whiteLevel.z}, bl[3] = {blackLevel.x, blackLevel.y, blackLevel.z};
    for (int c = 0; c < 3; c++) {
        glv.white[c] = wl[c];
        glv.black[c] = bl[c];
        lv.black[c] = bl[c];
        lv.invWhite[c] = 1.0f / wl[c];
    }
    const int hrW = 2 * dimX, hrH = 2 * dimY;
    // HR row window [rowBegin, rowEnd): whole 4-row tile rows (the caller aligns it to 16 rows or the frame's end)
    const int rowBlock0 = rowBegin / 4, rowBlocks = mfsr_cdiv(rowEnd, 4) - rowBlock0;
    dim3 block(64, 4), grid(mfsr_cdiv(hrW / 4, 64), rowBlocks);
    hipStream_t st = mfsr_s(stream);
    pix3* pI = (pix3*)imgOut;
    pix3* pT = (pix3*)totalWeights;
    const int cp = mfsr_cfa_packed();
    // two to four frames in one pass over the accumulators: the LDS tile kernel's geometry only
    bool pair = nFrames >= 2;
    for (int n = 0; n < nFrames; n++) pair = pair && tile_kernel_ok(kernelParam, shifts[n], dimX, dimY);
    // the monochrome pipeline (fields at the raw resolution = HR/2): its own instantiation of the tile kernel, one or two frames
    constexpr int kMono = pack_cfa(MFSR_GREEN, MFSR_GREEN, MFSR_GREEN, MFSR_GREEN);
    bool mono2 = packed2 == kMono && nFrames <= 2;
    for (int n = 0; n < nFrames; n++) mono2 = mono2 && tile_kernel_ok_fr2(kernelParam, shifts[n], dimX, dimY);
    if (nFrames > 2 && !pair) return 0;  // the caller splits the group
    // fresh accumulators ("as if zeroed", never read): the tile kernels write every row outside the top
    // and bottom margin bands themselves, the bands are zeroed here; the other kernels get a full memset
    const bool tileFirst = pair || mono2 || (nFrames == 1 && tile_kernel_ok(kernelParam, shifts[0], dimX, dimY));
    int tileFresh = 0;
    if (fresh) {
        if (tileFirst) {
            tileFresh = 1;
            if (zero_margin_bands(imgOut, totalWeights, hrH, strideOut, rowBegin, rowEnd, st) != 0) return -1;
        } else {
            const size_t off = (size_t)rowBegin * strideOut, bytes = (size_t)(rowEnd - rowBegin) * strideOut;
            if (hipMemsetAsync((char*)imgOut + off, 0, bytes, st) != hipSuccess) return -1;
            if (hipMemsetAsync((char*)totalWeights + off, 0, bytes, st) != hipSuccess) return -1;
        }
    }
    // margin launches: on the side stream (forked here, joined after the last one) when a tile kernel leads the call
    MarginStream* msx = (g_margin_overlap && tileFirst && !tileFresh) ? margin_stream() : nullptr;
    if (msx && (hipEventRecord(msx->fork, st) != hipSuccess || hipStreamWaitEvent(msx->stream, msx->fork, 0) != hipSuccess)) msx = nullptr;
    const hipStream_t mst = msx ? msx->stream : st;
    auto launch_margin = [&](int n) {
        const int M = STRIP_MARGIN;
        const long long cnt = 2LL * (M - 1) * (hrW - 2) + (long long)(hrH - 2 * M) * 2 * (M - 1);
        hipLaunchKernelGGL(k_accumulateMargin<2>, dim3(mfsr_cdiv(cnt, 256)), dim3(256), 0, mst, dataIn[n], pI, pT,
                           (const float4*)certaintyMask[n], kernelParam, shifts[n], glv, dimX, dimY, strideOut, strideMask, cp,
                           rowBegin, rowEnd);
        if (msx && n == nFrames - 1) {
            (void)hipEventRecord(msx->join, msx->stream);
            (void)hipStreamWaitEvent(st, msx->join, 0);
        }
    };
    // NF frames: the tile kernel, then their margin pixels in one launch
    auto launch_group = [&](auto cfaTag, auto nfTag) {
        constexpr int CFA = decltype(cfaTag)::value, NF = decltype(nfTag)::value;
        TileFrames<NF> fr;
        for (int n = 0; n < NF; n++) {
            fr.f[n].raw = dataIn[n];
            fr.f[n].mask = (const float4*)certaintyMask[n];
            fr.f[n].shifts = shifts[n];
        }
        launch_tile<CFA, NF>(grid, block, st, fr, pI, pT, kernelParam, glv, lv, dimX, dimY, strideOut, strideMask, cp, tileFresh,
                             rowBlock0);
        const int M = STRIP_MARGIN;
        const long long cnt = 2LL * (M - 1) * (hrW - 2) + (long long)(hrH - 2 * M) * 2 * (M - 1);
        hipLaunchKernelGGL((k_accumulateMarginN<NF, 2>), dim3(mfsr_cdiv(cnt, 64)), dim3(64, NF), 0, mst, fr, pI, pT, kernelParam, glv,
                           dimX, dimY, strideOut, strideMask, cp, rowBegin, rowEnd);
        if (msx) {
            (void)hipEventRecord(msx->join, msx->stream);
            (void)hipStreamWaitEvent(st, msx->join, 0);
        }
    };
    if (mono2) {
        auto launch_mono = [&](auto nfTag) {
            constexpr int NF = decltype(nfTag)::value;
            TileFrames<NF> fr;
            for (int n = 0; n < NF; n++) {
                fr.f[n].raw = dataIn[n];
                fr.f[n].mask = (const float4*)certaintyMask[n];
                fr.f[n].shifts = shifts[n];
            }
            launch_tile<kMono, NF, 2>(grid, block, st, fr, pI, pT, kernelParam, glv, lv, dimX, dimY, strideOut, strideMask, cp, tileFresh,
                                      rowBlock0);
            if (NF == 1) {
                launch_margin(0);
                return;
            }
            const int M = STRIP_MARGIN;
            const long long cnt = 2LL * (M - 1) * (hrW - 2) + (long long)(hrH - 2 * M) * 2 * (M - 1);
            hipLaunchKernelGGL((k_accumulateMarginN<NF, 2>), dim3(mfsr_cdiv(cnt, 64)), dim3(64, NF), 0, mst, fr, pI, pT, kernelParam, glv,
                               dimX, dimY, strideOut, strideMask, cp, rowBegin, rowEnd);
            if (msx) {
                (void)hipEventRecord(msx->join, msx->stream);
                (void)hipStreamWaitEvent(st, msx->join, 0);
            }
        };
        if (nFrames == 1) launch_mono(std::integral_constant<int, 1>{});
        if (nFrames == 2) launch_mono(std::integral_constant<int, 2>{});
        return 1;
    }
#define STRIP_CASE(a, b, c, d)                                                                                         \
    case pack_cfa(a, b, c, d):                                                                                         \
        if (pair) {                                                                                                    \
            using CfaTag = std::integral_constant<int, pack_cfa(a, b, c, d)>;                                          \
            if (nFrames == 2) launch_group(CfaTag{}, std::integral_constant<int, 2>{});                                \
            if (nFrames == 3) launch_group(CfaTag{}, std::integral_constant<int, 3>{});                                \
            if (nFrames == 4) launch_group(CfaTag{}, std::integral_constant<int, 4>{});                                \
        } else if (nFrames == 2) {                                                                                     \
            TileFrames<2> fr;                                                                                          \
            for (int n = 0; n < 2; n++) {                                                                              \
                fr.f[n].raw = dataIn[n];                                                                               \
                fr.f[n].mask = (const float4*)certaintyMask[n];                                                        \
                fr.f[n].shifts = shifts[n];                                                                            \
            }                                                                                                          \
            launch_strip_regs<pack_cfa(a, b, c, d), 2>(grid, block, st, fr, pI, pT, kernelParam, glv, lv, dimX, dimY,  \
                                                       strideOut, strideMask, cp, rowBlock0);                                    \
            launch_margin(0);                                                                                          \
            launch_margin(1);                                                                                          \
        } else {                                                                                                       \
            TileFrames<1> fr;                                                                                          \
            fr.f[0].raw = dataIn[0];                                                                                   \
            fr.f[0].mask = (const float4*)certaintyMask[0];                                                            \
            fr.f[0].shifts = shifts[0];                                                                                \
            if (tile_kernel_ok(kernelParam, shifts[0], dimX, dimY))                                                    \
                launch_tile<pack_cfa(a, b, c, d), 1>(grid, block, st, fr, pI, pT, kernelParam, glv, lv, dimX, dimY,    \
                                                     strideOut, strideMask, cp, tileFresh, rowBlock0);                           \
            else                                                                                                       \
                launch_strip_regs<pack_cfa(a, b, c, d), 1>(grid, block, st, fr, pI, pT, kernelParam, glv, lv, dimX,    \
                                                           dimY, strideOut, strideMask, cp, rowBlock0);                           \
            launch_margin(0);                                                                                          \
        }                                                                                                              \
        return 1;
    switch (packed2) {
        STRIP_CASE(MFSR_RED, MFSR_GREEN, MFSR_GREEN, MFSR_BLUE)   // RGGB
        STRIP_CASE(MFSR_BLUE, MFSR_GREEN, MFSR_GREEN, MFSR_RED)   // BGGR
        STRIP_CASE(MFSR_GREEN, MFSR_RED, MFSR_BLUE, MFSR_GREEN)   // GRBG
        STRIP_CASE(MFSR_GREEN, MFSR_BLUE, MFSR_RED, MFSR_GREEN)   // GBRG
        STRIP_CASE(MFSR_GREEN, MFSR_GREEN, MFSR_GREEN, MFSR_GREEN) // monochrome
        default: break;
    }
#undef STRIP_CASE
    return 0;
}

// x4: returns 1 if the x4 tile kernel took the nFrames (1 to 4) frames, 0 if the geometry is not its
// (fields at HR/8, even dimensions); the caller then uses the straight kernel.
int mfsr_try_launch_accumulate4x_tile(int nFrames, const uint16_t* const* dataIn, mfsr_float3* imgOut,
                                      mfsr_float3* totalWeights, const mfsr_float4* const* certaintyMask,
                                      mfsr_tex2d kernelParam, const mfsr_tex2d* shifts, mfsr_float3 whiteLevel,
                                      mfsr_float3 blackLevel, int dimX, int dimY, int strideOut, int strideMask,
                                      int fresh, int rowBegin, int rowEnd, mfsr_stream_t stream)
{
    read_env_once();
    if (nFrames < 1 || nFrames > 4 || !g_strip_use_tile) return 0;
    int cfa[4];
    mfsr_get_cfa_pattern(cfa);
    for (int i = 0; i < 4; i++)
        if (cfa[i] > MFSR_BLUE) return 0;
    const int packed2 = pack_cfa(cfa[0], cfa[1], cfa[2], cfa[3]);
    const int hrW = 4 * dimX, hrH = 4 * dimY;
    if ((dimX & 1) || (dimY & 1) || ((uintptr_t)imgOut & 15) || ((uintptr_t)totalWeights & 15) || (strideOut & 15)) return 0;
    if (hrW < 4 * STRIP_MARGIN || hrH < 4 * STRIP_MARGIN) return 0;
    for (int n = 0; n < nFrames; n++)
        if (!(kernelParam.width == shifts[n].width && kernelParam.height == shifts[n].height && kernelParam.width * 8 == hrW &&
              kernelParam.height * 8 == hrH && kernelParam.width >= 4))
            return 0;
    Levels3 glv;
    StripLevels lv;
    const float wl[3] = {whiteLevel.x, whiteLevel.y, whiteLevel.z}, bl[3] = {blackLevel.x, blackLevel.y, blackLevel.z};
    for (int c = 0; c < 3; c++) {
        glv.white[c] = wl[c];
        glv.black[c] = bl[c];
        lv.black[c] = bl[c];
        lv.invWhite[c] = 1.0f / wl[c];
    }
    hipStream_t st = mfsr_s(stream);
    pix3* pI = (pix3*)imgOut;
    pix3* pT = (pix3*)totalWeights;
    const int cp = mfsr_cfa_packed();
    // HR row window [rowBegin, rowEnd): whole 2-row tile rows
    const int tileY0 = rowBegin / 2;
    const int tilesX = mfsr_cdiv(hrW, 512), tilesY = mfsr_cdiv(rowEnd, 2) - tileY0;
    const dim3 block(64, 4), grid(tilesX * tilesY);
    if (fresh) {  // the top and bottom margin bands are the only rows the tile kernel does not write
        if (zero_margin_bands(imgOut, totalWeights, hrH, strideOut, rowBegin, rowEnd, st) != 0) return -1;
    }
    MarginStream* msx = (g_margin_overlap && !fresh) ? margin_stream() : nullptr;
    if (msx && (hipEventRecord(msx->fork, st) != hipSuccess || hipStreamWaitEvent(msx->stream, msx->fork, 0) != hipSuccess)) msx = nullptr;
    const hipStream_t mst = msx ? msx->stream : st;
    auto launch_margin = [&](int n) {
        const int M = STRIP_MARGIN;
        const long long cnt = 2LL * (M - 1) * (hrW - 2) + (long long)(hrH - 2 * M) * 2 * (M - 1);
        hipLaunchKernelGGL(k_accumulateMargin<4>, dim3(mfsr_cdiv(cnt, 256)), dim3(256), 0, mst, dataIn[n], pI, pT,
                           (const float4*)certaintyMask[n], kernelParam, shifts[n], glv, dimX, dimY, strideOut, strideMask, cp,
                           rowBegin, rowEnd);
        if (msx && n == nFrames - 1) {
            (void)hipEventRecord(msx->join, msx->stream);
            (void)hipStreamWaitEvent(st, msx->join, 0);
        }
    };
    // NF frames: the tile kernel, then their margin pixels in one launch (one frame: the single-frame margin kernel)
    auto launch_group = [&](auto cfaTag, auto nfTag) {
        constexpr int CFA = decltype(cfaTag)::value, NF = decltype(nfTag)::value;
        TileFrames<NF> fr;
        for (int n = 0; n < NF; n++) {
            fr.f[n].raw = dataIn[n];
            fr.f[n].mask = (const float4*)certaintyMask[n];
            fr.f[n].shifts = shifts[n];
        }
        hipLaunchKernelGGL((k_accumulate4xTile<CFA, NF>), grid, block, 0, st, fr, pI, pT, kernelParam, glv, lv, dimX, dimY, strideOut,
                           strideMask, cp, tilesX, fresh ? 1 : 0, tileY0);
        if (NF == 1) {
            launch_margin(0);
            return;
        }
        const int M = STRIP_MARGIN;
        const long long cnt = 2LL * (M - 1) * (hrW - 2) + (long long)(hrH - 2 * M) * 2 * (M - 1);
        hipLaunchKernelGGL((k_accumulateMarginN<NF, 4>), dim3(mfsr_cdiv(cnt, 64)), dim3(64, NF), 0, mst, fr, pI, pT, kernelParam, glv,
                           dimX, dimY, strideOut, strideMask, cp, rowBegin, rowEnd);
        if (msx) {
            (void)hipEventRecord(msx->join, msx->stream);
            (void)hipStreamWaitEvent(st, msx->join, 0);
        }
    };
#define X4_CASE(a, b, c, d)                                                                                            \
    case pack_cfa(a, b, c, d): {                                                                                       \
        using CfaTag = std::integral_constant<int, pack_cfa(a, b, c, d)>;                                              \
        if (nFrames == 1) launch_group(CfaTag{}, std::integral_constant<int, 1>{});                                    \
        if (nFrames == 2) launch_group(CfaTag{}, std::integral_constant<int, 2>{});                                    \
        if (nFrames == 3) launch_group(CfaTag{}, std::integral_constant<int, 3>{});                                    \
        if (nFrames == 4) launch_group(CfaTag{}, std::integral_constant<int, 4>{});                                    \
        return 1;                                                                                                      \
    }
    switch (packed2) {
        X4_CASE(MFSR_RED, MFSR_GREEN, MFSR_GREEN, MFSR_BLUE)   // RGGB
        X4_CASE(MFSR_BLUE, MFSR_GREEN, MFSR_GREEN, MFSR_RED)   // BGGR
        X4_CASE(MFSR_GREEN, MFSR_RED, MFSR_BLUE, MFSR_GREEN)   // GRBG
        X4_CASE(MFSR_GREEN, MFSR_BLUE, MFSR_RED, MFSR_GREEN)   // GBRG
        X4_CASE(MFSR_GREEN, MFSR_GREEN, MFSR_GREEN, MFSR_GREEN) // monochrome
        default: break;
    }
#undef X4_CASE
    return 0;
}
