// jpeg_baseline.hpp -- minimal baseline-JPEG reader for the CLI (the reference loads its "car" burst with cv::imread,
// finalProject/Project/multi_frame_sr.cpp:155-159,172: car/1.jpg .. car/4.jpg are baseline SOF0, 8 bit, YCbCr 4:2:0).
//
// Scope: sequential baseline DCT (SOF0), 8-bit precision, 1 or 3 components, sampling factors 1 or 2, Huffman coding,
// restart intervals.  No progressive / arithmetic / 12-bit / CMYK.  Decoding follows ITU-T T.81: Huffman decode (F.2.2),
// dequantisation, 8x8 inverse DCT (A.3.3, evaluated in double precision and rounded), level shift, chroma upsampling by
// pixel replication, JFIF YCbCr -> RGB.  OpenCV decodes through libjpeg (integer IDCT, "fancy" triangular chroma
// upsampling), so individual samples differ by a few levels from cv::imread -- this is the file-format boundary, not the
// hot path; tests compare against PIL's libjpeg decode with that tolerance.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace jpegb {

struct Huff {
    // canonical code tables per code length (T.81 C.2 / F.2.2.3)
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    bool present = false;
};

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int pred = 0;
    int bw = 0, bh = 0;  // blocks per row / column (padded to whole MCUs)
    std::vector<uint8_t> px;  // bw*8 x bh*8 samples
};

struct Bits {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int n = 0;
    bool eof = false;
    int bit()
    {
        if (n == 0) {
            if (p >= end) {
                eof = true;
                return 0;
            }
            uint8_t b = *p++;
            if (b == 0xFF) {
                if (p < end && *p == 0x00)
                    p++;  // stuffed zero
                else {
                    eof = true;  // a marker inside entropy data: pad with zeros (T.81 F.2.2.5)
                    p--;
                    return 0;
                }
            }
            acc = b;
            n = 8;
        }
        n--;
        return (acc >> n) & 1;
    }
    int receive(int s)
    {
        int v = 0;
        for (int i = 0; i < s; i++) v = (v << 1) | bit();
        return v;
    }
    void reset()
    {
        n = 0;
        acc = 0;
        eof = false;
    }
};

inline int extend(int v, int t) { return (t == 0) ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }

inline int decode(Bits& br, const Huff& h)
{
    int code = br.bit(), i = 1;
    while (i <= 16 && (h.maxcode[i] < 0 || code > h.maxcode[i])) {
        code = (code << 1) | br.bit();
        i++;
    }
    if (i > 16) return -1;
    return h.vals[h.valptr[i] + code - h.mincode[i]];
}

inline void idct8x8(const int* in, uint8_t* out, int stride)
{
    static double c[8][8];
    static bool init = false;
    if (!init) {
        for (int x = 0; x < 8; x++)
            for (int u = 0; u < 8; u++) c[x][u] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * M_PI / 16.0);
        init = true;
    }
    double tmp[64];
    for (int y = 0; y < 8; y++)      // rows: over u
        for (int x = 0; x < 8; x++) {
            double s = 0;
            for (int u = 0; u < 8; u++) s += c[x][u] * in[y * 8 + u];
            tmp[y * 8 + x] = s;
        }
    for (int x = 0; x < 8; x++)      // columns: over v
        for (int y = 0; y < 8; y++) {
            double s = 0;
            for (int v = 0; v < 8; v++) s += c[y][v] * tmp[v * 8 + x];
            int q = (int)std::lrint(s + 128.0);
            out[y * stride + x] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
        }
}

// Decodes `buf` into interleaved 8-bit samples (ch = 1 or 3, RGB).  Returns false on anything outside the scope above.
inline bool decode_jpeg(const std::vector<uint8_t>& buf, int& w, int& h, int& ch, std::vector<uint8_t>& out)
{
    static const int zz[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                               41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                               30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    if (buf.size() < 4 || buf[0] != 0xFF || buf[1] != 0xD8) return false;
    uint16_t qt[4][64];
    bool haveQ[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    Comp comp[3];
    int nc = 0, restart = 0, hmax = 1, vmax = 1;
    bool haveSof = false;
    size_t pos = 2;
    w = h = ch = 0;
    while (pos + 4 <= buf.size()) {
        if (buf[pos] != 0xFF) return false;
        const int m = buf[pos + 1];
        if (m == 0xFF) {
            pos++;
            continue;
        }
        const size_t len = ((size_t)buf[pos + 2] << 8) | buf[pos + 3];
        if (len < 2 || pos + 2 + len > buf.size()) return false;
        const uint8_t* d = &buf[pos + 4];
        const size_t dl = len - 2;
        if (m == 0xDB) {  // DQT
            size_t i = 0;
            while (i < dl) {
                const int pq = d[i] >> 4, tq = d[i] & 15;
                i++;
                if (tq > 3 || i + (pq ? 128 : 64) > dl) return false;
                for (int k = 0; k < 64; k++) {
                    qt[tq][zz[k]] = pq ? (uint16_t)((d[i] << 8) | d[i + 1]) : d[i];
                    i += pq ? 2 : 1;
                }
                haveQ[tq] = true;
            }
        } else if (m == 0xC4) {  // DHT
            size_t i = 0;
            while (i + 17 <= dl) {
                const int tc = d[i] >> 4, th = d[i] & 15;
                if (tc > 1 || th > 3) return false;
                Huff& t = tc ? ac[th] : dc[th];
                int counts[17], total = 0;
                for (int k = 1; k <= 16; k++) {
                    counts[k] = d[i + k];
                    total += counts[k];
                }
                if (total > 256 || i + 17 + total > dl) return false;
                memcpy(t.vals, &d[i + 17], total);
                int code = 0, p = 0;
                for (int k = 1; k <= 16; k++) {
                    t.valptr[k] = p;
                    t.mincode[k] = code;
                    code += counts[k];
                    p += counts[k];
                    t.maxcode[k] = counts[k] ? code - 1 : -1;
                    code <<= 1;
                }
                t.maxcode[17] = 0x7fffffff;
                t.present = true;
                i += 17 + total;
            }
        } else if (m == 0xC0) {  // SOF0: baseline
            if (dl < 6 || d[0] != 8) return false;
            h = (d[1] << 8) | d[2];
            w = (d[3] << 8) | d[4];
            nc = d[5];
            if ((nc != 1 && nc != 3) || w <= 0 || h <= 0 || w > 65500 || h > 65500 || dl < 6 + 3 * (size_t)nc) return false;
            // every 8x8 block costs at least two Huffman codes (>= 2 bits) in the scan: a frame header that promises more
            // blocks than the file's bits can encode is refused before its planes (up to 4 x 65500^2 bytes) are allocated
            if ((size_t)w * (size_t)h / 64 > buf.size() * 4 + 1024) return false;
            for (int k = 0; k < nc; k++) {
                comp[k].id = d[6 + 3 * k];
                comp[k].h = d[7 + 3 * k] >> 4;
                comp[k].v = d[7 + 3 * k] & 15;
                comp[k].tq = d[8 + 3 * k];
                if (comp[k].h < 1 || comp[k].h > 2 || comp[k].v < 1 || comp[k].v > 2 || comp[k].tq > 3) return false;
                hmax = comp[k].h > hmax ? comp[k].h : hmax;
                vmax = comp[k].v > vmax ? comp[k].v : vmax;
            }
            haveSof = true;
        } else if (m == 0xC1 || m == 0xC2 || m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
            return false;  // extended / progressive / lossless / arithmetic: out of scope
        } else if (m == 0xDD) {
            if (dl < 2) return false;
            restart = (d[0] << 8) | d[1];
        } else if (m == 0xDA) {  // SOS: the (single) scan of a baseline file
            if (!haveSof || dl < 1 || d[0] != nc || dl < 1 + 2 * (size_t)nc + 3) return false;
            for (int k = 0; k < nc; k++) {
                int idx = -1;
                for (int j = 0; j < nc; j++)
                    if (comp[j].id == d[1 + 2 * k]) idx = j;
                if (idx < 0) return false;
                comp[idx].td = d[2 + 2 * k] >> 4;
                comp[idx].ta = d[2 + 2 * k] & 15;
                if (comp[idx].td > 3 || comp[idx].ta > 3 || !dc[comp[idx].td].present || !ac[comp[idx].ta].present || !haveQ[comp[idx].tq])
                    return false;
            }
            const int mcuW = 8 * hmax, mcuH = 8 * vmax;
            const int mx = (w + mcuW - 1) / mcuW, my = (h + mcuH - 1) / mcuH;
            for (int k = 0; k < nc; k++) {
                comp[k].bw = mx * comp[k].h;
                comp[k].bh = my * comp[k].v;
                comp[k].px.assign((size_t)comp[k].bw * 8 * comp[k].bh * 8, 0);
                comp[k].pred = 0;
            }
            Bits br{&buf[pos + 2 + len], buf.data() + buf.size()};
            int count = 0;
            for (int my_ = 0; my_ < my; my_++)
                for (int mx_ = 0; mx_ < mx; mx_++) {
                    if (restart && count == restart) {
                        // RSTn marker: byte-align, skip it, reset predictors
                        br.reset();
                        while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) br.p++;
                        if (br.p + 1 < br.end) br.p += 2;
                        for (int k = 0; k < nc; k++) comp[k].pred = 0;
                        count = 0;
                    }
                    count++;
                    for (int k = 0; k < nc; k++)
                        for (int by = 0; by < comp[k].v; by++)
                            for (int bx = 0; bx < comp[k].h; bx++) {
                                int blk[64];
                                memset(blk, 0, sizeof(blk));
                                const int t = decode(br, dc[comp[k].td]);
                                if (t < 0 || t > 11) return false;
                                comp[k].pred += extend(br.receive(t), t);
                                blk[0] = comp[k].pred * qt[comp[k].tq][0];
                                for (int kk = 1; kk < 64;) {
                                    const int rs = decode(br, ac[comp[k].ta]);
                                    if (rs < 0) return false;
                                    const int r = rs >> 4, s = rs & 15;
                                    if (s == 0) {
                                        if (r == 15) {
                                            kk += 16;
                                            continue;
                                        }
                                        break;  // EOB
                                    }
                                    kk += r;
                                    if (kk > 63) return false;
                                    blk[zz[kk]] = extend(br.receive(s), s) * qt[comp[k].tq][zz[kk]];
                                    kk++;
                                }
                                const int px0 = (mx_ * comp[k].h + bx) * 8, py0 = (my_ * comp[k].v + by) * 8;
                                idct8x8(blk, &comp[k].px[(size_t)py0 * comp[k].bw * 8 + px0], comp[k].bw * 8);
                            }
                }
            // upsample by replication + colour conversion (JFIF: full-range BT.601)
            ch = nc;
            out.assign((size_t)w * h * ch, 0);
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    int s[3] = {0, 128, 128};
                    for (int k = 0; k < nc; k++) {
                        const int sx = x * comp[k].h / hmax, sy = y * comp[k].v / vmax;
                        s[k] = comp[k].px[(size_t)sy * comp[k].bw * 8 + sx];
                    }
                    uint8_t* o = &out[((size_t)y * w + x) * ch];
                    if (nc == 1) {
                        o[0] = (uint8_t)s[0];
                    } else {
                        const double Y = s[0], cb = s[1] - 128.0, cr = s[2] - 128.0;
                        const double rgb[3] = {Y + 1.402 * cr, Y - 0.344136 * cb - 0.714136 * cr, Y + 1.772 * cb};
                        for (int c = 0; c < 3; c++) {
                            const int q = (int)std::lrint(rgb[c]);
                            o[c] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
                        }
                    }
                }
            return true;
        } else if (m == 0xD9) {
            return false;  // EOI before any scan
        }
        pos += 2 + len;
    }
    return false;
}

}  // namespace jpegb
