// image_io.hpp -- image file I/O of the CLI tools (the reference uses cv::imread / cv::imwrite,
// finalProject/Project/multi_frame_sr.cpp:172,207-209): 8-bit PNG (zlib) read + write, binary PGM/PPM read, baseline JPEG
// read (jpeg_baseline.hpp: the "car" burst), uncompressed 8 / 16-bit TIFF read (raw-like inputs, test_opencv/main.cpp:346-368).
// Host code only.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "jpeg_baseline.hpp"

struct Image8 {
    int w = 0, h = 0, ch = 0;
    std::vector<uint8_t> px;      // 8 bits per sample (for 16-bit sources: the high bytes)
    std::vector<uint16_t> px16;   // 16-bit sources only: the full samples, same layout
};

static uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | p[1] << 16 | p[2] << 8 | p[3]; }

static bool read_file(const std::string& path, std::vector<uint8_t>& out)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    if (n < 0 || n > (1L << 30)) {  // unseekable, or nothing this tool reads is a gigabyte
        fclose(f);
        return false;
    }
    fseek(f, 0, SEEK_SET);
    out.resize(n);
    bool ok = fread(out.data(), 1, n, f) == (size_t)n;
    fclose(f);
    return ok;
}

// 8-bit gray / RGB / RGBA, non-interlaced PNG
static bool read_png(const std::string& path, Image8& img)
{
    std::vector<uint8_t> buf;
    if (!read_file(path, buf) || buf.size() < 33 || memcmp(buf.data(), "\x89PNG\r\n\x1a\n", 8)) return false;
    size_t pos = 8;
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat;
    while (pos + 12 <= buf.size()) {
        uint32_t len = be32(&buf[pos]);
        const char* type = (const char*)&buf[pos + 4];
        const uint8_t* data = &buf[pos + 8];
        if (pos + 12 + len > buf.size()) return false;
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) return false;
            const uint32_t uw = be32(data), uh = be32(data + 4);
            if (uw == 0 || uh == 0 || uw > 65536 || uh > 65536) return false;  // also keeps (stride+1)*h far from overflow
            w = be32(data);
            h = be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    if (w <= 0 || h <= 0 || depth != 8 || interlace != 0) return false;
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 6 ? 4 : ctype == 4 ? 2 : 0;
    if (!ch) return false;
    const size_t stride = (size_t)w * ch;
    // a deflate stream expands by at most ~1032 : 1: a header that promises more than the IDAT bytes can hold is refused
    // BEFORE anything of that size is allocated (65 536 x 65 536 x 4 would be 17 GB)
    if ((stride + 1) * (size_t)h > idat.size() * 1032 + 65536) return false;
    std::vector<uint8_t> rawpx((stride + 1) * h);
    uLongf dlen = rawpx.size();
    if (uncompress(rawpx.data(), &dlen, idat.data(), idat.size()) != Z_OK || dlen != rawpx.size()) return false;
    img.w = w;
    img.h = h;
    img.ch = ch;
    img.px.assign(stride * h, 0);
    std::vector<uint8_t> prev(stride, 0);
    for (int y = 0; y < h; y++) {
        const uint8_t* in = &rawpx[(stride + 1) * y];
        uint8_t* out = &img.px[stride * y];
        const int ft = in[0];
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= (size_t)ch ? out[i - ch] : 0, b = prev[i], c = i >= (size_t)ch ? prev[i - ch] : 0;
            int pr = 0;
            if (ft == 1) pr = a;
            else if (ft == 2) pr = b;
            else if (ft == 3) pr = (a + b) / 2;
            else if (ft == 4) {
                const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            out[i] = (uint8_t)(in[1 + i] + pr);
        }
        memcpy(prev.data(), out, stride);
    }
    return true;
}

static bool read_pnm(const std::string& path, Image8& img)
{
    std::vector<uint8_t> buf;
    if (!read_file(path, buf) || buf.size() < 8 || buf[0] != 'P' || (buf[1] != '5' && buf[1] != '6')) return false;
    int vals[3], nv = 0;
    size_t pos = 2;
    while (nv < 3 && pos < buf.size()) {
        while (pos < buf.size() && (buf[pos] == ' ' || buf[pos] == '\n' || buf[pos] == '\r' || buf[pos] == '\t')) pos++;
        if (pos >= buf.size()) return false;  // the header ends in white space
        if (buf[pos] == '#') {
            while (pos < buf.size() && buf[pos] != '\n') pos++;
            continue;
        }
        if (buf[pos] < '0' || buf[pos] > '9') return false;
        int v = 0;
        while (pos < buf.size() && buf[pos] >= '0' && buf[pos] <= '9') {
            v = v * 10 + (buf[pos++] - '0');
            if (v > 65535) return false;  // no overflow, and no frame this tool takes is larger
        }
        vals[nv++] = v;
    }
    pos++;  // the single white-space byte after maxval
    if (nv != 3 || vals[2] != 255 || vals[0] <= 0 || vals[1] <= 0 || pos > buf.size()) return false;
    img.w = vals[0];
    img.h = vals[1];
    img.ch = buf[1] == '5' ? 1 : 3;
    const size_t n = (size_t)img.w * img.h * img.ch;
    if (pos + n > buf.size()) return false;
    img.px.assign(buf.begin() + pos, buf.begin() + pos + n);
    return true;
}

static bool read_jpeg(const std::string& path, Image8& img)
{
    std::vector<uint8_t> buf;
    if (!read_file(path, buf)) return false;
    return jpegb::decode_jpeg(buf, img.w, img.h, img.ch, img.px);
}

// Baseline TIFF (classic, II or MM): first image, uncompressed, chunky, 8 or 16 bits per sample, 1 / 3 / 4 samples per
// pixel, any strip layout.  Everything else (LZW / deflate, tiles, planar, BigTIFF) is refused.
static bool read_tiff(const std::string& path, Image8& img)
{
    std::vector<uint8_t> buf;
    if (!read_file(path, buf) || buf.size() < 8) return false;
    const bool le = buf[0] == 'I' && buf[1] == 'I', be = buf[0] == 'M' && buf[1] == 'M';
    if (!le && !be) return false;
    auto u16 = [&](size_t o) -> uint32_t { return o + 2 <= buf.size() ? (le ? buf[o] | buf[o + 1] << 8 : buf[o] << 8 | buf[o + 1]) : 0; };
    auto u32 = [&](size_t o) -> uint32_t {
        if (o + 4 > buf.size()) return 0;
        return le ? (uint32_t)buf[o] | (uint32_t)buf[o + 1] << 8 | (uint32_t)buf[o + 2] << 16 | (uint32_t)buf[o + 3] << 24
                  : (uint32_t)buf[o] << 24 | (uint32_t)buf[o + 1] << 16 | (uint32_t)buf[o + 2] << 8 | (uint32_t)buf[o + 3];
    };
    if (u16(2) != 42) return false;
    const size_t ifd = u32(4);
    if (ifd == 0 || ifd + 2 > buf.size()) return false;
    const uint32_t n = u16(ifd);
    if (ifd + 2 + (size_t)n * 12 > buf.size()) return false;
    uint32_t width = 0, height = 0, bits = 1, compression = 1, spp = 1, rowsPerStrip = 0xffffffffu, planar = 1;
    std::vector<uint32_t> offsets, counts;
    // value(s) of an entry: type 3 = SHORT, 4 = LONG; inline when they fit 4 bytes
    auto values = [&](size_t e, std::vector<uint32_t>& out) {
        const uint32_t type = u16(e + 2), cnt = u32(e + 4);
        const size_t sz = type == 3 ? 2 : (type == 4 ? 4 : (type == 1 ? 1 : 0));
        if (sz == 0 || cnt > (1u << 24)) return false;
        const size_t base = (size_t)cnt * sz <= 4 ? e + 8 : u32(e + 8);
        if (base + (size_t)cnt * sz > buf.size()) return false;
        out.resize(cnt);
        for (uint32_t i = 0; i < cnt; i++) out[i] = sz == 2 ? u16(base + 2 * i) : (sz == 4 ? u32(base + 4 * (size_t)i) : buf[base + i]);
        return true;
    };
    for (uint32_t i = 0; i < n; i++) {
        const size_t e = ifd + 2 + (size_t)i * 12;
        const uint32_t tag = u16(e);
        std::vector<uint32_t> v;
        if (tag != 256 && tag != 257 && tag != 258 && tag != 259 && tag != 273 && tag != 277 && tag != 278 && tag != 279 && tag != 284)
            continue;
        if (!values(e, v) || v.empty()) return false;
        switch (tag) {
            case 256: width = v[0]; break;
            case 257: height = v[0]; break;
            case 258:
                bits = v[0];
                for (uint32_t b : v)
                    if (b != bits) return false;
                break;
            case 259: compression = v[0]; break;
            case 273: offsets = v; break;
            case 277: spp = v[0]; break;
            case 278: rowsPerStrip = v[0]; break;
            case 279: counts = v; break;
            case 284: planar = v[0]; break;
        }
    }
    if (width == 0 || height == 0 || width > 65535 || height > 65535 || compression != 1 || planar != 1) return false;
    if ((bits != 8 && bits != 16) || (spp != 1 && spp != 3 && spp != 4) || offsets.empty()) return false;
    if (rowsPerStrip == 0 || rowsPerStrip > height) rowsPerStrip = height;
    const size_t rowBytes = (size_t)width * spp * (bits / 8);
    const size_t strips = ((size_t)height + rowsPerStrip - 1) / rowsPerStrip;
    if (offsets.size() < strips) return false;
    if ((size_t)height * rowBytes > buf.size()) return false;  // uncompressed: the samples are in the file, or the header lies
    img.w = (int)width;
    img.h = (int)height;
    img.ch = (int)spp;
    img.px.resize((size_t)width * height * spp);
    if (bits == 16) img.px16.resize(img.px.size());
    else img.px16.clear();
    for (size_t st = 0; st < strips; st++) {
        const size_t y0 = st * rowsPerStrip, rows = y0 + rowsPerStrip <= height ? rowsPerStrip : height - y0;
        const size_t need = rows * rowBytes;
        if ((size_t)offsets[st] + need > buf.size() || (!counts.empty() && st < counts.size() && counts[st] < need)) return false;
        const uint8_t* src = buf.data() + offsets[st];
        const size_t o = y0 * (size_t)width * spp, cnt = rows * (size_t)width * spp;
        if (bits == 8) {
            memcpy(&img.px[o], src, cnt);
        } else {
            for (size_t i = 0; i < cnt; i++) {
                const uint16_t v = le ? (uint16_t)(src[2 * i] | src[2 * i + 1] << 8) : (uint16_t)(src[2 * i] << 8 | src[2 * i + 1]);
                img.px16[o + i] = v;
                img.px[o + i] = (uint8_t)(v >> 8);
            }
        }
    }
    return true;
}

static bool read_image(const std::string& path, Image8& img)
{
    img.px16.clear();
    try {  // the readers bound what they allocate by the file's size; an allocation failure is still a refusal, not an abort
        return read_png(path, img) || read_pnm(path, img) || read_jpeg(path, img) || read_tiff(path, img);
    } catch (const std::bad_alloc&) {
        return false;
    }
}

static void put32(std::vector<uint8_t>& v, uint32_t x)
{
    v.push_back(x >> 24);
    v.push_back(x >> 16);
    v.push_back(x >> 8);
    v.push_back(x);
}

static void png_chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data)
{
    put32(out, (uint32_t)data.size());
    std::vector<uint8_t> td(type, type + 4);
    td.insert(td.end(), data.begin(), data.end());
    out.insert(out.end(), td.begin(), td.end());
    put32(out, (uint32_t)crc32(0, td.data(), (uInt)td.size()));
}

[[maybe_unused]] static bool write_png(const std::string& path, const uint8_t* px, int w, int h, int ch)
{
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::vector<uint8_t> ihdr;
    put32(ihdr, w);
    put32(ihdr, h);
    ihdr.push_back(8);
    ihdr.push_back(ch == 1 ? 0 : 2);
    ihdr.push_back(0);
    ihdr.push_back(0);
    ihdr.push_back(0);
    png_chunk(out, "IHDR", ihdr);
    const size_t stride = (size_t)w * ch;
    std::vector<uint8_t> rawpx((stride + 1) * h);
    for (int y = 0; y < h; y++) {
        rawpx[(stride + 1) * y] = 0;
        memcpy(&rawpx[(stride + 1) * y + 1], px + stride * y, stride);
    }
    uLongf clen = compressBound(rawpx.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, rawpx.data(), rawpx.size(), 3) != Z_OK) return false;
    comp.resize(clen);
    png_chunk(out, "IDAT", comp);
    png_chunk(out, "IEND", {});
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fwrite(out.data(), 1, out.size(), f);
    fclose(f);
    return true;
}

