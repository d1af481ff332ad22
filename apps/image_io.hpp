// image_io.hpp -- image file I/O of the CLI tools (the reference uses cv::imread / cv::imwrite,
// finalProject/Project/multi_frame_sr.cpp:172,207-209): 8-bit PNG (zlib) read + write, binary PGM/PPM read, baseline JPEG
// read (jpeg_baseline.hpp: the "car" burst).  Host code only.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "jpeg_baseline.hpp"

struct Image8 {
    int w = 0, h = 0, ch = 0;
    std::vector<uint8_t> px;
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
        if (buf[pos] == '#') {
            while (pos < buf.size() && buf[pos] != '\n') pos++;
            continue;
        }
        int v = 0;
        while (pos < buf.size() && buf[pos] >= '0' && buf[pos] <= '9') v = v * 10 + (buf[pos++] - '0');
        vals[nv++] = v;
    }
    pos++;
    if (nv != 3 || vals[2] != 255) return false;
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

static bool read_image(const std::string& path, Image8& img)
{
    return read_png(path, img) || read_pnm(path, img) || read_jpeg(path, img);
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

