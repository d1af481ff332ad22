// multi_frame_sr -- drop-in for the reference CLI
//   finalProject/Project/multi_frame_sr.cpp:122-210
//   ./multi_frame_sr optFlowName inputName iterations
// Same argv, same dataset table (city / car / iso), same outputs
// (<input>_<flow>_sr_result.png and the sharpenImg2'ed <input>_<flow>_sr2_result.png)
// and the same "sec" / "FPS" prints, but the burst goes through the MI355X hot
// path (C-ABI of include/mfsr.h) instead of OpenCV's BTVL1.
//
// Differences that are inherent to the swap (documented, not hidden):
//   * optFlowName (farneback|tvl1|brox|pyrlk) selected an OpenCV optical-flow
//     back end; here every name maps to the built-in tile tracker + Lucas-Kanade
//     refinement, `iterations` sets the number of LK iterations.
//   * frames are 8-bit RGB files; they are re-mosaicked to an RGGB 12-bit raw
//     frame (value*16) because the hot path consumes raw frames.
//   * the reference replays the burst num_times=10 times through a temporal
//     window and times the second half; here every replay is one whole burst and
//     the second half of the replays is timed (same warm-up/timed split, :146-149,:188-206).
//   * image I/O: PNG (zlib) and binary PGM/PPM; JPEG ("car") is not decoded.
#include <hip/hip_runtime.h>
#include <zlib.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/mfsr.h"

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

static bool read_image(const std::string& path, Image8& img) { return read_png(path, img) || read_pnm(path, img); }

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

static bool write_png(const std::string& path, const uint8_t* px, int w, int h, int ch)
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

#define HIP_OK(x)                                                                   \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));          \
            return 1;                                                               \
        }                                                                           \
    } while (0)
#define MFSR_OK_OR_DIE(x)                                                           \
    do {                                                                            \
        int rc_ = (x);                                                              \
        if (rc_ != MFSR_OK) {                                                       \
            fprintf(stderr, "%s failed: %s\n", #x, mfsr_error_string(rc_));         \
            return 1;                                                               \
        }                                                                           \
    } while (0)

int main(int argc, char** argv)
{
    std::string optFlowName, inputName;
    int iterations = 10;
    if (argc == 1) {  // multi_frame_sr.cpp:126-128
        optFlowName = "farneback";
        inputName = "city";
    } else if (argc == 4) {
        optFlowName = argv[1];
        inputName = argv[2];
        iterations = atoi(argv[3]);
        if (iterations < 1) iterations = 1;
    } else {  // :138-142
        printf("./multi_frame_sr optFlowName inputName iterations\n");
        printf("\toptFlowName: farneback, tvl1, brox, pyrlk\n");
        printf("\tinputName: city, car, iso\n");
        printf("\titerations: integer, 1, 10, etc.\n");
        return -1;
    }
    if (optFlowName != "farneback" && optFlowName != "tvl1" && optFlowName != "brox" && optFlowName != "pyrlk") {
        fprintf(stderr, "Incorrect Optical Flow algorithm - %s\n", optFlowName.c_str());  // :84
        return -1;
    }
    const int scale = 2, num_times = 10, real_times = 5;  // :146-149
    int num_images = 5;
    std::string filenameFormat;
    if (inputName == "city") {
        num_images = 5;
        filenameFormat = "img_%06d.png";
    } else if (inputName == "car") {
        num_images = 4;
        filenameFormat = "car/%d.jpg";
    } else if (inputName == "iso") {
        num_images = 4;
        filenameFormat = "iso/%06d.png";
    } else {
        printf("wrong input\n");
        return -1;
    }

    // load the burst.  The reference indexes its frames i%num_images+1 (multi_frame_sr.cpp:171) while the frames bundled
    // with it are numbered from 0 (test_opencv/img_00000[0-4].png): use 1-based names when the last 1-based file exists,
    // 0-based names otherwise.
    std::vector<Image8> imgs(num_images);
    char buf[BUFSIZ];
    int firstIndex = 1;
    {
        snprintf(buf, sizeof(buf), filenameFormat.c_str(), num_images);
        FILE* probe = fopen(buf, "rb");
        if (probe)
            fclose(probe);
        else
            firstIndex = 0;
    }
    for (int i = 0; i < num_images; i++) {
        snprintf(buf, sizeof(buf), filenameFormat.c_str(), i + firstIndex);
        if (!read_image(buf, imgs[i])) {
            fprintf(stderr, "cannot read frame %d of '%s' (%s): PNG or binary PNM expected\n", i, inputName.c_str(), buf);
            return 1;
        }
        printf("%s, [%d x %d]\n", buf, imgs[i].w, imgs[i].h);
        if (imgs[i].w != imgs[0].w || imgs[i].h != imgs[0].h) {
            fprintf(stderr, "frame sizes differ\n");
            return 1;
        }
    }
    const int W = imgs[0].w & ~3, H = imgs[0].h & ~3;  // the pipeline needs multiples of 4
    if (mfsr_device_count() <= 0) {
        fprintf(stderr, "no HIP device: this build has no CPU fallback\n");
        return 1;
    }

    // re-mosaic 8-bit RGB (or gray) to a 12-bit RGGB raw frame
    std::vector<std::vector<uint16_t>> raws(num_images, std::vector<uint16_t>((size_t)W * H));
    for (int k = 0; k < num_images; k++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const uint8_t* p = &imgs[k].px[((size_t)y * imgs[k].w + x) * imgs[k].ch];
                const int c = (y & 1) + (x & 1);  // RGGB
                const int v = imgs[k].ch >= 3 ? p[c] : p[0];
                raws[k][(size_t)y * W + x] = (uint16_t)(v * 16);
            }

    mfsr_config cfg;
    MFSR_OK_OR_DIE(mfsr_config_default(&cfg, W, H, num_images, scale, 0));
    cfg.lkIterations = iterations;
    cfg.preAlign = 1;  // hand-held bursts: base shift + rotation per frame before the tile tracker (the bundled city frames
                       // are rotated by up to 15 degrees, test_opencv/main.cpp:1896)
    for (int c = 0; c < 3; c++) {
        cfg.black[c] = 0.0f;
        cfg.white[c] = 4080.0f;
    }
    cfg.maxVal = 4080.0f;
    const size_t wsBytes = mfsr_burst_workspace_bytes(&cfg), accBytes = mfsr_burst_accumulator_bytes(&cfg);
    void *ws = nullptr, *imgOut = nullptr, *weights = nullptr, *outF = nullptr;
    HIP_OK(hipMalloc(&ws, wsBytes));
    HIP_OK(hipMalloc(&imgOut, accBytes));
    HIP_OK(hipMalloc(&weights, accBytes));
    HIP_OK(hipMalloc(&outF, accBytes));
    std::vector<uint16_t*> dframes(num_images);
    for (int k = 0; k < num_images; k++) {
        HIP_OK(hipMalloc((void**)&dframes[k], (size_t)W * H * 2));
        HIP_OK(hipMemcpy(dframes[k], raws[k].data(), (size_t)W * H * 2, hipMemcpyHostToDevice));  // :172 upload
    }
    mfsr_burst* b = nullptr;
    MFSR_OK_OR_DIE(mfsr_burst_create(&b, &cfg, ws, wsBytes));

    const int hrW = W * scale, hrH = H * scale;
    const int start_i = num_times - real_times;
    std::chrono::steady_clock::time_point t0;
    for (int rep = 0; rep < num_times; rep++) {
        if (rep == start_i) {
            HIP_OK(hipDeviceSynchronize());
            t0 = std::chrono::steady_clock::now();  // tm1.start(), :188-190
        }
        HIP_OK(hipMemsetAsync(imgOut, 0, accBytes, nullptr));
        HIP_OK(hipMemsetAsync(weights, 0, accBytes, nullptr));
        MFSR_OK_OR_DIE(mfsr_burst_set_reference(b, dframes[cfg.reference], nullptr));
        for (int k = 0; k < num_images; k++)
            MFSR_OK_OR_DIE(mfsr_burst_add_frame(b, dframes[k], k == cfg.reference, (mfsr_float3*)imgOut,
                                                (mfsr_float3*)weights, nullptr));
        MFSR_OK_OR_DIE(mfsr_burst_finish(b, (const mfsr_float3*)imgOut, (const mfsr_float3*)weights, (mfsr_float3*)outF,
                                         nullptr, nullptr));
    }
    HIP_OK(hipDeviceSynchronize());
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%g sec\n", sec);                                               // :205
    printf("%g FPS\n", (double)(num_images * real_times) / sec);           // :206

    // result -> 8-bit RGB (D2H), then sharpenImg2 on the device
    uint8_t *d8 = nullptr, *d8s = nullptr;
    HIP_OK(hipMalloc((void**)&d8, (size_t)hrW * hrH * 3));
    HIP_OK(hipMalloc((void**)&d8s, (size_t)hrW * hrH * 3));
    MFSR_OK_OR_DIE(mfsr_quantize((const mfsr_float3*)outF, 12 * hrW, nullptr, d8, hrW, hrH, 255.0f, nullptr));
    MFSR_OK_OR_DIE(mfsr_sharpenImg2(d8, d8s, hrH, hrW, 3, hrW * 3, hrW * 3, nullptr));
    std::vector<uint8_t> h8((size_t)hrW * hrH * 3), h8s(h8.size());
    HIP_OK(hipMemcpy(h8.data(), d8, h8.size(), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h8s.data(), d8s, h8s.size(), hipMemcpyDeviceToHost));
    if (!write_png(inputName + "_" + optFlowName + "_sr_result.png", h8.data(), hrW, hrH, 3) ||   // :207
        !write_png(inputName + "_" + optFlowName + "_sr2_result.png", h8s.data(), hrW, hrH, 3)) {  // :209
        fprintf(stderr, "cannot write result PNGs\n");
        return 1;
    }
    mfsr_burst_destroy(b);
    return 0;
}
