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
//   * image I/O (apps/image_io.hpp): PNG (zlib), binary PGM/PPM and baseline JPEG (the "car" burst) in, PNG out.
//   * MFSR_GPUS=n in the environment (argv stays the reference's) shards the burst over n GPUs from this ONE process:
//     mfsr_dist_group (include/mfsr_dist.h) -- one worker thread per GPU inside the library, alignment sharded over
//     frames, fuse over HR row stripes, peer copies over xGMI; the u16 result is bit-identical to the 1-GPU burst.
//     MFSR_VIRTUAL_RANKS=1 puts all n ranks on device 0 (test rehearsal on a one-GPU box).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/mfsr.h"
#include "../include/mfsr_dist.h"
#include "image_io.hpp"

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
            fprintf(stderr, "cannot read frame %d of '%s' (%s): PNG, binary PNM, baseline JPEG or uncompressed TIFF expected\n", i, inputName.c_str(), buf);
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

    // re-mosaic 8-bit RGB (or gray) to a 12-bit RGGB raw frame; 16-bit sources (TIFF) keep their upper 12 bits, and a
    // single-channel 16-bit frame IS the raw frame (an RGGB mosaic as the sensor delivers it)
    bool wide = !imgs[0].px16.empty();
    for (int k = 0; k < num_images; k++)
        if (wide != !imgs[k].px16.empty() || imgs[k].ch != imgs[0].ch) {
            fprintf(stderr, "frames differ in sample depth or channel count\n");
            return 1;
        }
    std::vector<std::vector<uint16_t>> raws(num_images, std::vector<uint16_t>((size_t)W * H));
    for (int k = 0; k < num_images; k++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const size_t o = ((size_t)y * imgs[k].w + x) * imgs[k].ch;
                const int c = imgs[k].ch >= 3 ? (y & 1) + (x & 1) : 0;  // RGGB
                raws[k][(size_t)y * W + x] = wide ? (uint16_t)(imgs[k].px16[o + c] >> 4) : (uint16_t)(imgs[k].px[o + c] * 16);
            }

    mfsr_config cfg;
    MFSR_OK_OR_DIE(mfsr_config_default(&cfg, W, H, num_images, scale, 0));
    cfg.lkIterations = iterations;
    cfg.preAlign = 1;  // hand-held bursts: base shift + rotation per frame before the tile tracker (the bundled city frames
                       // are rotated by up to 15 degrees, test_opencv/main.cpp:1896)
    const float whiteLevel = wide ? 4095.0f : 4080.0f;  // 255 * 16, or the 12 bits kept of a 16-bit sample
    for (int c = 0; c < 3; c++) {
        cfg.black[c] = 0.0f;
        cfg.white[c] = whiteLevel;
    }
    cfg.maxVal = whiteLevel;
    const int hrW = W * scale, hrH = H * scale;
    const int start_i = num_times - real_times;
    std::chrono::steady_clock::time_point t0;
    std::vector<uint8_t> h8((size_t)hrW * hrH * 3), h8s(h8.size());
    uint8_t *d8 = nullptr, *d8s = nullptr;

    int gpus = 1;
    if (const char* e = getenv("MFSR_GPUS")) gpus = atoi(e);
    if (gpus > 1) {
        // ---- the burst sharded over `gpus` GPUs from this one process ------------------------------------------------
        const bool virt = getenv("MFSR_VIRTUAL_RANKS") && getenv("MFSR_VIRTUAL_RANKS")[0] == '1';
        if (!virt && mfsr_device_count() < gpus) {
            fprintf(stderr, "MFSR_GPUS=%d but only %d HIP device(s) are visible\n", gpus, mfsr_device_count());
            return 1;
        }
        std::vector<int> devs(gpus);
        for (int r = 0; r < gpus; r++) devs[r] = virt ? 0 : r;
        const size_t dwsBytes = mfsr_dist_workspace_bytes(&cfg, gpus);
        if (!dwsBytes) {
            fprintf(stderr, "mfsr_dist_workspace_bytes: invalid configuration\n");
            return 1;
        }
        std::vector<void*> dws(gpus, nullptr);
        std::vector<const uint16_t*> table((size_t)gpus * num_images, nullptr);  // row r: rank r's frames + the reference
        std::vector<int*> dstatus(gpus, nullptr);
        for (int r = 0; r < gpus; r++) {
            HIP_OK(hipSetDevice(devs[r]));
            HIP_OK(hipMalloc(&dws[r], dwsBytes));
            HIP_OK(hipMalloc((void**)&dstatus[r], sizeof(int)));
            for (int k = 0; k < num_images; k++) {
                if (k % gpus != r && k != cfg.reference) continue;
                uint16_t* p = nullptr;
                HIP_OK(hipMalloc((void**)&p, (size_t)W * H * 2));
                HIP_OK(hipMemcpy(p, raws[k].data(), (size_t)W * H * 2, hipMemcpyHostToDevice));  // :172 upload
                table[(size_t)r * num_images + k] = p;
            }
        }
        HIP_OK(hipSetDevice(devs[0]));
        uint16_t* d16 = nullptr;
        HIP_OK(hipMalloc((void**)&d16, (size_t)hrW * hrH * 6));
        mfsr_dist_group* g = nullptr;
        MFSR_OK_OR_DIE(mfsr_dist_group_create(&g, &cfg, gpus, devs.data(), dws.data(), dwsBytes));
        bool wholeFrames = false;
        for (int rep = 0; rep < num_times; rep++) {
            if (rep == start_i) {
                MFSR_OK_OR_DIE(mfsr_dist_group_synchronize(g, nullptr));
                t0 = std::chrono::steady_clock::now();
            }
            MFSR_OK_OR_DIE(mfsr_dist_group_process_burst(g, table.data(), MFSR_DIST_STRIPES, d16, dstatus.data(), nullptr));
            if (rep == 0 && !wholeFrames) {  // a flow beyond the raw halo of the stripe exchange: exchange whole raw frames
                MFSR_OK_OR_DIE(mfsr_dist_group_synchronize(g, nullptr));
                int st = 0;
                HIP_OK(hipMemcpy(&st, dstatus[0], sizeof(int), hipMemcpyDeviceToHost));
                if (st != 0) {
                    MFSR_OK_OR_DIE(mfsr_dist_group_set_raw_halo(g, H));
                    wholeFrames = true;
                    rep = -1;
                }
            }
        }
        MFSR_OK_OR_DIE(mfsr_dist_group_synchronize(g, nullptr));
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%g sec\n", sec);                                               // :205
        printf("%g FPS\n", (double)(num_images * real_times) / sec);           // :206
        std::vector<uint16_t> h16((size_t)hrW * hrH * 3);
        HIP_OK(hipMemcpy(h16.data(), d16, h16.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h16.size(); i++) h8[i] = (uint8_t)(((unsigned)h16[i] * 255u + 32767u) / 65535u);
        HIP_OK(hipMalloc((void**)&d8, h8.size()));
        HIP_OK(hipMalloc((void**)&d8s, h8.size()));
        HIP_OK(hipMemcpy(d8, h8.data(), h8.size(), hipMemcpyHostToDevice));
        MFSR_OK_OR_DIE(mfsr_sharpenImg2(d8, d8s, hrH, hrW, 3, hrW * 3, hrW * 3, nullptr));
        HIP_OK(hipMemcpy(h8s.data(), d8s, h8s.size(), hipMemcpyDeviceToHost));
        mfsr_dist_group_destroy(g);
        if (!write_png(inputName + "_" + optFlowName + "_sr_result.png", h8.data(), hrW, hrH, 3) ||   // :207
            !write_png(inputName + "_" + optFlowName + "_sr2_result.png", h8s.data(), hrW, hrH, 3)) {  // :209
            fprintf(stderr, "cannot write result PNGs\n");
            return 1;
        }
        return 0;
    }

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
    HIP_OK(hipMalloc((void**)&d8, (size_t)hrW * hrH * 3));
    HIP_OK(hipMalloc((void**)&d8s, (size_t)hrW * hrH * 3));
    MFSR_OK_OR_DIE(mfsr_quantize((const mfsr_float3*)outF, 12 * hrW, nullptr, d8, hrW, hrH, 255.0f, nullptr));
    MFSR_OK_OR_DIE(mfsr_sharpenImg2(d8, d8s, hrH, hrW, 3, hrW * 3, hrW * 3, nullptr));
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
