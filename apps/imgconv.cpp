// imgconv -- decode an image with the CLI's readers (apps/image_io.hpp: PNG, binary PGM/PPM, baseline JPEG, uncompressed
// TIFF) and write it as a binary PPM/PGM (16-bit sources: maxval 65535, big-endian samples).  Host-only helper: it lets the CPU test suite check the readers against an independent decoder, and
// converts a burst to a format any tool reads.   usage: imgconv in.{png,ppm,pgm,jpg} out.{ppm,pgm}
#include "image_io.hpp"

int main(int argc, char** argv)
{
    if (argc != 3) {
        fprintf(stderr, "usage: imgconv input output.ppm\n");
        return 2;
    }
    Image8 img;
    if (!read_image(argv[1], img)) {
        fprintf(stderr, "cannot decode %s\n", argv[1]);
        return 1;
    }
    const bool wide = !img.px16.empty();
    if (img.ch != 1 && img.ch != 3) {  // drop alpha / expand gray+alpha
        const int oc = img.ch == 2 ? 1 : 3;
        std::vector<uint8_t> px((size_t)img.w * img.h * oc);
        std::vector<uint16_t> px16(wide ? px.size() : 0);
        for (size_t i = 0; i < (size_t)img.w * img.h; i++)
            for (int c = 0; c < oc; c++) {
                px[i * oc + c] = img.px[i * img.ch + c];
                if (wide) px16[i * oc + c] = img.px16[i * img.ch + c];
            }
        img.px.swap(px);
        img.px16.swap(px16);
        img.ch = oc;
    }
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 1;
    fprintf(f, "P%d\n%d %d\n%d\n", img.ch == 1 ? 5 : 6, img.w, img.h, wide ? 65535 : 255);
    if (wide) {
        std::vector<uint8_t> be(img.px16.size() * 2);
        for (size_t i = 0; i < img.px16.size(); i++) {
            be[2 * i] = (uint8_t)(img.px16[i] >> 8);
            be[2 * i + 1] = (uint8_t)(img.px16[i] & 255);
        }
        fwrite(be.data(), 1, be.size(), f);
    } else {
        fwrite(img.px.data(), 1, img.px.size(), f);
    }
    fclose(f);
    return 0;
}
