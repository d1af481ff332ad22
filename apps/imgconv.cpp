// imgconv -- decode an image with the CLI's readers (apps/image_io.hpp: PNG, binary PGM/PPM, baseline JPEG) and write it as
// a binary PPM/PGM.  Host-only helper: it lets the CPU test suite check the readers against an independent decoder, and
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
    if (img.ch != 1 && img.ch != 3) {  // drop alpha / expand gray+alpha
        std::vector<uint8_t> px((size_t)img.w * img.h * (img.ch == 2 ? 1 : 3));
        const int oc = img.ch == 2 ? 1 : 3;
        for (size_t i = 0; i < (size_t)img.w * img.h; i++)
            for (int c = 0; c < oc; c++) px[i * oc + c] = img.px[i * img.ch + c];
        img.px.swap(px);
        img.ch = oc;
    }
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 1;
    fprintf(f, "P%d\n%d %d\n255\n", img.ch == 1 ? 5 : 6, img.w, img.h);
    fwrite(img.px.data(), 1, img.px.size(), f);
    fclose(f);
    return 0;
}
