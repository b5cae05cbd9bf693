/* oracle/ref_exr.cpp -- CHECKER ONLY (test infrastructure): the reference's own EXR reader.
 * ext/tinyexr/tinyexr.h and ext/stb/*.h are vendored in the reference tree and self-contained; they are compiled where
 * they lie, configured as src/image.cpp:1-9 configures them (stb's zlib, no miniz), into oracle/_ref/libref_exr.so (the
 * Makefile's `ref` target).  Used by tests/test_exr_cpu.py (container only) and tests/golden/make_exr_golden.py to produce
 * the committed vectors the product's reader (csrc/jtx_exr.cpp) is checked against.  Never linked into the product. */
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include "stb_image_write.h"
#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"
#define TINYEXR_USE_MINIZ 0
#define TINYEXR_USE_STB_ZLIB 1
#define TINYEXR_IMPLEMENTATION
#include "tinyexr.h"

extern "C" int ref_exr_from_memory(const unsigned char *b, size_t n, float **rgba, int *w, int *h, const char **err) {
    return LoadEXRFromMemory(rgba, w, h, b, n, err);
}
extern "C" int ref_exr_from_file(const char *path, float **rgba, int *w, int *h, const char **err) { return LoadEXR(rgba, w, h, path, err); }
extern "C" void ref_exr_free(float *p) { free(p); }

/* the reference's WRITER, to make golden files in compressions this repo's own test writer does not produce (PIZ):
 * interleaved RGB(A) / grey floats -> scan-line EXR, channels (A) B G R as tinyexr's own SaveEXRToMemory lays them out */
extern "C" size_t ref_exr_save(const float *data, int w, int h, int comps, int fp16, int compression, unsigned char **out, const char **err) {
    EXRHeader header; InitEXRHeader(&header);
    header.compression_type = compression;
    EXRImage image; InitEXRImage(&image);
    image.num_channels = comps;
    std::vector<std::vector<float>> planes((size_t) comps, std::vector<float>((size_t) w * h));
    for (size_t i = 0; i < (size_t) w * h; ++i) for (int c = 0; c < comps; ++c) planes[(size_t) c][i] = data[(size_t) comps * i + c];
    float *ptr[4] = {0, 0, 0, 0};
    const char *names4[4] = {"A", "B", "G", "R"}, *names3[3] = {"B", "G", "R"};
    for (int c = 0; c < comps; ++c) ptr[c] = planes[(size_t) (comps == 1 ? 0 : comps - 1 - c)].data();      /* file order (A) B G R */
    image.images = reinterpret_cast<unsigned char **>(ptr); image.width = w; image.height = h;
    header.num_channels = comps;
    header.channels = static_cast<EXRChannelInfo *>(malloc(sizeof(EXRChannelInfo) * (size_t) comps));
    header.pixel_types = static_cast<int *>(malloc(sizeof(int) * (size_t) comps));
    header.requested_pixel_types = static_cast<int *>(malloc(sizeof(int) * (size_t) comps));
    for (int c = 0; c < comps; ++c) {
        memset(&header.channels[c], 0, sizeof(EXRChannelInfo));
        strncpy(header.channels[c].name, comps == 4 ? names4[c] : comps == 3 ? names3[c] : "Y", 255);
        header.pixel_types[c] = TINYEXR_PIXELTYPE_FLOAT;
        header.requested_pixel_types[c] = fp16 ? TINYEXR_PIXELTYPE_HALF : TINYEXR_PIXELTYPE_FLOAT;
    }
    const size_t n = SaveEXRImageToMemory(&image, &header, out, err);
    free(header.channels); free(header.pixel_types); free(header.requested_pixel_types);
    return n;
}
extern "C" void ref_exr_free_bytes(unsigned char *p) { free(p); }
