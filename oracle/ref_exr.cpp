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
