/* oracle/ref_stb.c -- CHECKER ONLY (test infrastructure): the reference's own image decoder.
 * ext/stb/stb_image.h is vendored in the reference tree and self-contained, so it is compiled where it lies
 * (-I /root/reference/ext/stb, see the Makefile's `ref` target) into oracle/_ref/libref_stb.so.  Used by
 * tests/test_jpeg_cpu.py (container only) and by tests/golden/make_jpeg_golden.py to produce the committed vectors the
 * product's decoder (csrc/jtx_jpeg.cpp) is checked against.  Never linked into the product. */
#define STB_IMAGE_IMPLEMENTATION
#define STBI_NO_STDIO
#include "stb_image.h"

unsigned char *ref_stbi_load_from_memory(const unsigned char *b, int n, int *x, int *y, int *c) { return stbi_load_from_memory(b, n, x, y, c, 0); }
float *ref_stbi_loadf_from_memory(const unsigned char *b, int n, int *x, int *y, int *c) { return stbi_loadf_from_memory(b, n, x, y, c, 0); }
void ref_stbi_free(void *p) { stbi_image_free(p); }
const char *ref_stbi_failure(void) { return stbi_failure_reason(); }
