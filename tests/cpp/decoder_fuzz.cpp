// decoder_fuzz.cpp -- the image readers of the library (csrc/jtx_jpeg.cpp, csrc/jtx_exr.cpp, csrc/jtx_png.cpp) compiled for the HOST with
// AddressSanitizer + UBSan and fed corrupted copies of valid files: an asset is untrusted input, a bad one must end in an
// error code or a picture, never in a memory error.  Built and run by tests/test_decoder_fuzz_cpu.py (CPU only).
#include "../../include/jtx_mi.h"
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <fstream>
#include <iterator>

int jtx_capi_fail(const std::string &) { return 1; }          // the library's error channel (jtx_capi.hip), stubbed

static uint32_t rngState = 12345u;
static uint32_t rnd() { rngState = rngState * 747796405u + 2891336453u; uint32_t w = ((rngState >> ((rngState >> 28u) + 4u)) ^ rngState) * 277803737u; return (w >> 22u) ^ w; }

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 200;
    long ok = 0, bad = 0;
    for (int a = 2; a < argc; ++a) {
        std::ifstream f(argv[a], std::ios::binary);
        std::vector<uint8_t> orig((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (orig.empty()) { std::printf("cannot read %s\n", argv[a]); return 2; }
        const bool exr = orig.size() > 4 && orig[0] == 0x76 && orig[1] == 0x2f;
        for (int r = 0; r < rounds; ++r) {
            std::vector<uint8_t> b = orig;
            const int kind = r % 4;
            if (kind == 0) { const int n = 1 + rnd() % 8; for (int i = 0; i < n; ++i) b[rnd() % b.size()] = (uint8_t) rnd(); }          // byte noise
            else if (kind == 1) b.resize(1 + rnd() % b.size());                                                                          // truncation
            else if (kind == 2) { const size_t at = rnd() % b.size(), n = 1 + rnd() % 64; for (size_t i = at; i < at + n && i < b.size(); ++i) b[i] = 0xff; }   // a run of 0xff
            else { const size_t at = rnd() % b.size(); b.insert(b.begin() + at, (size_t) (rnd() % 16), (uint8_t) rnd()); }               // inserted bytes
            int32_t w = 0, h = 0, c = 0; int rc;
            const bool png = orig.size() > 4 && orig[0] == 0x89 && orig[1] == 'P';
            if (png) {
                rc = jtx_mi_decode_png(b.data(), (int64_t) b.size(), &w, &h, &c, nullptr, 0);
                if (rc == 0 && (int64_t) w * h * c <= (1 << 24)) { std::vector<uint8_t> out((size_t) w * h * c); rc = jtx_mi_decode_png(b.data(), (int64_t) b.size(), &w, &h, &c, out.data(), (int64_t) out.size()); }
            } else if (exr) {
                rc = jtx_mi_decode_exr(b.data(), (int64_t) b.size(), &w, &h, nullptr, 0);
                if (rc == 0 && (int64_t) w * h <= (1 << 22)) { std::vector<float> out((size_t) 4 * w * h); rc = jtx_mi_decode_exr(b.data(), (int64_t) b.size(), &w, &h, out.data(), (int64_t) out.size()); }
            } else {
                rc = jtx_mi_decode_jpeg(b.data(), (int64_t) b.size(), &w, &h, &c, nullptr, 0);
                if (rc == 0 && (int64_t) w * h * c <= (1 << 24)) { std::vector<uint8_t> out((size_t) w * h * c); rc = jtx_mi_decode_jpeg(b.data(), (int64_t) b.size(), &w, &h, &c, out.data(), (int64_t) out.size()); }
            }
            if (rc == 0) ++ok; else ++bad;
        }
    }
    std::printf("decoded %ld, refused %ld\n", ok, bad);
    return 0;
}
