// jtx_jpeg.cpp -- JPEG -> 8-bit samples, host only.  Part of the ingestion row.
//
// The reference decodes embedded glTF images with stbi_loadf_from_memory (image.cpp:97; ext/stb/stb_image.h), and
// helmet.glb -- the one textured asset it ships -- carries four baseline JPEGs (4:2:0) and one PROGRESSIVE JPEG
// (the metallic-roughness map).  Texel values feed the shading bit for bit, so this decoder follows the same
// arithmetic the reference's decoder uses, not just the JPEG standard:
//   * entropy decoding, progressive refinement and dequantisation: ITU T.81 (baseline + progressive Huffman, 8-bit,
//     restart intervals); coefficients are kept as 16-bit values, value x quantiser truncated to short as stb does
//   * the 8x8 inverse DCT is the fixed-point "jidctint / ISLOW" variant stb_image uses (12-bit constants, a column pass
//     that keeps 2 extra bits and shortcuts all-zero AC columns, row pass rounding at 1 << 17 with the +128 level shift)
//   * chroma upsampling: stb's JFIF-centred triangle filters (h2v1: (3a+b+2)>>2; h2v2: 3:1 vertically then
//     (3a+b+8)>>4 horizontally, edge samples (a*4+2... )>>2), nearest for other factors
//   * YCbCr -> RGB: stb's reduced-precision fixed point (20-bit, the cb term of green masked to its upper 16 bits)
// Output: 3 interleaved components for a 3-component image, 1 for greyscale (what stbi_load(.., req_comp = 0) returns).
// tests/test_jpeg_cpu.py checks it byte for byte against golden vectors produced by the reference's own stb_image.h
// (compiled where it lies by the test infrastructure, container only) incl. all five images of helmet.glb where the
// reference tree is present.
#include "../../include/jtx_mi.h"

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

int jtx_capi_fail(const std::string &msg);           // jtx_capi.hip

namespace {

const uint8_t kDezigzag[64 + 15] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};   // runs past the end land on 63

struct Huff {
    uint8_t size[257] = {}; uint16_t code[256] = {}; uint8_t values[256] = {};
    unsigned maxcode[18] = {}; int delta[17] = {};
    bool present = false;
    void build(const int *count) {
        int k = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < count[i]; ++j) { size[k++] = (uint8_t) (i + 1); if (k >= 257) throw std::runtime_error("bad huffman size list"); }
        size[k] = 0;
        unsigned c = 0; k = 0;
        for (int j = 1; j <= 16; ++j) {
            delta[j] = k - (int) c;
            if (size[k] == j) {
                while (size[k] == j) code[k++] = (uint16_t) c++;
                if (c - 1 >= (1u << j)) throw std::runtime_error("bad huffman code lengths");
            }
            maxcode[j] = c << (16 - j);
            c <<= 1;
        }
        maxcode[17] = 0xffffffffu;
        present = true;
    }
};

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, dc_pred = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0, coeff_w = 0, coeff_h = 0;
    std::vector<uint8_t> data;        // decoded plane, w2 x h2
    std::vector<short> coeff;         // progressive: 64 per block
};

struct Decoder {
    const uint8_t *p, *end;
    Huff hdc[4], hac[4];
    uint16_t dequant[4][64] = {};
    Comp comp[4];
    int ncomp = 0, width = 0, height = 0, hmax = 1, vmax = 1, mcu_x = 0, mcu_y = 0;
    bool progressive = false, jfif = false;
    int app14 = -1, restart_interval = 0;
    // scan state
    int scan_n = 0, order[4] = {}, spec_start = 0, spec_end = 0, succ_high = 0, succ_low = 0, eob_run = 0, todo = 0;
    uint32_t code_buffer = 0; int code_bits = 0; bool nomore = false; int marker = -1;
    int rgb = 0;

    Decoder(const uint8_t *b, size_t n) : p(b), end(b + n) {}
    int get8() { return p < end ? *p++ : 0; }
    int get16() { const int a = get8(); return (a << 8) | get8(); }

    // ---- bit reader (bytes 0xFF00 are stuffed 0xFF; any other marker ends the entropy-coded segment) ----
    void grow() {
        do {
            unsigned b = nomore ? 0u : (unsigned) get8();
            if (b == 0xff) {
                int c = get8();
                while (c == 0xff) c = get8();
                if (c != 0) { marker = c; nomore = true; return; }
            }
            code_buffer |= b << (24 - code_bits);
            code_bits += 8;
        } while (code_bits <= 24);
    }
    int huffDecode(const Huff &h) {
        if (code_bits < 16) grow();
        const unsigned temp = code_buffer >> 16;
        int k = 1;
        while (k <= 16 && temp >= h.maxcode[k]) ++k;
        if (k == 17 || k > code_bits) { code_bits -= 16; return -1; }
        const int c = (int) ((code_buffer >> (32 - k)) & ((1u << k) - 1)) + h.delta[k];
        if (c < 0 || c >= 256) return -1;
        code_bits -= k; code_buffer <<= k;
        return h.values[c];
    }
    int extendReceive(int n) {                      // n bits, sign-extended as T.81 F.2.2.1
        if (code_bits < n) grow();
        if (code_bits < n) return 0;
        const unsigned v = code_buffer >> (32 - n);
        code_buffer <<= n; code_bits -= n;
        return (int) v < (1 << (n - 1)) ? (int) v - (1 << n) + 1 : (int) v;
    }
    int getBits(int n) {
        if (code_bits < n) grow();
        if (code_bits < n) return 0;
        const unsigned v = code_buffer >> (32 - n);
        code_buffer <<= n; code_bits -= n;
        return (int) v;
    }
    int getBit() { return getBits(1); }

    void resetScan() {
        code_bits = 0; code_buffer = 0; nomore = false;
        for (auto &c : comp) c.dc_pred = 0;
        marker = -1;
        todo = restart_interval ? restart_interval : 0x7fffffff;
        eob_run = 0;
    }

    // ---- blocks ----
    void blockBaseline(short *data, const Huff &dc, const Huff &ac, Comp &c, const uint16_t *dq) {
        const int t = huffDecode(dc);
        if (t < 0 || t > 15) throw std::runtime_error("bad huffman code");
        std::memset(data, 0, 64 * sizeof(short));
        const int diff = t ? extendReceive(t) : 0;
        c.dc_pred += diff;
        data[0] = (short) ((unsigned) c.dc_pred * (unsigned) dq[0]);              // unsigned: a hostile stream must not overflow a signed int
        int k = 1;
        do {
            const int rs = huffDecode(ac);
            if (rs < 0) throw std::runtime_error("bad huffman code");
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) { if (rs != 0xf0) break; k += 16; }
            else { k += r; const unsigned zig = kDezigzag[k++]; data[zig] = (short) (extendReceive(s) * dq[zig]); }
        } while (k < 64);
    }
    void blockProgDC(short *data, const Huff &dc, Comp &c) {
        if (spec_end != 0) throw std::runtime_error("can't merge dc and ac");
        if (succ_high == 0) {
            std::memset(data, 0, 64 * sizeof(short));
            const int t = huffDecode(dc);
            if (t < 0 || t > 15) throw std::runtime_error("bad huffman code");
            const int diff = t ? extendReceive(t) : 0;
            c.dc_pred += diff;
            data[0] = (short) ((unsigned) c.dc_pred << succ_low);
        } else if (getBit()) data[0] += (short) (1 << succ_low);
    }
    void blockProgAC(short *data, const Huff &ac) {
        if (spec_start == 0) throw std::runtime_error("can't merge dc and ac");
        if (succ_high == 0) {
            const int shift = succ_low;
            if (eob_run) { --eob_run; return; }
            int k = spec_start;
            do {
                const int rs = huffDecode(ac);
                if (rs < 0) throw std::runtime_error("bad huffman code");
                const int s = rs & 15; int r = rs >> 4;
                if (s == 0) {
                    if (r < 15) { eob_run = 1 << r; if (r) eob_run += getBits(r); --eob_run; break; }
                    k += 16;
                } else { k += r; const unsigned zig = kDezigzag[k++]; data[zig] = (short) (extendReceive(s) * (1 << shift)); }
            } while (k <= spec_end);
        } else {
            const short bit = (short) (1 << succ_low);
            auto refine = [&](short *q) {
                if (getBit() && (*q & bit) == 0) { if (*q > 0) *q += bit; else *q -= bit; }
            };
            if (eob_run) {
                --eob_run;
                for (int k = spec_start; k <= spec_end; ++k) { short *q = &data[kDezigzag[k]]; if (*q != 0) refine(q); }
            } else {
                int k = spec_start;
                do {
                    const int rs = huffDecode(ac);
                    if (rs < 0) throw std::runtime_error("bad huffman code");
                    int s = rs & 15, r = rs >> 4;
                    if (s == 0) {
                        if (r < 15) { eob_run = (1 << r) - 1; if (r) eob_run += getBits(r); r = 64; }
                    } else {
                        if (s != 1) throw std::runtime_error("bad huffman code");
                        s = getBit() ? bit : -bit;
                    }
                    while (k <= spec_end) {
                        short *q = &data[kDezigzag[k++]];
                        if (*q != 0) refine(q);
                        else { if (r == 0) { *q = (short) s; break; } --r; }
                    }
                } while (k <= spec_end);
            }
        }
    }

    // ---- the fixed-point inverse DCT of the reference's decoder ----
    static int f2f(double x) { return (int) (x * 4096 + 0.5); }
    static uint8_t clamp8(int x) { return (unsigned) x > 255u ? (x < 0 ? 0 : 255) : (uint8_t) x; }
    // (corrupted coefficients can overflow the 32-bit intermediates; stb_image wraps there, and so does this: unsigned arithmetic)
    static int wadd(int a, int b) { return (int) ((unsigned) a + (unsigned) b); }
    static int wsub(int a, int b) { return (int) ((unsigned) a - (unsigned) b); }
    static int wmul(int a, int b) { return (int) ((unsigned) a * (unsigned) b); }
    static void idct1d(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7, int &x0, int &x1, int &x2, int &x3,
                       int &t0, int &t1, int &t2, int &t3) {
        static const int c0541 = f2f(0.5411961f), cm1847 = f2f(-1.847759065f), c0765 = f2f(0.765366865f), c1175 = f2f(1.175875602f),
                         c0298 = f2f(0.298631336f), c2053 = f2f(2.053119869f), c3072 = f2f(3.072711026f), c1501 = f2f(1.501321110f),
                         cm0899 = f2f(-0.899976223f), cm2562 = f2f(-2.562915447f), cm1961 = f2f(-1.961570560f), cm0390 = f2f(-0.390180644f);
        int p1, p2, p3, p4, p5;
        p2 = s2; p3 = s6;
        p1 = wmul(wadd(p2, p3), c0541);
        t2 = wadd(p1, wmul(p3, cm1847));
        t3 = wadd(p1, wmul(p2, c0765));
        p2 = s0; p3 = s4;
        t0 = wmul(wadd(p2, p3), 4096);
        t1 = wmul(wsub(p2, p3), 4096);
        x0 = wadd(t0, t3); x3 = wsub(t0, t3); x1 = wadd(t1, t2); x2 = wsub(t1, t2);
        t0 = s7; t1 = s5; t2 = s3; t3 = s1;
        p3 = wadd(t0, t2); p4 = wadd(t1, t3); p1 = wadd(t0, t3); p2 = wadd(t1, t2);
        p5 = wmul(wadd(p3, p4), c1175);
        t0 = wmul(t0, c0298); t1 = wmul(t1, c2053); t2 = wmul(t2, c3072); t3 = wmul(t3, c1501);
        p1 = wadd(p5, wmul(p1, cm0899)); p2 = wadd(p5, wmul(p2, cm2562)); p3 = wmul(p3, cm1961); p4 = wmul(p4, cm0390);
        t3 = wadd(t3, wadd(p1, p4)); t2 = wadd(t2, wadd(p2, p3)); t1 = wadd(t1, wadd(p2, p4)); t0 = wadd(t0, wadd(p1, p3));
    }
    static void idct(uint8_t *out, int stride, const short *d) {
        int val[64];
        for (int i = 0; i < 8; ++i) {
            const short *c = d + i; int *v = val + i;
            if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
                const int dc = c[0] * 4;
                v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
            } else {
                int x0, x1, x2, x3, t0, t1, t2, t3;
                idct1d(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56], x0, x1, x2, x3, t0, t1, t2, t3);
                x0 = wadd(x0, 512); x1 = wadd(x1, 512); x2 = wadd(x2, 512); x3 = wadd(x3, 512);
                v[0] = wadd(x0, t3) >> 10; v[56] = wsub(x0, t3) >> 10; v[8] = wadd(x1, t2) >> 10; v[48] = wsub(x1, t2) >> 10;
                v[16] = wadd(x2, t1) >> 10; v[40] = wsub(x2, t1) >> 10; v[24] = wadd(x3, t0) >> 10; v[32] = wsub(x3, t0) >> 10;
            }
        }
        for (int i = 0; i < 8; ++i) {
            const int *v = val + 8 * i; uint8_t *o = out + (size_t) i * stride;
            int x0, x1, x2, x3, t0, t1, t2, t3;
            idct1d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], x0, x1, x2, x3, t0, t1, t2, t3);
            const int bias = 65536 + (128 << 17);
            x0 = wadd(x0, bias); x1 = wadd(x1, bias); x2 = wadd(x2, bias); x3 = wadd(x3, bias);
            o[0] = clamp8(wadd(x0, t3) >> 17); o[7] = clamp8(wsub(x0, t3) >> 17); o[1] = clamp8(wadd(x1, t2) >> 17); o[6] = clamp8(wsub(x1, t2) >> 17);
            o[2] = clamp8(wadd(x2, t1) >> 17); o[5] = clamp8(wsub(x2, t1) >> 17); o[3] = clamp8(wadd(x3, t0) >> 17); o[4] = clamp8(wsub(x3, t0) >> 17);
        }
    }

    // ---- markers ----
    void parseDQT(int len) {
        len -= 2;
        while (len > 0) {
            const int q = get8(), prec = q >> 4, t = q & 15;
            if ((prec != 0 && prec != 1) || t > 3) throw std::runtime_error("bad DQT");
            for (int i = 0; i < 64; ++i) dequant[t][kDezigzag[i]] = (uint16_t) (prec ? get16() : get8());
            len -= prec ? 129 : 65;
        }
    }
    void parseDHT(int len) {
        len -= 2;
        while (len > 0) {
            const int q = get8(), tc = q >> 4, th = q & 15;
            if (tc > 1 || th > 3) throw std::runtime_error("bad DHT header");
            int sizes[16], n = 0;
            for (int i = 0; i < 16; ++i) { sizes[i] = get8(); n += sizes[i]; }
            if (n > 256) throw std::runtime_error("bad DHT header");
            Huff &h = tc == 0 ? hdc[th] : hac[th];
            h.build(sizes);
            for (int i = 0; i < n; ++i) h.values[i] = (uint8_t) get8();
            len -= 17 + n;
        }
    }
    void parseSOF(int len, bool prog) {
        (void) len;
        progressive = prog;
        if (get8() != 8) throw std::runtime_error("only 8-bit JPEG");
        height = get16(); width = get16();
        if (height == 0 || width == 0) throw std::runtime_error("0-size JPEG");
        ncomp = get8();
        if (ncomp != 1 && ncomp != 3 && ncomp != 4) throw std::runtime_error("bad component count");
        rgb = 0;
        static const char tag[3] = {'R', 'G', 'B'};
        for (int i = 0; i < ncomp; ++i) {
            Comp &c = comp[i];
            c.id = get8();
            if (ncomp == 3 && c.id == tag[i]) ++rgb;
            const int q = get8(); c.h = q >> 4; c.v = q & 15;
            if (!c.h || c.h > 4 || !c.v || c.v > 4) throw std::runtime_error("bad sampling factor");
            c.tq = get8(); if (c.tq > 3) throw std::runtime_error("bad quantiser index");
        }
        hmax = vmax = 1;
        for (int i = 0; i < ncomp; ++i) { if (comp[i].h > hmax) hmax = comp[i].h; if (comp[i].v > vmax) vmax = comp[i].v; }
        for (int i = 0; i < ncomp; ++i) if (hmax % comp[i].h || vmax % comp[i].v) throw std::runtime_error("bad sampling factor");
        const int mcu_w = hmax * 8, mcu_h = vmax * 8;
        mcu_x = (width + mcu_w - 1) / mcu_w; mcu_y = (height + mcu_h - 1) / mcu_h;
        for (int i = 0; i < ncomp; ++i) {
            Comp &c = comp[i];
            c.x = (width * c.h + hmax - 1) / hmax; c.y = (height * c.v + vmax - 1) / vmax;
            c.w2 = mcu_x * c.h * 8; c.h2 = mcu_y * c.v * 8;
            c.data.assign((size_t) c.w2 * c.h2, 0);
            if (progressive) { c.coeff_w = c.w2 / 8; c.coeff_h = c.h2 / 8; c.coeff.assign((size_t) c.w2 * c.h2, 0); }
        }
    }
    void parseSOS() {
        get16();
        scan_n = get8();
        if (scan_n < 1 || scan_n > 4 || scan_n > ncomp) throw std::runtime_error("bad SOS component count");
        for (int i = 0; i < scan_n; ++i) {
            const int id = get8(), q = get8();
            int which = 0;
            for (; which < ncomp; ++which) if (comp[which].id == id) break;
            if (which == ncomp) throw std::runtime_error("bad SOS component");
            comp[which].hd = q >> 4; comp[which].ha = q & 15;
            if (comp[which].hd > 3 || comp[which].ha > 3) throw std::runtime_error("bad SOS table index");
            order[i] = which;
        }
        spec_start = get8(); spec_end = get8();
        const int aa = get8(); succ_high = aa >> 4; succ_low = aa & 15;
        if (progressive) { if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) throw std::runtime_error("bad SOS"); }
        else { if (spec_start != 0 || succ_high != 0 || succ_low != 0) throw std::runtime_error("bad SOS"); spec_end = 63; }
    }
    // true while the interval counter allows going on
    bool afterMCU() {
        if (--todo <= 0) {
            if (code_bits < 24) grow();
            if (!(marker >= 0xd0 && marker <= 0xd7)) return false;
            resetScan();
        }
        return true;
    }
    void entropy() {
        for (int i = 0; i < scan_n; ++i) {                                        // an SOS naming a table no DHT defined
            const Comp &c = comp[order[i]];
            const bool needDC = !progressive || (spec_start == 0 && succ_high == 0), needAC = !progressive || spec_start != 0;
            if ((needDC && !hdc[c.hd].present) || (needAC && !hac[c.ha].present)) throw std::runtime_error("missing Huffman table");
        }
        resetScan();
        short block[64];
        if (scan_n == 1) {
            Comp &c = comp[order[0]];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i) {
                    if (!progressive) {
                        blockBaseline(block, hdc[c.hd], hac[c.ha], c, dequant[c.tq]);
                        idct(c.data.data() + (size_t) c.w2 * j * 8 + i * 8, c.w2, block);
                    } else {
                        short *d = c.coeff.data() + 64 * ((size_t) i + (size_t) j * c.coeff_w);
                        if (spec_start == 0) blockProgDC(d, hdc[c.hd], c); else blockProgAC(d, hac[c.ha]);
                    }
                    if (!afterMCU()) return;
                }
        } else {
            for (int j = 0; j < mcu_y; ++j)
                for (int i = 0; i < mcu_x; ++i) {
                    for (int k = 0; k < scan_n; ++k) {
                        Comp &c = comp[order[k]];
                        for (int y = 0; y < c.v; ++y)
                            for (int x = 0; x < c.h; ++x) {
                                const int bx = i * c.h + x, by = j * c.v + y;
                                if (!progressive) {
                                    blockBaseline(block, hdc[c.hd], hac[c.ha], c, dequant[c.tq]);
                                    idct(c.data.data() + (size_t) c.w2 * by * 8 + bx * 8, c.w2, block);
                                } else blockProgDC(c.coeff.data() + 64 * ((size_t) bx + (size_t) by * c.coeff_w), hdc[c.hd], c);
                            }
                    }
                    if (!afterMCU()) return;
                }
        }
    }
    void finishProgressive() {
        for (int n = 0; n < ncomp; ++n) {
            Comp &c = comp[n];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i) {
                    short *d = c.coeff.data() + 64 * ((size_t) i + (size_t) j * c.coeff_w);
                    for (int k = 0; k < 64; ++k) d[k] = (short) (d[k] * dequant[c.tq][k]);
                    idct(c.data.data() + (size_t) c.w2 * j * 8 + i * 8, c.w2, d);
                }
        }
    }
    int nextMarker() {
        if (marker != -1) { const int m = marker; marker = -1; return m; }
        int x = get8();
        if (x != 0xff) return -1;
        while (x == 0xff) x = get8();
        return x;
    }
    void decode() {
        if (get8() != 0xff || get8() != 0xd8) throw std::runtime_error("not a JPEG");
        bool haveFrame = false;
        int m = nextMarker();
        while (true) {
            if (m == -1) { if (p >= end) throw std::runtime_error("truncated JPEG"); m = nextMarker(); continue; }
            if (m == 0xd9) break;
            if (m == 0xda) {
                if (!haveFrame) throw std::runtime_error("SOS before SOF");
                parseSOS();
                entropy();
                if (marker == -1) {       // find the next marker behind the scan (stb: skips stray bytes)
                    while (p < end) { int x = get8(); if (x == 0xff) { int y = get8(); while (y == 0xff) y = get8(); if (y != 0 ) { marker = y; break; } } }
                    if (marker == -1) break;
                }
                m = nextMarker();
                if (m >= 0xd0 && m <= 0xd7) m = nextMarker();
                continue;
            }
            const int len = get16();
            if (len < 2) throw std::runtime_error("bad marker length");
            const uint8_t *next = p + len - 2;
            if (next > end) throw std::runtime_error("truncated JPEG");
            if (m == 0xc0 || m == 0xc1 || m == 0xc2) { parseSOF(len, m == 0xc2); haveFrame = true; }
            else if (m == 0xc4) parseDHT(len);
            else if (m == 0xdb) parseDQT(len);
            else if (m == 0xdd) { restart_interval = get16(); }
            else if (m == 0xe0 && len >= 7) { jfif = p[0] == 'J' && p[1] == 'F' && p[2] == 'I' && p[3] == 'F' && p[4] == 0; }
            else if (m == 0xee && len >= 14) { if (!std::memcmp(p, "Adobe", 6)) app14 = p[11]; }
            else if ((m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)) throw std::runtime_error("unsupported JPEG coding process");
            p = next;
            m = nextMarker();
        }
        if (!haveFrame) throw std::runtime_error("no frame in JPEG");
        if (progressive) finishProgressive();
    }

    // ---- upsampling + colour ----
    struct Resample { int hs, vs, ystep, w_lores, ypos; const uint8_t *line0, *line1; };
    static const uint8_t *resampleRow(uint8_t *out, const uint8_t *nearp, const uint8_t *farp, int w, int hs, int vs) {
        if (hs == 1 && vs == 1) return nearp;
        if (hs == 1 && vs == 2) { for (int i = 0; i < w; ++i) out[i] = (uint8_t) ((3 * nearp[i] + farp[i] + 2) >> 2); return out; }
        if (hs == 2 && vs == 1) {
            const uint8_t *in = nearp;
            if (w == 1) { out[0] = out[1] = in[0]; return out; }
            out[0] = in[0];
            out[1] = (uint8_t) ((in[0] * 3 + in[1] + 2) >> 2);
            int i;
            for (i = 1; i < w - 1; ++i) {
                const int n = 3 * in[i] + 2;
                out[i * 2] = (uint8_t) ((n + in[i - 1]) >> 2);
                out[i * 2 + 1] = (uint8_t) ((n + in[i + 1]) >> 2);
            }
            out[i * 2] = (uint8_t) ((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
            out[i * 2 + 1] = in[w - 1];
            return out;
        }
        if (hs == 2 && vs == 2) {
            if (w == 1) { out[0] = out[1] = (uint8_t) ((3 * nearp[0] + farp[0] + 2) >> 2); return out; }
            int t1 = 3 * nearp[0] + farp[0];
            out[0] = (uint8_t) ((t1 + 2) >> 2);
            for (int i = 1; i < w; ++i) {
                const int t0 = t1;
                t1 = 3 * nearp[i] + farp[i];
                out[i * 2 - 1] = (uint8_t) ((3 * t0 + t1 + 8) >> 4);
                out[i * 2] = (uint8_t) ((3 * t1 + t0 + 8) >> 4);
            }
            out[w * 2 - 1] = (uint8_t) ((t1 + 2) >> 2);
            return out;
        }
        for (int i = 0; i < w; ++i) for (int j = 0; j < hs; ++j) out[i * hs + j] = nearp[i];
        return out;
    }
    static int f2fix(float x) { return ((int) (x * 4096.0f + 0.5f)) << 8; }
    static uint8_t blinn(uint8_t x, uint8_t y) { const unsigned t = x * y + 128; return (uint8_t) ((t + (t >> 8)) >> 8); }

    void output(std::vector<uint8_t> &out, int &ocomp) {
        const int n = ncomp >= 3 ? 3 : 1;
        ocomp = n;
        const bool isRgb = ncomp == 3 && (rgb == 3 || (app14 == 0 && !jfif));
        const int decode_n = ncomp;
        out.assign((size_t) n * width * height, 0);
        Resample rs[4];
        std::vector<uint8_t> line[4];
        for (int k = 0; k < decode_n; ++k) {
            Resample &r = rs[k];
            line[k].assign((size_t) width + 3 + 8, 0);
            r.hs = hmax / comp[k].h; r.vs = vmax / comp[k].v; r.ystep = r.vs >> 1;
            r.w_lores = (width + r.hs - 1) / r.hs; r.ypos = 0;
            r.line0 = r.line1 = comp[k].data.data();
        }
        const int cr_r = f2fix(1.40200f), cr_g = -f2fix(0.71414f), cb_g = -f2fix(0.34414f), cb_b = f2fix(1.77200f);
        for (int j = 0; j < height; ++j) {
            uint8_t *o = out.data() + (size_t) n * width * j;
            const uint8_t *co[4] = {nullptr, nullptr, nullptr, nullptr};
            for (int k = 0; k < decode_n; ++k) {
                Resample &r = rs[k];
                const bool ybot = r.ystep >= (r.vs >> 1);
                co[k] = resampleRow(line[k].data(), ybot ? r.line1 : r.line0, ybot ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
                if (++r.ystep >= r.vs) {
                    r.ystep = 0; r.line0 = r.line1;
                    if (++r.ypos < comp[k].y) r.line1 += comp[k].w2;
                }
            }
            if (n == 1) { std::memcpy(o, co[0], (size_t) width); continue; }
            auto ycc = [&](uint8_t *dst) {
                for (int i = 0; i < width; ++i) {
                    const int yf = (co[0][i] << 20) + (1 << 19);
                    const int cr = co[2][i] - 128, cb = co[1][i] - 128;
                    int r = yf + cr * cr_r;
                    int g = yf + cr * cr_g + (int) (((unsigned) (cb * cb_g)) & 0xffff0000u);
                    int b = yf + cb * cb_b;
                    r >>= 20; g >>= 20; b >>= 20;
                    dst[3 * i] = clamp8(r); dst[3 * i + 1] = clamp8(g); dst[3 * i + 2] = clamp8(b);
                }
            };
            if (ncomp == 3) {
                if (isRgb) for (int i = 0; i < width; ++i) { o[3 * i] = co[0][i]; o[3 * i + 1] = co[1][i]; o[3 * i + 2] = co[2][i]; }
                else ycc(o);
            } else {              // 4 components
                if (app14 == 0) for (int i = 0; i < width; ++i) { const uint8_t m = co[3][i]; o[3 * i] = blinn(co[0][i], m); o[3 * i + 1] = blinn(co[1][i], m); o[3 * i + 2] = blinn(co[2][i], m); }
                else if (app14 == 2) { ycc(o); for (int i = 0; i < width; ++i) { const uint8_t m = co[3][i]; for (int c = 0; c < 3; ++c) o[3 * i + c] = blinn((uint8_t) (255 - o[3 * i + c]), m); } }
                else ycc(o);
            }
        }
    }
};

} // namespace

extern "C" int jtx_mi_decode_jpeg(const uint8_t *bytes, int64_t num_bytes, int32_t *width, int32_t *height, int32_t *components,
                                  uint8_t *out, int64_t capacity) {
    try {
        if (!bytes || num_bytes < 4 || !width || !height || !components) throw std::runtime_error("null argument");
        Decoder d(bytes, (size_t) num_bytes);
        if (!out) {                                  // size query: headers only would do; decode is cheap enough to keep one path honest
            const uint8_t *p = bytes + 2, *end = bytes + num_bytes;
            while (p + 4 <= end && p[0] == 0xff) {
                const int m = p[1], len = (p[2] << 8) | p[3];
                if (m == 0xc0 || m == 0xc1 || m == 0xc2) {
                    if (p + 10 > end) break;
                    *height = (p[5] << 8) | p[6]; *width = (p[7] << 8) | p[8]; *components = p[9] >= 3 ? 3 : 1;
                    return 0;
                }
                if (m == 0xda) break;
                p += 2 + len;
            }
            throw std::runtime_error("no frame header in JPEG");
        }
        d.decode();
        std::vector<uint8_t> px; int oc = 0;
        d.output(px, oc);
        if ((int64_t) px.size() > capacity) throw std::runtime_error("jtx_mi_decode_jpeg: output buffer too small");
        std::memcpy(out, px.data(), px.size());
        *width = d.width; *height = d.height; *components = oc;
        return 0;
    } catch (const std::exception &e) { return jtx_capi_fail(std::string("jtx_mi_decode_jpeg: ") + e.what()); }
}
