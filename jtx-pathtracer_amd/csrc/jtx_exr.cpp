// jtx_exr.cpp -- host-side OpenEXR reader for textures: what TextureImage::load gets from tinyexr's LoadEXR /
// LoadEXRFromMemory (image.cpp:63-66, 81-95, 108-121): one RGBA float image, rows top to bottom.
//
// Scope = what the reference's assets use and a little more: single-part SCANLINE files, compression NONE / RLE / ZIPS /
// ZIP / PIZ (all eleven maps under assets/scenes/shaderball/maps are ZIP, three HALF channels B, G, R), HALF / FLOAT / UINT
// channels, any data window, either line order (decreasing-Y files upside down, as tinyexr returns them).  Tiled, deep, multi-part files and the lossy PXR24 / B44 / DWA blocks are
// refused with a message.  The channel mapping is tinyexr's (tinyexr.h:6638-6794): channels named R, G, B (A optional,
// 1.0 when absent); a single channel of any name is copied to all FOUR outputs.  HALF -> float is tinyexr's
// half_to_float (an exact conversion; NaN payloads shifted up by 13 bits), UINT -> float a plain conversion.
// The zlib inflate below is a plain RFC 1950 / 1951 decoder written for this file.
#include "../../include/jtx_mi.h"
#include "jtx_inflate.hpp"

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

int jtx_capi_fail(const std::string &msg);           // jtx_capi.hip: sets the thread's error text, returns 1

namespace {

using jtxz::Fail; using jtxz::fail; using jtxz::inflateZlib;

// OpenEXR's byte predictor + interleave, undone (tinyexr.h:1536-1570, from ImfZipCompressor.cpp)
void unpredict(std::vector<uint8_t> &tmp, uint8_t *dst) {
    const size_t n = tmp.size();
    for (size_t i = 1; i < n; ++i) tmp[i] = (uint8_t) ((int) tmp[i - 1] + (int) tmp[i] - 128);
    const uint8_t *t1 = tmp.data(), *t2 = tmp.data() + (n + 1) / 2;
    size_t s = 0;
    while (true) {
        if (s < n) dst[s++] = *t1++; else break;
        if (s < n) dst[s++] = *t2++; else break;
    }
}

void unRle(const uint8_t *src, size_t srcLen, std::vector<uint8_t> &out) {      // ImfRle.cpp: count < 0 -> literal run
    size_t o = 0, i = 0;
    while (i < srcLen) {
        const int c = (int8_t) src[i++];
        if (c < 0) { const size_t n = (size_t) -c; if (i + n > srcLen || o + n > out.size()) fail("EXR: bad RLE run"); std::memcpy(&out[o], src + i, n); o += n; i += n; }
        else { const size_t n = (size_t) c + 1; if (i >= srcLen || o + n > out.size()) fail("EXR: bad RLE run"); std::memset(&out[o], src[i++], n); o += n; }
    }
    if (o != out.size()) fail("EXR: RLE block shorter than the scan lines it holds");
}

uint32_t rd32(const uint8_t *p) { return (uint32_t) p[0] | (uint32_t) p[1] << 8 | (uint32_t) p[2] << 16 | (uint32_t) p[3] << 24; }
uint64_t rd64(const uint8_t *p) { return (uint64_t) rd32(p) | (uint64_t) rd32(p + 4) << 32; }

// ---- PIZ blocks (ImfPizCompressor / ImfHuf / ImfWav, as tinyexr.h:1896-3420 carries them) ----
// block = [minNonZero u16][maxNonZero u16][bitmap bytes min..max][length i32][Huffman stream]; the stream decodes to the
// block's 16-bit words channel by channel, which a 2-D Haar-like wavelet (14-bit arithmetic when every value is below 2^14,
// modulo 2^16 otherwise) and the bitmap's value table turn back into the samples.
struct PizBits {                                           // MSB-first reader over the first `nbits` bits
    const uint8_t *p; size_t nbits, pos = 0;
    bool more() const { return pos < nbits; }
    unsigned bit() { const unsigned b = (p[pos >> 3] >> (7 - (pos & 7))) & 1u; ++pos; return b; }
    unsigned bits(int k) { unsigned v = 0; while (k--) { if (pos >= nbits) fail("EXR: PIZ stream ends inside a code"); v = (v << 1) | bit(); } return v; }
};

void pizHuffman(const uint8_t *src, size_t len, std::vector<uint16_t> &out) {
    if (len < 20) fail("EXR: PIZ Huffman block too short");
    const uint32_t im = rd32(src), iM = rd32(src + 4), nBits = rd32(src + 12);
    constexpr uint32_t ENC = 65537;                          // 2^16 literals + the run-length symbol
    if (im >= ENC || iM >= ENC || im > iM) fail("EXR: bad PIZ symbol range");
    // code lengths: 6 bits each, 59..62 = a run of 2..5 zero lengths, 63 = a run of 6 + (8 bits) zero lengths
    std::vector<uint8_t> lens(ENC, 0);
    PizBits tb{src + 20, (len - 20) * 8};
    for (uint32_t i = im; i <= iM; ++i) {
        const unsigned l = tb.bits(6);
        if (l == 63u) { const unsigned z = tb.bits(8) + 6u; if (i + z > iM + 1) fail("EXR: bad PIZ zero run"); i += z - 1; }
        else if (l >= 59u) { const unsigned z = l - 59u + 2u; if (i + z > iM + 1) fail("EXR: bad PIZ zero run"); i += z - 1; }
        else lens[i] = (uint8_t) l;
    }
    const size_t tableBytes = (tb.pos + 7) / 8;
    const uint8_t *data = src + 20 + tableBytes;
    if ((size_t) nBits > (len - 20 - tableBytes) * 8) fail("EXR: PIZ stream longer than its block");
    // canonical codes (hufCanonicalCodeTable): longer codes take the numerically lower values, equal lengths ascend with the symbol
    uint64_t count[59] = {}, base[59] = {};
    for (uint32_t i = 0; i < ENC; ++i) count[lens[i]]++;
    { uint64_t c = 0; for (int l = 58; l > 0; --l) { const uint64_t nc = (c + count[l]) >> 1; base[l] = c; c = nc; } }
    std::vector<uint32_t> first(60, 0), sorted;               // symbols by (length, value)
    sorted.reserve(ENC);
    for (int l = 1; l <= 58; ++l) { first[l] = (uint32_t) sorted.size(); if (count[l]) for (uint32_t i = im; i <= iM; ++i) if (lens[i] == l) sorted.push_back(i); }
    // 12-bit look-up for the short codes
    constexpr int FAST = 12;
    std::vector<uint32_t> fast(1u << FAST, 0u);               // (symbol << 6 | length), 0 = longer code
    for (int l = 1; l <= FAST; ++l)
        for (uint64_t k = 0; k < count[l]; ++k) {
            const uint64_t code = base[l] + k;
            if (code >> l) fail("EXR: bad PIZ code table");
            const uint32_t v = (sorted[first[l] + k] << 6) | (uint32_t) l;
            const uint64_t lo = code << (FAST - l);
            for (uint64_t f = 0; f < (1ull << (FAST - l)); ++f) fast[lo + f] = v;
        }
    PizBits br{data, nBits};
    size_t o = 0;
    const size_t want = out.size();
    while (br.more()) {
        uint32_t sym = ENC;
        if (br.nbits - br.pos >= (size_t) FAST) {               // peek FAST bits
            unsigned v = 0; for (int k = 0; k < FAST; ++k) v = (v << 1) | ((br.p[(br.pos + k) >> 3] >> (7 - ((br.pos + k) & 7))) & 1u);
            const uint32_t e = fast[v];
            if (e) { sym = e >> 6; br.pos += e & 63u; }
        }
        if (sym == ENC) {                                       // bit by bit: a code of length l ends where its value reaches base[l]
            uint64_t v = 0; int l = 0;
            while (true) {
                if (!br.more() || l >= 58) fail("EXR: bad PIZ code");
                v = (v << 1) | br.bit(); ++l;
                if (count[l] && v >= base[l]) { if (v - base[l] >= count[l]) fail("EXR: bad PIZ code"); sym = sorted[first[l] + (v - base[l])]; break; }
            }
        }
        if (sym == iM) {                                        // run-length symbol: repeat the previous word (8-bit count)
            const unsigned n = br.bits(8);
            if (o == 0 || o + n > want) fail("EXR: bad PIZ run");
            for (unsigned k = 0; k < n; ++k, ++o) out[o] = out[o - 1];
        } else { if (o >= want) fail("EXR: PIZ block longer than the scan lines it holds"); out[o++] = (uint16_t) sym; }
    }
    if (o != want) fail("EXR: PIZ block shorter than the scan lines it holds");
}

inline void wdec14(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {       // tinyexr.h:1896-1909
    const int ls = (int16_t) l, hi = (int16_t) h;
    const int ai = ls + (hi & 1) + (hi >> 1);
    a = (uint16_t) (int16_t) ai; b = (uint16_t) (int16_t) (ai - hi);
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {       // tinyexr.h:1936-1944
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & 0xffff, aa = (d + bb - 0x8000) & 0xffff;
    b = (uint16_t) bb; a = (uint16_t) aa;
}
void wav2Decode(uint16_t *in, int nx, int ox, int ny, int oy, uint16_t mx) {   // tinyexr.h:2059-2176
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx;
    int p = 1, p2;
    while (p <= n) p <<= 1;
    p >>= 1; p2 = p; p >>= 1;
    while (p >= 1) {
        uint16_t *py = in, *ey = in + (ptrdiff_t) oy * (ny - p2);
        const int oy1 = oy * p, oy2 = oy * p2, ox1 = ox * p, ox2 = ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t *px = py, *ex = py + (ptrdiff_t) ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
                if (w14) { wdec14(*px, *p10, i00, i10); wdec14(*p01, *p11, i01, i11); wdec14(i00, i01, *px, *p01); wdec14(i10, i11, *p10, *p11); }
                else { wdec16(*px, *p10, i00, i10); wdec16(*p01, *p11, i01, i11); wdec16(i00, i01, *px, *p01); wdec16(i10, i11, *p10, *p11); }
            }
            if (nx & p) { uint16_t *p10 = px + oy1; if (w14) wdec14(*px, *p10, i00, *p10); else wdec16(*px, *p10, i00, *p10); *px = i00; }
        }
        if (ny & p) {
            uint16_t *px = py, *ex = py + (ptrdiff_t) ox * (nx - p2);
            for (; px <= ex; px += ox2) { uint16_t *p01 = px + ox1; if (w14) wdec14(*px, *p01, i00, *p01); else wdec16(*px, *p01, i00, *p01); *px = i00; }
        }
        p2 = p; p >>= 1;
    }
}

// one PIZ block -> the scan lines' bytes (line by line, channel by channel); words[c] = 16-bit words per sample of channel c
void unPiz(const uint8_t *src, size_t len, const std::vector<int> &words, int nx, int ny, uint8_t *dst, size_t want) {
    if (len < 4) fail("EXR: PIZ block too short");
    const unsigned minNZ = src[0] | src[1] << 8, maxNZ = src[2] | src[3] << 8;
    if (maxNZ >= 8192u) fail("EXR: bad PIZ bitmap range");
    std::vector<uint8_t> bitmap(8192, 0);
    size_t at = 4;
    if (minNZ <= maxNZ) { const size_t n = maxNZ - minNZ + 1; if (at + n > len) fail("EXR: PIZ bitmap runs past the block"); std::memcpy(&bitmap[minNZ], src + at, n); at += n; }
    else if (!(minNZ == 8191u && maxNZ == 0u)) fail("EXR: bad PIZ bitmap range");
    std::vector<uint16_t> lut(65536, 0);
    int k = 0;
    for (int i = 0; i < 65536; ++i) if (i == 0 || (bitmap[i >> 3] & (1 << (i & 7)))) lut[k++] = (uint16_t) i;
    const uint16_t maxValue = (uint16_t) (k - 1);
    if (at + 4 > len) fail("EXR: PIZ block too short");
    const uint32_t hlen = rd32(src + at); at += 4;
    if (hlen > len - at) fail("EXR: PIZ Huffman block runs past the end");
    std::vector<uint16_t> buf(want / 2);
    pizHuffman(src + at, hlen, buf);
    std::vector<size_t> start(words.size());
    { size_t o = 0; for (size_t c = 0; c < words.size(); ++c) { start[c] = o; o += (size_t) nx * ny * words[c]; } if (o != buf.size()) fail("EXR: PIZ block of the wrong size"); }
    for (size_t c = 0; c < words.size(); ++c)
        for (int j = 0; j < words[c]; ++j) wav2Decode(buf.data() + start[c] + j, nx, words[c], ny, nx * words[c], maxValue);
    for (uint16_t &v : buf) v = lut[v];
    uint8_t *o = dst;
    for (int y = 0; y < ny; ++y)
        for (size_t c = 0; c < words.size(); ++c) {
            const size_t n = (size_t) nx * words[c];
            std::memcpy(o, buf.data() + start[c] + (size_t) y * n, 2 * n); o += 2 * n;
        }
}

float halfToFloat(uint16_t h) {                            // tinyexr.h half_to_float
    uint32_t o = (uint32_t) (h & 0x7fffu) << 13;
    const uint32_t shiftedExp = 0x7c00u << 13, e = shiftedExp & o;
    o += (127u - 15u) << 23;
    if (e == shiftedExp) o += (128u - 16u) << 23;          // Inf / NaN
    else if (e == 0) {                                     // zero / denormal: renormalise
        o += 1u << 23;
        float f; std::memcpy(&f, &o, 4);
        const uint32_t mu = 113u << 23; float magic; std::memcpy(&magic, &mu, 4);
        f -= magic;
        std::memcpy(&o, &f, 4);
    }
    o |= (uint32_t) (h & 0x8000u) << 16;
    float r; std::memcpy(&r, &o, 4);
    return r;
}

struct Channel { std::string name; int type; int xs, ys; };

void decodeExr(const uint8_t *b, size_t n, int32_t *width, int32_t *height, float *out, int64_t capacity) {
    if (n < 8 || rd32(b) != 20000630u) fail("EXR: bad magic number");
    const uint32_t ver = rd32(b + 4);
    if ((ver & 0xffu) != 2) fail("EXR: unsupported file version");
    if (ver & 0x200u) fail("EXR: tiled files are not supported");
    if (ver & 0x800u) fail("EXR: deep data is not supported");
    if (ver & 0x1000u) fail("EXR: multi-part files are not supported");
    size_t p = 8;
    std::vector<Channel> ch;
    int comp = -1, lineOrder = 0;
    int32_t dw[4] = {0, 0, -1, -1};
    bool haveDw = false;
    auto cstr = [&](size_t &q) { const size_t s = q; while (q < n && b[q]) ++q; if (q >= n) fail("EXR: header runs past the end"); std::string r((const char *) b + s, q - s); ++q; return r; };
    while (true) {
        if (p >= n) fail("EXR: header runs past the end");
        if (b[p] == 0) { ++p; break; }
        const std::string name = cstr(p), type = cstr(p);
        if (p + 4 > n) fail("EXR: header runs past the end");
        const uint32_t sz = rd32(b + p); p += 4;
        if (sz > n - p) fail("EXR: attribute runs past the end");
        const uint8_t *a = b + p;
        if (name == "channels") {
            size_t q = p;
            while (q < p + sz && b[q]) {
                Channel c; c.name = cstr(q);
                if (q + 16 > p + sz) fail("EXR: bad channel list");
                c.type = (int) rd32(b + q); c.xs = (int) rd32(b + q + 8); c.ys = (int) rd32(b + q + 12); q += 16;
                ch.push_back(c);
            }
        } else if (name == "compression") { if (sz < 1) fail("EXR: bad compression attribute"); comp = a[0]; }
        else if (name == "dataWindow") { if (sz < 16) fail("EXR: bad dataWindow"); for (int i = 0; i < 4; ++i) dw[i] = (int32_t) rd32(a + 4 * i); haveDw = true; }
        else if (name == "lineOrder") { if (sz < 1) fail("EXR: bad lineOrder"); lineOrder = a[0]; }
        p += sz;
    }
    // rows are placed by the y every block carries -- and, as tinyexr does it (tinyexr.h:3868-3892: row height - 1 - y for any
    // lineOrder != 0), a DECREASING_Y file comes out upside down: a quirk of the reference's reader, reproduced
    if (!haveDw || ch.empty() || comp < 0) fail("EXR: header lacks channels / compression / dataWindow");
    if (dw[2] < dw[0] || dw[3] < dw[1]) fail("EXR: empty data window");
    const int64_t W = (int64_t) dw[2] - dw[0] + 1, H = (int64_t) dw[3] - dw[1] + 1;
    if (W > 65536 || H > 65536) fail("EXR: image too large");
    int linesPerBlock;
    if (comp == 0 || comp == 1 || comp == 2) linesPerBlock = 1;
    else if (comp == 3) linesPerBlock = 16;
    else if (comp == 4) linesPerBlock = 32;
    else fail("EXR: this compression is not supported (NONE, RLE, ZIPS, ZIP, PIZ are)");
    size_t rowBytes = 0;
    std::vector<size_t> chOff(ch.size());
    for (size_t c = 0; c < ch.size(); ++c) {
        if (ch[c].xs != 1 || ch[c].ys != 1) fail("EXR: sub-sampled channels are not supported");
        if (ch[c].type < 0 || ch[c].type > 2) fail("EXR: bad pixel type");
        chOff[c] = rowBytes;
        rowBytes += (size_t) W * (ch[c].type == 1 ? 2 : 4);
    }
    *width = (int32_t) W; *height = (int32_t) H;
    if (!out) return;
    if (capacity < 4 * W * H) fail("EXR: output buffer too small");
    int iR = -1, iG = -1, iB = -1, iA = -1;                 // tinyexr.h:6638-6653
    for (size_t c = 0; c < ch.size(); ++c) {
        if (ch[c].name == "R") iR = (int) c; else if (ch[c].name == "G") iG = (int) c;
        else if (ch[c].name == "B") iB = (int) c; else if (ch[c].name == "A") iA = (int) c;
    }
    const bool grey = ch.size() == 1;
    if (!grey) { if (iR < 0) fail("R channel not found"); if (iG < 0) fail("G channel not found"); if (iB < 0) fail("B channel not found"); }
    const int64_t nblocks = (H + linesPerBlock - 1) / linesPerBlock;
    if (p + 8 * (size_t) nblocks > n) fail("EXR: offset table runs past the end");
    const size_t table = p;
    std::vector<uint8_t> raw, tmp;
    std::vector<char> rowSeen((size_t) H, 0);
    auto sample = [&](const uint8_t *row, int c, int64_t x) -> float {
        const uint8_t *q = row + chOff[c];
        if (ch[c].type == 1) return halfToFloat((uint16_t) (q[2 * x] | q[2 * x + 1] << 8));
        const uint32_t u = rd32(q + 4 * x);
        if (ch[c].type == 2) { float f; std::memcpy(&f, &u, 4); return f; }
        return (float) u;
    };
    for (int64_t blk = 0; blk < nblocks; ++blk) {
        const uint64_t off = rd64(b + table + 8 * (size_t) blk);
        if (off > n || n - off < 8) fail("EXR: block offset past the end");      // (off + 8 could wrap)
        const int32_t y0 = (int32_t) rd32(b + off);
        const uint32_t dsz = rd32(b + off + 4);
        if (dsz > n - off - 8) fail("EXR: block runs past the end");
        const int64_t r0 = (int64_t) y0 - dw[1];
        if (r0 < 0 || r0 >= H) fail("EXR: block outside the data window");
        const int64_t lines = r0 + linesPerBlock <= H ? linesPerBlock : H - r0;
        const size_t want = rowBytes * (size_t) lines;
        const uint8_t *src = b + off + 8;
        raw.resize(want);
        if (comp == 0) { if (dsz != want) fail("EXR: uncompressed block of the wrong size"); std::memcpy(raw.data(), src, want); }
        else if (dsz == want) std::memcpy(raw.data(), src, want);                 // stored as is (tinyexr.h:1494-1498)
        else {
            if (comp == 4) {
                std::vector<int> words(ch.size());
                for (size_t c = 0; c < ch.size(); ++c) words[c] = ch[c].type == 1 ? 1 : 2;
                unPiz(src, dsz, words, (int) W, (int) lines, raw.data(), want);
            } else {
                tmp.resize(want);
                if (comp == 1) unRle(src, dsz, tmp); else inflateZlib(src, dsz, tmp.data(), want);
                unpredict(tmp, raw.data());
            }
        }
        for (int64_t l = 0; l < lines; ++l) {
            const uint8_t *row = raw.data() + rowBytes * (size_t) l;
            const int64_t rowOut = lineOrder == 0 ? r0 + l : H - 1 - (r0 + l);
            float *o = out + 4 * W * rowOut;
            rowSeen[(size_t) (r0 + l)] = 1;
            for (int64_t x = 0; x < W; ++x) {
                if (grey) { const float v = sample(row, 0, x); o[4 * x] = v; o[4 * x + 1] = v; o[4 * x + 2] = v; o[4 * x + 3] = v; }
                else {
                    o[4 * x] = sample(row, iR, x); o[4 * x + 1] = sample(row, iG, x); o[4 * x + 2] = sample(row, iB, x);
                    o[4 * x + 3] = iA >= 0 ? sample(row, iA, x) : 1.0f;
                }
            }
        }
    }
    for (int64_t y = 0; y < H; ++y) if (!rowSeen[(size_t) y]) fail("EXR: a scan line is missing");
}

} // namespace

extern "C" int jtx_mi_decode_exr(const uint8_t *bytes, int64_t num_bytes, int32_t *width, int32_t *height, float *rgba_out, int64_t capacity) {
    if (!bytes || num_bytes <= 0 || !width || !height) return jtx_capi_fail("jtx_mi_decode_exr: null argument");
    try {
        decodeExr(bytes, (size_t) num_bytes, width, height, rgba_out, capacity);
        return 0;
    } catch (const Fail &f) {
        return jtx_capi_fail(f.msg);
    } catch (const std::exception &e) {
        return jtx_capi_fail(e.what());
    }
}
