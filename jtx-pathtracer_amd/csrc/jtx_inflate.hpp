// jtx_inflate.hpp -- a plain RFC 1950 / 1951 (zlib / deflate) decoder shared by the EXR and PNG readers (host code).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace jtxz {

struct Fail { std::string msg; };
[[noreturn]] inline void fail(const std::string &m) { throw Fail{m}; }

struct BitReader {
    const uint8_t *p, *end;
    uint64_t acc = 0; int n = 0;
    void need(int k) { while (n < k) { const uint64_t b = p < end ? *p++ : 0u; acc |= b << n; n += 8; } }
    unsigned bits(int k) { if (k == 0) return 0u; need(k); const unsigned v = (unsigned) (acc & ((1ull << k) - 1)); acc >>= k; n -= k; return v; }
    void alignByte() { const int r = n & 7; acc >>= r; n -= r; }
};

struct Huff {
    uint16_t count[16] = {}, symbol[320] = {};
    void build(const uint8_t *lens, int num) {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < num; ++i) count[lens[i]]++;
        count[0] = 0;
        uint16_t offs[16]; offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t) (offs[l] + count[l]);
        for (int i = 0; i < num; ++i) if (lens[i]) symbol[offs[lens[i]]++] = (uint16_t) i;
    }
    int decode(BitReader &br) const {                     // canonical code, one bit at a time (blocks are small)
        int code = 0, first = 0, index = 0;
        for (int l = 1; l <= 15; ++l) {
            code |= (int) br.bits(1);
            const int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        fail("bad Huffman code in a zlib block");
    }
};

// exact = false: output beyond dstLen is dropped (PNG readers tolerate a longer stream); a shorter one is always an error
inline void inflateZlib(const uint8_t *src, size_t srcLen, uint8_t *dst, size_t dstLen, bool exact = true) {
    if (srcLen < 6) fail("zlib block too short");
    if ((src[0] & 0x0f) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) fail("not a zlib stream");
    BitReader br{src + 2, src + srcLen};
    size_t out = 0;
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    bool last = false;
    while (!last) {
        last = br.bits(1) != 0;
        const unsigned type = br.bits(2);
        if (type == 0) {
            br.alignByte();
            const unsigned len = br.bits(16), nlen = br.bits(16);
            if ((len ^ nlen) != 0xffffu) fail("bad stored block");
            if (out + len > dstLen) { if (exact) fail("zlib block longer than the scan lines it holds"); for (unsigned i = 0; i < len && out < dstLen; ++i) dst[out++] = (uint8_t) br.bits(8); return; }
            for (unsigned i = 0; i < len; ++i) dst[out++] = (uint8_t) br.bits(8);
            continue;
        }
        if (type == 3) fail("bad zlib block type");
        Huff lit, dist;
        if (type == 1) {
            uint8_t l[288];
            for (int i = 0; i < 144; ++i) l[i] = 8;
            for (int i = 144; i < 256; ++i) l[i] = 9;
            for (int i = 256; i < 280; ++i) l[i] = 7;
            for (int i = 280; i < 288; ++i) l[i] = 8;
            lit.build(l, 288);
            uint8_t d[30]; for (int i = 0; i < 30; ++i) d[i] = 5;
            dist.build(d, 30);
        } else {
            const int hlit = (int) br.bits(5) + 257, hdist = (int) br.bits(5) + 1, hclen = (int) br.bits(4) + 4;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {};
            for (int i = 0; i < hclen; ++i) cl[order[i]] = (uint8_t) br.bits(3);
            Huff clh; clh.build(cl, 19);
            uint8_t lens[320] = {};
            int i = 0;
            while (i < hlit + hdist) {
                const int sym = clh.decode(br);
                if (sym < 16) lens[i++] = (uint8_t) sym;
                else {
                    int rep; uint8_t v = 0;
                    if (sym == 16) { if (i == 0) fail("bad code-length repeat"); v = lens[i - 1]; rep = 3 + (int) br.bits(2); }
                    else if (sym == 17) rep = 3 + (int) br.bits(3);
                    else rep = 11 + (int) br.bits(7);
                    if (i + rep > hlit + hdist) fail("code lengths overrun");
                    while (rep--) lens[i++] = v;
                }
            }
            if (hlit > 286 || hdist > 30) fail("too many Huffman codes");
            lit.build(lens, hlit); dist.build(lens + hlit, hdist);
        }
        while (true) {
            const int sym = lit.decode(br);
            if (sym < 256) { if (out >= dstLen) { if (exact) fail("zlib block longer than the scan lines it holds"); return; } dst[out++] = (uint8_t) sym; continue; }
            if (sym == 256) break;
            if (sym > 285) fail("bad length symbol");
            const unsigned len = lbase[sym - 257] + br.bits(lext[sym - 257]);
            const int ds = dist.decode(br);
            if (ds > 29) fail("bad distance symbol");
            const size_t d = dbase[ds] + br.bits(dext[ds]);
            if (d > out) fail("distance beyond the start of the block");
            if (out + len > dstLen) { if (exact) fail("zlib block longer than the scan lines it holds"); for (; out < dstLen; ++out) dst[out] = dst[out - d]; return; }
            for (unsigned k = 0; k < len; ++k, ++out) dst[out] = dst[out - d];
        }
    }
    if (out != dstLen) fail("zlib block shorter than the scan lines it holds");
}


} // namespace jtxz
