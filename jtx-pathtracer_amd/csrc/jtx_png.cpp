// jtx_png.cpp -- host-side PNG reader for textures: the 8-bit samples stbi_load_from_memory(.., req_comp = 0) of the reference's
// ext/stb/stb_image.h returns (image.cpp:70,97 turn them into floats with pow(v / 255, 2.2)); pinned by
// tests/golden/png_cases.npz.  Colour types 0 / 2 / 3 / 4 / 6, 1 - 16 bits, Adam7; a tRNS chunk gives a palette its alpha
// and a grey / RGB image a colour key (alpha 0 where the pixel equals the key, compared on 16 bits for 16-bit files);
// 1 / 2 / 4-bit grey is scaled to 0..255 (x 255 / 85 / 17), 16-bit samples keep their high byte.  CRCs are not checked
// (stb does not check them either).
#include "../../include/jtx_mi.h"
#include "jtx_inflate.hpp"

#include <cstdlib>
#include <stdexcept>

int jtx_capi_fail(const std::string &msg);           // jtx_capi.hip: sets the thread's error text, returns 1

namespace {

using jtxz::Fail; using jtxz::fail;

uint32_t be32(const uint8_t *p) { return (uint32_t) p[0] << 24 | (uint32_t) p[1] << 16 | (uint32_t) p[2] << 8 | p[3]; }

// h filtered lines of `stride` bytes at raw[p...] -> out (h x stride), returns the next p
size_t unfilter(const std::vector<uint8_t> &raw, size_t p, size_t h, size_t stride, size_t bpp, std::vector<uint8_t> &out) {
    out.assign(h * stride, 0);
    for (size_t y = 0; y < h; ++y) {
        if (p + 1 + stride > raw.size()) fail("PNG: data ends inside a scan line");
        const int ft = raw[p]; const uint8_t *line = &raw[p + 1]; p += 1 + stride;
        uint8_t *cur = &out[y * stride]; const uint8_t *prev = y ? cur - stride : nullptr;
        if (ft < 0 || ft > 4) fail("PNG: bad filter type");
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= bpp ? cur[x - bpp] : 0, b = prev ? prev[x] : 0, c = (prev && x >= bpp) ? prev[x - bpp] : 0;
            int pr = 0;
            if (ft == 1) pr = a; else if (ft == 2) pr = b; else if (ft == 3) pr = (a + b) >> 1;
            else if (ft == 4) { const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c); pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            cur[x] = (uint8_t) (line[x] + pr);
        }
    }
    return p;
}

void decodePng(const uint8_t *d, size_t n, int32_t *width, int32_t *height, int32_t *comps, uint8_t *out, int64_t capacity) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (n < 8 || std::memcmp(d, sig, 8) != 0) fail("PNG: bad signature");
    size_t pos = 8;
    uint32_t w = 0, h = 0; int depth = 0, ctype = -1, interlace = 0; bool haveHdr = false;
    std::vector<uint8_t> idat, plte, trns;
    while (pos + 8 <= n) {
        const uint32_t len = be32(d + pos); const uint8_t *tag = d + pos + 4;
        if (len > n - pos - 8) fail("PNG: chunk runs past the end");
        const uint8_t *body = d + pos + 8;
        pos += 12 + (size_t) len;
        if (!std::memcmp(tag, "IHDR", 4)) {
            if (len < 13) fail("PNG: bad IHDR");
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; haveHdr = true;
            if (body[10] || body[11]) fail("PNG: bad compression / filter method");
        } else if (!std::memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(tag, "PLTE", 4)) plte.assign(body, body + len);
        else if (!std::memcmp(tag, "tRNS", 4)) trns.assign(body, body + len);
        else if (!std::memcmp(tag, "IEND", 4)) break;
    }
    if (!haveHdr) fail("PNG: no IHDR");
    if (w == 0 || h == 0 || w > (1u << 24) || h > (1u << 24) || (uint64_t) w * h > (1ull << 28)) fail("PNG: bad size");
    if (!(ctype == 0 || ctype == 2 || ctype == 3 || ctype == 4 || ctype == 6) || interlace > 1) fail("PNG: bad header");
    const bool depthOk = depth == 8 || (depth == 16 && ctype != 3) || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4));
    if (!depthOk) fail("PNG: unsupported bit depth");
    if (ctype == 3 && (plte.size() < 3 || plte.size() % 3)) fail("PNG: palette image without a palette");
    const int nch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4;
    bool key = false; int key16[3] = {0, 0, 0};
    if (!trns.empty() && (ctype == 0 || ctype == 2) && trns.size() >= (size_t) 2 * nch) {
        key = true;
        for (int k = 0; k < nch; ++k) { const int v = trns[2 * k] << 8 | trns[2 * k + 1]; key16[k] = depth == 16 ? v : (v & 255); }
    }
    const int outc = ctype == 3 ? (trns.empty() ? 3 : 4) : nch + (key ? 1 : 0);
    *width = (int32_t) w; *height = (int32_t) h; *comps = outc;
    if (!out) return;
    if (capacity < (int64_t) w * h * outc) fail("PNG: output buffer too small");
    const size_t bpp = (size_t) (nch * depth >= 8 ? nch * depth / 8 : 1);
    // size of the filtered data
    static const int px0[7] = {0, 4, 0, 2, 0, 1, 0}, py0[7] = {0, 0, 4, 0, 2, 0, 1}, pdx[7] = {8, 8, 4, 4, 2, 2, 1}, pdy[7] = {8, 8, 8, 4, 4, 2, 2};
    size_t total = 0;
    if (!interlace) total = (size_t) h * (1 + ((size_t) w * nch * depth + 7) / 8);
    else for (int k = 0; k < 7; ++k) {
        const size_t sw = (w > (uint32_t) px0[k]) ? (w - px0[k] + pdx[k] - 1) / pdx[k] : 0, sh = (h > (uint32_t) py0[k]) ? (h - py0[k] + pdy[k] - 1) / pdy[k] : 0;
        if (sw && sh) total += sh * (1 + (sw * nch * depth + 7) / 8);
    }
    std::vector<uint8_t> raw(total);
    if (idat.empty()) fail("PNG: no image data");
    jtxz::inflateZlib(idat.data(), idat.size(), raw.data(), raw.size(), false);
    std::vector<int32_t> val((size_t) w * h * nch);           // samples, 16-bit values whole
    auto scatter = [&](const std::vector<uint8_t> &rows, size_t sw, size_t sh, size_t x0, size_t y0, size_t dx, size_t dy) {
        const size_t stride = (sw * nch * depth + 7) / 8;
        for (size_t y = 0; y < sh; ++y) for (size_t x = 0; x < sw; ++x) for (int c = 0; c < nch; ++c) {
            const uint8_t *r = &rows[y * stride]; int v;
            const size_t i = x * nch + c;
            if (depth == 16) v = r[2 * i] << 8 | r[2 * i + 1];
            else if (depth == 8) v = r[i];
            else { const size_t bit = i * depth; v = (r[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1); }
            val[((y0 + y * dy) * w + (x0 + x * dx)) * nch + c] = v;
        }
    };
    std::vector<uint8_t> rows;
    if (!interlace) { unfilter(raw, 0, h, ((size_t) w * nch * depth + 7) / 8, bpp, rows); scatter(rows, w, h, 0, 0, 1, 1); }
    else {
        size_t p = 0;
        for (int k = 0; k < 7; ++k) {
            const size_t sw = (w > (uint32_t) px0[k]) ? (w - px0[k] + pdx[k] - 1) / pdx[k] : 0, sh = (h > (uint32_t) py0[k]) ? (h - py0[k] + pdy[k] - 1) / pdy[k] : 0;
            if (!sw || !sh) continue;
            p = unfilter(raw, p, sh, (sw * nch * depth + 7) / 8, bpp, rows);
            scatter(rows, sw, sh, px0[k], py0[k], pdx[k], pdy[k]);
        }
    }
    const int scale = depth == 1 ? 255 : depth == 2 ? 85 : depth == 4 ? 17 : 1;
    for (size_t i = 0; i < (size_t) w * h; ++i) {
        const int32_t *v = &val[i * nch]; uint8_t *o = out + i * outc;
        if (ctype == 3) {
            if ((size_t) v[0] * 3 + 2 >= plte.size()) fail("PNG: palette index out of range");
            o[0] = plte[3 * v[0]]; o[1] = plte[3 * v[0] + 1]; o[2] = plte[3 * v[0] + 2];
            if (outc == 4) o[3] = (size_t) v[0] < trns.size() ? trns[v[0]] : 255;
            continue;
        }
        bool isKey = key;
        for (int c = 0; c < nch; ++c) {
            if (key && v[c] != key16[c]) isKey = false;
            o[c] = depth == 16 ? (uint8_t) (v[c] >> 8) : (uint8_t) (ctype == 0 && depth < 8 ? v[c] * scale : v[c]);
        }
        if (key) o[nch] = isKey ? 0 : 255;
    }
}

} // namespace

extern "C" int jtx_mi_decode_png(const uint8_t *bytes, int64_t num_bytes, int32_t *width, int32_t *height, int32_t *components,
                                 uint8_t *out, int64_t capacity) {
    if (!bytes || num_bytes <= 0 || !width || !height || !components) return jtx_capi_fail("jtx_mi_decode_png: null argument");
    try {
        decodePng(bytes, (size_t) num_bytes, width, height, components, out, capacity);
        return 0;
    } catch (const Fail &f) {
        return jtx_capi_fail(f.msg);
    } catch (const std::exception &e) {
        return jtx_capi_fail(e.what());
    }
}
