// jtx_host_loader.hpp -- loadScene(path, scene) for a C++ host (src/loader.hpp:6), Wavefront OBJ only: the geometry side of
// the reference's Assimp import (loader.cpp:21: Triangulate | FlipUVs | GenNormals | PreTransformVertices) without Assimp,
// and its material rule for OBJ files (loader.cpp:105-149: one DIFFUSE WHITE material per material name, the diffuse map as
// albedo texture).  Same semantics as scenes.load_obj of the Python mirror (tests/test_host_loader_cpu.py compares them):
//   * one mesh per `o` / `g` group and `usemtl` run, in file order; faces fan-triangulated; NO vertex joining -- three fresh
//     vertices per face corner in face order; one Triangle ref per face (loader.cpp:216-222);
//   * a face without `vn` gets its flat normal (GenNormals), `vt` becomes (u, 1 - v) (FlipUVs); a mesh has uvs only if every
//     corner of its faces has one;
//   * `map_Kd` of a `mtllib` entry is read as TextureImage::load reads it (image.cpp:59-74): .exr through jtx_mi_decode_exr
//     (RGBA floats as stored), JPEG / PNG through jtx_mi_decode_jpeg / jtx_mi_decode_png + stbi_loadf's pow(v / 255, 2.2); paths stay inside the
//     asset's directory; anything unreadable leaves albedoTexId at -1 ("Failed to load texture", loader.cpp:98).
// glTF 2.0 / GLB (loadGltf below): the same restatement of the Assimp pipeline as jtx_pathtracer_amd.gltf.load_gltf -- one mesh
// per primitive in scene-graph order, node transforms baked in (PreTransformVertices: points by the matrix, normals by its
// inverse transpose, re-normalised), flat normals where NORMAL is missing, v -> 1 - v, every material METALLIC_ROUGHNESS with
// albedo WHITE, alphaX = metallicFactor, alphaY = roughnessFactor (loader.cpp:122-143), base-colour and metallic-roughness
// maps through the PNG / JPEG readers + stbi_loadf's float conversion; a primitive without material gets DIFFUSE (1, 0.3, 0.5).
#pragma once
#include "jtx_host_api.hpp"

#include <cctype>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace jtxmi {

namespace loader_detail {

inline std::string dirOf(const std::string &p) { const size_t k = p.find_last_of("/\\"); return k == std::string::npos ? std::string() : p.substr(0, k + 1); }
inline bool confined(const std::string &rel) {                        // relative, and no way up out of the asset's directory
    if (rel.empty() || rel[0] == '/' || rel[0] == '\\' || rel.find(':') != std::string::npos) return false;
    int depth = 0; std::stringstream ss(rel); std::string part;
    while (std::getline(ss, part, '/')) { if (part == "..") { if (--depth < 0) return false; } else if (!part.empty() && part != ".") ++depth; }
    return true;
}
inline bool readFile(const std::string &p, std::vector<uint8_t> &out) {
    std::ifstream f(p, std::ios::binary); if (!f) return false;
    out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>()); return true;
}
// TextureImage::load(path): EXR as stored; JPEG -> stbi_loadf (ldr_to_hdr: pow(v / 255, 2.2f) evaluated in double)
inline bool loadTexture(const std::string &path, TextureImage &t) {
    std::vector<uint8_t> bytes;
    if (!readFile(path, bytes) || bytes.empty()) return false;
    const std::string ext = path.substr(path.find_last_of('.') + 1);
    int32_t w = 0, h = 0, c = 0;
    if (ext == "exr" || ext == "EXR") {
        if (jtx_mi_decode_exr(bytes.data(), (int64_t) bytes.size(), &w, &h, nullptr, 0)) return false;
        t.data_.resize((size_t) 4 * w * h);
        if (jtx_mi_decode_exr(bytes.data(), (int64_t) bytes.size(), &w, &h, t.data_.data(), (int64_t) t.data_.size())) return false;
        t.width_ = w; t.height_ = h; t.channels_ = 4; return true;
    }
    if (bytes.size() > 2 && bytes[0] == 0xff && bytes[1] == 0xd8) {
        if (jtx_mi_decode_jpeg(bytes.data(), (int64_t) bytes.size(), &w, &h, &c, nullptr, 0)) return false;
        std::vector<uint8_t> px((size_t) w * h * c);
        if (jtx_mi_decode_jpeg(bytes.data(), (int64_t) bytes.size(), &w, &h, &c, px.data(), (int64_t) px.size())) return false;
        float lut[256];
        for (int v = 0; v < 256; ++v) lut[v] = (float) std::pow((double) ((float) v / 255.0f), (double) 2.2f);
        t.data_.resize(px.size());
        for (size_t i = 0; i < px.size(); ++i) t.data_[i] = lut[px[i]];
        t.width_ = w; t.height_ = h; t.channels_ = c; return c >= 3;
    }
    if (bytes.size() > 8 && bytes[0] == 0x89 && bytes[1] == 'P' && bytes[2] == 'N' && bytes[3] == 'G') {
        if (jtx_mi_decode_png(bytes.data(), (int64_t) bytes.size(), &w, &h, &c, nullptr, 0)) return false;
        std::vector<uint8_t> px((size_t) w * h * c);
        if (jtx_mi_decode_png(bytes.data(), (int64_t) bytes.size(), &w, &h, &c, px.data(), (int64_t) px.size())) return false;
        float lut[256];
        for (int v = 0; v < 256; ++v) lut[v] = (float) std::pow((double) ((float) v / 255.0f), (double) 2.2f);
        const int ncol = (c & 1) ? c : c - 1;                          // stbi__ldr_to_hdr: an alpha channel stays linear (v / 255)
        t.data_.resize(px.size());
        for (size_t i = 0; i < px.size(); ++i) t.data_[i] = (int) (i % (size_t) c) < ncol ? lut[px[i]] : (float) px[i] / 255.0f;
        t.width_ = w; t.height_ = h; t.channels_ = c; return c >= 3;
    }
    return false;
}

} // namespace loader_detail

inline void loadGltf(const std::string &path, Scene &scene);

inline void loadScene(const std::string &path, Scene &scene) {
    using namespace loader_detail;
    std::ifstream in(path);
    if (!in) throw std::runtime_error("loadScene: cannot open " + path);
    { const std::string ext = path.size() >= 5 ? path.substr(path.find_last_of('.') + 1) : std::string();
      if (ext == "glb" || ext == "GLB" || ext == "gltf" || ext == "GLTF") { in.close(); loadGltf(path, scene); return; }
      if (ext != "obj" && ext != "OBJ") throw std::runtime_error("loadScene: Wavefront OBJ, glTF and GLB are read here, not ." + ext); }
    const std::string base = dirOf(path);
    struct Corner { int v, vt, vn; };
    struct Group { std::string name, mtl; std::vector<Corner> corners; };   // 3 corners per triangle
    std::vector<Vec3> V, VN; std::vector<Vec2f> VT;
    std::vector<Group> groups; Group *cur = nullptr;
    std::string curName = "default", curMtl;
    std::vector<std::string> libs;
    std::string line;
    while (std::getline(in, line)) {
        std::stringstream ss(line); std::string tag; ss >> tag;
        if (tag.empty() || tag[0] == '#') continue;
        if (tag == "v") { Vec3 p; ss >> p.x >> p.y >> p.z; V.push_back(p); }
        else if (tag == "vn") { Vec3 p; ss >> p.x >> p.y >> p.z; VN.push_back(p); }
        else if (tag == "vt") { Vec2f p; ss >> p.x; if (!(ss >> p.y)) p.y = 0; VT.push_back(p); }
        else if (tag == "o" || tag == "g") { if (!(ss >> curName)) curName = "default"; cur = nullptr; }
        else if (tag == "usemtl") { curMtl.clear(); ss >> curMtl; cur = nullptr; }
        else if (tag == "mtllib") { std::string rest; std::getline(ss, rest); const size_t a = rest.find_first_not_of(" \t"); if (a != std::string::npos) { rest = rest.substr(a); while (!rest.empty() && (rest.back() == '\r' || rest.back() == ' ')) rest.pop_back(); libs.push_back(rest); } }
        else if (tag == "f") {
            if (!cur) { groups.push_back({curName, curMtl, {}}); cur = &groups.back(); }
            std::vector<Corner> cs; std::string tok;
            while (ss >> tok) {
                int idx[3] = {0, 0, 0}; int k = 0; std::string num;
                for (size_t i = 0; i <= tok.size() && k < 3; ++i) {
                    if (i == tok.size() || tok[i] == '/') { if (!num.empty()) idx[k] = std::stoi(num); num.clear(); ++k; }
                    else num += tok[i];
                }
                Corner c;
                c.v = idx[0] > 0 ? idx[0] - 1 : (int) V.size() + idx[0];
                c.vt = idx[1] ? (idx[1] > 0 ? idx[1] - 1 : (int) VT.size() + idx[1]) : -1;
                c.vn = idx[2] ? (idx[2] > 0 ? idx[2] - 1 : (int) VN.size() + idx[2]) : -1;
                if (c.v < 0 || c.v >= (int) V.size() || c.vt >= (int) VT.size() || c.vn >= (int) VN.size()) throw std::runtime_error("loadScene: index out of range in " + path);
                cs.push_back(c);
            }
            for (size_t k = 1; k + 1 < cs.size(); ++k) { cur->corners.push_back(cs[0]); cur->corners.push_back(cs[k]); cur->corners.push_back(cs[k + 1]); }   // fan
        }
    }
    // materials: one DIFFUSE WHITE per name, with its diffuse map (loader.cpp:105-149)
    std::map<std::string, std::string> mapKd;
    for (const std::string &lib : libs) {
        if (!confined(lib)) continue;
        std::ifstream mf(base + lib); std::string name;
        while (std::getline(mf, line)) {
            std::stringstream ss(line); std::string tag; ss >> tag;
            if (tag == "newmtl") { name.clear(); ss >> name; }
            else if (tag == "map_Kd" && !name.empty()) { std::string tok, last; while (ss >> tok) last = tok; if (!last.empty()) mapKd[name] = last; }
        }
    }
    scene.materials.reserve(scene.materials.size() + groups.size() + 1);   // Mesh::material points into the vector
    std::map<std::string, int> matOf, texOf;
    auto materialFor = [&](const std::string &mtl) {
        auto it = matOf.find(mtl);
        if (it != matOf.end()) return it->second;
        Material m; m.type = Material::DIFFUSE; m.albedo = Vec3(1, 1, 1);
        auto kd = mapKd.find(mtl);
        if (kd != mapKd.end()) {
            const std::string full = base + kd->second;
            auto t = texOf.find(full);
            if (t == texOf.end()) {
                TextureImage img; int id = -1;
                if (confined(kd->second) && loadTexture(full, img)) { scene.textures.push_back(std::move(img)); id = (int) scene.textures.size() - 1; }
                t = texOf.emplace(full, id).first;
            }
            m.albedoTexId = t->second;
        }
        scene.materials.push_back(m);
        return matOf[mtl] = (int) scene.materials.size() - 1;
    };
    for (const Group &g : groups) {
        const size_t nv = g.corners.size();
        if (nv == 0) continue;
        std::shared_ptr<Vec3> pos(new Vec3[nv], std::default_delete<Vec3[]>()), nrm(new Vec3[nv], std::default_delete<Vec3[]>());
        std::shared_ptr<Vec3i> idx(new Vec3i[nv / 3], std::default_delete<Vec3i[]>());
        bool hasUv = true;
        for (const Corner &c : g.corners) hasUv = hasUv && c.vt >= 0;
        std::shared_ptr<Vec2f> uvs;
        if (hasUv) uvs.reset(new Vec2f[nv], std::default_delete<Vec2f[]>());
        for (size_t f = 0; f < nv / 3; ++f) {
            const Corner *c = &g.corners[3 * f];
            Vec3 p[3] = {V[c[0].v], V[c[1].v], V[c[2].v]};
            const bool haveN = c[0].vn >= 0 && c[1].vn >= 0 && c[2].vn >= 0;
            Vec3 flat;
            if (!haveN) {                                              // GenNormals: the face normal, float32 throughout
                const Vec3 a(p[1].x - p[0].x, p[1].y - p[0].y, p[1].z - p[0].z), b(p[2].x - p[0].x, p[2].y - p[0].y, p[2].z - p[0].z);
                flat = Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
                const float l = std::sqrt(flat.x * flat.x + flat.y * flat.y + flat.z * flat.z);
                if (l > 0) flat = Vec3(flat.x / l, flat.y / l, flat.z / l);
            }
            for (int k = 0; k < 3; ++k) {
                pos.get()[3 * f + k] = p[k];
                nrm.get()[3 * f + k] = haveN ? VN[c[k].vn] : flat;
                if (hasUv) { uvs.get()[3 * f + k].x = VT[c[k].vt].x; uvs.get()[3 * f + k].y = 1.0f - VT[c[k].vt].y; }   // FlipUVs
            }
            idx.get()[f] = Vec3i((int) (3 * f), (int) (3 * f + 1), (int) (3 * f + 2));
        }
        Material *mat = &scene.materials[materialFor(g.mtl)];
        Mesh mesh(idx.get(), (int) (nv / 3), pos.get(), (int) nv, nrm.get(), uvs.get(), mat);
        mesh.name = g.name;
        scene.meshes.push_back(mesh);
        scene.storage.push_back(pos); scene.storage.push_back(nrm); scene.storage.push_back(idx); if (uvs) scene.storage.push_back(uvs);
        const int meshIndex = (int) scene.meshes.size() - 1;
        for (size_t f = 0; f < nv / 3; ++f) scene.triangles.push_back(Triangle{(int) f, meshIndex});
    }
}


// ------------------------------------------------------------------------------------------------ glTF 2.0 / GLB
namespace loader_detail {

struct Json {                                                          // a small JSON value (objects keep insertion order)
    enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
    double num = 0; bool b = false; std::string str; std::vector<Json> arr; std::vector<std::pair<std::string, Json>> obj;
    const Json *get(const std::string &k) const { if (kind != OBJ) return nullptr; for (const auto &kv : obj) if (kv.first == k) return &kv.second; return nullptr; }
    bool has(const std::string &k) const { return get(k) != nullptr; }
    const Json &at(const std::string &k) const { const Json *j = get(k); if (!j) throw std::runtime_error("glTF: missing \"" + k + "\""); return *j; }
    const Json &at(size_t i) const { if (kind != ARR || i >= arr.size()) throw std::runtime_error("glTF: index out of range"); return arr[i]; }
    double numOr(const std::string &k, double d) const { const Json *j = get(k); return j && j->kind == NUM ? j->num : d; }
    long intOr(const std::string &k, long d) const { return (long) numOr(k, (double) d); }
    std::string strOr(const std::string &k, const std::string &d) const { const Json *j = get(k); return j && j->kind == STR ? j->str : d; }
};
struct JsonParser {
    const char *p, *e;
    [[noreturn]] void bad(const char *m) { throw std::runtime_error(std::string("glTF: JSON ") + m); }
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
    Json value(int depth = 0) {
        if (depth > 64) bad("nested too deep");
        ws(); if (p >= e) bad("ends early");
        Json j;
        if (*p == '{') { ++p; j.kind = Json::OBJ; ws(); if (p < e && *p == '}') { ++p; return j; }
            while (true) { ws(); if (p >= e || *p != '"') bad("key expected"); Json k = string(); ws(); if (p >= e || *p != ':') bad("colon expected"); ++p;
                           j.obj.emplace_back(k.str, value(depth + 1)); ws(); if (p < e && *p == ',') { ++p; continue; } if (p < e && *p == '}') { ++p; return j; } bad("bad object"); } }
        if (*p == '[') { ++p; j.kind = Json::ARR; ws(); if (p < e && *p == ']') { ++p; return j; }
            while (true) { j.arr.push_back(value(depth + 1)); ws(); if (p < e && *p == ',') { ++p; continue; } if (p < e && *p == ']') { ++p; return j; } bad("bad array"); } }
        if (*p == '"') return string();
        if (e - p >= 4 && !std::strncmp(p, "true", 4)) { p += 4; j.kind = Json::BOOL; j.b = true; return j; }
        if (e - p >= 5 && !std::strncmp(p, "false", 5)) { p += 5; j.kind = Json::BOOL; return j; }
        if (e - p >= 4 && !std::strncmp(p, "null", 4)) { p += 4; return j; }
        const char *s0 = p; while (p < e && (std::strchr("+-0123456789.eE", *p))) ++p;
        if (p == s0) bad("unexpected character");
        j.kind = Json::NUM; j.num = std::strtod(std::string(s0, p).c_str(), nullptr); return j;
    }
    Json string() {
        Json j; j.kind = Json::STR; ++p;
        while (p < e && *p != '"') {
            if (*p == '\\') { if (++p >= e) bad("bad escape");
                switch (*p) { case 'n': j.str += '\n'; break; case 't': j.str += '\t'; break; case 'r': j.str += '\r'; break; case 'b': j.str += '\b'; break; case 'f': j.str += '\f'; break;
                              case 'u': { if (e - p < 5) bad("bad escape"); const unsigned c = (unsigned) std::strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16); p += 4;
                                          if (c < 0x80) j.str += (char) c; else if (c < 0x800) { j.str += (char) (0xc0 | c >> 6); j.str += (char) (0x80 | (c & 63)); } else { j.str += (char) (0xe0 | c >> 12); j.str += (char) (0x80 | ((c >> 6) & 63)); j.str += (char) (0x80 | (c & 63)); } break; }
                              default: j.str += *p; }
                ++p; }
            else j.str += *p++;
        }
        if (p >= e) bad("unterminated string");
        ++p;
        return j;
    }
};
inline std::vector<uint8_t> base64(const std::string &t) {
    std::vector<uint8_t> o; unsigned acc = 0; int nb = 0;
    for (char ch : t) { int v; if (ch >= 'A' && ch <= 'Z') v = ch - 'A'; else if (ch >= 'a' && ch <= 'z') v = ch - 'a' + 26; else if (ch >= '0' && ch <= '9') v = ch - '0' + 52; else if (ch == '+' || ch == '-') v = 62; else if (ch == '/' || ch == '_') v = 63; else continue;
                      acc = acc << 6 | (unsigned) v; nb += 6; if (nb >= 8) { nb -= 8; o.push_back((uint8_t) (acc >> nb)); } }
    return o;
}
inline std::string unquote(const std::string &u) { std::string o; for (size_t i = 0; i < u.size(); ++i) { if (u[i] == '%' && i + 2 < u.size() + 0 && std::isxdigit((unsigned char) u[i + 1]) && std::isxdigit((unsigned char) u[i + 2])) { o += (char) std::strtoul(u.substr(i + 1, 2).c_str(), nullptr, 16); i += 2; } else o += u[i]; } return o; }

struct Mat4d { double m[4][4]; };
inline Mat4d identity4() { Mat4d r{}; for (int i = 0; i < 4; ++i) r.m[i][i] = 1; return r; }
inline Mat4d mul4(const Mat4d &a, const Mat4d &b) { Mat4d r{}; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[k][j]; r.m[i][j] = s; } return r; }
inline Mat4d nodeMatrix(const Json &n) {
    Mat4d r = identity4();
    if (const Json *mm = n.get("matrix")) { if (mm->arr.size() != 16) throw std::runtime_error("glTF: node matrix needs 16 numbers"); for (int c = 0; c < 4; ++c) for (int rr = 0; rr < 4; ++rr) r.m[rr][c] = mm->arr[4 * c + rr].num; return r; }   // column-major
    double t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, sc[3] = {1, 1, 1};
    if (const Json *j = n.get("translation")) for (int i = 0; i < 3 && i < (int) j->arr.size(); ++i) t[i] = j->arr[i].num;
    if (const Json *j = n.get("rotation")) for (int i = 0; i < 4 && i < (int) j->arr.size(); ++i) q[i] = j->arr[i].num;
    if (const Json *j = n.get("scale")) for (int i = 0; i < 3 && i < (int) j->arr.size(); ++i) sc[i] = j->arr[i].num;
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)}, {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                            {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) r.m[i][j] = R[i][j] * sc[j]; r.m[i][3] = t[i]; }
    return r;
}

struct GltfDoc {
    Json doc; std::vector<std::vector<uint8_t>> bufs; std::string base;
    // accessor -> rows of doubles (count x ncomp), normalised integers mapped as glTF asks
    std::vector<double> accessor(long i, int &ncomp, size_t &count) const {
        const Json &a = doc.at("accessors").at((size_t) i);
        const long ct = a.intOr("componentType", 0); const std::string ty = a.strOr("type", "");
        ncomp = ty == "SCALAR" ? 1 : ty == "VEC2" ? 2 : ty == "VEC3" ? 3 : ty == "VEC4" ? 4 : ty == "MAT4" ? 16 : 0;
        const int isz = ct == 5120 || ct == 5121 ? 1 : ct == 5122 || ct == 5123 ? 2 : ct == 5125 || ct == 5126 ? 4 : 0;
        if (!ncomp || !isz) throw std::runtime_error("glTF: unsupported accessor type");
        count = (size_t) a.intOr("count", 0);
        std::vector<double> out(count * ncomp, 0.0);
        if (!a.has("bufferView")) return out;
        const Json &v = doc.at("bufferViews").at((size_t) a.intOr("bufferView", 0));
        const size_t start = (size_t) v.intOr("byteOffset", 0) + (size_t) a.intOr("byteOffset", 0);
        size_t stride = (size_t) v.intOr("byteStride", 0); if (!stride) stride = (size_t) isz * ncomp;
        const std::vector<uint8_t> &buf = bufs.at((size_t) v.intOr("buffer", 0));
        if (count && start + (count - 1) * stride + (size_t) isz * ncomp > buf.size()) throw std::runtime_error("glTF: accessor runs past its buffer");
        const bool norm = a.get("normalized") && a.get("normalized")->b;
        for (size_t k = 0; k < count; ++k) for (int c = 0; c < ncomp; ++c) {
            const uint8_t *q = &buf[start + k * stride + (size_t) c * isz]; double x;
            if (ct == 5126) { float f; std::memcpy(&f, q, 4); x = f; }
            else if (ct == 5125) { uint32_t u; std::memcpy(&u, q, 4); x = u; }
            else if (ct == 5123) { uint16_t u; std::memcpy(&u, q, 2); x = u; if (norm) x = (double) ((float) u / 65535.0f); }
            else if (ct == 5122) { int16_t u; std::memcpy(&u, q, 2); x = u; if (norm) { const float f = (float) u / 32767.0f; x = f < -1.0f ? -1.0 : f; } }
            else if (ct == 5121) { x = *q; if (norm) x = (double) ((float) *q / 255.0f); }
            else { const int8_t u = (int8_t) *q; x = u; if (norm) { const float f = (float) u / 127.0f; x = f < -1.0f ? -1.0 : f; } }
            out[k * ncomp + c] = x;
        }
        return out;
    }
    std::vector<uint8_t> external(const std::string &uri) const {
        const std::string rel = unquote(uri);
        if (uri.find("://") != std::string::npos || !confined(rel)) throw std::runtime_error("glTF external URI escapes the asset directory: " + uri);
        std::vector<uint8_t> b; if (!readFile(base + rel, b)) throw std::runtime_error("glTF: cannot read " + rel); return b;
    }
    std::vector<uint8_t> imageBytes(const Json &img) const {
        if (img.has("bufferView")) { const Json &v = doc.at("bufferViews").at((size_t) img.intOr("bufferView", 0)); const std::vector<uint8_t> &b = bufs.at((size_t) v.intOr("buffer", 0));
            const size_t s0 = (size_t) v.intOr("byteOffset", 0), n = (size_t) v.intOr("byteLength", 0); if (s0 + n > b.size()) throw std::runtime_error("glTF: image runs past its buffer"); return std::vector<uint8_t>(b.begin() + s0, b.begin() + s0 + n); }
        const std::string uri = img.strOr("uri", "");
        if (uri.compare(0, 5, "data:") == 0) return base64(uri.substr(uri.find(',') + 1));
        return external(uri);
    }
};

// TextureImage::load(buffer, size, STB) -> stbi_loadf_from_memory (image.cpp:97): PNG / JPEG bytes -> floats
inline bool loadTextureBytes(const std::vector<uint8_t> &bytes, TextureImage &t) {
    int32_t w = 0, h = 0, c = 0; std::vector<uint8_t> px;
    const bool png = bytes.size() > 8 && bytes[0] == 0x89 && bytes[1] == 'P', jpg = bytes.size() > 2 && bytes[0] == 0xff && bytes[1] == 0xd8;
    if (!png && !jpg) return false;
    auto dec = [&](uint8_t *o, int64_t cap) { return png ? jtx_mi_decode_png(bytes.data(), (int64_t) bytes.size(), &w, &h, &c, o, cap) : jtx_mi_decode_jpeg(bytes.data(), (int64_t) bytes.size(), &w, &h, &c, o, cap); };
    if (dec(nullptr, 0)) return false;
    px.resize((size_t) w * h * c);
    if (dec(px.data(), (int64_t) px.size())) return false;
    float lut[256];
    for (int v = 0; v < 256; ++v) lut[v] = (float) std::pow((double) ((float) v / 255.0f), (double) 2.2f);
    const int ncol = (c & 1) ? c : c - 1;                              // stbi__ldr_to_hdr: an alpha channel stays linear
    t.data_.resize(px.size());
    for (size_t i = 0; i < px.size(); ++i) t.data_[i] = (int) (i % (size_t) c) < ncol ? lut[px[i]] : (float) px[i] / 255.0f;
    t.width_ = w; t.height_ = h; t.channels_ = c;
    return true;
}

} // namespace loader_detail

inline void loadGltf(const std::string &path, Scene &scene) {
    using namespace loader_detail;
    std::vector<uint8_t> blob;
    if (!readFile(path, blob)) throw std::runtime_error("loadScene: cannot open " + path);
    GltfDoc g; g.base = dirOf(path);
    std::vector<uint8_t> binChunk; bool haveBin = false;
    auto rd = [&](size_t at) { if (at + 4 > blob.size()) throw std::runtime_error("glTF: truncated GLB"); return (uint32_t) blob[at] | (uint32_t) blob[at + 1] << 8 | (uint32_t) blob[at + 2] << 16 | (uint32_t) blob[at + 3] << 24; };
    if (blob.size() >= 12 && !std::memcmp(blob.data(), "glTF", 4)) {
        size_t total = rd(8); if (total > blob.size()) total = blob.size();
        size_t pos = 12; bool haveDoc = false;
        while (pos + 8 <= total) {
            const size_t n = rd(pos); const uint32_t kind = rd(pos + 4);
            if (pos + 8 + n > blob.size()) throw std::runtime_error("glTF: GLB chunk runs past the end");
            if (kind == 0x4E4F534Au) { JsonParser jp{(const char *) &blob[pos + 8], (const char *) &blob[pos + 8] + n}; g.doc = jp.value(); haveDoc = true; }
            else if (kind == 0x004E4942u && !haveBin) { binChunk.assign(blob.begin() + pos + 8, blob.begin() + pos + 8 + n); haveBin = true; }
            pos += 8 + n;
        }
        if (!haveDoc) throw std::runtime_error("glTF: GLB without a JSON chunk");
    } else { JsonParser jp{(const char *) blob.data(), (const char *) blob.data() + blob.size()}; g.doc = jp.value(); }
    if (const Json *bs = g.doc.get("buffers")) for (const Json &b : bs->arr) {
        const std::string uri = b.strOr("uri", "");
        if (!b.has("uri")) g.bufs.push_back(binChunk);
        else if (uri.compare(0, 5, "data:") == 0) g.bufs.push_back(base64(uri.substr(uri.find(',') + 1)));
        else g.bufs.push_back(g.external(uri));
    }
    std::map<long, int> texOfImage;
    auto textureId = [&](const Json *info) -> int {
        if (!info || !info->has("index")) return -1;
        const Json &tx = g.doc.at("textures").at((size_t) info->intOr("index", 0));
        if (!tx.has("source")) return -1;
        const long src = tx.intOr("source", 0);
        auto it = texOfImage.find(src); if (it != texOfImage.end()) return it->second;
        TextureImage img; int id = -1;
        if (loadTextureBytes(g.imageBytes(g.doc.at("images").at((size_t) src)), img) && img.channels_ >= 3) { scene.textures.push_back(std::move(img)); id = (int) scene.textures.size() - 1; }
        return texOfImage[src] = id;
    };
    size_t nprims = 0;
    if (const Json *ms = g.doc.get("meshes")) for (const Json &m : ms->arr) if (const Json *ps = m.get("primitives")) nprims += ps->arr.size();
    const Json *mats = g.doc.get("materials");
    scene.materials.reserve(scene.materials.size() + (mats ? mats->arr.size() : 0) + nprims * 4 + 1);   // Mesh::material points into the vector
    std::map<std::string, int> matIndex;
    if (mats) for (const Json &m : mats->arr) {                         // loader.cpp:104-150
        const std::string name = m.strOr("name", "");
        if (matIndex.count(name)) continue;
        const Json *pbr = m.get("pbrMetallicRoughness"); const Json none;
        const Json &pb = pbr ? *pbr : none;
        Material mt; mt.type = Material::METALLIC_ROUGHNESS; mt.albedo = Vec3(1, 1, 1);
        mt.alphaX = (float) pb.numOr("metallicFactor", 1.0); mt.alphaY = (float) pb.numOr("roughnessFactor", 1.0);
        mt.albedoTexId = textureId(pb.get("baseColorTexture")); mt.metallicRoughnessTexId = textureId(pb.get("metallicRoughnessTexture"));
        scene.materials.push_back(mt); matIndex[name] = (int) scene.materials.size() - 1;
    }
    struct Visit { long node; Mat4d parent; };
    std::vector<Visit> stack;
    const Json &sc = g.doc.at("scenes").at((size_t) g.doc.intOr("scene", 0));
    { const Json &roots = sc.at("nodes"); for (size_t i = roots.arr.size(); i-- > 0;) stack.push_back({(long) roots.arr[i].num, identity4()}); }
    size_t guard = 0;
    while (!stack.empty()) {                                            // depth-first, children in order (the Python loader's recursion)
        if (++guard > 1000000) throw std::runtime_error("glTF: node graph too large or cyclic");
        const Visit v = stack.back(); stack.pop_back();
        const Json &n = g.doc.at("nodes").at((size_t) v.node);
        const Mat4d m = mul4(v.parent, nodeMatrix(n));
        if (n.has("mesh")) {
            const Json &mesh = g.doc.at("meshes").at((size_t) n.intOr("mesh", 0));
            for (const Json &prim : mesh.at("primitives").arr) {
                if (prim.intOr("mode", 4) != 4) continue;
                const Json &att = prim.at("attributes");
                int nc; size_t np;
                std::vector<double> pos = g.accessor(att.intOr("POSITION", -1), nc, np); if (nc != 3) throw std::runtime_error("glTF: POSITION must be VEC3");
                std::vector<long> idx;
                if (prim.has("indices")) { int ic; size_t ni; const std::vector<double> iv = g.accessor(prim.intOr("indices", 0), ic, ni); idx.resize(ni * ic); for (size_t k = 0; k < idx.size(); ++k) idx[k] = (long) iv[k]; }
                else { idx.resize(np); for (size_t k = 0; k < np; ++k) idx[k] = (long) k; }
                idx.resize(idx.size() / 3 * 3);
                for (long k : idx) if (k < 0 || (size_t) k >= np) throw std::runtime_error("glTF: index out of range");
                std::vector<double> uv, nrm; bool haveUv = false, haveN = false; size_t cnt;
                if (att.has("TEXCOORD_0")) { uv = g.accessor(att.intOr("TEXCOORD_0", 0), nc, cnt); haveUv = nc == 2 && cnt >= np; }
                if (att.has("NORMAL")) { nrm = g.accessor(att.intOr("NORMAL", 0), nc, cnt); haveN = nc == 3 && cnt >= np; }
                if (!haveN) {                                            // GenNormals: flat, vertices un-shared
                    std::vector<double> p2(idx.size() * 3), u2(haveUv ? idx.size() * 2 : 0);
                    for (size_t k = 0; k < idx.size(); ++k) { for (int a = 0; a < 3; ++a) p2[3 * k + a] = pos[3 * idx[k] + a]; if (haveUv) { u2[2 * k] = uv[2 * idx[k]]; u2[2 * k + 1] = uv[2 * idx[k] + 1]; } }
                    pos.swap(p2); if (haveUv) uv.swap(u2); np = idx.size();
                    for (size_t k = 0; k < np; ++k) idx[k] = (long) k;
                    nrm.assign(np * 3, 0.0);
                    for (size_t f = 0; f + 2 < np; f += 3) {
                        const double *a = &pos[3 * f], *b = &pos[3 * f + 3], *c = &pos[3 * f + 6];
                        const double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
                        double fn[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                        const double l = std::sqrt(fn[0] * fn[0] + fn[1] * fn[1] + fn[2] * fn[2]);
                        for (int k = 0; k < 3; ++k) for (int a2 = 0; a2 < 3; ++a2) nrm[3 * (f + k) + a2] = l > 0 ? fn[a2] / l : 0.0;
                    }
                }
                // inverse transpose of the upper 3x3 (cofactors / determinant)
                const double (*M)[4] = m.m; double co[3][3];
                co[0][0] = M[1][1] * M[2][2] - M[1][2] * M[2][1]; co[0][1] = M[1][2] * M[2][0] - M[1][0] * M[2][2]; co[0][2] = M[1][0] * M[2][1] - M[1][1] * M[2][0];
                co[1][0] = M[0][2] * M[2][1] - M[0][1] * M[2][2]; co[1][1] = M[0][0] * M[2][2] - M[0][2] * M[2][0]; co[1][2] = M[0][1] * M[2][0] - M[0][0] * M[2][1];
                co[2][0] = M[0][1] * M[1][2] - M[0][2] * M[1][1]; co[2][1] = M[0][2] * M[1][0] - M[0][0] * M[1][2]; co[2][2] = M[0][0] * M[1][1] - M[0][1] * M[1][0];
                const double det = M[0][0] * co[0][0] + M[0][1] * co[0][1] + M[0][2] * co[0][2];
                if (det == 0) throw std::runtime_error("glTF: singular node transform");
                std::shared_ptr<Vec3> wp(new Vec3[np], std::default_delete<Vec3[]>()), wn(new Vec3[np], std::default_delete<Vec3[]>());
                std::shared_ptr<Vec2f> wuv; if (haveUv) wuv.reset(new Vec2f[np], std::default_delete<Vec2f[]>());
                std::shared_ptr<Vec3i> wi(new Vec3i[idx.size() / 3], std::default_delete<Vec3i[]>());
                for (size_t k = 0; k < np; ++k) {
                    const double *p = &pos[3 * k], *q = &nrm[3 * k];
                    wp.get()[k] = Vec3((float) (M[0][0] * p[0] + M[0][1] * p[1] + M[0][2] * p[2] + M[0][3]), (float) (M[1][0] * p[0] + M[1][1] * p[1] + M[1][2] * p[2] + M[1][3]),
                                       (float) (M[2][0] * p[0] + M[2][1] * p[1] + M[2][2] * p[2] + M[2][3]));
                    double t[3]; for (int r = 0; r < 3; ++r) t[r] = (co[r][0] * q[0] + co[r][1] * q[1] + co[r][2] * q[2]) / det;
                    const double l = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
                    wn.get()[k] = l > 0 ? Vec3((float) (t[0] / l), (float) (t[1] / l), (float) (t[2] / l)) : Vec3((float) t[0], (float) t[1], (float) t[2]);
                    if (haveUv) { wuv.get()[k].x = (float) uv[2 * k]; wuv.get()[k].y = 1.0f - (float) uv[2 * k + 1]; }
                }
                for (size_t f = 0; f < idx.size() / 3; ++f) wi.get()[f] = Vec3i((int) idx[3 * f], (int) idx[3 * f + 1], (int) idx[3 * f + 2]);
                int mat;
                if (prim.has("material")) mat = matIndex.at(g.doc.at("materials").at((size_t) prim.intOr("material", 0)).strOr("name", ""));
                else { Material dm; dm.type = Material::DIFFUSE; dm.albedo = Vec3(1.0f, 0.3f, 0.5f); scene.materials.push_back(dm); mat = (int) scene.materials.size() - 1; }   // loader.cpp:206-209
                Mesh out(wi.get(), (int) (idx.size() / 3), wp.get(), (int) np, wn.get(), wuv.get(), &scene.materials[mat]);
                out.name = mesh.strOr("name", "mesh_" + std::to_string(scene.meshes.size()));
                scene.meshes.push_back(out);
                scene.storage.push_back(wp); scene.storage.push_back(wn); scene.storage.push_back(wi); if (wuv) scene.storage.push_back(wuv);
                const int meshIndex = (int) scene.meshes.size() - 1;
                for (size_t f = 0; f < idx.size() / 3; ++f) scene.triangles.push_back(Triangle{(int) f, meshIndex});
            }
        }
        if (const Json *ch = n.get("children")) for (size_t i = ch->arr.size(); i-- > 0;) stack.push_back({(long) ch->arr[i].num, m});
    }
}

} // namespace jtxmi
