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
// glTF / GLB: the Python mirror (jtx_pathtracer_amd.gltf) -- a JSON + PNG reader is not worth a second copy in C++.
#pragma once
#include "jtx_host_api.hpp"

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

inline void loadScene(const std::string &path, Scene &scene) {
    using namespace loader_detail;
    std::ifstream in(path);
    if (!in) throw std::runtime_error("loadScene: cannot open " + path);
    if (path.size() < 4 || (path.substr(path.size() - 4) != ".obj" && path.substr(path.size() - 4) != ".OBJ"))
        throw std::runtime_error("loadScene: only Wavefront OBJ on the C++ side (glTF / GLB: jtx_pathtracer_amd.gltf.load_gltf)");
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

} // namespace jtxmi
