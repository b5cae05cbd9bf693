// host_loader_test.cpp -- loadScene (host/jtx_host_loader.hpp) on an OBJ file: prints what it built so that
// tests/test_host_loader_cpu.py can compare it with the Python mirror's scenes.load_obj.  No GPU (the decoders are host code).
#include "../../jtx-pathtracer_amd/host/jtx_host_loader.hpp"
#include <cstdio>
static unsigned crc(unsigned c, const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *) p;
    c = ~c;
    for (size_t i = 0; i < n; ++i) { c ^= b[i]; for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u))); }
    return ~c;
}
int main(int argc, char **argv) {
    jtxmi::Scene s;
    try { jtxmi::loadScene(argv[1], s); } catch (const std::exception &e) { std::printf("error %s\n", e.what()); return 1; }
    std::printf("meshes %zu triangles %zu materials %zu textures %zu\n", s.meshes.size(), s.triangles.size(), s.materials.size(), s.textures.size());
    for (const auto &m : s.meshes) {
        unsigned c = crc(0, m.vertices, sizeof(float) * 3 * m.numVertices);
        c = crc(c, m.normals, sizeof(float) * 3 * m.numVertices);
        if (m.uvs) c = crc(c, m.uvs, sizeof(float) * 2 * m.numVertices);
        c = crc(c, m.indices, sizeof(int) * 3 * m.numIndices);
        std::printf("mesh %s %d %d uv %d mat %d tex %d crc %08x\n", m.name.c_str(), m.numVertices, m.numIndices, m.uvs ? 1 : 0,
                    (int) (m.material - s.materials.data()), m.material->albedoTexId, c);
    }
    if (argc > 2) {                                            // raw dump for comparisons with a tolerance
        FILE *f = std::fopen(argv[2], "wb");
        for (const auto &m : s.meshes) {
            const int hdr[4] = {m.numVertices, m.numIndices, m.uvs ? 1 : 0, (int) (m.material - s.materials.data())};
            std::fwrite(hdr, 4, 4, f); std::fwrite(m.vertices, 12, m.numVertices, f); std::fwrite(m.normals, 12, m.numVertices, f);
            if (m.uvs) std::fwrite(m.uvs, 8, m.numVertices, f);
            std::fwrite(m.indices, 12, m.numIndices, f);
        }
        for (const auto &mt : s.materials) { const float v[8] = {(float) mt.type, mt.albedo.x, mt.albedo.y, mt.albedo.z, mt.alphaX, mt.alphaY, (float) mt.albedoTexId, (float) mt.metallicRoughnessTexId}; std::fwrite(v, 4, 8, f); }
        std::fclose(f);
    }
    for (const auto &t : s.textures) std::printf("texture %d %d %d crc %08x\n", t.width_, t.height_, t.channels_, crc(0, t.data_.data(), sizeof(float) * t.data_.size()));
    return 0;
}
