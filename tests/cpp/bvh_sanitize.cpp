// bvh_sanitize.cpp -- the host BVH builder (csrc/jtx_bvh_build.cpp: big subtrees on std::async threads, stitched) compiled
// for the host under ThreadSanitizer or AddressSanitizer + UBSan (tests/test_bvh_sanitize_cpu.py builds it both ways): the
// parallel build must be race-free and must give the tree of the one-thread build, node for node.
#include "../../jtx-pathtracer_amd/csrc/jtx_host.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>

static uint32_t st = 99u;
static float rnd() { st = st * 747796405u + 2891336453u; uint32_t w = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u; return (float) (((w >> 22u) ^ w) & 0xffffffu) / 16777216.0f; }

int main(int argc, char **argv) {
    const int ntri = argc > 1 ? std::atoi(argv[1]) : 60000;
    std::vector<float> v((size_t) 9 * ntri), nrm((size_t) 9 * ntri, 0.0f);
    std::vector<int32_t> idx((size_t) 3 * ntri);
    for (int t = 0; t < ntri; ++t) {
        const float c[3] = {rnd() * 100.0f, rnd() * 40.0f, rnd() * 100.0f};
        for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) v[(size_t) 9 * t + 3 * k + a] = c[a] + (rnd() - 0.5f) * (t % 97 == 0 ? 20.0f : 0.8f);
        for (int k = 0; k < 3; ++k) { idx[(size_t) 3 * t + k] = 3 * t + k; nrm[(size_t) 9 * t + 3 * k + 1] = 1.0f; }
    }
    for (int t = 0; t < 64 && t + 1 < ntri; t += 2) std::memcpy(&v[(size_t) 9 * (t + 1)], &v[(size_t) 9 * t], 9 * sizeof(float));   // duplicates: equal centroids
    jtx_mi_mesh mesh{};
    mesh.num_triangles = ntri; mesh.num_vertices = 3 * ntri; mesh.indices = idx.data(); mesh.vertices = v.data(); mesh.normals = nrm.data(); mesh.material = 0;
    for (int i = 0; i < 4; ++i) mesh.transform[5 * i] = 1.0f;
    std::vector<jtx_mi_tri_ref> refs((size_t) ntri);
    for (int t = 0; t < ntri; ++t) refs[(size_t) t] = jtx_mi_tri_ref{t, 0};
    jtx_mi_material mat{};
    jtx_mi_scene_desc d{};
    d.num_meshes = 1; d.meshes = &mesh; d.num_tri_refs = ntri; d.tri_refs = refs.data(); d.num_materials = 1; d.materials = &mat; d.max_prims_in_node = 1;
    jtxh::BvhResult par, seq;
    setenv("JTX_BVH_THREADS", "8", 1);
    jtxh::buildBVH(d, par);
    setenv("JTX_BVH_THREADS", "1", 1);
    jtxh::buildBVH(d, seq);
    if (par.nodes.size() != seq.nodes.size() || par.max_depth != seq.max_depth ||
        std::memcmp(par.nodes.data(), seq.nodes.data(), par.nodes.size() * sizeof(jtx_mi_bvh_node)) != 0 ||
        std::memcmp(par.refs.data(), seq.refs.data(), par.refs.size() * sizeof(jtx_mi_tri_ref)) != 0) { std::printf("parallel and sequential trees differ\n"); return 1; }
    // a bad description must throw, not read out of bounds
    int bad = 0;
    { jtx_mi_tri_ref r2 = {ntri, 0}; jtx_mi_scene_desc e = d; e.num_tri_refs = 1; e.tri_refs = &r2; jtxh::BvhResult o; try { jtxh::buildBVH(e, o); } catch (const std::exception &) { ++bad; } }
    { std::vector<int32_t> i2(idx); i2[7] = 3 * ntri; jtx_mi_mesh m2 = mesh; m2.indices = i2.data(); jtx_mi_scene_desc e = d; e.meshes = &m2; jtxh::BvhResult o; try { jtxh::buildBVH(e, o); } catch (const std::exception &) { ++bad; } }
    { jtx_mi_tri_ref r2 = {0, 3}; jtx_mi_scene_desc e = d; e.num_tri_refs = 1; e.tri_refs = &r2; jtxh::BvhResult o; try { jtxh::buildBVH(e, o); } catch (const std::exception &) { ++bad; } }
    std::printf("nodes %zu depth %d bad-input errors %d\n", par.nodes.size(), par.max_depth, bad);
    return bad == 3 ? 0 : 1;
}
