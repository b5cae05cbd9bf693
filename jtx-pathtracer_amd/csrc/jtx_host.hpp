// jtx_host.hpp -- host-side pieces of the core shared between the BVH builder and the C-ABI.
#pragma once
#include "../../include/jtx_mi.h"

#include <stdexcept>
#include <vector>

namespace jtxh {

struct BvhResult {
    std::vector<jtx_mi_bvh_node> nodes;   // depth-first, first child at i+1 (bvh.cpp:135-149)
    std::vector<jtx_mi_tri_ref>  refs;    // Scene::triangles_ after the build
    std::vector<int32_t>         orig;    // refs[i]'s index in the scene's own Scene::triangles (the builder's input order)
    int   max_depth = 0;                  // deepest node (root = 0) = traversal stack bound
    float scene_radius = 0;
};

void buildBVH(const jtx_mi_scene_desc &desc, BvhResult &out);
void meshVertices(const jtx_mi_mesh &m, int tri, float v0[3], float v1[3], float v2[3]);

} // namespace jtxh
