// jtx_capi.hip -- implementation of the C-ABI declared in include/jtx_mi.h.
//
// Host responsibilities (all that is left of Scene / StaticCamera on the CPU): build the BVH, bake
// mesh transforms, lay the scene out for the kernels, derive the camera basis (Camera::init,
// camera.cpp:7-31), launch, and move the film.  There is deliberately no CPU rendering path.
#include "jtx_host.hpp"
#include "jtx_launch.hpp"
#include "jtx_wide_quant.hpp"
#include "jtx_progressive.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <csignal>
#include <cxxabi.h>
#include <deque>
#include <execinfo.h>
#include <fcntl.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

using namespace jtx;

namespace {

thread_local std::string g_err;

int fail(const std::string &msg) { g_err = msg; return 1; }

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Host memory the library does not own (std::vector storage, caller arrays) never reaches a HIP copy directly: bulk transfers go
// through a page-locked buffer of the library's own, so the GPU only ever touches host pages that this library allocated page-locked
// and that stay put until it frees them.  (Round 4: an intermittent "Memory access fault by GPU ... on address <a malloc-heap
// address>" during jtx_mi_scene_create -- twice in ~35 runs of the GPU suite -- while the only GPU work in flight were this file's
// uploads from vectors; the suite also page-locks camera buffers that live in the malloc heap.  Whatever the runtime does with pageable
// pointers -- pin them, cache the pin, look them up again after the heap has been trimmed and re-grown -- it now does not get any
// from here.  The Python mirror's camera buffers moved out of the malloc heap for the same reason: api.py.)
struct HostStage {
    std::mutex mu; void *p = nullptr; size_t cap = 0;
    void *get(size_t bytes) {                       // caller holds mu
        if (bytes > cap) {
            if (p) (void) hipHostFree(p);
            p = nullptr; cap = 0;
            HIPCHK(hipHostMalloc(&p, bytes, hipHostMallocPortable));
            cap = bytes;
        }
        return p;
    }
};
// one stage per DEVICE (the current one of the calling thread): the per-device worker threads of a multi-device scene creation and of
// jtx_mi_multi_render's delivery copy side by side instead of queueing behind one process-wide lock.  Never freed: the
// stages must outlive every scene, and static destructors run after the runtime's.
HostStage &hostStage() {
    static HostStage *st = new HostStage[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return st[dev & 15];
}
constexpr size_t kStageChunk = (size_t) 8 << 20, kStageMin = 4096;                   // (copies of a few words stay direct: the runtime embeds / stages them itself)

// host -> device, complete on return.  The destination must not be in use by work in flight.
void stagedH2D(void *dst, const void *src, size_t bytes) {
    if (!bytes) return;
    if (bytes < kStageMin) { HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return; }
    HostStage &st = hostStage();
    std::lock_guard<std::mutex> lk(st.mu);
    char *stage = (char *) st.get(bytes < kStageChunk ? bytes : kStageChunk);
    for (size_t off = 0; off < bytes; off += kStageChunk) {
        const size_t n = bytes - off < kStageChunk ? bytes - off : kStageChunk;
        std::memcpy(stage, (const char *) src + off, n);
        HIPCHK(hipMemcpy((char *) dst + off, stage, n, hipMemcpyHostToDevice));
    }
}
// device -> host, complete on return.  The caller has synchronised with whatever wrote the source.
void stagedD2H(void *dst, const void *src, size_t bytes) {
    if (!bytes) return;
    if (bytes < kStageMin) { HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return; }
    HostStage &st = hostStage();
    std::lock_guard<std::mutex> lk(st.mu);
    char *stage = (char *) st.get(bytes < kStageChunk ? bytes : kStageChunk);
    for (size_t off = 0; off < bytes; off += kStageChunk) {
        const size_t n = bytes - off < kStageChunk ? bytes - off : kStageChunk;
        HIPCHK(hipMemcpy(stage, (const char *) src + off, n, hipMemcpyDeviceToHost));
        std::memcpy((char *) dst + off, stage, n);
    }
}

// a page-locked host buffer owned by a scene (grow-only): the target of its asynchronous read-backs
struct PinBuf {
    void *p = nullptr; size_t cap = 0;
    void *ensure(size_t bytes) {
        if (bytes > cap) { if (p) (void) hipHostFree(p); p = nullptr; cap = 0; HIPCHK(hipHostMalloc(&p, bytes, hipHostMallocPortable)); cap = bytes; }
        return p;
    }
    ~PinBuf() { if (p) (void) hipHostFree(p); }
    PinBuf() = default; PinBuf(const PinBuf &) = delete; PinBuf &operator=(const PinBuf &) = delete;
};

template <class T> struct DevBuf {
    T *p = nullptr; size_t n = 0, cap = 0;       // n: elements in use; cap: elements allocated
    void alloc(size_t count) { release(); if (count) HIPCHK(hipMalloc((void **) &p, count * sizeof(T))); n = cap = count; }
    void ensure(size_t count) { if (count <= cap) n = count; else alloc(count); }     // grow-only: an edit loop re-uses its buffers
    void upload(const std::vector<T> &v) { alloc(v.size()); stagedH2D(p, v.data(), v.size() * sizeof(T)); }
    void fill(const std::vector<T> &v) { ensure(v.size()); stagedH2D(p, v.data(), v.size() * sizeof(T)); }
    void release() { if (p) (void) hipFree(p); p = nullptr; n = cap = 0; }
    void swap(DevBuf &o) noexcept { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); }
    ~DevBuf() { release(); }
};

// Every scene lives on the device that was current at jtx_mi_scene_create; the current HIP device is per THREAD
// (DynamicCamera's worker thread never called hipSetDevice), so each scene-taking entry point switches to the
// scene's device for its duration and restores the caller's.
struct DeviceGuard {
    int prev = -1; bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) { HIPCHK(hipSetDevice(dev)); switched = true; }
    }
    ~DeviceGuard() { if (switched && prev >= 0) (void) hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete; DeviceGuard &operator=(const DeviceGuard &) = delete;
};

constexpr size_t kMaxRadBytes = (size_t) 8 << 30;   // per-path radiance buffer of one pass (k_render_paths); opts.max_record_mb / env JTX_MAX_RAD_MB override
constexpr int kResolverMax = 128;                 // workgroups of the progressive resolver (words of prog_host per kind)
constexpr int kWorkRing = 1024;                   // chunk counters of k_render_paths launches (power of two): at most half of them per pass
constexpr size_t kLdsThreadedBudget = 20 * 1024; // 8 threaded node orderings + tris staged in LDS when they fit this (8 blocks/CU)

} // namespace

namespace {
bool hostPinned(const void *p) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void) hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}
}

int jtx_capi_fail(const std::string &msg) { return fail(msg); }     // for the other translation units of the library
// device -> caller memory, complete on return: direct when the caller's buffer is page-locked, through the library's staging otherwise (throws)
void jtx_capi_d2h(void *dst, const void *src, size_t bytes, hipStream_t st) {
    if (hostPinned(dst)) { HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st)); return; }
    HIPCHK(hipStreamSynchronize(st));
    stagedD2H(dst, src, bytes);
}

namespace {
// after the stream has drained: did the last persistent launch stop on the cancellation flag (k_render_paths pushes its
// chunk counter past 2^30 then, and k_resolve_samples skips the pass)?  Launches that never poll (counting, alternate
// integrators, wavefront) complete, and a cancellation is honoured between their passes.
bool passAbandoned(jtx_mi_scene &s, int slot);
}

struct jtx_mi_scene {
    jtxh::BvhResult bvh;
    DevBuf<float4> tnodes, tris, shade;
    DevBuf<uint4> wide;
    DevBuf<DMaterial> materials;
    DevBuf<DLight> lights;
    DevBuf<DTexture> textures;
    DevBuf<float> texels;
    DevBuf<unsigned long long> counters;
    // device refit (jtx_refit.hip): what a transform edit needs to recompute every position-dependent record in place
    DevBuf<float4> prim_src, pbox, nbox;
    DevBuf<float> mesh_xf;
    DevBuf<int> leaf_nodes, level_nodes, rec_node, wide_map, wide_fail;
    std::vector<float> mesh_xf_host;
    PinBuf pin_nb;                    // page-locked landing buffer of the refit's read-back (node boxes)
    float *mesh_xf_pinned = nullptr;  // page-locked copy the uploads read: the first PAGEABLE host-to-device copy after a render took 8-20 ms (round 4 trace)
    std::vector<int> level_begin;    // per interior depth: offsets into level_nodes
    DevBuf<int> orig_id;             // per BVH-ordered primitive: its index in the scene's own Scene::triangles (input order of a device rebuild)
    int num_leaves = 0, num_wide = 0, refitted = 0, device_built = 0;
    bool xf_dirty = false;
    DevBuf<float4> lw_box; DevBuf<unsigned> lw_tab;   // flat leaf list of tiny scenes (traverseLeaves)
    // jtx_mi_scene_rebuild: the SECOND set of every structure a rebuild replaces, plus the builder's scratch.  A rebuild writes the
    // spare set while the live one stays untouched, and swaps the two when everything stands (failure-atomic; no allocation from the
    // second rebuild on).  Costs the geometry's device memory twice -- HBM is sized for it.
    struct RebuildSpare {
        DevBuf<float4> src, tris, shade, nbox, tnodes, lw_box; DevBuf<int> orig, leaves, levels, rec_node, map; DevBuf<uint4> wide;
        DevBuf<unsigned> lw_tab; DevBuf<DLight> lights;
        DevBuf<int> order, pos, size; DevBuf<jtx_mi_bvh_node> hn;       // scratch
        PinBuf pin_nodes, pin_ord;       // page-locked landing buffers of the rebuild's read-backs (nodes, primitive order)
        DevBuildArena arena;
        ~RebuildSpare() { if (arena.base) (void) hipFree(arena.base); }
        size_t bytes() const {           // device memory the set holds (what jtx_mi_scene_release_rebuild gives back)
            return (src.cap + tris.cap + shade.cap + nbox.cap + tnodes.cap + lw_box.cap) * sizeof(float4) +
                   (orig.cap + leaves.cap + levels.cap + rec_node.cap + map.cap + order.cap + pos.cap + size.cap) * sizeof(int) +
                   wide.cap * sizeof(uint4) + lw_tab.cap * sizeof(unsigned) + lights.cap * sizeof(DLight) + hn.cap * sizeof(jtx_mi_bvh_node) + arena.cap;
        }
        void release() {
            src.release(); tris.release(); shade.release(); nbox.release(); tnodes.release(); lw_box.release(); orig.release(); leaves.release();
            levels.release(); rec_node.release(); map.release(); wide.release(); lw_tab.release(); lights.release(); order.release(); pos.release();
            size.release(); hn.release();
            if (arena.base) { (void) hipFree(arena.base); arena.base = nullptr; arena.cap = 0; }
            if (pin_nodes.p) { (void) hipHostFree(pin_nodes.p); pin_nodes.p = nullptr; pin_nodes.cap = 0; }
            if (pin_ord.p) { (void) hipHostFree(pin_ord.p); pin_ord.p = nullptr; pin_ord.cap = 0; }
        }
    } spare;
    // Per-path radiance records of k_render_paths (and of the strata-split mode): JTX_MI_FRAME_SLOTS sets, so that several frames of one
    // scene can be in flight at once (opts.frame_slot): the last chunks of frame i -- every persistent wave spends its final ~0.4 ms with
    // ever fewer live lanes -- run beside the first chunks of frame i + 1, launched on another stream into another set, and the resolve
    // pass of frame i (which finds no free wave slot while frame i + 1 fills the chip) beside the first chunks of frame i + 2.
    // slot_done[k]: recorded behind the last launch that used set k; a launch waits for it first, so renders of one slot are ordered
    // by the library whatever streams they come on, and launches that use the scene's singletons (ray counters, wavefront arrays,
    // strata-split buffers) wait for, and record into, all of them.
    DevBuf<float4> rad[JTX_MI_FRAME_SLOTS];
    hipEvent_t slot_done[JTX_MI_FRAME_SLOTS] = {};
    DevBuf<unsigned> work;           // chunk counters of the persistent k_render_paths launches (kWorkRing, used round-robin)
    unsigned work_slot = 0;
    // Per frame slot, the last pass launched in it.  last_work: chunk counter of its last k_render_paths launch (null: a launch without
    // one): >= 2^30 after its stream drained = that pass was abandoned.  A pass whose radiance records exceed the buffer cap goes in several
    // launches of consecutive strata, each resolved on its own: parts = (chunk counter, first stratum, one past the last) of every launch
    // of the pass, in order.  resolved_end: strata of the pass that are in the film (set by passAbandoned).
    struct PassPart { unsigned *work; volatile unsigned *abandoned; int begin, end; };   // (abandoned: the launch's word of abandon_host)
    struct PassRec { unsigned *last_work = nullptr; std::vector<PassPart> parts; int resolved_end = 0; } pass[JTX_MI_FRAME_SLOTS];
    int last_slot = 0;               // the slot of the last launch (jtx_mi_cancel_pending looks at that pass)
    DevBuf<float> film_acc;          // device film for jtx_mi_render (host-buffer variant)
    DevBuf<unsigned char> film_img;  // RGB8 preview
    // progressive launches (jtx_mi_render with a callback: one k_render_paths<.., PROG> launch for all passes + k_resolve_progressive beside it)
    DevBuf<unsigned> prog_ctl;       // [0] chunk counter, [32] closed-at, [64] the resolver leader's word, [128 ..] one word per persistent wave (RenderParams::prog_slots)
    hipStream_t resolve_stream = nullptr, copy_stream = nullptr;   // the resolver's stream; the stream previews travel on
    hipEvent_t prog_ready = nullptr, prog_paths_done = nullptr, prog_resolved = nullptr;
    unsigned *prog_host = nullptr;   // host-mapped: [0, 128) started, [128, 256) progress: one word per resolver workgroup; [256] the host's word for the path kernel
    unsigned prog_live_epoch = 0;
    unsigned *prog_host_dev = nullptr;
    unsigned prog_epoch = 0;
    DevScene dev{};
    // wavefront integrator state (sized for pixels * strata-per-batch slots)
    DevBuf<float> wf_floats;         // all float SoA arrays, carved
    DevBuf<float4> wf_hit;
    DevBuf<int> wf_ints;             // flags, depth, rng
    DevBuf<unsigned> wf_heads;       // one work-queue head per launch
    size_t wf_slots = 0, wf_nheads = 0;
    int num_cus = 256;
    float ms_by_kind[5] = {0, 0, 0, 0, 0};   // generate, trace-closest, shade, trace-any, resolve (profiled renders)
    int   n_by_kind[5] = {0, 0, 0, 0, 0};
    hipStream_t stream = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending, free_events;     // timing pairs of the launches since the last jtx_mi_kernel_time

    size_t device_bytes = 0;
    int device = 0;
    std::mutex mu;
    unsigned char *pin_img = nullptr; float *pin_acc = nullptr; size_t pin_pixels = 0;   // pinned staging for pageable caller buffers
    int last_completed = 0;              // strata in the film after the last jtx_mi_render (== sample_end unless cancelled)
    unsigned *stop_host = nullptr;       // host-mapped cancellation word (jtx_mi_cancel), read by the persistent kernels
    const unsigned *stop_dev = nullptr;
    unsigned *abandon_host = nullptr;    // host-mapped: one word per launch of the work ring, set by k_resolve_samples when its pass was abandoned
    unsigned *abandon_dev = nullptr;

    ~jtx_mi_scene() {
        for (auto &e : pending) { (void) hipEventDestroy(e.first); (void) hipEventDestroy(e.second); }
        for (auto &e : free_events) { (void) hipEventDestroy(e.first); (void) hipEventDestroy(e.second); }
        for (auto &e : slot_done) if (e) (void) hipEventDestroy(e);
        for (hipEvent_t e : {prog_ready, prog_paths_done, prog_resolved}) if (e) (void) hipEventDestroy(e);
        for (hipStream_t x : {resolve_stream, copy_stream}) if (x) (void) hipStreamDestroy(x);
        if (prog_host) (void) hipHostFree(prog_host);
        if (stream) (void) hipStreamDestroy(stream);
        if (stop_host) (void) hipHostFree(stop_host);
        if (abandon_host) (void) hipHostFree(abandon_host);
        if (mesh_xf_pinned) (void) hipHostFree(mesh_xf_pinned);
        if (pin_img) (void) hipHostFree(pin_img);
        if (pin_acc) (void) hipHostFree(pin_acc);
    }
};

namespace {

inline void xformNormal(const float m[16], const float *v, float out[3]) {     // Transform::applyToNormal: upper 3x3
    for (int r = 0; r < 3; ++r) out[r] = m[4 * r + 0] * v[0] + m[4 * r + 1] * v[1] + m[4 * r + 2] * v[2];
}

inline float srgbToLinear(float v) {                                            // color.hpp:13-23
    return v <= 0.04045f ? v / 12.92f : std::pow((v + 0.055f) / 1.055f, 2.4f);
}

void validate(const jtx_mi_scene_desc &d) {
    if (d.num_meshes < 0 || d.num_tri_refs < 0 || d.num_materials < 0 || d.num_lights < 0 || d.num_textures < 0)
        throw std::runtime_error("negative count in scene description");
    for (int i = 0; i < d.num_meshes; ++i) {
        const jtx_mi_mesh &m = d.meshes[i];
        if (!m.indices || !m.vertices || !m.normals) throw std::runtime_error("mesh with null indices/vertices/normals");
        if (m.material < 0 || m.material >= d.num_materials) throw std::runtime_error("mesh.material out of range");
    }
    for (int i = 0; i < d.num_materials; ++i) {
        const jtx_mi_material &m = d.materials[i];
        if (m.type < 0 || m.type > 4) throw std::runtime_error("material.type out of range (0..3 Material::Type, 4 THIN_DIELECTRIC)");
        if (m.albedo_tex < -1 || m.albedo_tex >= d.num_textures || m.mr_tex < -1 || m.mr_tex >= d.num_textures)
            throw std::runtime_error("material texture id out of range (-1 = none)");
    }
    for (int i = 0; i < d.num_textures; ++i) {
        const jtx_mi_texture &t = d.textures[i];
        if (t.width <= 0 || t.height <= 0 || t.channels < 3 || !t.texels) throw std::runtime_error("texture needs w,h > 0, >= 3 channels and texels");
    }
    for (int i = 0; i < d.num_lights; ++i)
        if (d.lights[i].type < 0 || d.lights[i].type > 1) throw std::runtime_error("light.type out of range");
}

// ---- wide-node builder (layout + proof sketch: traverseWide in jtx_scene_dev.hpp) ----
constexpr int kMaxWideDepth = 24;          // stack = depth x 8 B x 256 lanes of LDS per workgroup: 11 levels (22 KB) still run 7 workgroups / CU,
                                           // 24 levels (48 KB) 3; deeper trees (degenerate input) keep the binary records

struct WideBuilder {
    const std::vector<jtx_mi_bvh_node> &nodes;
    std::vector<uint4> &out;
    std::vector<int> leaves;           // leaf count of every binary subtree
    // optimal cut (surface-area cost): F[8 n + k - 1] = the least sum of box areas of all wide nodes needed below binary
    // node n when n's subtree may occupy at most k child slots of its parent's wide node (k = 1: n itself is the child)
    std::vector<float> F;
    std::vector<int32_t> *refit = nullptr;   // optional: 16 ints per wide node {granule, binary node, #interior, #leaves, leaf-record granule, -, -, -, children[8]}
    int depth = 0;
    bool ok = true;

    bool leaf(int i) const { return nodes[i].num_prims != 0; }

    void writeLeaf(size_t at, int i) {
        const jtx_mi_bvh_node &n = nodes[i];
        auto fb = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
        out[at] = make_uint4(fb(n.pmin[0]), fb(n.pmax[0]), fb(n.pmin[1]), fb(n.pmax[1]));
        out[at + 1] = make_uint4(fb(n.pmin[2]), fb(n.pmax[2]), (uint32_t) n.offset, (uint32_t) n.num_prims);
    }

    static double area(const jtx_mi_bvh_node &n) {
        const double dx = (double) n.pmax[0] - n.pmin[0], dy = (double) n.pmax[1] - n.pmin[1], dz = (double) n.pmax[2] - n.pmin[2];
        return dx * dy + dy * dz + dz * dx;
    }

    float Fk(int n, int k) const { return F[8 * (size_t) n + k - 1]; }

    // bottom-up (children stand behind their parent in the depth-first array): W(n) = area(n) + the best split of
    // at most 8 slots over n's two subtrees; F(n, k) = min(W(n), best split of at most k slots); leaves cost nothing here
    // (their boxes are tested whatever the cut).  The classic surface-area collapse of wide-BVH builders, applied to the
    // reference's binary tree -- any cut gives the same hits (leaf-walk argument in jtx_scene_dev.hpp), this one the fewest node steps.
    void prepareCuts() {
        const size_t n = nodes.size();
        F.assign(8 * n, 0.0f);
        for (size_t i = n; i-- > 0;) {
            if (leaf((int) i)) continue;
            const int L = (int) i + 1, R = nodes[i].offset;
            float best[9];                                  // best[k]: best split with at most k slots, k >= 2
            for (int k = 2; k <= 8; ++k) {
                float c = 3.0e38f;
                for (int a = 1; a < k; ++a) { const float v = Fk(L, a) + Fk(R, k - a); if (v < c) c = v; }
                best[k] = c;
            }
            const float W = (float) area(nodes[i]) + best[8];
            F[8 * i] = W;
            for (int k = 2; k <= 8; ++k) F[8 * i + k - 1] = best[k] < W ? best[k] : W;
        }
    }

    void fill(int b, size_t at, int level) {
        if (!ok) return;
        if (level + 1 > depth) depth = level + 1;
        const jtx_mi_bvh_node &nb = nodes[b];
        // treelet: open the binary subtree below b, always at the child with the largest box, until 8 children stand
        struct TNode { int node, left, right; };           // left/right: treelet indices, -1 = a child of the wide node
        TNode t[15]; int nt = 0;
        auto add = [&](int node) { t[nt] = {node, -1, -1}; return nt++; };
        add(b);
        t[0].left = add(b + 1); t[0].right = add(nb.offset);
        if (!F.empty()) {
            // the cut of at most 8 subtrees that minimises the summed box area of the wide nodes below (prepareCuts)
            nt = 1; t[0].left = t[0].right = -1;
            struct Job { int ti, k; } jobs[16]; int nj = 0;
            jobs[nj++] = {0, 8};
            bool root = true;
            while (nj) {
                const Job j = jobs[--nj];
                const int node = t[j.ti].node;
                if (leaf(node)) continue;
                const int L = node + 1, R = nodes[node].offset;
                int bi = 0; float bc = root ? 3.0e38f : Fk(node, 1);            // bi = 0: stays a child of this wide node
                for (int i = 1; i < j.k; ++i) { const float c = Fk(L, i) + Fk(R, j.k - i); if (c < bc) { bc = c; bi = i; } }
                root = false;
                if (bi == 0) continue;
                const int l = add(L), r = add(R);
                t[j.ti].left = l; t[j.ti].right = r;
                jobs[nj++] = {r, j.k - bi}; jobs[nj++] = {l, bi};
            }
        } else {
        int nchild = 2;
        // 1. swallow whole subtrees that fit into the free slots, smallest first (a subtree of m leaves costs m - 1
        //    slots and saves a wide node that would test only m boxes); 2. otherwise open the child with the largest box
        while (nchild < 8) {
            int best = -1, bestLeaves = 1 << 30;
            for (int i = 1; i < nt; ++i)
                if (t[i].left < 0 && !leaf(t[i].node) && leaves[t[i].node] - 1 <= 8 - nchild && leaves[t[i].node] < bestLeaves) {
                    best = i; bestLeaves = leaves[t[i].node];
                }
            if (best >= 0) {
                int st[16], sp = 0; st[sp++] = best;
                while (sp) {
                    const int i = st[--sp];
                    if (leaf(t[i].node)) continue;
                    const int l = add(t[i].node + 1), r = add(nodes[t[i].node].offset);
                    t[i].left = l; t[i].right = r; ++nchild;
                    st[sp++] = r; st[sp++] = l;
                }
                continue;
            }
            double bestArea = -1.0;
            for (int i = 1; i < nt; ++i)
                if (t[i].left < 0 && !leaf(t[i].node) && area(nodes[t[i].node]) > bestArea) { best = i; bestArea = area(nodes[t[i].node]); }
            if (best < 0) break;
            const int l = add(t[best].node + 1), r = add(nodes[t[best].node].offset);
            t[best].left = l; t[best].right = r;
            ++nchild;
        }
        }
        // children left to right; slots: wide (interior) children first, then leaf records, each in that order
        int order[8], n = 0;
        { int st[16], sp = 0; st[sp++] = 0;
          while (sp) { const int i = st[--sp]; if (t[i].left < 0) order[n++] = i; else { st[sp++] = t[i].right; st[sp++] = t[i].left; } } }
        int slotOf[15]; for (int &v : slotOf) v = -1;
        int child[8], ni = 0, nl = 0;
        for (int k = 0; k < n; ++k) if (!leaf(t[order[k]].node)) { slotOf[order[k]] = ni; child[ni++] = t[order[k]].node; }
        for (int k = 0; k < n; ++k) if (leaf(t[order[k]].node)) { slotOf[order[k]] = ni + nl; child[ni + nl] = t[order[k]].node; ++nl; }
        // grid: plane = (k + q) 2^e per axis, children rounded outward on it (jtx_wide_quant.hpp: the same code builds and refits
        // the node on the device)
        jtxq::NodeGrid grid;
        if (!jtxq::nodeGrid(nb.pmin, nb.pmax, grid)) { ok = false; return; }
        uint8_t qlo[3][8] = {}, qhi[3][8] = {};
        for (int s = 0; s < ni + nl; ++s) {
            const jtx_mi_bvh_node &c = nodes[child[s]];
            uint8_t lo3[3], hi3[3];
            if (!jtxq::quantiseChild(grid, nb.pmin, nb.pmax, c.pmin, c.pmax, lo3, hi3)) { ok = false; return; }
            for (int k = 0; k < 3; ++k) { qlo[k][s] = lo3[k]; qhi[k][s] = hi3[k]; }
        }
        // visiting order per octant: the reference's near-first rule (scene.cpp:40-46) applied inside the treelet
        uint32_t perm[8];
        for (int o = 0; o < 8; ++o) {
            uint32_t pm = 0; int cnt = 0;
            int st[16], sp = 0; st[sp++] = 0;
            while (sp) {
                const int i = st[--sp];
                if (t[i].left < 0) { pm |= (uint32_t) slotOf[i] << (3 * cnt++); continue; }
                const bool neg = (o >> nodes[t[i].node].axis) & 1;
                st[sp++] = neg ? t[i].left : t[i].right;             // far child: after the near subtree
                st[sp++] = neg ? t[i].right : t[i].left;
            }
            perm[o] = pm;
        }
        // children block: [ni nodes][nl leaf records] (jtx_wide_quant.hpp)
        const size_t base = out.size();
        if (base + jtxq::blockGranules(ni, nl) >= jtxq::kMaxGranules) { ok = false; return; }
        out.resize(base + jtxq::blockGranules(ni, nl), make_uint4(0u, 0u, 0u, 0u));
        uint32_t nd[16], tw[4 * jtxq::kTails];
        jtxq::encodeGridAndPlanes(nd, grid, ni, ni + nl, qlo, qhi);
        if (!jtxq::encodeTail(tw, (uint32_t) base, perm, ni + nl)) { ok = false; return; }
        for (int g = 0; g < 4; ++g) out[at + g] = make_uint4(nd[4 * g], nd[4 * g + 1], nd[4 * g + 2], nd[4 * g + 3]);
        for (uint32_t t = 0; t < jtxq::kTails; ++t) out[at + 4 + t] = make_uint4(tw[4 * t], tw[4 * t + 1], tw[4 * t + 2], tw[4 * t + 3]);
        if (b == 0) {                                                   // the root-peel record: group word, orders, the children's exact boxes
            uint32_t rec[4 * 14] = {};
            jtxq::encodePeelHeader(rec, (uint32_t) base, ni, ni + nl, perm);
            for (int s = 0; s < ni + nl; ++s) jtxq::encodePeelBox(rec, s, nodes[child[s]].pmin, nodes[child[s]].pmax);
            for (int g = 0; g < 14; ++g) out[jtxq::kPeelRec + g] = make_uint4(rec[4 * g], rec[4 * g + 1], rec[4 * g + 2], rec[4 * g + 3]);
        }
        if (refit) {
            int32_t rec[16] = {(int32_t) at, b, ni, nl, (int32_t) jtxq::leafAt((uint32_t) base, ni, 0), 0, 0, 0, -1, -1, -1, -1, -1, -1, -1, -1};
            for (int s2 = 0; s2 < ni + nl; ++s2) rec[8 + s2] = child[s2];
            refit->insert(refit->end(), rec, rec + 16);
        }
        for (int s = ni; s < ni + nl; ++s) writeLeaf(jtxq::leafAt((uint32_t) base, ni, s - ni), child[s]);
        for (int s = 0; s < ni; ++s) fill(child[s], jtxq::nodeAt((uint32_t) base, s), level + 1);
    }
};

bool buildWide(const std::vector<jtx_mi_bvh_node> &nodes, std::vector<uint4> &out, int &depth, std::vector<int32_t> *refitMap = nullptr) {
    out.clear(); depth = 0;
    if (refitMap) refitMap->clear();
    if (nodes.size() < 2 || nodes[0].num_prims != 0) return false;      // a single leaf: nothing to collapse
    WideBuilder wb{nodes, out};
    wb.refit = refitMap;
    wb.leaves.assign(nodes.size(), 1);
    for (size_t i = nodes.size(); i-- > 0;)
        if (nodes[i].num_prims == 0) {
            // depth-first layout: first child at i + 1, second child behind the first subtree -- also rules out cycles
            if (nodes[i].offset <= (int) i + 1 || (size_t) nodes[i].offset >= nodes.size()) return false;
            wb.leaves[i] = wb.leaves[i + 1] + wb.leaves[nodes[i].offset];
        }
    static const int sahCut = [] { const char *e = getenv("JTX_WIDE_SAH_CUT"); return e ? atoi(e) : 1; }();
    if (sahCut) wb.prepareCuts();                                     // JTX_WIDE_SAH_CUT=0: the greedy largest-box cut of round 1
    out.assign(jtxq::kFirstBlock, make_uint4(0u, 0u, 0u, 0u));         // root-peel record, root node (jtx_wide_quant.hpp)
    wb.fill(0, jtxq::kRootNode, 0);
    depth = wb.depth;
    return wb.ok;
}

// Tiny scenes: the flat leaf list of traverseLeaves (leaves in b.nodes order; per octant: leaf at position p, position of
// leaf l -- read off the threaded orderings).  pos[k * nn + g]: node g's place in octant k's order.
// -> number of leaves in the list (0: none -- more than 32 leaves, or a single node); box / tab filled (grow-only) when > 0
static bool leafPlanesInRange(const std::vector<jtx_mi_bvh_node> &nodes) {
    for (const jtx_mi_bvh_node &n : nodes) if (n.num_prims)
        for (int a = 0; a < 3; ++a) if (!(std::fabs(n.pmin[a]) <= jtx::LEAF_RANGE && std::fabs(n.pmax[a]) <= jtx::LEAF_RANGE)) return false;
    return true;
}
int buildLeafTables(const std::vector<jtx_mi_bvh_node> &nodes, const std::vector<int> &pos, DevBuf<float4> &box, DevBuf<unsigned> &tabBuf) {
    const size_t nn = nodes.size();
    std::vector<int> leafId(nn, -1); int nl = 0;
    for (size_t i = 0; i < nn; ++i) if (nodes[i].num_prims) leafId[i] = nl++;
    if (!(nl > 0 && nl <= 32 && nn > 1)) return 0;
    // the list's planes within +-2^60 (jtx::LEAF_RANGE): with a ray origin bounded the same way plane - o stays finite, which the
    // fma form of the leaf-box test relies on (jtx_scene_dev.hpp: slabRegularSel); scenes beyond that walk the binary records
    if (!leafPlanesInRange(nodes)) return 0;
    const int npad = (nl + 3) & ~3;                                     // phase A of traverseLeaves runs in groups of four
    std::vector<float4> lb(2 * (size_t) npad, make_float4(0.f, 0.f, 0.f, 0.f));
    for (size_t i = 0; i < nn; ++i) if (leafId[i] >= 0) {
        const jtx_mi_bvh_node &n = nodes[i];
        const int z = n.offset, w = n.num_prims; float fz, fw; std::memcpy(&fz, &z, 4); std::memcpy(&fw, &w, 4);
        lb[2 * (size_t) leafId[i]] = make_float4(n.pmin[0], n.pmax[0], n.pmin[1], n.pmax[1]);
        lb[2 * (size_t) leafId[i] + 1] = make_float4(n.pmin[2], n.pmax[2], fz, fw);
    }
    std::vector<unsigned> tab(128, 0u);
    for (int k = 0; k < 8; ++k) {
        std::vector<int> at(nn, -1);                                    // node standing at position i of octant k's order
        for (size_t g = 0; g < nn; ++g) at[(size_t) pos[(size_t) k * nn + g]] = (int) g;
        int p2 = 0;
        for (size_t i = 0; i < nn; ++i) {
            const int l = leafId[at[i]];
            if (l < 0) continue;
            tab[16 * k + (p2 >> 2)] |= (unsigned) l << (8 * (p2 & 3));                    // leaf visited at position p2
            tab[16 * k + 8 + (l >> 2)] |= (unsigned) p2 << (8 * (l & 3));                 // position of leaf l
            ++p2;
        }
        for (int l = nl; l < npad; ++l) tab[16 * k + 8 + (l >> 2)] |= (unsigned) l << (8 * (l & 3));   // padding: positions >= nl
    }
    box.fill(lb); tabBuf.fill(tab);
    return nl;
}
void buildLeafTables(jtx_mi_scene &s, const std::vector<int> &pos) {
    const int nl = buildLeafTables(s.bvh.nodes, pos, s.lw_box, s.lw_tab);
    s.dev.lw_box = nl ? s.lw_box.p : nullptr; s.dev.lw_tab = nl ? s.lw_tab.p : nullptr; s.dev.lw_leaves = nl;
}

// Flatten the scene into the kernel layout documented in jtx_scene_dev.hpp.
void flatten(const jtx_mi_scene_desc &d, jtx_mi_scene &s) {
    const jtxh::BvhResult &b = s.bvh;
    const size_t nn = b.nodes.size(), np = b.refs.size();
    static const bool traceCreate = getenv("JTX_TRACE_CREATE") != nullptr;    // where scene creation spends its time (stderr)
    auto lapT = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!traceCreate) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[jtx create] %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(n - lapT).count());
        lapT = n;
    };
    // the 8-ary node set depends on the binary nodes only: its thread starts first and runs beside everything below
    std::vector<int32_t> wideMap;
    std::vector<uint4> wide; int wideDepth = 0; bool wideOk = false;
    std::exception_ptr err = nullptr; std::mutex errMu;
    auto guarded = [&](auto fn) { return [&, fn] { try { fn(); } catch (...) { std::lock_guard<std::mutex> g(errMu); err = std::current_exception(); } }; };
    const char *off = getenv("JTX_NO_WIDE");
    const bool wantWide = nn && !(off && atoi(off));
    // (std::thread's constructor throws when the process may not have another thread -- a container's process / thread limit --:
    //  every helper thread of this function is optional, the creating thread then does the work itself; and a vector of
    //  joinable threads must never be unwound, that is std::terminate)
    auto wideJob = guarded([&] { wideOk = buildWide(b.nodes, wide, wideDepth, &wideMap) && wideDepth <= kMaxWideDepth; });
    std::thread wideThread;
    bool wideInline = false;
    if (wantWide) { try { wideThread = std::thread(wideJob); } catch (const std::system_error &) { wideInline = true; } }
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } wideJoin{wideThread};   // also on a throw below
    // per-primitive loops: split over host threads (each index writes its own records)
    auto forPrims = [&](auto body) {
        int nt = (int) std::thread::hardware_concurrency(); if (nt > 16) nt = 16; if (nt < 1 || np < 8192) nt = 1;
        std::vector<std::thread> ts;
        struct JoinAll { std::vector<std::thread> &v; ~JoinAll() { for (auto &t : v) if (t.joinable()) t.join(); } } joinAll{ts};
        ts.reserve(nt);
        int started = 1;
        for (int t = 1; t < nt; ++t) {
            try { ts.emplace_back(guarded([&, t] { for (size_t i = np * t / nt; i < np * (t + 1) / nt; ++i) body(i); })); started = t + 1; }
            catch (const std::system_error &) { break; }
        }
        for (size_t i = 0; i < np / nt; ++i) body(i);
        for (size_t i = np * started / nt; i < np; ++i) body(i);                   // the ranges no thread could be created for
        for (auto &t : ts) t.join();
        if (err) std::rethrow_exception(err);
    };
    std::vector<float4> tris(3 * np), shade(4 * np);
    forPrims([&](size_t i) {
        const jtx_mi_mesh &m = d.meshes[b.refs[i].mesh_index];
        const int tri = b.refs[i].index;
        float v0[3], v1[3], v2[3];
        jtxh::meshVertices(m, tri, v0, v1, v2);
        const float e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};      // v0v1, mesh.hpp:109
        const float e2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};      // v0v2, mesh.hpp:110
        tris[3 * i + 0] = make_float4(v0[0], v0[1], v0[2], e1[0]);
        tris[3 * i + 1] = make_float4(e1[1], e1[2], e2[0], e2[1]);
        float fty; const int mty = d.materials[m.material].type; std::memcpy(&fty, &mty, 4);
        tris[3 * i + 2] = make_float4(e2[2], fty, 0.f, 0.f);                    // .y: Material::type of the primitive (shade sorting)
        const int32_t *ix = m.indices + 3 * (size_t) tri;
        float n0[3], n1[3], n2[3];
        xformNormal(m.transform, m.normals + 3 * (size_t) ix[0], n0);           // getNormals mesh.hpp:92-97
        xformNormal(m.transform, m.normals + 3 * (size_t) ix[1], n1);
        xformNormal(m.transform, m.normals + 3 * (size_t) ix[2], n2);
        float uv[6] = {0, 0, 0, 0, 0, 0};                                       // no uvs => (0,0) (Q3)
        if (m.uvs) for (int k = 0; k < 3; ++k) { uv[2 * k] = m.uvs[2 * (size_t) ix[k]]; uv[2 * k + 1] = m.uvs[2 * (size_t) ix[k] + 1]; }
        float fmat; const int mat = m.material; std::memcpy(&fmat, &mat, 4);
        shade[4 * i + 0] = make_float4(n0[0], n0[1], n0[2], n1[0]);
        shade[4 * i + 1] = make_float4(n1[1], n1[2], n2[0], n2[1]);
        shade[4 * i + 2] = make_float4(n2[2], uv[0], uv[1], uv[2]);
        shade[4 * i + 3] = make_float4(uv[3], uv[4], uv[5], fmat);
    });
    lap("triangle + shading records");
    s.tris.upload(tris); s.shade.upload(shade);
    lap("  upload");
    // ---- refit sources: object-space vertices / normals per BVH-ordered primitive, the mesh transforms, node lists ----
    {
        std::vector<float4> src(5 * np);
        forPrims([&](size_t i) {
            const jtx_mi_mesh &m = d.meshes[b.refs[i].mesh_index];
            const int32_t *ix = m.indices + 3 * (size_t) b.refs[i].index;
            const float *p0 = m.vertices + 3 * (size_t) ix[0], *p1 = m.vertices + 3 * (size_t) ix[1], *p2 = m.vertices + 3 * (size_t) ix[2];
            const float *q0 = m.normals + 3 * (size_t) ix[0], *q1 = m.normals + 3 * (size_t) ix[1], *q2 = m.normals + 3 * (size_t) ix[2];
            float fm; const int mi = b.refs[i].mesh_index; std::memcpy(&fm, &mi, 4);
            src[5 * i + 0] = make_float4(p0[0], p0[1], p0[2], p1[0]);
            src[5 * i + 1] = make_float4(p1[1], p1[2], p2[0], p2[1]);
            src[5 * i + 2] = make_float4(p2[2], q0[0], q0[1], q0[2]);
            src[5 * i + 3] = make_float4(q1[0], q1[1], q1[2], q2[0]);
            src[5 * i + 4] = make_float4(q2[1], q2[2], fm, 0.f);
        });
        s.prim_src.upload(src);
        s.orig_id.upload(b.orig);
        s.pbox.alloc(2 * np);
        s.mesh_xf_host.assign((size_t) 16 * d.num_meshes, 0.f);
        for (int i = 0; i < d.num_meshes; ++i) std::memcpy(&s.mesh_xf_host[16 * (size_t) i], d.meshes[i].transform, 16 * sizeof(float));
        s.mesh_xf.upload(s.mesh_xf_host);
        std::vector<float4> nb(2 * nn);
        std::vector<int> leaves, depth(nn, 0);
        for (size_t i = 0; i < nn; ++i) {
            const jtx_mi_bvh_node &n = b.nodes[i];
            const int z = n.offset, w = n.num_prims;
            float fz, fw; std::memcpy(&fz, &z, 4); std::memcpy(&fw, &w, 4);
            nb[2 * i] = make_float4(n.pmin[0], n.pmax[0], n.pmin[1], n.pmax[1]);
            nb[2 * i + 1] = make_float4(n.pmin[2], n.pmax[2], fz, fw);
            if (n.num_prims) leaves.push_back((int) i);
            else { depth[i + 1] = depth[i] + 1; depth[n.offset] = depth[i] + 1; }
        }
        int maxd = 0;
        for (size_t i = 0; i < nn; ++i) if (!b.nodes[i].num_prims && depth[i] > maxd) maxd = depth[i];
        s.level_begin.assign(nn ? maxd + 2 : 1, 0);
        for (size_t i = 0; i < nn; ++i) if (!b.nodes[i].num_prims) s.level_begin[depth[i] + 1]++;
        for (size_t l = 1; l < s.level_begin.size(); ++l) s.level_begin[l] += s.level_begin[l - 1];
        std::vector<int> lv(s.level_begin.empty() ? 0 : s.level_begin.back()), cursor(s.level_begin.begin(), s.level_begin.end());
        for (size_t i = 0; i < nn; ++i) if (!b.nodes[i].num_prims) lv[cursor[depth[i]]++] = (int) i;
        s.nbox.upload(nb); s.leaf_nodes.upload(leaves); s.level_nodes.upload(lv);
        s.num_leaves = (int) leaves.size();
    }
    lap("refit sources + upload");

    // ---- threaded node records: one near-first depth-first ordering per direction-sign octant ----
    // (layout and rationale: traverseThreaded in jtx_scene_dev.hpp).  The 8 orderings are independent of each other and of
    // the wide-node build: one host thread each.
    std::vector<int> pos(8 * nn);                                   // pos[k * nn + g]: node g's place in octant k's order
    {
        std::vector<int> size(nn, 1);                               // subtree sizes (children follow their parent in b.nodes)
        for (size_t i = nn; i-- > 0;)
            if (b.nodes[i].num_prims == 0) size[i] = 1 + size[i + 1] + size[b.nodes[i].offset];
        // position of every node in octant k's near-first depth-first order, top-down in index order (a parent stands before
        // its children in b.nodes): the near child follows its parent, the far child follows the near subtree
        // (dirIsNeg[axis] ? (second, first) : (first, second), scene.cpp:40-46); the records themselves are written on the
        // device (k_build_threaded) from s.nbox, these positions and the sizes
        auto positions = [&](int k) {
            int *p = pos.data() + (size_t) k * nn;
            if (nn) p[0] = 0;
            for (size_t g = 0; g < nn; ++g) {
                const jtx_mi_bvh_node &n = b.nodes[g];
                if (n.num_prims != 0) continue;
                const int first = (int) g + 1, second = n.offset;
                const bool neg = (k >> n.axis) & 1;
                const int near = neg ? second : first, far = neg ? first : second;
                p[near] = p[g] + 1;
                p[far] = p[g] + 1 + size[near];
            }
        };
        {
            std::vector<std::thread> pool;
            struct JoinAll { std::vector<std::thread> &v; ~JoinAll() { for (auto &t : v) if (t.joinable()) t.join(); } } joinAll{pool};
            pool.reserve(8);
            int started = 1;
            for (int k = 1; k < 8; ++k) {
                try { pool.emplace_back(guarded([&, k] { positions(k); })); started = k + 1; }
                catch (const std::system_error &) { break; }
            }
            positions(0);
            for (int k = started; k < 8; ++k) positions(k);
            for (auto &t : pool) t.join();
        }
        if (err) std::rethrow_exception(err);
        lap("positions in the 8 orderings");
        DevBuf<int> dpos, dsize;
        dpos.upload(pos); dsize.upload(size);
        s.tnodes.alloc(2 * 8 * nn); s.rec_node.alloc(8 * nn);
        HIPCHK(jtx_launch_build_threaded(s.nbox.p, dpos.p, dsize.p, (int) nn, s.tnodes.p, s.rec_node.p, nullptr));
        HIPCHK(hipStreamSynchronize(nullptr));
        lap("8 threaded orderings");
        if (wideThread.joinable()) wideThread.join();
        if (wideInline) wideJob();
        lap("wait for the wide nodes");
        if (err) std::rethrow_exception(err);
    }
    s.dev.tnodes = s.tnodes.p;
    buildLeafTables(s, pos);

    // ---- wide (8-ary, quantised) nodes for the uncounted kernels of HBM-resident scenes (traverseWide) ----
    s.wide.release(); s.dev.wide = nullptr; s.dev.wide_depth = 0;
    s.num_wide = 0;
    if (wideOk) {
        s.wide.upload(wide);
        s.dev.wide = s.wide.p; s.dev.wide_depth = wideDepth;
        s.wide_map.upload(wideMap); s.num_wide = (int) (wideMap.size() / 16);
    }
    if (!s.wide_fail.p) s.wide_fail.alloc(1);

    std::vector<DMaterial> mats(d.num_materials);
    std::vector<char> usedAsAlbedo(d.num_textures, 0);
    for (int i = 0; i < d.num_materials; ++i) {
        const jtx_mi_material &m = d.materials[i];
        DMaterial &o = mats[i];
        std::memset(&o, 0, sizeof o);
        o.type = m.type;
        for (int k = 0; k < 3; ++k) { o.albedo[k] = m.albedo[k]; o.ior[k] = m.ior[k]; o.k[k] = m.k[k]; o.emission[k] = m.emission[k]; }
        o.alpha_x = m.alpha_x; o.alpha_y = m.alpha_y;
        o.albedo_tex = m.albedo_tex; o.mr_tex = m.mr_tex;
        if (m.albedo_tex >= 0) usedAsAlbedo[m.albedo_tex] = 1;
    }
    s.materials.upload(mats);

    std::vector<DLight> lights(d.num_lights);
    for (int i = 0; i < d.num_lights; ++i) {
        const jtx_mi_light &l = d.lights[i];
        DLight &o = lights[i];
        std::memset(&o, 0, sizeof o);
        o.type = l.type; o.scale = l.scale;
        for (int k = 0; k < 3; ++k) { o.position[k] = l.position[k]; o.intensity[k] = l.intensity[k]; }
        o.scene_radius = l.type == 1 ? b.scene_radius : l.scene_radius;         // scene.cpp:128-134
    }
    s.lights.upload(lights);

    std::vector<DTexture> tex(d.num_textures);
    std::vector<float> texels;
    for (int i = 0; i < d.num_textures; ++i) {
        const jtx_mi_texture &t = d.textures[i];
        const size_t count = (size_t) t.width * t.height * t.channels;
        tex[i].w = t.width; tex[i].h = t.height; tex[i].c = t.channels; tex[i].pad = 0;
        tex[i].raw_off = (long long) texels.size();
        texels.insert(texels.end(), t.texels, t.texels + count);
        tex[i].linear_off = tex[i].raw_off;
        if (usedAsAlbedo[i]) {
            tex[i].linear_off = (long long) texels.size();
            for (size_t k = 0; k < count; ++k) texels.push_back(srgbToLinear(t.texels[k]));
        }
    }
    s.textures.upload(tex); s.texels.upload(texels);

    DevScene &ds = s.dev;
    ds.tris = s.tris.p; ds.shade = s.shade.p;
    ds.materials = s.materials.p; ds.lights = s.lights.p; ds.textures = s.textures.p; ds.texels = s.texels.p;
    ds.num_nodes = (int) nn; ds.num_prims = (int) np; ds.num_lights = d.num_lights; ds.num_materials = d.num_materials;
    ds.lds_threaded = (nn > 0 && 8 * nn * 32 + np * 48 <= kLdsThreadedBudget) ? 1 : 0;
    ds.material_mask = 0;
    for (int i = 0; i < d.num_materials; ++i) ds.material_mask |= 1 << d.materials[i].type;
    for (int k = 0; k < 3; ++k) ds.sky[k] = d.sky_color[k];
    lap("upload the rest");
    s.device_bytes = ((size_t) 16 * nn + tris.size() + shade.size() + s.wide.n) * sizeof(float4) + mats.size() * sizeof(DMaterial) +
                     lights.size() * sizeof(DLight) + tex.size() * sizeof(DTexture) + texels.size() * sizeof(float);
}

// Camera::init (camera.cpp:7-31)
DCam deriveCamera(const jtx_mi_camera_desc &c) {
    auto sub = [](const float a[3], const float b[3], float o[3]) { for (int i = 0; i < 3; ++i) o[i] = a[i] - b[i]; };
    auto cross = [](const float a[3], const float b[3], float o[3]) {
        o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; };
    auto norm = [](float v[3]) { const float l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); for (int i = 0; i < 3; ++i) v[i] = v[i] / l; };
    const float PI = 3.14159265358979323846f;
    DCam k{};
    const float aspect = (float) c.width / (float) c.height;
    const float h = std::tan((c.yfov * PI / 180.0f) / 2);
    const float vh = 2 * h * c.focus_distance;
    const float vw = vh * aspect;
    float w[3], u[3], v[3];
    sub(c.center, c.target, w); norm(w);
    cross(c.up, w, u); norm(u);
    cross(w, u, v);
    float vu[3], vv[3];
    for (int i = 0; i < 3; ++i) { vu[i] = vw * u[i]; vv[i] = vh * v[i]; }
    for (int i = 0; i < 3; ++i) { k.du[i] = vu[i] / (float) c.width; k.dv[i] = vv[i] / (float) c.height; }
    for (int i = 0; i < 3; ++i) {
        const float ul = c.center[i] - (c.focus_distance * w[i]) - vu[i] / 2 - vv[i] / 2;
        k.vp00[i] = ul + 0.5f * (k.du[i] + k.dv[i]);
    }
    const float dr = c.focus_distance * std::tan((c.defocus_angle / 2) * PI / 180.0f);
    for (int i = 0; i < 3; ++i) { k.defocus_u[i] = dr * u[i]; k.defocus_v[i] = dr * v[i]; k.center[i] = c.center[i]; }
    k.defocus_angle = c.defocus_angle;
    k.xs = c.x_pixel_samples; k.ys = c.y_pixel_samples;
    return k;
}

void checkCamera(const jtx_mi_camera_desc &c) {
    if (c.width <= 0 || c.height <= 0) throw std::runtime_error("camera width/height must be > 0");
    if (c.x_pixel_samples <= 0 || c.y_pixel_samples <= 0) throw std::runtime_error("camera pixel samples must be > 0");
    if (c.max_depth < 0) throw std::runtime_error("camera max_depth must be >= 0");
}

std::pair<hipEvent_t, hipEvent_t> takeEvents(jtx_mi_scene &s) {
    if (!s.free_events.empty()) { auto e = s.free_events.back(); s.free_events.pop_back(); return e; }
    std::pair<hipEvent_t, hipEvent_t> e;
    HIPCHK(hipEventCreate(&e.first)); HIPCHK(hipEventCreate(&e.second));
    return e;
}

// opts.integrator == 0: the measured policy of DESIGN.md "Integrators" (env JTX_INTEGRATOR overrides)
int autoIntegrator(const jtx_mi_scene &s) {
    const char *e = getenv("JTX_INTEGRATOR");
    const int v = e ? atoi(e) : 0;
    if (v >= 1 && v <= 2) return v;
    return (s.dev.lds_threaded || s.dev.material_mask == MAT_DIFFUSE_ONLY || s.dev.wide) ? 1 : 2;
}

// ---- wavefront integrator orchestration ----
// strata per batch: the batch's slot arrays (~132 B/slot) should stay near the 256 MiB Infinity Cache
int wfStrataPerBatch(int pixels, int nstrata) {
    const char *e = getenv("JTX_WF_BATCH");
    int b = e ? atoi(e) : 0;
    if (b <= 0) { b = (int) (((size_t) 16 << 20) / (size_t) (pixels > 0 ? pixels : 1)); if (b < 1) b = 1; }
    if (b > nstrata) b = nstrata;
    return b;
}

void wfEnsureBuffers(jtx_mi_scene &s, size_t slots, size_t nheads) {
    if (s.wf_slots < slots) {
        s.wf_floats.alloc(25 * slots); s.wf_hit.alloc(slots); s.wf_ints.alloc(4 * slots);
        s.wf_slots = slots;
    }
    if (s.wf_nheads < nheads) { s.wf_heads.alloc(nheads); s.wf_nheads = nheads; }
}

struct KindTimer {          // optional per-kernel HIP events (opts.reserved & 1)
    jtx_mi_scene &s; bool on; hipStream_t st;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> evs;
    void begin(int kind) { if (!on) return; std::pair<hipEvent_t, hipEvent_t> e; HIPCHK(hipEventCreate(&e.first)); HIPCHK(hipEventCreate(&e.second));
                           HIPCHK(hipEventRecord(e.first, st)); evs.push_back({kind, e}); }
    void end() { if (!on) return; HIPCHK(hipEventRecord(evs.back().second.second, st)); }
    void collect() {
        if (!on) return;
        for (auto &k : evs) {
            HIPCHK(hipEventSynchronize(k.second.second));
            float ms = 0; HIPCHK(hipEventElapsedTime(&ms, k.second.first, k.second.second));
            s.ms_by_kind[k.first] += ms; s.n_by_kind[k.first]++;
            (void) hipEventDestroy(k.second.first); (void) hipEventDestroy(k.second.second);
        }
        evs.clear();
    }
};

void launchWavefront(jtx_mi_scene &s, const jtx_mi_camera_desc &cam, const jtx_mi_render_opts &o, int sb, int se,
                     float *d_acc, unsigned char *d_img, hipStream_t stream, int rank, int world) {
    WfParams p{};
    p.scene = s.dev;
    p.cam = deriveCamera(cam);
    p.width = cam.width; p.height = cam.height; p.max_depth = cam.max_depth;
    p.tile_rank = rank; p.tile_world = world;
    p.tiles_x = (cam.width + 31) / 32;
    const int tiles = p.tiles_x * ((cam.height + 31) / 32);
    const int owned = tiles > rank ? (tiles - rank + world - 1) / world : 0;
    if (owned == 0) return;
    p.pixels = owned * 1024;
    const int nstrata = se - sb;
    const int B = wfStrataPerBatch(p.pixels, nstrata);
    p.num_slots = p.pixels * B;
    const int batches = (nstrata + B - 1) / B;
    const int D = cam.max_depth;
    wfEnsureBuffers(s, (size_t) p.num_slots, 0);
    const size_t N = s.wf_slots;
    float *f = s.wf_floats.p;
    WfBuffers &b = p.b;
    b.rox = f + 0 * N; b.roy = f + 1 * N; b.roz = f + 2 * N; b.rdx = f + 3 * N; b.rdy = f + 4 * N; b.rdz = f + 5 * N;
    b.betax = f + 6 * N; b.betay = f + 7 * N; b.betaz = f + 8 * N; b.radx = f + 9 * N; b.rady = f + 10 * N; b.radz = f + 11 * N;
    b.sox = f + 12 * N; b.soy = f + 13 * N; b.soz = f + 14 * N; b.sdx = f + 15 * N; b.sdy = f + 16 * N; b.sdz = f + 17 * N;
    b.stmax = f + 18 * N; b.pendx = f + 19 * N; b.pendy = f + 20 * N; b.pendz = f + 21 * N;
    b.hit = s.wf_hit.p;
    b.flags = s.wf_ints.p; b.depth = s.wf_ints.p + N; b.rng = (unsigned *) (s.wf_ints.p + 2 * N); b.sflags = s.wf_ints.p + 3 * N;
    p.acc = d_acc; p.img = d_img;
    // Material-sorted shade launches (one specialised launch per Material::type): measured 10 % SLOWER than one
    // launch on the mixed scene (shade is bound by its slot traffic, not by BxDF divergence) -- opt-in only.
    { const char *e = getenv("JTX_WF_SORT_SHADE"); p.sort_shade = (e && atoi(e) != 0 && __builtin_popcount(s.dev.material_mask) > 1) ? 1 : 0; }
    const bool count = o.count_rays != 0;
    if (count) {
        if (!s.counters.p) s.counters.alloc(64);
        HIPCHK(hipMemsetAsync(s.counters.p, 0, 64 * sizeof(unsigned long long), stream));
    }
    p.counters = count ? s.counters.p : nullptr;
    const int grid = s.num_cus * 8;
    KindTimer kt{s, (o.reserved & 1) != 0, stream, {}};
    const bool hasLights = s.dev.num_lights > 0;
    for (int bi = 0; bi < batches; ++bi) {
        const int s0 = sb + bi * B;
        const int ns = (s0 + B <= se) ? B : se - s0;
        kt.begin(0); HIPCHK(jtx_wf_generate(p, s0, ns, stream)); kt.end();
        for (int r = 0; r <= D; ++r) {
            kt.begin(1); HIPCHK(jtx_wf_trace(p, 0, grid, count, stream)); kt.end();
            // material-sorted shading: one launch per Material::type present (the closest-hit stage tagged each
            // slot with the type it hit), each running the kernel specialised for that BxDF; misses ride with the first
            if (!p.sort_shade) {
                kt.begin(2); HIPCHK(jtx_wf_shade(p, grid, count, -1, s.dev.material_mask ? s.dev.material_mask : 15, stream)); kt.end();
            } else {
                bool first = true;
                for (int ty = 0; ty < 4; ++ty) {
                    if (!(s.dev.material_mask & (1 << ty))) continue;
                    kt.begin(2); HIPCHK(jtx_wf_shade(p, grid, count, (ty + 1) | (first ? 16 : 0), 1 << ty, stream)); kt.end();
                    first = false;
                }
            }
            if (hasLights && r < D) { kt.begin(3); HIPCHK(jtx_wf_trace(p, 1, grid, count, stream)); kt.end(); }
        }
        kt.begin(4); HIPCHK(jtx_wf_resolve(p, s0, ns, (s0 + ns == se) ? 1 : 0, stream)); kt.end();
    }
    kt.collect();
}

// Does this render go through the persistent path kernel (working memory per frame slot)?  Everything else -- counting launches, the alternate
// Li, integrator 2, JTX_DYNAMIC_PATHS=0 -- uses per-scene singletons.
bool usesPathKernel(const jtx_mi_scene &s, const jtx_mi_render_opts &o) {
    static const int dynamicPaths = [] { const char *e = getenv("JTX_DYNAMIC_PATHS"); return e ? atoi(e) : 1; }();
    const int integ = o.integrator != 0 ? o.integrator : autoIntegrator(s);
    const bool alt = o.path_integrator != 0 || (s.dev.material_mask & 16) != 0;
    return !alt && integ == 1 && dynamicPaths && o.count_rays == 0;
}

// One launch of the integrator over [sb, se) on `stream`, bracketed by HIP events on that stream.
// prog != nullptr: a PROGRESSIVE launch -- all passes of [sb, se) in one k_render_paths<.., PROG> launch on `stream`, k_resolve_progressive beside it
// on the scene's resolver stream (the caller has checked that the persistent path kernel takes this render and that the range's records fit)
struct ProgLaunch { int tick = 1; int resolver_wgs = 0; int spg = 1, groups = 0; unsigned epoch = 0; int extra_leave_waves = 0; };     // in: tick, resolver_wgs, spg, extra_leave_waves; out: groups, epoch
void launchRender(jtx_mi_scene &s, const jtx_mi_camera_desc &cam, const jtx_mi_render_opts &o, int sb, int se,
                  float *d_acc, unsigned char *d_img, hipStream_t stream, ProgLaunch *prog = nullptr) {
    const int slot = o.frame_slot;
    if (slot < 0 || slot >= JTX_MI_FRAME_SLOTS) throw std::runtime_error("frame_slot: 0 .. " + std::to_string(JTX_MI_FRAME_SLOTS - 1));
    if (o.integrator < 0 || o.integrator > 2) throw std::runtime_error("integrator: 0 (auto), 1 (pixel-persistent) or 2 (HBM wavefront)");
    if (o.path_integrator < 0 || o.path_integrator > 2) throw std::runtime_error("path_integrator: 0 integrateMIS, 1 integrate, 2 integrateBasic");
    if (o.path_integrator == 1 && s.dev.num_lights == 0)
        throw std::runtime_error("integrate (integrator.cpp:85) indexes scene.lights without a guard: the scene needs at least one light");
    const int integ = o.integrator != 0 ? o.integrator : autoIntegrator(s);
    const bool count = o.count_rays != 0;
    const bool alt = o.path_integrator != 0 || (s.dev.material_mask & 16) != 0;
    // uncounted launches: dynamic path assignment (k_render_paths: a wave hands the paths of its pixel block x strata
    // range to whichever lane is free); JTX_DYNAMIC_PATHS=0 and the counting launches: one lane per pixel
    static const int dynamicPaths = [] { const char *e = getenv("JTX_DYNAMIC_PATHS"); return e ? atoi(e) : 1; }();
    for (auto &e : s.slot_done) if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // only the persistent path kernel keeps its working memory per slot; everything else uses per-scene singletons
    const bool slotted = !alt && integ == 1 && dynamicPaths && !count;
    struct SlotFence {               // wait before, record behind -- also when the launch throws half way (what was enqueued still runs)
        jtx_mi_scene &s; hipStream_t st; int slot; bool both;
        SlotFence(jtx_mi_scene &s_, hipStream_t st_, int slot_, bool both_) : s(s_), st(st_), slot(slot_), both(both_) {
            for (int k = 0; k < JTX_MI_FRAME_SLOTS; ++k) if (both || k == slot) HIPCHK(hipStreamWaitEvent(st, s.slot_done[k], 0));
        }
        ~SlotFence() { for (int k = 0; k < JTX_MI_FRAME_SLOTS; ++k) if (both || k == slot) (void) hipEventRecord(s.slot_done[k], st); }
    } fence(s, stream, slot, !slotted);
    RenderParams p{};
    jtx_mi_scene::PassRec &rec = s.pass[slot];
    s.last_slot = slot;
    rec.last_work = nullptr;          // only a launch that owns a chunk counter arms passAbandoned() (a stale one would void later passes)
    rec.parts.clear(); rec.resolved_end = se;
    p.prev_work = nullptr; p.abandoned = nullptr;
    p.scene = s.dev;
    p.cam = deriveCamera(cam);
    p.width = cam.width; p.height = cam.height; p.max_depth = cam.max_depth;
    p.sample_begin = sb; p.sample_end = se;
    const int world = o.tile_world > 1 ? o.tile_world : 1;
    const int rank = o.tile_world > 1 ? o.tile_rank : 0;
    if (rank < 0 || rank >= world) throw std::runtime_error("tile_rank must be in [0, tile_world)");
    p.tile_rank = rank; p.tile_world = world;
    p.tiles_x = (cam.width + 31) / 32;
    const int tiles = p.tiles_x * ((cam.height + 31) / 32);
    const int owned = tiles > rank ? (tiles - rank + world - 1) / world : 0;
    p.acc = d_acc; p.img = d_img;
    p.stop = s.stop_dev;
    if (count) {
        if (!s.counters.p) s.counters.alloc(64);
        HIPCHK(hipMemsetAsync(s.counters.p, 0, 64 * sizeof(unsigned long long), stream));
    }
#ifdef JTX_PROFILE_TIMELINE     /* diagnostic build (jtx_profile.hpp): room for a (start, end) pair per wave */
    if (s.counters.n < 64 + 2 * 65536) { s.counters.alloc(64 + 2 * 65536); HIPCHK(hipMemsetAsync(s.counters.p, 0, (64 + 2 * 65536) * sizeof(unsigned long long), stream)); }
#endif
    p.counters = s.counters.p;
    // timing pairs are only drained by jtx_mi_kernel_time (bench / tools): a UI that never asks keeps the newest 64
    while (s.pending.size() >= 64 && hipEventQuery(s.pending.front().second) == hipSuccess) {
        s.free_events.push_back(s.pending.front()); s.pending.erase(s.pending.begin());
    }
    auto ev = takeEvents(s);
    struct EvReturn { jtx_mi_scene &s; std::pair<hipEvent_t, hipEvent_t> ev; bool armed = true;
                      ~EvReturn() { if (armed) s.free_events.push_back(ev); } } evGuard{s, ev};   // a throw below must not leak the pair
    bool evClosed = false;
    HIPCHK(hipEventRecord(ev.first, stream));
    if (owned == 0) {                 // a shard without tiles (more ranks than 32x32 tiles): nothing to launch, an empty timing pair
        HIPCHK(hipEventRecord(ev.second, stream));
        evGuard.armed = false;
        s.pending.push_back(ev);
        return;
    }
    if (alt) {
        HIPCHK(jtx_launch_render_alt(p, owned, count, o.path_integrator, stream));
    } else if (integ == 1) {
        // strata groups: the strata of a pixel block are spread over `groups` waves (gridDim.y); every path's clamped
        // radiance goes to rad[stratum][pixel] and k_resolve_samples adds them to the film in sample order.
        // One-lane-per-pixel launches (counting, JTX_DYNAMIC_PATHS=0) split only small shards / frames: 32 or 64 ways.
        const long waves = (long) owned * 16;
        int groups = 1;
        { const char *e = getenv("JTX_STRATA_GROUPS"); if (e) groups = atoi(e); else if (waves < (long) s.num_cus * 24) groups = 64; else if (waves < (long) s.num_cus * 96) groups = 32; }
        if (groups < 1) groups = 1;
        if (groups > se - sb) groups = se - sb;
        if (slotted) {
            // chunk = (8x8 pixel block, strata group) = 64 x strata paths, fetched by persistent waves: ~250 k chunks per
            // launch keep the end of the launch short at every shard size (C2: 1 group 44.2 ms, 8 groups 38.1 ms;
            // 1/8 shard: 64 groups 5.5 ms): the smallest power of two that gives that many, at most one group per stratum.
            // With ANOTHER FRAME OF THIS SCENE IN FLIGHT (a launch in another frame slot that has not finished) the end of this launch
            // is filled by the next one, and the kernels that stage the scene in LDS do better with FEWER, LARGER chunks (fewer
            // fetches: a fetch stalls the wave for the counter's round trip): ~6 chunks per persistent wave, at least 512 paths each
            // (profiles/r05_frames_in_flight.md: C2 26.5 against 27.8 ms, a 1/8 shard 3.43 against 4.06 = 1/8 of the frame's kernel time)
            if (!getenv("JTX_STRATA_GROUPS")) {
                bool pipelined = false;
                if (!o.sequence_end) for (int k = 0; k < JTX_MI_FRAME_SLOTS; ++k) if (k != slot && hipEventQuery(s.slot_done[k]) == hipErrorNotReady) pipelined = true;
                (void) hipGetLastError();
                groups = 1;
                if (pipelined && s.dev.lds_threaded) {
                    int bs = 0; const long pw = (long) jtx_render_paths_grid(s.dev, s.num_cus, &bs) * (bs / 64);
                    while ((long) groups * waves < 6 * pw && groups * 2 <= se - sb && (se - sb) / (groups * 2) >= 8) groups *= 2;
                } else
                    while ((long) groups * waves < 250000 && groups * 2 <= se - sb) groups *= 2;
            }
            p.rad_stride = owned * 1024;
            // the per-path radiance buffer holds (strata of a pass) x (owned pixels) x 16 B: a frame that would need more
            // than kMaxRadBytes goes in several passes of consecutive strata (the resolve continues the sums in order)
            size_t maxRad = kMaxRadBytes;
            { const char *e = getenv("JTX_MAX_RAD_MB"); if (e && atol(e) > 0) maxRad = (size_t) atol(e) << 20; }
            if (o.max_record_mb > 0) maxRad = (size_t) o.max_record_mb << 20;
            const long perPass = (long) (maxRad / ((size_t) p.rad_stride * sizeof(float4)));
            const int chunk = perPass >= se - sb ? se - sb : (perPass > 1 ? (int) perPass : 1);
            const size_t need = (size_t) p.rad_stride * (size_t) chunk;
            DevBuf<float4> &rad = s.rad[slot];
            if (rad.n < need || rad.n / 4 > need) {                         // grow, and give memory back when the frame shrank a lot
                if (rad.p) HIPCHK(hipStreamSynchronize(stream));             // (a pass in flight may still write the old buffer: this stream waits for slot_done)
                rad.alloc(need);
            }
            p.rad = rad.p;
            // one chunk counter per launch: a pass split into several launches (radiance buffer cap) and the pass pipelined behind it
            // must never share one (passAbandoned() reads them per part) -- a ring of kWorkRing, at most half of it per pass
            const int nparts = (se - sb + chunk - 1) / chunk;
            // (every frame slot may have a pass in flight, and a resolve still reads the counter of the pass before its own: a pass
            //  takes at most 1 / (slots + 1) of the ring, so that no counter comes round while a launch in flight can read it)
            constexpr int kPartsMax = kWorkRing / (JTX_MI_FRAME_SLOTS + 1);
            if (nparts > kPartsMax)
                throw std::runtime_error("the radiance buffer cap splits this pass into " + std::to_string(nparts) + " launches (more than " +
                                         std::to_string(kPartsMax) + "): raise max_record_mb / JTX_MAX_RAD_MB or render fewer strata per pass");
            if (prog) {
                // ---- one launch for all passes; the resolver beside it ----
                if (nparts != 1) throw std::runtime_error("progressive launch: the range's radiance records exceed the cap (the caller splits the range)");
                RenderParams q = p;
                q.strata_per_group = prog->spg;
                q.num_groups = (se - sb + prog->spg - 1) / prog->spg;
                q.num_subblocks = owned * 16;
                prog->groups = q.num_groups;
                if (q.num_groups > 32767) throw std::runtime_error("progressive launch: more than 32767 strata groups (raise samples_per_tick)");
                q.prog_groups_per_pass = prog->tick > prog->spg ? prog->tick / prog->spg : 1;
                const int leave = jtx_resolve_progressive_waves(prog->resolver_wgs) + prog->extra_leave_waves;
                const int nwaves = jtx_render_paths_waves(q, s.num_cus, 1, leave);
                // the chunk counter, the closed-at word, the resolver leader's word and the waves' words on cache lines of their own: the
                // counter takes every fetch of every wave, and a poll of a word on ITS line queues up with them
                constexpr int kClosedAt = jtx::PROG_CTL_CLOSED_AT, kLeader = jtx::PROG_CTL_LEADER, kSlots = jtx::PROG_CTL_SLOTS;
#ifdef JTX_DBG_PROG
                const size_t words = kSlots + 65536;                       // (+ the diagnostic build's per-wave fetch counts, hardware ids and the leader's snapshot)
#else
                const size_t words = kSlots + (size_t) nwaves;
#endif
                if (s.prog_ctl.n < words) { if (s.prog_ctl.p) HIPCHK(hipDeviceSynchronize()); s.prog_ctl.alloc(words + 1024); }
                // closed-at and every wave's word: "none" (a wave that has not started holds no path); chunk counter and the resolver leader's word: 0
                HIPCHK(hipMemsetAsync(s.prog_ctl.p, 0xff, words * sizeof(unsigned), stream));
                HIPCHK(hipMemsetAsync(s.prog_ctl.p, 0, sizeof(unsigned), stream));
                HIPCHK(hipMemsetAsync(s.prog_ctl.p + kLeader, 0, (kSlots - kLeader) * sizeof(unsigned), stream));
                q.work = s.prog_ctl.p; q.prog_closed_at = s.prog_ctl.p + kClosedAt; q.prog_leader = s.prog_ctl.p + kLeader; q.prog_slots = s.prog_ctl.p + kSlots;
                rec.last_work = nullptr;                                   // (progress comes from the resolver's words, not from a chunk counter)
                if (!s.resolve_stream) {
                    int least = 0, greatest = 0;
                    (void) hipDeviceGetStreamPriorityRange(&least, &greatest);
                    HIPCHK(hipStreamCreateWithPriority(&s.resolve_stream, hipStreamNonBlocking, greatest));
                }
                for (hipEvent_t *e : {&s.prog_ready, &s.prog_paths_done, &s.prog_resolved}) if (!*e) HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
                if (!s.prog_host) {
                    HIPCHK(hipHostMalloc((void **) &s.prog_host, (2 * kResolverMax + 16) * sizeof(unsigned), hipHostMallocMapped));
                    std::memset(s.prog_host, 0, (2 * kResolverMax + 16) * sizeof(unsigned));
                    void *d = nullptr; HIPCHK(hipHostGetDevicePointer(&d, s.prog_host, 0)); s.prog_host_dev = (unsigned *) d;
                }
                prog->epoch = (++s.prog_epoch & 0x7fffu) + 1u;
                // (the resolver's words of the launch before: none may read as this launch's when the 15-bit epoch comes round -- the launch before has ended)
                for (int w = 0; w < 2 * kResolverMax; ++w) __atomic_store_n(s.prog_host + w, 0u, __ATOMIC_RELAXED);
                s.prog_live_epoch = prog->epoch;                           // (jtx_prog_finished vouches for this launch's path kernel while it has not ended)
                __atomic_store_n(s.prog_host + 2 * kResolverMax, prog->epoch, __ATOMIC_RELEASE);
                const int nr = prog->resolver_wgs;
                // The PATH KERNEL FIRST, then the resolver on its own stream (not before the film and the control words stand).  The path grid
                // leaves the resolver's wave slots free, so the resolver starts beside it at once -- when the two streams run side by side.
                // HIP promises no such thing (streams may share a hardware queue whose packets run in order), and the order of the two launches
                // is what makes that harmless: the path kernel waits for nobody, so a resolver that only gets to run AFTER it adds all passes
                // then -- the right film, previews late.  (Round 6's first version launched the resolver first and waited for it to check in:
                // tools/soak.py met the mapping where the path kernel then queued BEHIND the resolver, which waited for it -- a minute, its
                // bounded wait, then an error.)  The resolver's stream is a high-priority one: a queue of its own where the runtime has one.
                HIPCHK(hipEventRecord(s.prog_ready, stream));
                bool launchPaths = true;
                unsigned patienceMs = 0;                                   // (0: the resolver's default, a minute)
                hipStream_t rstream = s.resolve_stream;
#ifdef JTX_TEST_HOOKS       /* libjtx_mi_testhooks.so only */
                if (getenv("JTX_TEST_PROGRESSIVE_ONE_STREAM")) rstream = stream;         // both kernels on ONE stream: the serialised case, on a box whose streams do run side by side
                if (const char *e = getenv("JTX_TEST_PROGRESSIVE_DELAY_PATH_MS")) {      // the path kernel comes late (as behind another process' long launch): a host function holds its stream
                    static std::atomic<int> delayMs{0}; delayMs = atoi(e);
                    HIPCHK(hipLaunchHostFunc(stream, [](void *) { std::this_thread::sleep_for(std::chrono::milliseconds(delayMs.load())); }, nullptr));
                }
                if (getenv("JTX_TEST_PROGRESSIVE_NO_PATH_KERNEL")) launchPaths = false;  // the path kernel never comes: the resolver's bounded wait ...
                if (const char *e = getenv("JTX_TEST_RESOLVER_PATIENCE_MS")) patienceMs = (unsigned) atoi(e);      // ... shortened from a minute
#endif
                if (launchPaths)
                    if (const hipError_t le = jtx_launch_render_paths(q, owned, s.num_cus, stream, 1, true, leave))
                        throw std::runtime_error(std::string("k_render_paths (progressive): ") + hipGetErrorString(le));
                if (rstream != stream) HIPCHK(hipStreamWaitEvent(rstream, s.prog_ready, 0));
                HIPCHK(jtx_launch_resolve_progressive(q, owned, nwaves, nr, s.prog_host_dev, s.prog_host_dev + kResolverMax, s.prog_host_dev + 2 * kResolverMax, prog->epoch, rstream, patienceMs));
                HIPCHK(hipEventRecord(s.prog_resolved, rstream));
                HIPCHK(hipEventRecord(ev.second, stream)); evClosed = true;
                HIPCHK(hipEventRecord(s.prog_paths_done, stream));
                HIPCHK(hipStreamWaitEvent(stream, s.prog_resolved, 0));    // the stream (and the slot's fence) stands for both kernels from here on
            } else
            for (int b0 = sb; b0 < se; b0 += chunk) {
                RenderParams q = p;
                q.sample_begin = b0; q.sample_end = b0 + chunk < se ? b0 + chunk : se;
                int g = groups; if (g > q.sample_end - q.sample_begin) g = q.sample_end - q.sample_begin;
                q.strata_per_group = (q.sample_end - q.sample_begin + g - 1) / g;
                q.num_groups = (q.sample_end - q.sample_begin + q.strata_per_group - 1) / q.strata_per_group;
                q.num_subblocks = owned * 16;
                if (!s.work.p) s.work.alloc(kWorkRing);
                const unsigned ring = s.work_slot++ & (kWorkRing - 1);
                q.work = s.work.p + ring;                                  // one counter per launch in flight
                HIPCHK(hipMemsetAsync(q.work, 0, sizeof(unsigned), stream));
                s.abandon_host[ring] = 0u;                                 // (this launch's resolve is the only writer, and it has not been enqueued yet)
                q.abandoned = s.abandon_dev + ring;
                q.prev_work = rec.parts.empty() ? nullptr : rec.parts.back().work;      // (a pass split by the record cap: its launches enter the film in order or not at all)
                rec.last_work = q.work;
                rec.parts.push_back({q.work, s.abandon_host + ring, q.sample_begin, q.sample_end});
                HIPCHK(jtx_launch_render_paths(q, owned, s.num_cus, stream));
                if (q.sample_end == se) { HIPCHK(hipEventRecord(ev.second, stream)); evClosed = true; }   // kernel_time: without the last resolve
                HIPCHK(jtx_launch_resolve_samples(q, owned, stream));
            }
        } else if (groups > 1) {
            p.strata_per_group = (se - sb + groups - 1) / groups;
            p.rad_stride = owned * 1024;
            const size_t need = (size_t) p.rad_stride * (size_t) (se - sb);
            if (s.rad[0].n < need) { if (s.rad[0].p) HIPCHK(hipStreamSynchronize(stream)); s.rad[0].alloc(need); }
            p.rad = s.rad[0].p;
            HIPCHK(jtx_launch_render_pixels(p, owned, count, stream));
            HIPCHK(jtx_launch_resolve_samples(p, owned, stream));
        } else {
            HIPCHK(jtx_launch_render_pixels(p, owned, count, stream));
        }
    }
    else launchWavefront(s, cam, o, sb, se, d_acc, d_img, stream, rank, world);
    if (!evClosed) HIPCHK(hipEventRecord(ev.second, stream));
    evGuard.armed = false;
    s.pending.push_back(ev);
}

} // namespace

// ---- one progressive launch (jtx_progressive.hpp): shared by jtx_mi_render and jtx_mi_multi_render ----
bool jtx_prog_usable(jtx_mi_scene *s, const jtx_mi_render_opts &o) {
    static const int progressiveOn = [] { const char *e = getenv("JTX_PROGRESSIVE_LAUNCH"); return e ? atoi(e) : 1; }();
    return progressiveOn && usesPathKernel(*s, o);
}

static int progOwnedTiles(const jtx_mi_camera_desc &cam, const jtx_mi_render_opts &o) {
    const int tiles = ((cam.width + 31) / 32) * ((cam.height + 31) / 32);
    const int world = o.tile_world > 1 ? o.tile_world : 1, rank = o.tile_world > 1 ? o.tile_rank : 0;
    return (rank >= 0 && tiles > rank) ? (tiles - rank + world - 1) / world : 0;
}

long jtx_prog_span(jtx_mi_scene *, const jtx_mi_camera_desc &cam, const jtx_mi_render_opts &o, int tick) {
    const int owned = progOwnedTiles(cam, o);
    size_t maxRad = kMaxRadBytes;
    { const char *e = getenv("JTX_MAX_RAD_MB"); if (e && atol(e) > 0) maxRad = (size_t) atol(e) << 20; }
    if (o.max_record_mb > 0) maxRad = (size_t) o.max_record_mb << 20;
    const size_t rowBytes = (size_t) (owned > 0 ? owned : 1) * 1024 * sizeof(float4);
    long perLaunch = (long) (maxRad / rowBytes) / tick * tick;                // whole passes
    if (perLaunch > 32767l * tick) perLaunch = 32767l * tick;                 // (the resolver's words count groups in 15 bits; a group is at least a pass, or a whole fraction of a LONG one)
    if (perLaunch < tick) throw std::runtime_error("one pass of " + std::to_string(tick) + " strata does not fit the radiance-record cap: raise max_record_mb or lower samples_per_tick");
    return perLaunch;
}

void jtx_prog_begin(jtx_mi_scene *s, const jtx_mi_camera_desc &cam, const jtx_mi_render_opts &o, int b0, int e0, int tick, float *d_acc, unsigned char *d_img,
                    hipStream_t stream, int extra_leave_waves, bool locked, JtxProgRun &run) {
    std::unique_lock<std::mutex> lk(s->mu, std::defer_lock);
    if (locked) lk.lock();
    DeviceGuard dg(s->device);
    const int owned = progOwnedTiles(cam, o);
    run = JtxProgRun{};
    run.begin = b0; run.end = e0; run.tick = tick;
    if (owned == 0) { run.nothing = true; return; }
    static const int resolverEnv = [] { const char *e = getenv("JTX_RESOLVER_WGS"); return e ? atoi(e) : 0; }();
    int nr = resolverEnv > 0 ? resolverEnv : 64;                        // one watches the path waves, the others add passes
    if (nr > kResolverMax) nr = kResolverMax;
    if (nr > owned * 4 + 1) nr = owned * 4 + 1;                         // (a worker per 256 pixel slots at most)
    if (nr < 2) nr = 2;
    ProgLaunch pl; pl.tick = tick; pl.resolver_wgs = nr; pl.extra_leave_waves = extra_leave_waves;
    // strata per group (= per chunk of an 8x8 block, and the step in which the film advances): a pass; a whole fraction of one
    // when passes are long and the launch would have few chunks; SEVERAL short passes when a pass alone makes chunks too small
    // for the persistent waves (C2, one stratum per chunk: kernel 25.2 ms against 22.7 at eight) -- the callback still runs
    // once per pass, the film and the preview then advance every few passes (at C2's 0.36 ms per pass, far above any display
    // rate: the reference's UI polls currentSample_ once per frame it draws, display.cpp:700-706)
    static const int minStrata = [] { const char *e = getenv("JTX_PROG_MIN_STRATA"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : v; }();
    pl.spg = tick;
    if ((e0 - b0) % tick == 0) while (pl.spg % 2 == 0 && pl.spg >= 16 && (long) owned * 16 * ((e0 - b0) / pl.spg) < 250000) pl.spg /= 2;
    while (pl.spg < minStrata && pl.spg * 2 <= e0 - b0) pl.spg += tick;
    launchRender(*s, cam, o, b0, e0, d_acc, d_img, stream, &pl);
    run.spg = pl.spg; run.groups = pl.groups; run.resolver_wgs = nr; run.epoch = pl.epoch;
}

int jtx_prog_completed(jtx_mi_scene *s, const JtxProgRun &run, bool *gave_up) {
    if (run.nothing) return run.end;
    unsigned g = 0x7fffu;
    for (int w = 0; w < run.resolver_wgs; ++w) {
        const unsigned v = __atomic_load_n(s->prog_host + kResolverMax + w, __ATOMIC_ACQUIRE);
        const bool mine = (v >> 16) == run.epoch;
        const unsigned gw = mine ? (v & 0x7fffu) : 0u;
        if (mine && (v & 0x8000u) && gave_up) *gave_up = true;
        g = gw < g ? gw : g;
    }
    const long d = (long) run.begin + (long) g * run.spg;               // (the resolver adds whole passes only)
    return d < run.end ? (int) d : run.end;
}

bool jtx_prog_finished(jtx_mi_scene *s) {
    if (!s->prog_resolved || !s->prog_paths_done) return true;
    DeviceGuard dg(s->device);
    const bool paths = hipEventQuery(s->prog_paths_done) == hipSuccess;
    // the host's word for a path kernel that has not ended (it may still be waiting its turn): the resolver does not give up on it
    if (!paths && s->prog_host) __atomic_store_n(s->prog_host + 2 * kResolverMax, s->prog_live_epoch, __ATOMIC_RELEASE);
    const bool f = paths && hipEventQuery(s->prog_resolved) == hipSuccess;
    (void) hipGetLastError();
    return f;
}

extern "C" {

const char *jtx_mi_last_error(void) { return g_err.c_str(); }
int jtx_mi_version(void) { return JTX_MI_VERSION; }

// JTX_ABORT_LOG=<file>: a std::terminate / SIGABRT inside the process leaves its reason and a native backtrace in that file.
// (A test runner that captures file descriptor 2 swallows the C++ runtime's last words; an intermittent abort inside
// jtx_mi_scene_create on the GPU box -- round 4 -- left nothing but "Fatal Python error: Aborted".  tests/conftest.py sets it.)
namespace {
int g_abortFd = -1;
void abortTrace(const char *why) {
    if (g_abortFd < 0) return;
    (void) !write(g_abortFd, why, strlen(why));
    void *frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, g_abortFd);
    (void) !write(g_abortFd, "\n", 1);
}
void onAbortSignal(int) { abortTrace("SIGABRT\n"); signal(SIGABRT, SIG_DFL); raise(SIGABRT); }
void onTerminate() {
    char msg[512] = "std::terminate without an active exception\n";
    if (std::exception_ptr e = std::current_exception()) {
        try { std::rethrow_exception(e); }
        catch (const std::exception &x) { snprintf(msg, sizeof msg, "std::terminate: %s\n", x.what()); }
        catch (...) { snprintf(msg, sizeof msg, "std::terminate: a non-std exception\n"); }
    }
    abortTrace(msg);
    signal(SIGABRT, SIG_DFL);
    abort();
}
struct AbortLogInstaller {
    AbortLogInstaller() {
        const char *f = getenv("JTX_ABORT_LOG");
        if (!f || !*f) return;
        g_abortFd = open(f, O_WRONLY | O_CREAT | O_APPEND, 0644);
        if (g_abortFd < 0) return;
        std::set_terminate(onTerminate);
        signal(SIGABRT, onAbortSignal);
    }
} g_abortLogInstaller;
}

int jtx_mi_device_count(int32_t *count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = n; return 0;
}
int jtx_mi_set_device(int32_t device) {
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail(std::string("hipSetDevice: ") + hipGetErrorString(e));
    return 0;
}

int jtx_mi_bvh_build(const jtx_mi_scene_desc *desc, jtx_mi_bvh_node *nodes_out, int32_t *num_nodes_out,
                     jtx_mi_tri_ref *refs_out, int32_t *max_depth_out) {
    try {
        if (!desc) throw std::runtime_error("null scene description");
        validate(*desc);
        jtxh::BvhResult r;
        jtxh::buildBVH(*desc, r);
        if (nodes_out) std::memcpy(nodes_out, r.nodes.data(), r.nodes.size() * sizeof(jtx_mi_bvh_node));
        if (refs_out) std::memcpy(refs_out, r.refs.data(), r.refs.size() * sizeof(jtx_mi_tri_ref));
        if (num_nodes_out) *num_nodes_out = (int32_t) r.nodes.size();
        if (max_depth_out) *max_depth_out = r.max_depth;
        return 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}

int jtx_mi_wide_build(const jtx_mi_bvh_node *nodes, int32_t num_nodes, uint32_t *granules_out, int64_t capacity,
                      int64_t *num_granules_out, int32_t *depth_out) {
    try {
        if (!nodes || num_nodes < 0) throw std::runtime_error("null nodes");
        std::vector<jtx_mi_bvh_node> v(nodes, nodes + num_nodes);
        std::vector<uint4> wide; int depth = 0;
        if (!buildWide(v, wide, depth)) { wide.clear(); depth = 0; }
        if (num_granules_out) *num_granules_out = (int64_t) wide.size();
        if (depth_out) *depth_out = depth;
        if (granules_out) {
            if ((int64_t) wide.size() > capacity) throw std::runtime_error("granules_out too small");
            std::memcpy(granules_out, wide.data(), wide.size() * sizeof(uint4));
        }
        return 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}

int jtx_mi_scene_create(const jtx_mi_scene_desc *desc, jtx_mi_scene **out) {
    jtx_mi_scene *s = nullptr;
    try {
        if (!desc || !out) throw std::runtime_error("null argument");
        *out = nullptr;
        const bool trace = getenv("JTX_TRACE_CREATE") != nullptr;
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!trace) return;
            const auto n = std::chrono::steady_clock::now();
            fprintf(stderr, "[jtx create] %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t0).count());
            t0 = n;
        };
        validate(*desc);
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
            throw std::runtime_error("no HIP device: the jtx_mi core has no CPU fallback");
        s = new jtx_mi_scene();
        HIPCHK(hipGetDevice(&s->device));
        lap("validate");
        jtxh::buildBVH(*desc, s->bvh);
        lap("buildBVH");
        flatten(*desc, *s);
        lap("flatten (all of the above)");
        HIPCHK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        HIPCHK(hipHostMalloc((void **) &s->stop_host, sizeof(unsigned), hipHostMallocMapped));
        *s->stop_host = 0u;
        { void *d = nullptr; HIPCHK(hipHostGetDevicePointer(&d, s->stop_host, 0)); s->stop_dev = (const unsigned *) d; }
        HIPCHK(hipHostMalloc((void **) &s->abandon_host, kWorkRing * sizeof(unsigned), hipHostMallocMapped));
        std::memset(s->abandon_host, 0, kWorkRing * sizeof(unsigned));
        { void *d = nullptr; HIPCHK(hipHostGetDevicePointer(&d, s->abandon_host, 0)); s->abandon_dev = (unsigned *) d; }
        { int cus = 0; HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device)); s->num_cus = cus > 0 ? cus : 256; }
        lap("stream, stop flag, CU count");
        *out = s;
        return 0;
    } catch (const std::exception &e) { delete s; return fail(e.what()); }
}

void jtx_mi_scene_destroy(jtx_mi_scene *scene) {
    if (!scene) return;
    int prev = -1;
    const bool sw = hipGetDevice(&prev) == hipSuccess && prev != scene->device && hipSetDevice(scene->device) == hipSuccess;
    if (scene->stream) (void) hipStreamSynchronize(scene->stream);
    delete scene;
    if (sw) (void) hipSetDevice(prev);
}

// Mesh::transform after Display's edit (recalculateTransform, display.cpp:547,568,586); takes effect at the next jtx_mi_scene_refit
int jtx_mi_scene_set_transform(jtx_mi_scene *s, int32_t mesh, const float *m16) {
    if (!s || !m16) return fail("null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    if (mesh < 0 || (size_t) mesh * 16 >= s->mesh_xf_host.size()) return fail("mesh index out of range");
    std::memcpy(&s->mesh_xf_host[16 * (size_t) mesh], m16, 16 * sizeof(float));
    s->xf_dirty = true;
    return 0;
}

namespace { void uploadTransforms(jtx_mi_scene &sc); }

// Scene::rebuildBVH's place in the edit loop (display.cpp:902-905), on the device and with the topology kept (jtx_refit.hip)
int jtx_mi_scene_refit(jtx_mi_scene *s) {
    try {
        if (!s) throw std::runtime_error("null scene");
        DeviceGuard dg(s->device);
        std::lock_guard<std::mutex> lk(s->mu);
        if (s->dev.num_nodes == 0) return 0;
        const bool trace = getenv("JTX_TRACE_CREATE") != nullptr;
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!trace) return;
            HIPCHK(hipStreamSynchronize(s->stream));
            const auto n = std::chrono::steady_clock::now();
            fprintf(stderr, "[jtx refit] %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t0).count());
            t0 = n;
        };
        uploadTransforms(*s);
        lap("transforms to the device");
        HIPCHK(hipMemsetAsync(s->wide_fail.p, 0, sizeof(int), s->stream));
        RefitArgs a{};
        a.prim_src = s->prim_src.p; a.mesh_xf = s->mesh_xf.p; a.tris = s->tris.p; a.shade = s->shade.p; a.pbox = s->pbox.p;
        a.nbox = s->nbox.p; a.leaf_nodes = s->leaf_nodes.p; a.level_nodes = s->level_nodes.p;
        a.tnodes = s->tnodes.p; a.rec_node = s->rec_node.p;
        a.wide = s->wide.p; a.wide_map = s->wide_map.p; a.wide_fail = s->wide_fail.p;
        a.num_prims = s->dev.num_prims; a.num_nodes = s->dev.num_nodes; a.num_leaves = s->num_leaves; a.num_wide = s->wide.p ? s->num_wide : 0;
        HIPCHK(jtx_launch_refit(a, s->level_begin.data(), (int) s->level_begin.size() - 1, s->stream));
        lap("kernels");
        // the host's copy of the nodes, the scene radius (scene.hpp:81-84) and with it the DISTANT lights (scene.cpp:128-134)
        const size_t nbCount = 2 * (size_t) s->dev.num_nodes;
        float4 *nb = (float4 *) s->pin_nb.ensure((nbCount + 1) * sizeof(float4));       // page-locked, the scene's own: the last granule takes the wide-refit flag
        int *wfailp = (int *) (nb + nbCount);
        HIPCHK(hipMemcpyAsync(nb, s->nbox.p, nbCount * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipMemcpyAsync(wfailp, s->wide_fail.p, sizeof(int), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        const int wfail = *wfailp;
        lap("node boxes to the host");
        for (size_t i = 0; i < s->bvh.nodes.size(); ++i) {
            jtx_mi_bvh_node &n = s->bvh.nodes[i];
            n.pmin[0] = nb[2 * i].x; n.pmax[0] = nb[2 * i].y; n.pmin[1] = nb[2 * i].z; n.pmax[1] = nb[2 * i].w;
            n.pmin[2] = nb[2 * i + 1].x; n.pmax[2] = nb[2 * i + 1].y;
        }
        {
            const jtx_mi_bvh_node &r = s->bvh.nodes[0];
            const float dx = r.pmax[0] - r.pmin[0], dy = r.pmax[1] - r.pmin[1], dz = r.pmax[2] - r.pmin[2];
            s->bvh.scene_radius = std::sqrt(dx * dx + dy * dy + dz * dz) / 2;
            if (s->lights.n) {
                std::vector<DLight> ls(s->lights.n);
                stagedD2H(ls.data(), s->lights.p, ls.size() * sizeof(DLight));
                bool any = false;
                for (auto &l : ls) if (l.type == 1) { l.scene_radius = s->bvh.scene_radius; any = true; }
                if (any) stagedH2D(s->lights.p, ls.data(), ls.size() * sizeof(DLight));
            }
        }
        if (s->dev.lw_leaves && !leafPlanesInRange(s->bvh.nodes)) { s->dev.lw_leaves = 0; s->dev.lw_box = nullptr; s->dev.lw_tab = nullptr; }   // refitted out of the list's range
        if (s->dev.lw_leaves) {                                                 // tiny scenes: the flat leaf list follows the refitted nodes
            std::vector<float4> lb(s->lw_box.n, make_float4(0.f, 0.f, 0.f, 0.f)); int l = 0;
            for (const jtx_mi_bvh_node &n : s->bvh.nodes) if (n.num_prims) {
                const int z = n.offset, w = n.num_prims; float fz, fw; std::memcpy(&fz, &z, 4); std::memcpy(&fw, &w, 4);
                lb[2 * (size_t) l] = make_float4(n.pmin[0], n.pmax[0], n.pmin[1], n.pmax[1]);
                lb[2 * (size_t) l + 1] = make_float4(n.pmin[2], n.pmax[2], fz, fw);
                ++l;
            }
            stagedH2D(s->lw_box.p, lb.data(), lb.size() * sizeof(float4));
        }
        if (wfail) { s->dev.wide = nullptr; s->dev.wide_depth = 0; }          // a node lost its grid (coordinates out of range): binary records only
        s->refitted = 1; s->xf_dirty = false;
        lap("host nodes, lights, leaf list");
        return 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}

// Scene::rebuildBVH (scene.hpp:66-69) in the edit loop (display.cpp:545-588, 902-905), ON THE DEVICE: a new TOPOLOGY for the
// edited geometry -- the reference's binned-SAH tree, node for node (jtx_build_dev.hip) -- and every structure derived from it.
namespace {
// the spare set of jtx_mi_scene_rebuild and the builder's scratch, sized for the scene's primitives (grow-only: a second call allocates nothing)
void reserveRebuild(jtx_mi_scene &sc, bool &wantWide, bool &wideOnDevice, size_t &wideCap) {
    jtx_mi_scene::RebuildSpare &sp = sc.spare;
    const int np = sc.dev.num_prims;
    const size_t maxN = 2 * (size_t) np;
    sp.src.ensure(5 * (size_t) np); sp.tris.ensure(3 * (size_t) np); sp.shade.ensure(4 * (size_t) np); sp.orig.ensure(np); sp.order.ensure(np);
    sp.nbox.ensure(2 * maxN); sp.hn.ensure(maxN); sp.leaves.ensure(maxN); sp.levels.ensure(maxN); sp.pos.ensure(8 * maxN); sp.size.ensure(maxN);
    const char *off = getenv("JTX_NO_WIDE");
    wantWide = !(off && atoi(off));
    // the device builder knows the area-optimal treelet cut only: under JTX_WIDE_SAH_CUT=0 (the greedy cut of round 1, a diagnostic)
    // the 8-ary nodes come from the host's buildWide, as at scene creation
    static const int sahCut = [] { const char *e = getenv("JTX_WIDE_SAH_CUT"); return e ? atoi(e) : 1; }();
    wideOnDevice = wantWide && sahCut;
    wideCap = jtxq::kNodeG * (size_t) np + 2 * (size_t) np + jtxq::kFirstBlock;     // every interior node its own wide node at worst
    if (wideOnDevice) { sp.wide.ensure(wideCap); sp.map.ensure(16 * (size_t) np); }
}
}

static int rebuildImpl(jtx_mi_scene *s, int32_t max_prims_in_node, bool commit);

// A DRY RUN of the rebuild: everything jtx_mi_scene_rebuild does except its commit section.  hipMalloc alone would not do: device
// memory is mapped at first touch and a translation unit's code object is loaded at its first launch -- measured on the atrium,
// allocation 0.9 ms, but the first rebuild still 26 ms against 7 for the second.  After the dry run the first rebuild is a second one.
int jtx_mi_scene_reserve_rebuild(jtx_mi_scene *s) { return rebuildImpl(s, 1, false); }
int jtx_mi_scene_release_rebuild(jtx_mi_scene *s) {
    try {
        if (!s) throw std::runtime_error("null scene");
        std::lock_guard<std::mutex> lk(s->mu);
        DeviceGuard dg(s->device);
        HIPCHK(hipStreamSynchronize(s->stream));                 // (a rebuild runs on the scene's stream and is complete on return; renders do not touch the spare set)
        s->spare.release();
        return 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}

namespace {
// Mesh::transform of every mesh to the device (enqueued on the scene's stream), from page-locked memory
void uploadTransforms(jtx_mi_scene &sc) {
    const size_t n = sc.mesh_xf_host.size();
    if (!n) return;
    if (!sc.mesh_xf_pinned) HIPCHK(hipHostMalloc((void **) &sc.mesh_xf_pinned, n * sizeof(float), hipHostMallocDefault));
    std::memcpy(sc.mesh_xf_pinned, sc.mesh_xf_host.data(), n * sizeof(float));
    HIPCHK(hipMemcpyAsync(sc.mesh_xf.p, sc.mesh_xf_pinned, n * sizeof(float), hipMemcpyHostToDevice, sc.stream));
}
}

int jtx_mi_scene_rebuild(jtx_mi_scene *s, int32_t max_prims_in_node) { return rebuildImpl(s, max_prims_in_node, true); }

static int rebuildImpl(jtx_mi_scene *s, int32_t max_prims_in_node, bool commit) {
    try {
        if (!s) throw std::runtime_error("null scene");
        DeviceGuard dg(s->device);
        std::lock_guard<std::mutex> lk(s->mu);
        const int np = s->dev.num_prims;
        if (np == 0) return 0;
        const bool trace = getenv("JTX_TRACE_CREATE") != nullptr;
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!trace) return;
            HIPCHK(hipStreamSynchronize(s->stream));
            const auto n = std::chrono::steady_clock::now();
            fprintf(stderr, "[jtx rebuild] %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t0).count());
            t0 = n;
        };
        // FAILURE-ATOMIC: everything is built into the spare set (jtx_mi_scene::spare) and into locals; the scene is touched
        // only in the commit section at the end, which cannot throw.  Whatever fails before leaves the scene as it was (copies that
        // were enqueued into locals are drained first: ~Drain).
        struct Drain { hipStream_t st; ~Drain() { (void) hipStreamSynchronize(st); } } drain{s->stream};
        jtx_mi_scene::RebuildSpare &sp = s->spare;
        HIPCHK(hipStreamSynchronize(s->stream));                  // frames in flight read the live set; the spare set may still be read by them too
        lap("wait for frames in flight");
        uploadTransforms(*s);
        lap("transforms to the device");
        bool wantWide, wideOnDevice; size_t wideCap;
        reserveRebuild(*s, wantWide, wideOnDevice, wideCap);
        lap("buffers (allocated by the first rebuild only)");
        DevBuildBuffers B{};
        B.prim_src = s->prim_src.p; B.tris = s->tris.p; B.shade = s->shade.p; B.orig = s->orig_id.p; B.mesh_xf = s->mesh_xf.p; B.np = np;
        B.max_prims = max_prims_in_node > 0 ? max_prims_in_node : 1;
        B.prim_src_out = sp.src.p; B.tris_out = sp.tris.p; B.shade_out = sp.shade.p; B.orig_out = sp.orig.p;
        B.nbox = sp.nbox.p; B.hnodes = sp.hn.p; B.order = sp.order.p; B.leaf_nodes = sp.leaves.p; B.level_nodes = sp.levels.p; B.pos = sp.pos.p; B.size = sp.size.p;
        B.wide = wideOnDevice ? sp.wide.p : nullptr; B.wide_map = sp.map.p; B.wide_cap = wideCap;
        DevBuildResult R;
        HIPCHK(jtx_device_build(B, sp.arena, R, s->stream));
        if (R.declined)
            throw std::runtime_error(std::string("the device builder declined: ") + R.declined + "; the scene is unchanged -- rebuild it on the host (Scene::buildBVH)");
        lap("device build");
        const int nn = R.nn;
        // ---- host copies (locals): nodes, Scene::triangles_ in leaf order ----
        std::vector<jtx_mi_bvh_node> nodes2((size_t) nn);
        jtx_mi_bvh_node *pinNodes = (jtx_mi_bvh_node *) sp.pin_nodes.ensure(2 * (size_t) np * sizeof(jtx_mi_bvh_node));   // page-locked landing buffers of the
        const int *ord = (const int *) sp.pin_ord.ensure((size_t) np * sizeof(int));                                     // spare set (sized once: <= 2 np nodes)
        HIPCHK(hipMemcpyAsync(pinNodes, sp.hn.p, (size_t) nn * sizeof(jtx_mi_bvh_node), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipMemcpyAsync((void *) ord, sp.order.p, (size_t) np * sizeof(int), hipMemcpyDeviceToHost, s->stream));
        // ---- triangle / shading records recomputed from the re-ordered sources; the 8 threaded orderings ----
        if (s->pbox.cap < 2 * (size_t) np) s->pbox.alloc(2 * (size_t) np);          // (scratch of the refit kernels, not read by any render)
        sp.tnodes.ensure(2 * 8 * (size_t) nn); sp.rec_node.ensure(8 * (size_t) nn);
        RefitArgs a{};
        a.prim_src = sp.src.p; a.mesh_xf = s->mesh_xf.p; a.tris = sp.tris.p; a.shade = sp.shade.p; a.pbox = s->pbox.p; a.num_prims = np;
        HIPCHK(jtx_launch_refit_prims(a, s->stream));
        HIPCHK(jtx_launch_build_threaded(sp.nbox.p, sp.pos.p, sp.size.p, nn, sp.tnodes.p, sp.rec_node.p, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        std::memcpy(nodes2.data(), pinNodes, (size_t) nn * sizeof(jtx_mi_bvh_node));
        lap("records, 8 threaded orderings");
        std::vector<jtx_mi_tri_ref> refs(np); std::vector<int32_t> orig(np);
        for (int i = 0; i < np; ++i) { refs[i] = s->bvh.refs[ord[i]]; orig[i] = s->bvh.orig[ord[i]]; }
        // ---- 8-ary nodes ----
        bool wideOk = wideOnDevice && R.wide_ok && R.wide_depth <= kMaxWideDepth;
        int wideDepth = R.wide_depth, numWide = R.num_wide; size_t wideGranules = R.wide_granules;
        if (wantWide && !wideOnDevice) {
            std::vector<uint4> wh; std::vector<int32_t> wm; int wd = 0;
            if (buildWide(nodes2, wh, wd, &wm) && wd <= kMaxWideDepth) {
                sp.wide.fill(wh); sp.map.ensure(wm.size());
                stagedH2D(sp.map.p, wm.data(), wm.size() * sizeof(int32_t));
                wideOk = true; wideDepth = wd; numWide = (int) (wm.size() / 16); wideGranules = wh.size();
            }
        }
        if (!s->wide_fail.p) s->wide_fail.alloc(1);
        // ---- tiny scenes: the flat leaf list needs the positions on the host ----
        int nleafList = 0;
        {
            int nl = 0; for (const jtx_mi_bvh_node &n : nodes2) if (n.num_prims) ++nl;
            if (nl > 0 && nl <= 32 && nn > 1) {
                std::vector<int> hpos(8 * (size_t) nn);
                stagedD2H(hpos.data(), sp.pos.p, hpos.size() * sizeof(int));
                nleafList = buildLeafTables(nodes2, hpos, sp.lw_box, sp.lw_tab);
            }
        }
        // ---- scene radius (scene.hpp:81-84) and with it the DISTANT lights (scene.cpp:128-134) ----
        const jtx_mi_bvh_node &r = nodes2[0];
        const float dx = r.pmax[0] - r.pmin[0], dy = r.pmax[1] - r.pmin[1], dz = r.pmax[2] - r.pmin[2];
        const float radius = std::sqrt(dx * dx + dy * dy + dz * dz) / 2;
        bool newLights = false;
        if (s->lights.n) {
            std::vector<DLight> ls(s->lights.n);
            stagedD2H(ls.data(), s->lights.p, ls.size() * sizeof(DLight));
            for (auto &l : ls) if (l.type == 1) { l.scene_radius = radius; newLights = true; }
            if (newLights) sp.lights.fill(ls);
        }
        if (!commit) return 0;                                                 // jtx_mi_scene_reserve_rebuild: memory touched, kernels loaded, scene untouched
#ifdef JTX_TEST_HOOKS    /* libjtx_mi_testhooks.so only (build.py: build_test_hooks); the product library has no fault injection */
        if (getenv("JTX_FAIL_REBUILD_BEFORE_COMMIT")) throw std::runtime_error("injected failure before the commit (JTX_FAIL_REBUILD_BEFORE_COMMIT)");   // tests: failure atomicity
#endif
        // ---- commit: swaps and assignments only -- nothing below can throw ----
        s->prim_src.swap(sp.src); s->tris.swap(sp.tris); s->shade.swap(sp.shade); s->orig_id.swap(sp.orig);
        s->nbox.swap(sp.nbox); s->leaf_nodes.swap(sp.leaves); s->level_nodes.swap(sp.levels);
        s->tnodes.swap(sp.tnodes); s->rec_node.swap(sp.rec_node);
        s->bvh.nodes.swap(nodes2); s->bvh.refs.swap(refs); s->bvh.orig.swap(orig);
        s->bvh.max_depth = R.max_depth; s->bvh.scene_radius = radius;
        s->level_begin.swap(R.level_begin);
        s->num_leaves = R.nleaves;
        s->dev.tnodes = s->tnodes.p; s->dev.tris = s->tris.p; s->dev.shade = s->shade.p;
        s->dev.num_nodes = nn;
        s->dev.lds_threaded = (nn > 0 && 8 * (size_t) nn * 32 + (size_t) np * 48 <= kLdsThreadedBudget) ? 1 : 0;
        s->dev.wide = nullptr; s->dev.wide_depth = 0; s->num_wide = 0;
        if (wideOk) {
            s->wide.swap(sp.wide); s->wide_map.swap(sp.map);
            s->wide.n = wideGranules;                                       // (the allocation is larger; n = granules in use, as after a host build)
            s->dev.wide = s->wide.p; s->dev.wide_depth = wideDepth; s->num_wide = numWide;
        }
        s->lw_box.swap(sp.lw_box); s->lw_tab.swap(sp.lw_tab);
        s->dev.lw_box = nleafList ? s->lw_box.p : nullptr; s->dev.lw_tab = nleafList ? s->lw_tab.p : nullptr; s->dev.lw_leaves = nleafList;
        if (newLights) { s->lights.swap(sp.lights); s->dev.lights = s->lights.p; }
        s->refitted = 0; s->device_built = 1; s->xf_dirty = false;
        lap("host copies, lights");
        return 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}

int jtx_mi_scene_get_info(const jtx_mi_scene *s, jtx_mi_scene_info *out) {
    if (!s || !out) return fail("null argument");
    out->num_nodes = s->dev.num_nodes; out->num_prims = s->dev.num_prims; out->max_depth = s->bvh.max_depth;
    out->lds_resident = s->dev.lds_threaded; out->scene_radius = s->bvh.scene_radius; out->device_bytes = s->device_bytes;
    out->auto_integrator = autoIntegrator(*s);
    // (what the kernels walk: nothing when a refit or a build could not quantise the tree -- jtx_mi_scene_get_wide then returns 0 granules too)
    const uint64_t wb = s->dev.wide ? (uint64_t) s->wide.n * sizeof(uint4) : 0;
    out->wide_depth = s->dev.wide_depth; out->wide_bytes = wb > 0x7fffffffull ? 0x7fffffff : (int32_t) wb; out->wide_bytes64 = wb;
    out->rebuild_spare_bytes = s->spare.bytes();
    out->frame_slot_bytes = (s->work.cap + s->prog_ctl.cap) * sizeof(unsigned);
    for (int k = 0; k < JTX_MI_FRAME_SLOTS; ++k) out->frame_slot_bytes += s->rad[k].cap * sizeof(float4);
    out->refitted = s->refitted;
    out->device_built = s->device_built;
    out->num_cus = s->num_cus;
    { int bs = 0; out->resident_workgroups = jtx_render_paths_grid(s->dev, s->num_cus, &bs); out->workgroup_size = bs; }
    return 0;
}
int jtx_mi_scene_get_wide(jtx_mi_scene *s, uint32_t *granules_out, int64_t capacity, int64_t *num_granules_out) {
    try {
        if (!s) throw std::runtime_error("null scene");
        DeviceGuard dg(s->device);
        std::lock_guard<std::mutex> lk(s->mu);
        const size_t n = s->dev.wide ? s->wide.n : 0;
        if (num_granules_out) *num_granules_out = (int64_t) n;
        if (granules_out && n) {
            if ((int64_t) n > capacity) throw std::runtime_error("granules_out too small");
            HIPCHK(hipStreamSynchronize(s->stream));
            stagedD2H(granules_out, s->wide.p, n * sizeof(uint4));
        }
        return 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}
int jtx_mi_scene_get_bvh(const jtx_mi_scene *s, jtx_mi_bvh_node *nodes_out, jtx_mi_tri_ref *refs_out) {
    if (!s) return fail("null scene");
    if (nodes_out) std::memcpy(nodes_out, s->bvh.nodes.data(), s->bvh.nodes.size() * sizeof(jtx_mi_bvh_node));
    if (refs_out) std::memcpy(refs_out, s->bvh.refs.data(), s->bvh.refs.size() * sizeof(jtx_mi_tri_ref));
    return 0;
}

int jtx_mi_render_device(jtx_mi_scene *s, const jtx_mi_camera_desc *cam, const jtx_mi_render_opts *opts,
                         void *d_acc_rgb, void *d_img_rgb, void *stream) {
    try {
        if (!s || !cam || !d_acc_rgb) throw std::runtime_error("null argument");
        DeviceGuard dg(s->device);
        checkCamera(*cam);
        jtx_mi_render_opts o{}; if (opts) o = *opts;
        const int spp = cam->x_pixel_samples * cam->y_pixel_samples;
        const int sb = o.sample_begin > 0 ? o.sample_begin : 0;
        const int se = (o.sample_end > 0 && o.sample_end < spp) ? o.sample_end : spp;
        if (sb >= se) throw std::runtime_error("empty sample range");
        std::lock_guard<std::mutex> lk(s->mu);
        hipStream_t st = stream ? (hipStream_t) stream : s->stream;
        if (o.tile_world > 1) {                 // non-owned pixels must read as exactly 0 for the reduce
            if (sb == 0) HIPCHK(hipMemsetAsync(d_acc_rgb, 0, sizeof(float) * 3 * (size_t) cam->width * cam->height, st));
            if (d_img_rgb) HIPCHK(hipMemsetAsync(d_img_rgb, 0, 3 * (size_t) cam->width * cam->height, st));
        }
        launchRender(*s, *cam, o, sb, se, (float *) d_acc_rgb, (unsigned char *) d_img_rgb, st);
        return 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}

int jtx_mi_sync(jtx_mi_scene *s) {
    if (!s) return fail("null scene");
    try {
        DeviceGuard dg(s->device);
        HIPCHK(hipStreamSynchronize(s->stream));
        HIPCHK(hipDeviceSynchronize());
        return 0;
    } catch (const std::exception &e) { return fail(std::string("sync: ") + e.what()); }
}

int jtx_mi_kernel_time(jtx_mi_scene *s, float *ms_total, int32_t *launches) {
    try {
        if (!s) throw std::runtime_error("null scene");
        std::lock_guard<std::mutex> lk(s->mu);
        DeviceGuard dg(s->device);
        float total = 0; int n = 0;
        for (auto &e : s->pending) {
            HIPCHK(hipEventSynchronize(e.second));
            float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e.first, e.second));
            total += ms; ++n;
            s->free_events.push_back(e);
        }
        s->pending.clear();
        if (ms_total) *ms_total = total;
        if (launches) *launches = n;
        return 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}

#include "jtx_profile_readers.hpp"     // jtx_mi_debug_*: diagnostic builds only, nothing in the product

int jtx_mi_kernel_time_by_kind(jtx_mi_scene *s, float *ms5, int32_t *n5) {
    if (!s || !ms5 || !n5) return fail("null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    for (int i = 0; i < 5; ++i) { ms5[i] = s->ms_by_kind[i]; n5[i] = s->n_by_kind[i]; s->ms_by_kind[i] = 0; s->n_by_kind[i] = 0; }
    return 0;
}

// the scene's 64-word counter block <-> jtx_mi_counters ([0..8] the nine ray counters, [CNT_SHADE_T..] / [CNT_EVAL_T..] the per-class tallies)
static void countersAddWords(jtx_mi_counters &c, const unsigned long long *h) {
    c.n_camera += h[0]; c.n_closest += h[1]; c.n_any += h[2]; c.n_nodes_closest += h[3]; c.n_tri_closest += h[4];
    c.n_accept += h[5]; c.n_nodes_any += h[6]; c.n_tri_any += h[7]; c.n_shade += h[8];
    for (int i = 0; i < 8; ++i) { c.n_shade_class[i] += h[CNT_SHADE_T + i]; c.n_eval_class[i] += h[CNT_EVAL_T + i]; }
}
static void countersToWords(const jtx_mi_counters &c, unsigned long long *h) {
    h[0] = c.n_camera; h[1] = c.n_closest; h[2] = c.n_any; h[3] = c.n_nodes_closest; h[4] = c.n_tri_closest;
    h[5] = c.n_accept; h[6] = c.n_nodes_any; h[7] = c.n_tri_any; h[8] = c.n_shade;
    for (int i = 0; i < 8; ++i) { h[CNT_SHADE_T + i] = c.n_shade_class[i]; h[CNT_EVAL_T + i] = c.n_eval_class[i]; }
}

int jtx_mi_get_counters(jtx_mi_scene *s, jtx_mi_counters *out) {
    try {
        if (!s || !out) throw std::runtime_error("null argument");
        if (!s->counters.p) throw std::runtime_error("no counted render has run (opts.count_rays)");
        DeviceGuard dg(s->device);
        HIPCHK(hipStreamSynchronize(s->stream));
        HIPCHK(hipDeviceSynchronize());
        unsigned long long h[64];
        HIPCHK(hipMemcpy(h, s->counters.p, sizeof h, hipMemcpyDeviceToHost));
        *out = jtx_mi_counters{};
        countersAddWords(*out, h);
        return 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}

namespace {
bool passAbandoned(jtx_mi_scene &s, int slot) {
    jtx_mi_scene::PassRec &rec = s.pass[slot];
    if (!rec.last_work) return false;
    // the launches of the pass in order: every one before the first abandoned one completed and was resolved into the film.  (The
    // verdict is read from HOST memory: every launch's resolve has written it there before its stream drained -- no device copy.)
    for (const auto &part : rec.parts)
        if (*part.abandoned != 0u) { rec.resolved_end = part.begin; return true; }
    if (!rec.parts.empty()) rec.resolved_end = rec.parts.back().end;
    return false;
}
}

// 1: a cancellation is pending (jtx_mi_cancel since the last reset); 2: and the last persistent pass was abandoned by it
int jtx_mi_cancel_pending(jtx_mi_scene *s, int32_t *out) {
    try {
        if (!s || !out || !s->stop_host) throw std::runtime_error("null argument");
        DeviceGuard dg(s->device);
        const bool pending = __atomic_load_n(s->stop_host, __ATOMIC_ACQUIRE) != 0;
        bool abandoned = false;
        if (pending) { std::lock_guard<std::mutex> lk(s->mu); abandoned = passAbandoned(*s, s->last_slot); }   // (reads and writes the pass records of the scene)
        *out = pending ? (abandoned ? 2 : 1) : 0;
        return 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}
int jtx_mi_cancel_reset(jtx_mi_scene *s) {
    if (!s || !s->stop_host) return fail("null scene");
    __atomic_store_n(s->stop_host, 0u, __ATOMIC_RELEASE);
    return 0;
}

int jtx_mi_cancel(jtx_mi_scene *s) {
    if (!s || !s->stop_host) return fail("null scene");
    __atomic_store_n(s->stop_host, 1u, __ATOMIC_RELEASE);
    return 0;
}

int jtx_mi_render(jtx_mi_scene *s, const jtx_mi_camera_desc *cam, const jtx_mi_render_opts *opts,
                  float *acc_rgb, uint8_t *img_rgb, jtx_mi_progress_cb cb, void *user) {
    try {
        if (!s || !cam || !acc_rgb) throw std::runtime_error("null argument");
        DeviceGuard dg(s->device);
        checkCamera(*cam);
        jtx_mi_render_opts o{}; if (opts) o = *opts;
        const int spp = cam->x_pixel_samples * cam->y_pixel_samples;
        const int sb = o.sample_begin > 0 ? o.sample_begin : 0;
        const int se = (o.sample_end > 0 && o.sample_end < spp) ? o.sample_end : spp;
        if (sb >= se) throw std::runtime_error("empty sample range");
        const size_t npix = (size_t) cam->width * cam->height;
        std::unique_lock<std::mutex> lk(s->mu);
        __atomic_store_n(s->stop_host, 0u, __ATOMIC_RELEASE);                  // stopRender_ = false (camera.cpp:48)
        if (s->film_acc.n != 3 * npix) { s->film_acc.alloc(3 * npix); s->film_img.alloc(3 * npix); }
        if (sb == 0) HIPCHK(hipMemsetAsync(s->film_acc.p, 0, sizeof(float) * 3 * npix, s->stream));
        else if (hostPinned(acc_rgb)) HIPCHK(hipMemcpyAsync(s->film_acc.p, acc_rgb, sizeof(float) * 3 * npix, hipMemcpyHostToDevice, s->stream));
        else { HIPCHK(hipStreamSynchronize(s->stream)); stagedH2D(s->film_acc.p, acc_rgb, sizeof(float) * 3 * npix); }      // a pageable caller buffer: through the library's staging
        HIPCHK(hipMemsetAsync(s->film_img.p, 0, 3 * npix, s->stream));
        const int tick = (cb && o.samples_per_tick > 0 && o.samples_per_tick < se - sb) ? o.samples_per_tick : (se - sb);
        jtx_mi_counters total{}; const bool count = o.count_rays != 0;
        unsigned char *dimg = img_rgb ? s->film_img.p : nullptr;
        struct Drain { hipStream_t a, b, c; ~Drain() { for (hipStream_t x : {a, b, c}) if (x) (void) hipStreamSynchronize(x); } };   // nothing of this call outlives it
        // Film delivery: a pinned caller buffer (jtx_mi_pin_host / hipHostRegister: what the Camera mirrors do with img_ /
        // acc_) is the DMA target itself; a pageable one is fed through the scene's pinned staging buffers and one host
        // memcpy -- hipMemcpy to pageable memory would stage through small driver buffers at a fraction of the PCIe rate.
        const bool imgDirect = img_rgb && hostPinned(img_rgb), accDirect = hostPinned(acc_rgb);
        if ((img_rgb && !imgDirect) || !accDirect) {
            if (s->pin_pixels != npix) {
                if (s->pin_img) (void) hipHostFree(s->pin_img);
                if (s->pin_acc) (void) hipHostFree(s->pin_acc);
                s->pin_img = nullptr; s->pin_acc = nullptr; s->pin_pixels = 0;
                HIPCHK(hipHostMalloc((void **) &s->pin_img, 3 * npix, hipHostMallocDefault));
                HIPCHK(hipHostMalloc((void **) &s->pin_acc, 3 * npix * sizeof(float), hipHostMallocDefault));
                s->pin_pixels = npix;
            }
        }
        static const bool trace = getenv("JTX_TRACE_RENDER") != nullptr;        // developer aid: host-side phase times to stderr
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double tMark = now();
        auto lap = [&](const char *what) { if (trace) { const double t = now(); fprintf(stderr, "[jtx_mi_render] %-18s %8.3f ms\n", what, t - tMark); tMark = t; } };
        bool cancelled = false;
        int done = sb;                                                          // strata whose sums are in the film of every pixel
        jtx_mi_render_opts oSlot0 = o; oSlot0.frame_slot = 0; oSlot0.sequence_end = 1;
        const bool ownsTiles = progOwnedTiles(*cam, o) > 0;                           // (a shard without tiles launches nothing: pass by pass below)
        if (cb && tick < se - sb && ownsTiles && jtx_prog_usable(s, o)) {
            // ---- PROGRESSIVE (round 6): ONE launch for all passes, whatever samplesPerPass_ is (before: a launch, a resolve, a preview copy and
            // a host turn-around per pass -- 46.5 ms per C2 frame at samplesPerPass_ = 1 for a 22.7 ms kernel).  k_render_paths<.., PROG> traces
            // pass after pass; k_resolve_progressive, beside it, adds every pass that is complete to the film and the preview and says, in
            // host-mapped memory, how many passes are in the film of EVERY pixel (currentSample_, camera.cpp:68-74).  This thread watches
            // that: when it has advanced, the RGB8 preview is copied as it stands -- every pixel of it shows at least the reported strata,
            // some already a pass more: the reference's UI reads img_ unsynchronised beside the tile workers too (display.cpp:702-703) --
            // and the callback runs once per pass, in order.  The launch does not wait for the callback.  A stop (the callback's return
            // value, jtx_mi_cancel from any thread) closes the chunk counter: the passes whose chunks were all dealt are finished and
            // added, the pass after them leaves no trace.
            // A range whose records exceed the cap goes in several such launches, one after the other.
            if (!s->copy_stream) HIPCHK(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
            if (!s->resolve_stream) {
                int least = 0, greatest = 0;
                (void) hipDeviceGetStreamPriorityRange(&least, &greatest);
                HIPCHK(hipStreamCreateWithPriority(&s->resolve_stream, hipStreamNonBlocking, greatest));
            }
            Drain drain{s->stream, s->resolve_stream, s->copy_stream};
            const long perLaunch = jtx_prog_span(s, *cam, o, tick);
            int reported = sb;                                                  // strata the callback has been told of
            bool stopAsked = false;
            for (int b0 = sb; b0 < se && !cancelled; ) {
                const int e0 = (long) b0 + perLaunch < se ? b0 + (int) perLaunch : se;
                JtxProgRun run;
                jtx_prog_begin(s, *cam, oSlot0, b0, e0, tick, s->film_acc.p, dimg, s->stream, 0, false, run);
                lap("enqueue");
                bool resolverGaveUp = false;
                unsigned idle = 0;
                while (true) {
                    const bool finished = jtx_prog_finished(s);                 // (first: the words read below are then final)
                    const int have = jtx_prog_completed(s, run, &resolverGaveUp);
                    if (have > reported && !stopAsked) {
                        idle = 0;
                        if (img_rgb) {                                          // the preview as it stands, to (pinned) host memory
                            HIPCHK(hipMemcpyAsync(imgDirect ? img_rgb : s->pin_img, dimg, 3 * npix, hipMemcpyDeviceToHost, s->copy_stream));
                            HIPCHK(hipStreamSynchronize(s->copy_stream));
                            if (!imgDirect) std::memcpy(img_rgb, s->pin_img, 3 * npix);
                        }
                        lap("preview");
                        while (reported < have && !stopAsked) {
                            reported = reported + tick < have ? reported + tick : have;
                            lk.unlock();
                            const int stop = cb(reported, spp, user);           // currentSample_ advance, camera.cpp:68-74
                            lk.lock();
                            if (stop) { stopAsked = true; __atomic_store_n(s->stop_host, 1u, __ATOMIC_RELEASE); }
                        }
                    }
                    if (finished) break;
                    if (++idle > 64) std::this_thread::sleep_for(std::chrono::microseconds(idle > 2048 ? 200 : 20)); else std::this_thread::yield();   // (a long pass: the poll backs off)
                }
                HIPCHK(hipStreamSynchronize(s->stream));
                done = jtx_prog_completed(s, run, &resolverGaveUp);
                if (resolverGaveUp) throw std::runtime_error("progressive launch: the resolver waited a minute for the path kernel and gave up (the film holds the passes added so far)");
#ifdef JTX_DBG_PROG      /* diagnostic build: the resolver leader's log of (time, groups dealt, groups out of every wave's hands) */
                { unsigned h[48]; HIPCHK(hipMemcpy(h, s->prog_ctl.p + 64 + 16, sizeof h, hipMemcpyDeviceToHost)); for (unsigned i = 0; i < h[0] && i < 14; ++i) fprintf(stderr, "[leader] t %8.3f ms dealt groups %u by waves %u\n", (h[1 + 3 * i] - h[1]) / 1000.0, h[2 + 3 * i], h[3 + 3 * i]);
                  unsigned g[128]; HIPCHK(hipMemcpy(g, s->prog_ctl.p + 128 + 49152, sizeof g, hipMemcpyDeviceToHost));
                  for (unsigned k = 0; k < 16; ++k) if (g[80 + k]) fprintf(stderr, "    SIMD wave slot %2u: %4u waves, mean fetches %6.2f, still holding group 0: %u\n", k, g[80 + k], (double) g[96 + k] / g[80 + k], g[112 + k]);
                  fprintf(stderr, "[snapshot at 2 groups dealt, %.3f ms before the first line] waves seen %u, fetches min %u max %u mean %.2f; waves still holding group 0: %u (mean fetches %.2f)\n",
                          (h[1] - g[7]) / 1000.0, g[5], g[2], g[3], g[5] ? (double) g[4] / g[5] : 0.0, g[0], g[0] ? (double) g[1] / g[0] : 0.0);
                  for (unsigned i = 0; i < g[6] && i < 16; ++i) fprintf(stderr, "    wave %5u fetches %3u hw_id %08x (wave %u simd %u cu %u sh %u se %u) word %u\n", g[8 + 4 * i], g[9 + 4 * i], g[10 + 4 * i],
                          g[10 + 4 * i] & 15u, (g[10 + 4 * i] >> 4) & 3u, (g[10 + 4 * i] >> 8) & 15u, (g[10 + 4 * i] >> 12) & 1u, (g[10 + 4 * i] >> 13) & 7u, g[11 + 4 * i]); }
#endif
                if (done < e0 || stopAsked) cancelled = true;
                b0 = e0;
                lap("launch");
            }
        } else {
            // ---- one pass at a time: a frame without passes (one launch), or launches on the scene's singletons (counting, the alternate
            // Li, integrator 2), which do not poll the cancellation word -- it is honoured between their passes ----
            Drain drain{s->stream, nullptr, nullptr};
            for (int b = sb; b < se; b += tick) {
                const int e = b + tick < se ? b + tick : se;
                if (b > sb && __atomic_load_n(s->stop_host, __ATOMIC_ACQUIRE) != 0) { cancelled = true; break; }
                launchRender(*s, *cam, o, b, e, s->film_acc.p, dimg, s->stream);
                if (img_rgb && cb) HIPCHK(hipMemcpyAsync(imgDirect ? img_rgb : s->pin_img, dimg, 3 * npix, hipMemcpyDeviceToHost, s->stream));
                HIPCHK(hipStreamSynchronize(s->stream));
                lap("pass");
                if (passAbandoned(*s, o.frame_slot)) {                         // the kernels saw the cancellation: the abandoned launch left no trace;
                    if (s->pass[o.frame_slot].resolved_end > done) done = s->pass[o.frame_slot].resolved_end;   // earlier launches of a split pass are in the film and count
                    cancelled = true; break;
                }
                done = e;
                if (count) {
                    unsigned long long h[64];
                    HIPCHK(hipMemcpy(h, s->counters.p, sizeof h, hipMemcpyDeviceToHost));
                    countersAddWords(total, h);
                }
                if (cb) {
                    if (img_rgb && !imgDirect) std::memcpy(img_rgb, s->pin_img, 3 * npix);
                    lk.unlock();
                    const int stop = cb(done, spp, user);                       // currentSample_ advance, camera.cpp:68-74
                    lk.lock();
                    if (stop) { cancelled = true; break; }
                }
            }
        }
        // the film as the launches left it: accumulation buffer and the preview that belongs to it
        HIPCHK(hipMemcpyAsync(accDirect ? acc_rgb : s->pin_acc, s->film_acc.p, sizeof(float) * 3 * npix, hipMemcpyDeviceToHost, s->stream));
        if (img_rgb) HIPCHK(hipMemcpyAsync(imgDirect ? img_rgb : s->pin_img, dimg, 3 * npix, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        lap("film D2H");
        if (!accDirect) std::memcpy(acc_rgb, s->pin_acc, sizeof(float) * 3 * npix);
        if (img_rgb && !imgDirect) std::memcpy(img_rgb, s->pin_img, 3 * npix);
        lap("film memcpy");
        s->last_completed = done;
        if (count) {   // leave the frame totals on the device for jtx_mi_get_counters
            unsigned long long h[64] = {};
            countersToWords(total, h);
            HIPCHK(hipMemcpy(s->counters.p, h, sizeof h, hipMemcpyHostToDevice));
        }
        return cancelled ? JTX_MI_CANCELLED : 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}

// Pin a caller buffer so that jtx_mi_render DMA-writes it directly (Camera::img_ / acc_ live as long as the camera).
int jtx_mi_pin_host(void *ptr, uint64_t bytes) {
    if (!ptr || !bytes) return fail("null argument");
    hipError_t e = hipHostRegister(ptr, (size_t) bytes, hipHostRegisterDefault);
    if (e != hipSuccess) { (void) hipGetLastError(); return fail(std::string("hipHostRegister: ") + hipGetErrorString(e)); }
    return 0;
}
int jtx_mi_unpin_host(void *ptr) {
    if (!ptr) return fail("null argument");
    hipError_t e = hipHostUnregister(ptr);
    if (e != hipSuccess) { (void) hipGetLastError(); return fail(std::string("hipHostUnregister: ") + hipGetErrorString(e)); }
    return 0;
}

// the frame slots' working memory (radiance records, chunk counters) back to the device; the next render allocates what it needs
int jtx_mi_scene_release_frames(jtx_mi_scene *s) {
    try {
        if (!s) throw std::runtime_error("null scene");
        std::lock_guard<std::mutex> lk(s->mu);
        DeviceGuard dg(s->device);
        HIPCHK(hipDeviceSynchronize());                 // launches in any slot, on any stream, have ended
        for (int k = 0; k < JTX_MI_FRAME_SLOTS; ++k) { s->rad[k].release(); s->pass[k] = jtx_mi_scene::PassRec{}; }
        s->work.release(); s->work_slot = 0; s->prog_ctl.release();
        return 0;
    } catch (const std::exception &e) { return fail(std::string("release_frames: ") + e.what()); }
}

int jtx_mi_last_completed_sample(const jtx_mi_scene *s, int32_t *out) {
    if (!s || !out) return fail("null argument");
    *out = s->last_completed;
    return 0;
}

} // extern "C"

// ---- fine-grained parity entry points: host buffers in, host buffers out ----
namespace {
template <class T> struct Tmp {
    T *p = nullptr; size_t n = 0;
    Tmp(size_t count) : n(count) { if (count) HIPCHK(hipMalloc((void **) &p, count * sizeof(T))); }
    Tmp(const T *h, size_t count) : n(count) {
        if (count) { HIPCHK(hipMalloc((void **) &p, count * sizeof(T))); stagedH2D(p, h, count * sizeof(T)); }
    }
    void down(T *h) { if (h && n) stagedD2H(h, p, n * sizeof(T)); }
    ~Tmp() { if (p) (void) hipFree(p); }
};
}


extern "C" {

// traversal: JTX_MI_TRAVERSAL_BINARY (0) the binary threaded records -- the reference's node visits, what the counted kernels walk;
// JTX_MI_TRAVERSAL_PRODUCTION (1) the structure the TIMED launch of this scene walks (flat leaf list / LDS copy / 8-ary nodes / binary
// records, by the rule of jtx_launch_render_paths); 2 + SRC: that source if the scene carries it, an error otherwise (tests)
static int batchSource(const jtx_mi_scene &s, int traversal) {
    if (traversal == 0) return SRC_GLOBAL;
    if (traversal == 1) return jtx_production_source(s.dev);
    const int src = traversal - 2;
    const bool ok = src == SRC_GLOBAL || (src == SRC_LDS && s.dev.lds_threaded) || (src == SRC_LEAF && s.dev.lds_threaded && s.dev.lw_leaves > 0) ||
                    (src == SRC_WIDE && !s.dev.lds_threaded && s.dev.wide);
    if (!ok) throw std::runtime_error("traversal " + std::to_string(traversal) + ": this scene does not carry that structure");
    return src;
}

int jtx_mi_closest_hit_batch_via(jtx_mi_scene *s, int32_t traversal, int32_t n, const float *o, const float *d, float tmin, float tmax,
                                 int32_t *hit, float *t, int32_t *prim, float *b1, float *b2, float *point, float *normal,
                                 float *uv, int32_t *source_out) {
    try {
        if (!s || n < 0 || (n && (!o || !d))) throw std::runtime_error("bad argument");
        DeviceGuard dg(s->device);
        const int src = batchSource(*s, traversal);
        if (source_out) *source_out = src;
        if (n == 0) return 0;
        Tmp<float> dO(o, 3 * (size_t) n), dD(d, 3 * (size_t) n), dT(n), dB1(n), dB2(n), dP(3 * (size_t) n), dN(3 * (size_t) n), dUV(2 * (size_t) n);
        Tmp<int> dHit(n), dPrim(n);
        HIPCHK(jtx_launch_closest_batch(s->dev, src, n, dO.p, dD.p, tmin, tmax, dHit.p, dT.p, dPrim.p, dB1.p, dB2.p, dP.p, dN.p, dUV.p, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        dHit.down(hit); dT.down(t); dPrim.down(prim); dB1.down(b1); dB2.down(b2); dP.down(point); dN.down(normal); dUV.down(uv);
        return 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}
int jtx_mi_closest_hit_batch(jtx_mi_scene *s, int32_t n, const float *o, const float *d, float tmin, float tmax,
                             int32_t *hit, float *t, int32_t *prim, float *b1, float *b2, float *point, float *normal,
                             float *uv) {
    return jtx_mi_closest_hit_batch_via(s, 0, n, o, d, tmin, tmax, hit, t, prim, b1, b2, point, normal, uv, nullptr);
}

int jtx_mi_any_hit_batch_via(jtx_mi_scene *s, int32_t traversal, int32_t n, const float *o, const float *d, const float *tmin, const float *tmax,
                             int32_t *hit, int32_t *source_out) {
    try {
        if (!s || n < 0 || (n && (!o || !d || !tmin || !tmax || !hit))) throw std::runtime_error("bad argument");
        DeviceGuard dg(s->device);
        const int src = batchSource(*s, traversal);
        if (source_out) *source_out = src;
        if (n == 0) return 0;
        Tmp<float> dO(o, 3 * (size_t) n), dD(d, 3 * (size_t) n), dA(tmin, n), dB(tmax, n);
        Tmp<int> dHit(n);
        HIPCHK(jtx_launch_any_batch(s->dev, src, n, dO.p, dD.p, dA.p, dB.p, dHit.p, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        dHit.down(hit);
        return 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}
int jtx_mi_any_hit_batch(jtx_mi_scene *s, int32_t n, const float *o, const float *d, const float *tmin, const float *tmax,
                         int32_t *hit) {
    return jtx_mi_any_hit_batch_via(s, 0, n, o, d, tmin, tmax, hit, nullptr);
}

static int bxdfBatch(jtx_mi_scene *s, int mode, int32_t material, int32_t n, const float *normal, const float *uv, const float *wo,
                     const float *wi_in, const float *uc, const float *u2, int32_t *ok, float *f, float *wi_out, float *pdf) {
    try {
        if (!s || n < 0 || (n && (!normal || !wo))) throw std::runtime_error("bad argument");
        if (material < 0 || material >= s->dev.num_materials) throw std::runtime_error("material out of range");
        DeviceGuard dg(s->device);
        if (n == 0) return 0;
        const size_t N = (size_t) n;
        Tmp<float> dN(normal, 3 * N), dUV(uv, uv ? 2 * N : 0), dWo(wo, 3 * N), dWi(wi_in, wi_in ? 3 * N : 0), dUc(uc, uc ? N : 0),
            dU2(u2, u2 ? 2 * N : 0), dF(3 * N), dWiOut(3 * N), dPdf(N);
        Tmp<int> dOk(N);
        HIPCHK(jtx_launch_bxdf_batch(s->dev, mode, material, n, dN.p, dUV.p, dWo.p, dWi.p, dUc.p, dU2.p, dOk.p, dF.p, dWiOut.p, dPdf.p, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        if (ok) dOk.down(ok);
        dF.down(f); dWiOut.down(wi_out); dPdf.down(pdf);
        return 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}
int jtx_mi_bxdf_sample_batch(jtx_mi_scene *s, int32_t material, int32_t n, const float *normal, const float *uv, const float *wo,
                             const float *uc, const float *u2, int32_t *ok, float *f, float *wi, float *pdf) {
    if (n > 0 && (!uc || !u2)) return fail("bad argument");
    return bxdfBatch(s, 0, material, n, normal, uv, wo, nullptr, uc, u2, ok, f, wi, pdf);
}
int jtx_mi_bxdf_eval_batch(jtx_mi_scene *s, int32_t material, int32_t n, const float *normal, const float *uv, const float *wo,
                           const float *wi, float *f) {
    if (n > 0 && !wi) return fail("bad argument");
    return bxdfBatch(s, 1, material, n, normal, uv, wo, wi, nullptr, nullptr, nullptr, f, nullptr, nullptr);
}
int jtx_mi_bxdf_pdf_batch(jtx_mi_scene *s, int32_t material, int32_t n, const float *normal, const float *uv, const float *wo,
                          const float *wi, float *pdf) {
    if (n > 0 && !wi) return fail("bad argument");
    return bxdfBatch(s, 1, material, n, normal, uv, wo, wi, nullptr, nullptr, nullptr, nullptr, nullptr, pdf);
}

int jtx_mi_camera_rays(const jtx_mi_camera_desc *cam, int32_t n, const int32_t *row, const int32_t *col, const int32_t *sample,
                       float *o, float *d) {
    try {
        if (!cam || n < 0 || (n && (!row || !col || !sample))) throw std::runtime_error("bad argument");
        checkCamera(*cam);
        if (n == 0) return 0;
        Tmp<int> dR(row, n), dC(col, n), dS(sample, n);
        Tmp<float> dO(3 * (size_t) n), dD(3 * (size_t) n);
        HIPCHK(jtx_launch_camera_rays(deriveCamera(*cam), n, dR.p, dC.p, dS.p, dO.p, dD.p, nullptr));
        HIPCHK(hipDeviceSynchronize());
        dO.down(o); dD.down(d);
        return 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}

int jtx_mi_radiance_samples(jtx_mi_scene *s, const jtx_mi_camera_desc *cam, int32_t n, const int32_t *row, const int32_t *col,
                            const int32_t *sample, float *rgb) {
    try {
        if (!s || !cam || n < 0 || (n && (!row || !col || !sample || !rgb))) throw std::runtime_error("bad argument");
        DeviceGuard dg(s->device);
        checkCamera(*cam);
        if (n == 0) return 0;
        Tmp<int> dR(row, n), dC(col, n), dS(sample, n);
        Tmp<float> dRGB(3 * (size_t) n);
        HIPCHK(jtx_launch_radiance_samples(s->dev, deriveCamera(*cam), cam->max_depth, n, dR.p, dC.p, dS.p, dRGB.p, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        dRGB.down(rgb);
        return 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}

int jtx_mi_radiance_samples_li(jtx_mi_scene *s, const jtx_mi_camera_desc *cam, int32_t li, int32_t n, const int32_t *row,
                               const int32_t *col, const int32_t *sample, float *rgb) {
    try {
        if (!s || !cam || n < 0 || (n && (!row || !col || !sample || !rgb))) throw std::runtime_error("bad argument");
        if (li < 0 || li > 2) throw std::runtime_error("li: 0 integrateMIS, 1 integrate, 2 integrateBasic");
        if (li == 1 && s->dev.num_lights == 0) throw std::runtime_error("integrate needs at least one light");
        DeviceGuard dg(s->device);
        checkCamera(*cam);
        if (n == 0) return 0;
        Tmp<int> dR(row, n), dC(col, n), dS(sample, n);
        Tmp<float> dRGB(3 * (size_t) n);
        HIPCHK(jtx_launch_radiance_samples_alt(s->dev, deriveCamera(*cam), cam->max_depth, li, n, dR.p, dC.p, dS.p, dRGB.p, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        dRGB.down(rgb);
        return 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}

int jtx_mi_rng_stream(uint32_t x, uint32_t y, uint32_t n, int32_t count, uint32_t *out_u32, float *out_f32) {
    try {
        if (count < 0) throw std::runtime_error("bad argument");
        if (count == 0) return 0;
        Tmp<uint32_t> dU(count); Tmp<float> dF(count);
        HIPCHK(jtx_launch_rng_stream(x, y, n, count, dU.p, dF.p, nullptr));
        HIPCHK(hipDeviceSynchronize());
        dU.down(out_u32); dF.down(out_f32);
        return 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}

int jtx_mi_sincos_batch(const float *x, int32_t n, float *out_sin, float *out_cos) {
    try {
        if (n < 0 || (n && !x)) throw std::runtime_error("bad argument");
        if (n == 0) return 0;
        Tmp<float> dX(x, n), dS(n), dC(n);
        HIPCHK(jtx_launch_sincos(dX.p, n, dS.p, dC.p, nullptr));
        HIPCHK(hipDeviceSynchronize());
        dS.down(out_sin); dC.down(out_cos);
        return 0;
    } catch (const std::exception &e) { return fail(e.what()); }
}

} // extern "C"
