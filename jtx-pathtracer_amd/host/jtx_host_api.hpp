// jtx_host_api.hpp -- host C++ mirror of the reference's Scene / Camera / integrator interface, over
// the C-ABI of include/jtx_mi.h.  Class, member and function names are the reference's
// (src/scene.hpp:24-91, src/camera.hpp:20-188, src/material.hpp:5-23, src/lights/lights.hpp:24-34,
// src/mesh.hpp:10-69, src/image.hpp:18-92, src/integrator.hpp:12, src/bsdf/bxdf.hpp:131-133) so that
// the reference's callers (Display::renderScene, main.cpp) compile against it; everything that
// computes is forwarded to the HIP library.  There is no CPU rendering path in here.
#pragma once
#include "../../include/jtx_mi.h"

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace jtxmi {

struct Vec3 { float x = 0, y = 0, z = 0; Vec3() = default; Vec3(float a, float b, float c) : x(a), y(b), z(c) {} explicit Vec3(float s) : x(s), y(s), z(s) {}
              float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); } float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); } };
struct Vec2f { float x = 0, y = 0; };
struct Vec3i { int x = 0, y = 0, z = 0; Vec3i() = default; Vec3i(int a, int b, int c) : x(a), y(b), z(c) {} };
struct Ray { Vec3 origin, dir; float time = 0; Ray() = default; Ray(Vec3 o, Vec3 d, float t = 0) : origin(o), dir(d), time(t) {} };
struct Interval { float min, max; Interval(float a, float b) : min(a), max(b) {} };
struct AABB {                                                         // util/aabb.hpp:7-16: the default box is the empty one
    Vec3 pmin{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    Vec3 pmax{std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest()};
    Vec3 diagonal() const { return Vec3(pmax.x - pmin.x, pmax.y - pmin.y, pmax.z - pmin.z); }
};
struct Transform { float m[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}; };
constexpr float INF = std::numeric_limits<float>::infinity();

struct CameraProperties { Vec3 center, target, up; float yfov = 20, defocusAngle = 0, focusDistance = 1; };   // scene.hpp:15-22

struct Material {                                                     // material.hpp:5-23; tex ids default to -1 (quirk Q4)
    enum Type { DIFFUSE = 0, DIELECTRIC = 1, CONDUCTOR = 2, METALLIC_ROUGHNESS = 3 };
    Type type = DIFFUSE; Vec3 albedo; float refractionIndex = 0; Vec3 IOR; Vec3 k; float alphaX = 0, alphaY = 0; Vec3 emission;
    int albedoTexId = -1; int metallicRoughnessTexId = -1;
};
struct Light { enum Type { POINT = 0, DISTANT = 1 }; Type type = POINT; Vec3 position, intensity; float scale = 1, sceneRadius = 0; };   // lights.hpp:24-34
struct SurfaceIntersection { Vec3 point, normal; Vec2f uv; const Material *material = nullptr; float t = 0; bool frontFace = false; };      // material.hpp:25-40
struct BSDFSample { Vec3 fSample, w_i; float pdf = 0, eta = 0; bool isSpecular = false, isTransmission = false; };                           // bxdf.hpp:118-125
struct TextureImage { int width_ = 0, height_ = 0, channels_ = 0; std::vector<float> data_; };                                              // image.hpp:99-226

struct Mesh {                                                         // mesh.hpp:10-69 (arrays owned by the caller, as in the reference)
    std::string name; int numVertices = 0, numIndices = 0;
    Vec3i *indices = nullptr; Vec3 *vertices = nullptr; Vec3 *normals = nullptr; Vec2f *uvs = nullptr; Material *material = nullptr;
    Transform transform;
    Mesh(Vec3i *i, int ni, Vec3 *v, int nv, Vec3 *n, Material *m) : numVertices(nv), numIndices(ni), indices(i), vertices(v), normals(n), material(m) {}
    Mesh(Vec3i *i, int ni, Vec3 *v, int nv, Vec3 *n, Vec2f *uv, Material *m) : numVertices(nv), numIndices(ni), indices(i), vertices(v), normals(n), uvs(uv), material(m) {}
};
struct Triangle { int index, meshIndex; };                            // mesh.hpp:202-204

inline void check(int rc) { if (rc) throw std::runtime_error(jtx_mi_last_error()); }

class Scene {                                                         // scene.hpp:24-91
public:
    std::string name;
    std::vector<Material> materials;
    std::vector<Light> lights;
    Vec3 skyColor;
    std::vector<Triangle> triangles;
    std::vector<Mesh> meshes;
    std::vector<TextureImage> textures;
    CameraProperties cameraProperties;
    std::vector<std::shared_ptr<void>> storage;                       // arrays a loader allocated for `meshes` (the reference leaks its new[])

    ~Scene() { destroyBVH(); }
    int numPrimitives() const { return (int) triangles.size(); }

    // Multi-GPU: name the devices BEFORE buildBVH(); StaticCamera::render then shards the frame over them
    // (jtx_mi_multi_render: tile k of camera.cpp:55-64 goes to devices[k % n]).  Empty = the current device only.
    void useDevices(std::vector<int> devices) { devices_ = std::move(devices); }

    void buildBVH(int maxPrimsInNode = 1) {                           // scene.cpp:96-135 -> jtx_mi_scene_create
        if (handle_) return;
        std::vector<jtx_mi_mesh> ms(meshes.size());
        for (size_t i = 0; i < meshes.size(); ++i) {
            const Mesh &m = meshes[i]; jtx_mi_mesh &o = ms[i];
            o.num_triangles = m.numIndices; o.num_vertices = m.numVertices;
            o.indices = &m.indices[0].x; o.vertices = &m.vertices[0].x; o.normals = &m.normals[0].x; o.uvs = m.uvs ? &m.uvs[0].x : nullptr;
            o.material = (int) (m.material - materials.data());       // Material* -> index (loader.cpp:15-17 keeps the vector stable)
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) o.transform[4 * r + c] = m.transform.m[r][c];
        }
        std::vector<jtx_mi_material> mats(materials.size());
        for (size_t i = 0; i < materials.size(); ++i) {
            const Material &m = materials[i]; jtx_mi_material &o = mats[i];
            o.type = m.type; o.alpha_x = m.alphaX; o.alpha_y = m.alphaY; o.albedo_tex = m.albedoTexId; o.mr_tex = m.metallicRoughnessTexId;
            for (int k = 0; k < 3; ++k) { o.albedo[k] = m.albedo[k]; o.ior[k] = m.IOR[k]; o.k[k] = m.k[k]; o.emission[k] = m.emission[k]; }
        }
        std::vector<jtx_mi_light> ls(lights.size());
        for (size_t i = 0; i < lights.size(); ++i) {
            ls[i].type = lights[i].type; ls[i].scale = lights[i].scale; ls[i].scene_radius = lights[i].sceneRadius;
            for (int k = 0; k < 3; ++k) { ls[i].position[k] = lights[i].position[k]; ls[i].intensity[k] = lights[i].intensity[k]; }
        }
        std::vector<jtx_mi_texture> ts(textures.size());
        for (size_t i = 0; i < textures.size(); ++i) ts[i] = {textures[i].width_, textures[i].height_, textures[i].channels_, textures[i].data_.data()};
        static_assert(sizeof(Triangle) == sizeof(jtx_mi_tri_ref), "Triangle layout");
        jtx_mi_scene_desc d{};
        d.num_meshes = (int) ms.size(); d.meshes = ms.data();
        d.num_tri_refs = (int) triangles.size(); d.tri_refs = (const jtx_mi_tri_ref *) triangles.data();
        d.num_materials = (int) mats.size(); d.materials = mats.data();
        d.num_lights = (int) ls.size(); d.lights = ls.data();
        d.num_textures = (int) ts.size(); d.textures = ts.data();
        d.sky_color[0] = skyColor.x; d.sky_color[1] = skyColor.y; d.sky_color[2] = skyColor.z;
        d.max_prims_in_node = maxPrimsInNode;
        check(jtx_mi_scene_create(&d, &handle_));
        if (devices_.size() > 1) check(jtx_mi_multi_create(&d, devices_.data(), (int) devices_.size(), &multi_));
        jtx_mi_scene_info info; check(jtx_mi_scene_get_info(handle_, &info));
        for (auto &l : lights) if (l.type == Light::DISTANT) l.sceneRadius = info.scene_radius;   // scene.cpp:128-134
    }
    void destroyBVH() { if (multi_) { jtx_mi_multi_destroy(multi_); multi_ = nullptr; } if (handle_) { jtx_mi_scene_destroy(handle_); handle_ = nullptr; } }
    jtx_mi_multi *multiHandle() const { return multi_; }
    void rebuildBVH(int maxPrimsInNode = 1) { destroyBVH(); buildBVH(maxPrimsInNode); }
    // The edit loop's cheap path (display.cpp:545-588 set rebuildBVH_ after every transform edit): push every Mesh::transform
    // and refit the device scene in place, topology kept (jtx_mi_scene_refit; a correct render of the edited scene, flagged
    // `refitted`; rebuildBVH() gives the reference's own tree again).  Single-device scenes only.
    void refitBVH() {
        if (!handle_) { buildBVH(); return; }
        for (size_t i = 0; i < meshes.size(); ++i) check(jtx_mi_scene_set_transform(handle_, (int) i, &meshes[i].transform.m[0][0]));
        check(jtx_mi_scene_refit(handle_));
        jtx_mi_scene_info info; check(jtx_mi_scene_get_info(handle_, &info));
        for (auto &l : lights) if (l.type == Light::DISTANT) l.sceneRadius = info.scene_radius;
    }
    // Scene::rebuildBVH for the edit loop without the host: the edited transforms go to the device, where the reference's
    // binned-SAH tree (bvh.cpp:9-133) is built anew, node for node and primitive for primitive, with everything derived from it
    // (jtx_mi_scene_rebuild).  Single-device scenes only.
    // the second buffer set of rebuildBVHOnDevice, allocated ahead of the first edit (jtx_mi_scene_reserve_rebuild); optional
    void reserveRebuild() { if (handle_) check(jtx_mi_scene_reserve_rebuild(handle_)); }
    // ... and given back when the editing is over (jtx_mi_scene_release_rebuild: about the geometry's device memory once more)
    void releaseRebuild() { if (handle_) check(jtx_mi_scene_release_rebuild(handle_)); }
    // the frame slots' working memory (per-path radiance records of the renders so far: jtx_mi_scene_info::frame_slot_bytes) back to the device
    void releaseFrames() { if (handle_) check(jtx_mi_scene_release_frames(handle_)); }
    void rebuildBVHOnDevice(int maxPrimsInNode = 1) {
        if (!handle_) { buildBVH(maxPrimsInNode); return; }
        for (size_t i = 0; i < meshes.size(); ++i) check(jtx_mi_scene_set_transform(handle_, (int) i, &meshes[i].transform.m[0][0]));
        check(jtx_mi_scene_rebuild(handle_, maxPrimsInNode));
        jtx_mi_scene_info info; check(jtx_mi_scene_get_info(handle_, &info));
        for (auto &l : lights) if (l.type == Light::DISTANT) l.sceneRadius = info.scene_radius;
    }
    void destroy() { destroyBVH(); }                                  // mesh arrays stay with the caller
    AABB bounds() const {                                             // scene.hpp:71-74: the root node's box (after a refit: the refitted one)
        AABB b;
        if (!handle_) return b;
        jtx_mi_scene_info i; check(jtx_mi_scene_get_info(handle_, &i));
        if (i.num_nodes <= 0) return b;
        std::vector<jtx_mi_bvh_node> nodes((size_t) i.num_nodes);
        check(jtx_mi_scene_get_bvh(handle_, nodes.data(), nullptr));
        b.pmin = Vec3(nodes[0].pmin[0], nodes[0].pmin[1], nodes[0].pmin[2]);
        b.pmax = Vec3(nodes[0].pmax[0], nodes[0].pmax[1], nodes[0].pmax[2]);
        return b;
    }
    float getSceneRadius() const { if (!handle_) return 0; jtx_mi_scene_info i; check(jtx_mi_scene_get_info(handle_, &i)); return i.scene_radius; }

    // Scene::closestHit / anyHit (scene.cpp:10-94): single-ray forms for API compatibility (one GPU batch of 1)
    bool closestHit(const Ray &r, Interval t, SurfaceIntersection &rec) const {
        int hit = 0, prim = -1; float tt, b1, b2, p[3], n[3], uv[2];
        check(jtx_mi_closest_hit_batch(need(), 1, &r.origin.x, &r.dir.x, t.min, t.max, &hit, &tt, &prim, &b1, &b2, p, n, uv));
        if (!hit) return false;
        rec.t = tt; rec.point = Vec3(p[0], p[1], p[2]); rec.normal = Vec3(n[0], n[1], n[2]); rec.uv = Vec2f{uv[0], uv[1]};
        rec.frontFace = (r.dir.x * n[0] + r.dir.y * n[1] + r.dir.z * n[2]) < 0;
        return true;
    }
    bool anyHit(const Ray &r, Interval t) const {
        int hit = 0;
        check(jtx_mi_any_hit_batch(need(), 1, &r.origin.x, &r.dir.x, &t.min, &t.max, &hit));
        return hit != 0;
    }
    jtx_mi_scene *handle() const { return need(); }
private:
    jtx_mi_scene *need() const { if (!handle_) throw std::runtime_error("Scene::buildBVH() has not been called"); return handle_; }
    jtx_mi_scene *handle_ = nullptr;
    jtx_mi_multi *multi_ = nullptr;
    std::vector<int> devices_;
};

struct RGB { unsigned char R, G, B; };
// Film storage the cameras page-lock (jtx_mi_pin_host): whole pages of its own.  Page-locking works on pages; a plain std::vector's
// block shares its first and last page with whatever else the allocator put there.
template <class T> struct PageAllocator {
    typedef T value_type;
    PageAllocator() = default;
    template <class U> PageAllocator(const PageAllocator<U> &) {}
    T *allocate(std::size_t n) {
        void *p = nullptr;
        const std::size_t bytes = (n * sizeof(T) + 4095) / 4096 * 4096;
        if (posix_memalign(&p, 4096, bytes ? bytes : 4096) != 0) throw std::bad_alloc();
        return (T *) p;
    }
    void deallocate(T *p, std::size_t) { std::free(p); }
    template <class U> bool operator==(const PageAllocator<U> &) const { return true; }
    template <class U> bool operator!=(const PageAllocator<U> &) const { return false; }
};

class RGB8Image {                                                     // image.hpp:18-58
public:
    int w_ = 0, h_ = 0;
    RGB8Image() = default; RGB8Image(int w, int h) : w_(w), h_(h), buffer(w * h) {}
    void resize(int w, int h) { w_ = w; h_ = h; buffer.resize(w * h); }
    void clear() { std::fill(buffer.begin(), buffer.end(), RGB{0, 0, 0}); }
    const RGB *data() const { return buffer.data(); }
    RGB *data() { return buffer.data(); }
    // RGB8Image::save (image.cpp:11-25: rows flipped, stbi_write_png): a PNG when the path ends in .png -- 8-bit RGB, filter 0,
    // stored deflate blocks: the same pixels as the reference's file, not the same bytes -- else a binary PPM
    void save(const char *path) const {
        FILE *f = std::fopen(path, "wb"); if (!f) return;
        const std::string p(path);
        const bool png = p.size() > 4 && (p.substr(p.size() - 4) == ".png" || p.substr(p.size() - 4) == ".PNG");
        if (!png) {
            std::fprintf(f, "P6\n%d %d\n255\n", w_, h_);
            for (int r = h_ - 1; r >= 0; --r) std::fwrite(&buffer[(size_t) r * w_], 3, w_, f);
            std::fclose(f); return;
        }
        auto crc = [](const std::vector<unsigned char> &d, size_t from) { unsigned c = 0xffffffffu; for (size_t i = from; i < d.size(); ++i) { c ^= d[i]; for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u))); } return ~c; };
        auto be32 = [](std::vector<unsigned char> &d, unsigned v) { d.push_back(v >> 24); d.push_back(v >> 16); d.push_back(v >> 8); d.push_back(v); };
        auto chunk = [&](const char *tag, const std::vector<unsigned char> &body) {
            std::vector<unsigned char> c; be32(c, (unsigned) body.size()); c.insert(c.end(), tag, tag + 4); c.insert(c.end(), body.begin(), body.end());
            be32(c, crc(c, 4)); std::fwrite(c.data(), 1, c.size(), f);
        };
        std::fwrite("\x89PNG\r\n\x1a\n", 1, 8, f);
        std::vector<unsigned char> ihdr; be32(ihdr, (unsigned) w_); be32(ihdr, (unsigned) h_); ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
        chunk("IHDR", ihdr);
        std::vector<unsigned char> raw; raw.reserve((size_t) h_ * (3 * w_ + 1));
        for (int r = h_ - 1; r >= 0; --r) { raw.push_back(0); const unsigned char *row = &buffer[(size_t) r * w_].R; raw.insert(raw.end(), row, row + 3 * (size_t) w_); }
        std::vector<unsigned char> z; z.push_back(0x78); z.push_back(0x01);
        unsigned a = 1, b = 0;
        for (size_t at = 0; at < raw.size() || at == 0; at += 65535) {
            const size_t n = raw.size() - at < 65535 ? raw.size() - at : 65535;
            z.push_back(at + n >= raw.size() ? 1 : 0); z.push_back(n & 255); z.push_back(n >> 8); z.push_back(~n & 255); z.push_back((~n >> 8) & 255);
            z.insert(z.end(), raw.begin() + at, raw.begin() + at + n);
            for (size_t i = at; i < at + n; ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
            if (raw.empty()) break;
        }
        be32(z, (b << 16) | a);
        chunk("IDAT", z); chunk("IEND", {});
        std::fclose(f);
    }
private:
    std::vector<RGB, PageAllocator<RGB>> buffer;
};
class AccumulationBuffer {                                            // image.hpp:64-92
public:
    int w_ = 0, h_ = 0;
    AccumulationBuffer() = default; AccumulationBuffer(int w, int h) : w_(w), h_(h), buffer_(w * h) {}
    void resize(int w, int h) { w_ = w; h_ = h; buffer_.resize(w * h); }
    void clear() { std::fill(buffer_.begin(), buffer_.end(), Vec3()); }
    const Vec3 *data() const { return buffer_.data(); }
    Vec3 *data() { return buffer_.data(); }
private:
    std::vector<Vec3, PageAllocator<Vec3>> buffer_;
};

class Camera {                                                        // camera.hpp:20-140
public:
    int width_, height_; float aspectRatio_;
    int xPixelSamples_, yPixelSamples_, maxDepth_;
    CameraProperties properties_;
    RGB8Image img_;
    std::atomic<int> currentSample_{0};
    Camera(int width, int height, CameraProperties cp, int xs, int ys, int maxDepth, int threadCount = 4)
        : width_(width), height_(height), aspectRatio_((float) width / (float) height), xPixelSamples_(xs), yPixelSamples_(ys),
          maxDepth_(maxDepth), properties_(cp), img_(width, height), acc_(width, height), threadCount_(threadCount) {}
    void save(const char *path) const { img_.save(path); }
    ~Camera() { unpin(); }
    void resize(int w, int h) { unpin(); width_ = w; height_ = h; aspectRatio_ = (float) w / (float) h; img_.clear(); img_.resize(w, h); acc_.clear(); acc_.resize(w, h); }
    void clear() { img_.clear(); }
    void terminateRender() { stopRender_ = true; }
    int getSpp() const { return xPixelSamples_ * yPixelSamples_; }
    int getThreadCount() const { return threadCount_; }
    const AccumulationBuffer &accumulation() const { return acc_; }
protected:
    AccumulationBuffer acc_;
    int threadCount_;
    std::atomic<bool> stopRender_{false};
    // img_ / acc_ live as long as the camera: page-lock them once so that jtx_mi_render DMA-writes them directly
    // (a refusal only costs the library's staging copy)
    bool pinnedImg_ = false, pinnedAcc_ = false;
    void pin() {
        if (!pinnedImg_) pinnedImg_ = jtx_mi_pin_host(&img_.data()[0].R, (uint64_t) width_ * height_ * 3) == 0;
        if (!pinnedAcc_) pinnedAcc_ = jtx_mi_pin_host(&acc_.data()[0].x, (uint64_t) width_ * height_ * 3 * sizeof(float)) == 0;
    }
    void unpin() {
        if (pinnedImg_) jtx_mi_unpin_host(&img_.data()[0].R);
        if (pinnedAcc_) jtx_mi_unpin_host(&acc_.data()[0].x);
        pinnedImg_ = pinnedAcc_ = false;
    }
    jtx_mi_camera_desc desc() const {
        jtx_mi_camera_desc c{};
        for (int k = 0; k < 3; ++k) { c.center[k] = properties_.center[k]; c.target[k] = properties_.target[k]; c.up[k] = properties_.up[k]; }
        c.yfov = properties_.yfov; c.defocus_angle = properties_.defocusAngle; c.focus_distance = properties_.focusDistance;
        c.width = width_; c.height = height_; c.x_pixel_samples = xPixelSamples_; c.y_pixel_samples = yPixelSamples_; c.max_depth = maxDepth_;
        return c;
    }
};

class StaticCamera : public Camera {                                  // camera.hpp:179-188
public:
    int samplesPerPass_ = 1;
    using Camera::Camera;
    // StaticCamera::render(const Scene&) camera.cpp:45-128: blocking; img_ and currentSample_ advance per pass so a UI
    // thread can keep showing the progressive image (display.cpp:702-703); terminateRender() stops after the pass.
    void render(const Scene &scene) {
        stopRender_ = false; currentSample_.store(0); acc_.clear(); pin();
        jtx_mi_render_opts o{}; o.samples_per_tick = samplesPerPass_ > 0 ? samplesPerPass_ : 1;
        jtx_mi_camera_desc c = desc();
        int rc;
        if (jtx_mi_multi *mh = scene.multiHandle()) {                    // Scene::useDevices: the frame is sharded over the GPUs
            activeMulti_ = mh;
            rc = jtx_mi_multi_render(mh, &c, &o, &acc_.data()[0].x, &img_.data()[0].R, &StaticCamera::tick, this);
            activeMulti_ = nullptr;
        } else {
            active_ = scene.handle();
            rc = jtx_mi_render(scene.handle(), &c, &o, &acc_.data()[0].x, &img_.data()[0].R, &StaticCamera::tick, this);
            active_ = nullptr;
        }
        if (rc != JTX_MI_CANCELLED) check(rc);
    }
    // Camera::terminateRender (camera.hpp:77): also reaches the pass in flight
    void terminateRender() {
        stopRender_ = true;
        if (jtx_mi_scene *h = active_) jtx_mi_cancel(h);
        if (jtx_mi_multi *h = activeMulti_) jtx_mi_multi_cancel(h);
    }
    // one-shot variant without per-pass host copies (what a non-interactive caller wants)
    void renderFinal(const Scene &scene) {
        stopRender_ = false; acc_.clear(); pin();
        jtx_mi_camera_desc c = desc();
        check(jtx_mi_render(scene.handle(), &c, nullptr, &acc_.data()[0].x, &img_.data()[0].R, nullptr, nullptr));
        currentSample_.store(getSpp());
    }
private:
    std::atomic<jtx_mi_scene *> active_{nullptr};
    std::atomic<jtx_mi_multi *> activeMulti_{nullptr};
    static int tick(int32_t cur, int32_t, void *user) { auto *self = (StaticCamera *) user; self->currentSample_.store(cur); return self->stopRender_ ? 1 : 0; }
};

// DynamicCamera (camera.hpp:198-256, camera.cpp:130-255): the interactive, restartable progressive render the UI
// drives.  render(scene) returns at once; a worker accumulates samplesPerPass strata per pass into acc_ / img_ and
// advances currentSample_ after every pass (the reference's end-of-pass barrier, camera.cpp:141-147); calling
// render() again (camera moved, scene edited) abandons the frame in flight after its current pass and starts over.
// The reference spreads a pass over threadCount CPU workers; here a pass is one launch on the GPU, so one host
// thread feeds it.
class DynamicCamera : public Camera {
public:
    DynamicCamera(int width, int height, CameraProperties cp, int xs, int ys, int maxDepth, int samplesPerPass = 1, int threadCount = 4)
        : Camera(width, height, cp, xs, ys, maxDepth, threadCount), samplesPerPass_(samplesPerPass > 0 ? samplesPerPass : 1) { startThreads(); }
    ~DynamicCamera() { stopThreads(); }
    void resize(int w, int h) {                                        // camera.cpp:191-194
        std::unique_lock<std::mutex> lk(mu_);
        pending_ = false; ++generation_; abandon(); idle_.wait(lk, [&] { return !busy_; });   // stop the pass in flight before the buffers move
        Camera::resize(w, h); scene_ = nullptr; currentSample_.store(0);
    }
    void render(const Scene &scene) {                                  // camera.cpp:196-211
        std::unique_lock<std::mutex> lk(mu_);
        pending_ = false; ++generation_; abandon(); idle_.wait(lk, [&] { return !busy_; });
        scene_ = &scene; acc_.clear(); img_.clear(); currentSample_.store(0); error_.clear();
        pending_ = true;
        lk.unlock(); wake_.notify_all();
    }
    void stopRender() { stopThreads(); }                               // camera.hpp:227
    bool finished() const { return currentSample_.load() >= getSpp(); }
    // not in the reference (its UI polls currentSample_): block until the frame is complete or failed
    void wait() { std::unique_lock<std::mutex> lk(mu_); idle_.wait(lk, [&] { return !busy_ && !pending_ && (scene_ == nullptr || finished() || !error_.empty()); }); if (!error_.empty()) throw std::runtime_error(error_); }
private:
    int samplesPerPass_;
    const Scene *scene_ = nullptr;
    std::thread thread_;
    std::mutex mu_;
    std::condition_variable wake_, idle_;
    unsigned long generation_ = 0, running_ = 0;
    bool busy_ = false, pending_ = false, stopThreads_ = false;
    std::string error_;
    void abandon() { if (busy_ && scene_) jtx_mi_cancel(scene_->handle()); }   // caller holds mu_: reach the pass in flight
    void startThreads() { stopThreads_ = false; thread_ = std::thread(&DynamicCamera::workerThread, this); }
    void stopThreads() {                                               // camera.cpp:178-189
        { std::unique_lock<std::mutex> lk(mu_); stopThreads_ = true; ++generation_; abandon(); }
        wake_.notify_all();
        if (thread_.joinable()) thread_.join();
    }
    static int tick(int32_t, int32_t, void *user) {                    // end of a pass: camera.cpp:141-147
        auto *self = (DynamicCamera *) user;
        self->currentSample_.fetch_add(self->samplesPerPass_);
        std::unique_lock<std::mutex> lk(self->mu_);
        return (self->generation_ != self->running_ || self->stopThreads_) ? 1 : 0;
    }
    void workerThread() {                                              // camera.cpp:213-255
        std::unique_lock<std::mutex> lk(mu_);
        while (true) {
            wake_.wait(lk, [&] { return stopThreads_ || pending_; });
            if (stopThreads_) break;
            pending_ = false; running_ = generation_; busy_ = true;
            const Scene *scene = scene_;
            jtx_mi_camera_desc c = desc();
            jtx_mi_render_opts o{}; o.samples_per_tick = samplesPerPass_;
            lk.unlock();
            const int rc = jtx_mi_render(scene->handle(), &c, &o, &acc_.data()[0].x, &img_.data()[0].R, &DynamicCamera::tick, this);
            lk.lock();
            if (rc && rc != JTX_MI_CANCELLED) error_ = jtx_mi_last_error();
            busy_ = false;
            idle_.notify_all();
        }
    }
};

// sampleBxdf / evalBxdf / pdfBxdf (bxdf.hpp:131-133), single-sample forms
inline int materialIndex(const Scene &s, const Material *m) { return (int) (m - s.materials.data()); }
inline bool sampleBxdf(const Scene &scene, const SurfaceIntersection &rec, const Vec3 &w_o, float uc, const Vec2f &u, BSDFSample &s) {
    int ok = 0; float f[3], wi[3], pdf;
    check(jtx_mi_bxdf_sample_batch(scene.handle(), materialIndex(scene, rec.material), 1, &rec.normal.x, &rec.uv.x, &w_o.x, &uc, &u.x, &ok, f, wi, &pdf));
    s.fSample = Vec3(f[0], f[1], f[2]); s.w_i = Vec3(wi[0], wi[1], wi[2]); s.pdf = pdf;
    return ok != 0;
}
inline Vec3 evalBxdf(const Scene &scene, const Material *mat, const SurfaceIntersection &rec, const Vec3 &w_o, const Vec3 &w_i) {
    float f[3]; check(jtx_mi_bxdf_eval_batch(scene.handle(), materialIndex(scene, mat), 1, &rec.normal.x, &rec.uv.x, &w_o.x, &w_i.x, f));
    return Vec3(f[0], f[1], f[2]);
}
inline float pdfBxdf(const Scene &scene, const Material *mat, const SurfaceIntersection &rec, const Vec3 &w_o, const Vec3 &w_i) {
    float p; check(jtx_mi_bxdf_pdf_batch(scene.handle(), materialIndex(scene, mat), 1, &rec.normal.x, &rec.uv.x, &w_o.x, &w_i.x, &p));
    return p;
}

} // namespace jtxmi
