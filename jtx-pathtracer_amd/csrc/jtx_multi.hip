// jtx_multi.hip -- one host process, N devices: the multi-GPU frame behind the C-ABI (jtx_mi_multi_*).
//
// The reference's caller is C++ (Display::renderScene on a detached thread, display.cpp:898-910); this gives it the
// 8 GPUs of a node without Python or torchrun.  Sharding is SURVEY 8e's: shard r owns the 32x32 tiles k with
// k % N == r (camera.cpp:55-64's tiles, interleaved), renders ALL strata of its own pixels in the reference's sample
// order -- per-pixel float sums bit-identical to one GPU -- and the scene is replicated.  Per pass:
//     every device: its shard (jtx_mi_render_device on its own stream), pack the own pixels into a compact slab (12 B
//                   accumulation + 3 B RGB8 per pixel, slot order) and push it to device 0 over xGMI (hipMemcpyPeerAsync
//                   on the SOURCE's stream: the N-1 slabs travel on N-1 different point-to-point links at once)
//     device 0    : when every shard has arrived and none was cancelled, scatter the slabs into the frame, copy to host
//                   (all-or-nothing per pass: an abandoned pass leaves no trace in the frame)
// One exchange per pass, no collective library needed for a gather onto one device; the torchrun path of bench.py
// (jtx_pathtracer_amd.distributed, RCCL) does the same exchange between processes.
// A device may be listed more than once (shards then share it): that is how the whole path runs on a one-GPU box.
#include "jtx_host.hpp"
#include "jtx_tiles.hpp"
#include "jtx_progressive.hpp"
#include <chrono>
#include <thread>
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

extern "C" const char *jtx_mi_last_error(void);
int jtx_capi_fail(const std::string &msg);           // jtx_capi.hip: sets the thread's error text, returns 1
void jtx_capi_d2h(void *dst, const void *src, size_t bytes, hipStream_t st);   // jtx_capi.hip: device -> caller memory (staged when pageable), complete on return; throws

namespace {

#define MHIPCHK(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define MCHK(expr) do { if ((expr) != 0) throw std::runtime_error(jtx_mi_last_error()); } while (0)

// own pixels of shard (rank, world) : frame -> slab, slot order (12 B + 3 B per slot; padding slots are skipped)
__global__ void __launch_bounds__(256) k_pack_shard(const float *acc, const unsigned char *img, float *slab_acc, unsigned char *slab_img,
                                                    int nslots, int rank, int world, int width, int height) {
    const int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= nslots) return;
    int row, col;
    if (!jtx::slotToPixel(slot, rank, world, width, height, row, col)) return;
    const size_t pix = (size_t) row * width + col;
    slab_acc[3 * (size_t) slot] = acc[3 * pix]; slab_acc[3 * (size_t) slot + 1] = acc[3 * pix + 1]; slab_acc[3 * (size_t) slot + 2] = acc[3 * pix + 2];
    if (img) { slab_img[3 * (size_t) slot] = img[3 * pix]; slab_img[3 * (size_t) slot + 1] = img[3 * pix + 1]; slab_img[3 * (size_t) slot + 2] = img[3 * pix + 2]; }
}
// slab of shard (rank, world) -> frame
__global__ void __launch_bounds__(256) k_scatter_shard(const float *slab_acc, const unsigned char *slab_img, float *acc, unsigned char *img,
                                                       int nslots, int rank, int world, int width, int height) {
    const int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= nslots) return;
    int row, col;
    if (!jtx::slotToPixel(slot, rank, world, width, height, row, col)) return;
    const size_t pix = (size_t) row * width + col;
    if (acc) { acc[3 * pix] = slab_acc[3 * (size_t) slot]; acc[3 * pix + 1] = slab_acc[3 * (size_t) slot + 1]; acc[3 * pix + 2] = slab_acc[3 * (size_t) slot + 2]; }
    if (img) { img[3 * pix] = slab_img[3 * (size_t) slot]; img[3 * pix + 1] = slab_img[3 * (size_t) slot + 1]; img[3 * pix + 2] = slab_img[3 * (size_t) slot + 2]; }
}

struct Shard {
    int device = 0;
    jtx_mi_scene *scene = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t xstream = nullptr;                                 // progressive renders: packs and pushes previews while `stream` carries the launch
    float *acc = nullptr; unsigned char *img = nullptr;            // full-size film of this shard (own pixels, zero elsewhere)
    float *slab_acc = nullptr; unsigned char *slab_img = nullptr;  // on this device
    float *recv_acc = nullptr; unsigned char *recv_img = nullptr;  // on device 0
    size_t film_pixels = 0, slab_slots = 0;
    // peer: this shard's device writes device 0's memory directly (the same device, or peer access granted: hipMemcpyPeerAsync over xGMI).
    // Otherwise the slab goes through page-locked host memory of the library's own: device -> stage on this shard's stream, stage -> device 0
    // on device 0's stream behind an event (a platform that refuses peer access -- IOMMU settings, a PCIe-only pair -- still renders).
    bool peer = true;
    float *stage_acc = nullptr; unsigned char *stage_img = nullptr; size_t stage_slots = 0;
    hipEvent_t staged = nullptr;
};

struct SetDev { int prev = -1; explicit SetDev(int d) { (void) hipGetDevice(&prev); MHIPCHK(hipSetDevice(d)); } ~SetDev() { if (prev >= 0) (void) hipSetDevice(prev); } };

} // namespace

struct jtx_mi_multi {
    std::vector<Shard> shards;
    float *frame_acc = nullptr; unsigned char *frame_img = nullptr;   // the assembled frame, on device 0
    size_t frame_pixels = 0;
    int last_completed = 0;
    float last_ms[64] = {};
};

namespace {

void releaseShard(Shard &s, int rootDevice) {
    if (hipSetDevice(s.device) == hipSuccess) {
        if (s.stream) (void) hipStreamSynchronize(s.stream);
        if (s.acc) (void) hipFree(s.acc);
        if (s.img) (void) hipFree(s.img);
        if (s.slab_acc) (void) hipFree(s.slab_acc);
        if (s.slab_img) (void) hipFree(s.slab_img);
        if (s.scene) jtx_mi_scene_destroy(s.scene);
        if (s.stream) (void) hipStreamDestroy(s.stream);
        if (s.xstream) { (void) hipStreamSynchronize(s.xstream); (void) hipStreamDestroy(s.xstream); }
        if (s.staged) (void) hipEventDestroy(s.staged);
        if (s.stage_acc) (void) hipHostFree(s.stage_acc);
        if (s.stage_img) (void) hipHostFree(s.stage_img);
    }
    if (hipSetDevice(rootDevice) == hipSuccess) {
        if (s.recv_acc) (void) hipFree(s.recv_acc);
        if (s.recv_img) (void) hipFree(s.recv_img);
    }
    s = Shard{};
}

void ensureBuffers(jtx_mi_multi &m, int width, int height) {
    const int n = (int) m.shards.size();
    const size_t npix = (size_t) width * height;
    for (int r = 0; r < n; ++r) {
        Shard &s = m.shards[r];
        const size_t slots = (size_t) jtx::ownedTiles(width, height, r, n) * 1024;
        if (s.film_pixels == npix && s.slab_slots == slots) continue;
        {
            SetDev sd(s.device);
            if (s.acc) (void) hipFree(s.acc); if (s.img) (void) hipFree(s.img);
            if (s.slab_acc) (void) hipFree(s.slab_acc); if (s.slab_img) (void) hipFree(s.slab_img);
            s.acc = nullptr; s.img = nullptr; s.slab_acc = nullptr; s.slab_img = nullptr;
            MHIPCHK(hipMalloc((void **) &s.acc, npix * 3 * sizeof(float)));
            MHIPCHK(hipMalloc((void **) &s.img, npix * 3));
            if (slots) { MHIPCHK(hipMalloc((void **) &s.slab_acc, slots * 3 * sizeof(float))); MHIPCHK(hipMalloc((void **) &s.slab_img, slots * 3)); }
        }
        if (r > 0) {
            SetDev sd(m.shards[0].device);
            if (s.recv_acc) (void) hipFree(s.recv_acc); if (s.recv_img) (void) hipFree(s.recv_img);
            s.recv_acc = nullptr; s.recv_img = nullptr;
            if (slots) { MHIPCHK(hipMalloc((void **) &s.recv_acc, slots * 3 * sizeof(float))); MHIPCHK(hipMalloc((void **) &s.recv_img, slots * 3)); }
        }
        if (r > 0 && !s.peer && s.stage_slots != slots) {
            SetDev sd(s.device);
            if (s.stage_acc) (void) hipHostFree(s.stage_acc); if (s.stage_img) (void) hipHostFree(s.stage_img);
            s.stage_acc = nullptr; s.stage_img = nullptr; s.stage_slots = 0;
            if (slots) {
                MHIPCHK(hipHostMalloc((void **) &s.stage_acc, slots * 3 * sizeof(float), hipHostMallocPortable));
                MHIPCHK(hipHostMalloc((void **) &s.stage_img, slots * 3, hipHostMallocPortable));
            }
            if (!s.staged) MHIPCHK(hipEventCreateWithFlags(&s.staged, hipEventDisableTiming));
            s.stage_slots = slots;
        }
        s.film_pixels = npix; s.slab_slots = slots;
    }
    if (m.frame_pixels != npix) {
        SetDev sd(m.shards[0].device);
        if (m.frame_acc) (void) hipFree(m.frame_acc); if (m.frame_img) (void) hipFree(m.frame_img);
        m.frame_acc = nullptr; m.frame_img = nullptr;
        MHIPCHK(hipMalloc((void **) &m.frame_acc, npix * 3 * sizeof(float)));
        MHIPCHK(hipMalloc((void **) &m.frame_img, npix * 3));
        m.frame_pixels = npix;
    }
}

} // namespace

extern "C" {

int jtx_mi_multi_create(const jtx_mi_scene_desc *desc, const int32_t *devices, int32_t n_devices, jtx_mi_multi **out) {
    jtx_mi_multi *m = nullptr;
    int prev = -1; (void) hipGetDevice(&prev);
    try {
        if (!desc || !out || n_devices < 1 || n_devices > 64) throw std::runtime_error("jtx_mi_multi_create: need 1..64 devices");
        *out = nullptr;
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw std::runtime_error("no HIP device: the jtx_mi core has no CPU fallback");
        m = new jtx_mi_multi();
        m->shards.resize(n_devices);
        for (int r = 0; r < n_devices; ++r) {
            Shard &s = m->shards[r];
            s.device = devices ? devices[r] : r;
            if (s.device < 0 || s.device >= ndev) throw std::runtime_error("jtx_mi_multi_create: device index out of range");
            MHIPCHK(hipSetDevice(s.device));
            MCHK(jtx_mi_scene_create(desc, &s.scene));                   // the scene is replicated
            MHIPCHK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
            MHIPCHK(hipStreamCreateWithFlags(&s.xstream, hipStreamNonBlocking));
        }
        // peer access to device 0 for the slab pushes (a no-op between shards of one device; not fatal when the platform
        // refuses it: hipMemcpyPeerAsync then stages through the host)
        for (int r = 1; r < n_devices; ++r) {
            Shard &s = m->shards[r];
            s.peer = s.device == m->shards[0].device;
            if (!s.peer) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, s.device, m->shards[0].device) == hipSuccess && can) {
                    MHIPCHK(hipSetDevice(s.device));
                    const hipError_t e = hipDeviceEnablePeerAccess(m->shards[0].device, 0);
                    s.peer = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
                    if (!s.peer) (void) hipGetLastError();
                }
            }
#ifdef JTX_TEST_HOOKS       /* libjtx_mi_testhooks.so only: the refusal, on a box where no pair of devices refuses */
            if (getenv("JTX_TEST_REFUSE_PEER_ACCESS")) s.peer = false;
#endif
        }
        if (prev >= 0) (void) hipSetDevice(prev);
        *out = m;
        return 0;
    } catch (const std::exception &e) {
        const std::string msg = e.what();
        if (m) { for (auto &s : m->shards) releaseShard(s, m->shards[0].device); delete m; }
        if (prev >= 0) (void) hipSetDevice(prev);
        return jtx_capi_fail(msg);
    }
}

void jtx_mi_multi_destroy(jtx_mi_multi *m) {
    if (!m) return;
    int prev = -1; (void) hipGetDevice(&prev);
    const int root = m->shards.empty() ? 0 : m->shards[0].device;
    for (auto &s : m->shards) releaseShard(s, root);
    if (hipSetDevice(root) == hipSuccess) { if (m->frame_acc) (void) hipFree(m->frame_acc); if (m->frame_img) (void) hipFree(m->frame_img); }
    delete m;
    if (prev >= 0) (void) hipSetDevice(prev);
}

int jtx_mi_multi_cancel(jtx_mi_multi *m) {
    if (!m) return jtx_capi_fail("null handle");
    for (auto &s : m->shards) if (s.scene) jtx_mi_cancel(s.scene);
    return 0;
}

int jtx_mi_multi_shard_time(jtx_mi_multi *m, float *ms_per_shard, int32_t n) {
    if (!m || !ms_per_shard) return jtx_capi_fail("null argument");
    for (int r = 0; r < n && r < (int) m->shards.size(); ++r) ms_per_shard[r] = m->last_ms[r];
    return 0;
}

int jtx_mi_multi_last_completed_sample(const jtx_mi_multi *m, int32_t *out) {
    if (!m || !out) return jtx_capi_fail("null argument");
    *out = m->last_completed;
    return 0;
}

int jtx_mi_multi_render(jtx_mi_multi *m, const jtx_mi_camera_desc *cam, const jtx_mi_render_opts *opts, float *acc_rgb,
                        uint8_t *img_rgb, jtx_mi_progress_cb cb, void *user) {
    int prev = -1; (void) hipGetDevice(&prev);
    try {
        if (!m || !cam || !acc_rgb) throw std::runtime_error("null argument");
        if (cam->width <= 0 || cam->height <= 0 || cam->x_pixel_samples <= 0 || cam->y_pixel_samples <= 0)
            throw std::runtime_error("camera width/height/pixel samples must be > 0");
        jtx_mi_render_opts o{}; if (opts) o = *opts;
        if (o.count_rays) throw std::runtime_error("jtx_mi_multi_render: count_rays is a single-device diagnostic");
        const int n = (int) m->shards.size();
        const int W = cam->width, H = cam->height;
        const int spp = cam->x_pixel_samples * cam->y_pixel_samples;
        const int sb = o.sample_begin > 0 ? o.sample_begin : 0;
        const int se = (o.sample_end > 0 && o.sample_end < spp) ? o.sample_end : spp;
        if (sb >= se) throw std::runtime_error("empty sample range");
        if (sb > 0) throw std::runtime_error("jtx_mi_multi_render: resuming (sample_begin > 0) needs the shard films of the previous call: render the frame in one call");
        ensureBuffers(*m, W, H);
        const size_t npix = (size_t) W * H;
        const int tick = (cb && o.samples_per_tick > 0) ? o.samples_per_tick : (se - sb);
        Shard &root = m->shards[0];
        for (int r = 0; r < n; ++r) MCHK(jtx_mi_cancel_reset(m->shards[r].scene));     // stopRender_ = false (camera.cpp:48)
        { SetDev sd(root.device);
          MHIPCHK(hipMemsetAsync(m->frame_acc, 0, npix * 3 * sizeof(float), root.stream));
          MHIPCHK(hipMemsetAsync(m->frame_img, 0, npix * 3, root.stream)); }
        bool cancelled = false;
        int done = sb;
        bool progressive = cb && tick < se - sb;
        for (int r = 0; r < n && progressive; ++r) progressive = jtx_prog_usable(m->shards[r].scene, o);
        if (progressive) {
            // ---- PROGRESSIVE (round 6): every shard traces ALL passes in one launch (k_render_paths<.., PROG> + its resolver, jtx_progressive.hpp) instead
            // of 64 x {launch per shard, pack, push, wait for all, scatter}.  This thread watches the shards: the frame's currentSample_ is the
            // minimum over them; when it has advanced, the shards' previews are packed and pushed on their exchange streams (the launch
            // streams are busy), scattered on device 0 and copied to the host, and the callback runs once per pass.  The exact film crosses
            // once, at the end.  A stop leaves the shards at different passes (each ends at a pass boundary of its own): strata can be added to
            // a film, not taken out, so the shards that are behind render on to the furthest one -- the frame then holds exactly [0, n).
            std::vector<JtxProgRun> runs(n);
            { SetDev sd(root.device); MHIPCHK(hipStreamSynchronize(root.stream)); }   // the frame is cleared before anything is scattered into it on the exchange stream
            auto exchange = [&](bool withFilm) {
                for (int r = 0; r < n; ++r) {
                    Shard &s = m->shards[r];
                    if (!s.slab_slots) continue;
                    SetDev sd(s.device);
                    const int nslots = (int) s.slab_slots;
                    hipLaunchKernelGGL(k_pack_shard, dim3((nslots + 255) / 256), dim3(256), 0, s.xstream, s.acc, img_rgb ? s.img : nullptr, s.slab_acc,
                                       s.slab_img, nslots, r, n, W, H);
                    MHIPCHK(hipGetLastError());
                    if (r > 0 && s.peer) {
                        if (withFilm) MHIPCHK(hipMemcpyPeerAsync(s.recv_acc, root.device, s.slab_acc, s.device, s.slab_slots * 3 * sizeof(float), s.xstream));
                        if (img_rgb) MHIPCHK(hipMemcpyPeerAsync(s.recv_img, root.device, s.slab_img, s.device, s.slab_slots * 3, s.xstream));
                    } else if (r > 0) {
                        if (withFilm) MHIPCHK(hipMemcpyAsync(s.stage_acc, s.slab_acc, s.slab_slots * 3 * sizeof(float), hipMemcpyDeviceToHost, s.xstream));
                        if (img_rgb) MHIPCHK(hipMemcpyAsync(s.stage_img, s.slab_img, s.slab_slots * 3, hipMemcpyDeviceToHost, s.xstream));
                        MHIPCHK(hipStreamSynchronize(s.xstream));
                        SetDev sr(root.device);
                        if (withFilm) MHIPCHK(hipMemcpyAsync(s.recv_acc, s.stage_acc, s.slab_slots * 3 * sizeof(float), hipMemcpyHostToDevice, root.xstream));
                        if (img_rgb) MHIPCHK(hipMemcpyAsync(s.recv_img, s.stage_img, s.slab_slots * 3, hipMemcpyHostToDevice, root.xstream));
                    }
                }
                for (int r = 0; r < n; ++r) { SetDev sd(m->shards[r].device); MHIPCHK(hipStreamSynchronize(m->shards[r].xstream)); }
                SetDev sd(root.device);
                for (int r = 0; r < n; ++r) {
                    Shard &s = m->shards[r];
                    if (!s.slab_slots) continue;
                    const int nslots = (int) s.slab_slots;
                    hipLaunchKernelGGL(k_scatter_shard, dim3((nslots + 255) / 256), dim3(256), 0, root.xstream, r ? s.recv_acc : s.slab_acc,
                                       r ? s.recv_img : s.slab_img, withFilm ? m->frame_acc : nullptr, img_rgb ? m->frame_img : nullptr, nslots, r, n, W, H);
                    MHIPCHK(hipGetLastError());
                }
                if (img_rgb) jtx_capi_d2h(img_rgb, m->frame_img, 3 * npix, root.xstream);
                MHIPCHK(hipStreamSynchronize(root.xstream));
            };
            long span = se - sb;
            for (int r = 0; r < n; ++r) {
                jtx_mi_render_opts q = o; q.tile_rank = r; q.tile_world = n;
                const long sp = jtx_prog_span(m->shards[r].scene, *cam, q, tick);
                span = sp < span ? sp : span;
            }
            int reported = sb;
            bool stopAsked = false;
            for (int b0 = sb; b0 < se && !cancelled; ) {
                const int e0 = (long) b0 + span < se ? b0 + (int) span : se;
                for (int r = 0; r < n; ++r) {
                    Shard &s = m->shards[r];
                    if (!s.slab_slots) { runs[r] = JtxProgRun{}; runs[r].begin = b0; runs[r].end = e0; runs[r].nothing = true; continue; }
                    SetDev sd(s.device);
                    if (b0 == sb) {                          // pixels of other shards read as exactly 0 (the slabs carry own pixels only; the films stay clean)
                        MHIPCHK(hipMemsetAsync(s.acc, 0, npix * 3 * sizeof(float), s.stream));
                        MHIPCHK(hipMemsetAsync(s.img, 0, npix * 3, s.stream));
                    }
                    jtx_mi_render_opts q = o; q.tile_rank = r; q.tile_world = n; q.frame_slot = 0; q.sequence_end = 1;
                    jtx_prog_begin(s.scene, *cam, q, b0, e0, tick, s.acc, s.img, s.stream, 128, true, runs[r]);
                }
                bool gaveUp = false;
                unsigned idle = 0;
                while (true) {
                    bool finished = true;
                    for (int r = 0; r < n; ++r) if (!runs[r].nothing) finished = finished && jtx_prog_finished(m->shards[r].scene);     // (first: the words read below are then final)
                    int have = e0;
                    for (int r = 0; r < n; ++r) { const int d = jtx_prog_completed(m->shards[r].scene, runs[r], &gaveUp); have = d < have ? d : have; }
                    if (have > reported && !stopAsked) {
                        idle = 0;
                        if (img_rgb) exchange(false);        // the previews as they stand
                        while (reported < have && !stopAsked) {
                            reported = reported + tick < have ? reported + tick : have;
                            if (cb(reported, spp, user)) { stopAsked = true; for (int r = 0; r < n; ++r) jtx_mi_cancel(m->shards[r].scene); }
                        }
                    }
                    if (finished) break;
                    if (++idle > 64) std::this_thread::sleep_for(std::chrono::microseconds(idle > 2048 ? 200 : 20)); else std::this_thread::yield();   // (a long pass: the poll backs off)
                }
                for (int r = 0; r < n; ++r) { SetDev sd(m->shards[r].device); MHIPCHK(hipStreamSynchronize(m->shards[r].stream)); }
                if (gaveUp) throw std::runtime_error("progressive launch: a shard's resolver waited a minute for its path kernel and gave up");
                int lo = e0, hi = b0;
                std::vector<int> at(n, e0);
                for (int r = 0; r < n; ++r) {
                    if (runs[r].nothing) continue;
                    at[r] = jtx_prog_completed(m->shards[r].scene, runs[r], &gaveUp);
                    lo = at[r] < lo ? at[r] : lo; hi = at[r] > hi ? at[r] : hi;
                }
                if (lo < e0 || stopAsked) {
                    // stopped: the shards that are behind render on to the furthest one (batch launches on their resumed films; the stop word is
                    // cleared for them -- a stop that arrives again meanwhile is honoured by the loop: they are rendered on until they stand)
                    cancelled = true;
                    if (hi < b0) hi = b0;
                    for (int guard = 0; guard < 8; ++guard) {
                        bool all = true;
                        for (int r = 0; r < n; ++r) {
                            Shard &s = m->shards[r];
                            if (runs[r].nothing || at[r] >= hi) continue;
                            all = false;
                            SetDev sd(s.device);
                            MCHK(jtx_mi_cancel_reset(s.scene));
                            jtx_mi_render_opts q = o; q.tile_rank = r; q.tile_world = n; q.sample_begin = at[r]; q.sample_end = hi;
                            MCHK(jtx_mi_render_device(s.scene, cam, &q, s.acc, s.img, s.stream));
                        }
                        if (all) break;
                        for (int r = 0; r < n; ++r) {
                            Shard &s = m->shards[r];
                            if (runs[r].nothing || at[r] >= hi) continue;
                            SetDev sd(s.device);
                            MHIPCHK(hipStreamSynchronize(s.stream));
                            int32_t flag = 0; MCHK(jtx_mi_cancel_pending(s.scene, &flag));
                            if (flag != 2) at[r] = hi;       // (2: this catch-up launch was abandoned too -- once more)
                        }
                    }
                    for (int r = 0; r < n; ++r)
                        if (!runs[r].nothing && at[r] < hi)
                            throw std::runtime_error("jtx_mi_multi_render: stopped again during each of 8 attempts to bring the shards level; the frame is not consistent");
                    done = hi;
                    break;
                }
                done = e0;
                b0 = e0;
            }
            exchange(true);                                  // the exact film, once
            {
                SetDev sd(root.device);
                jtx_capi_d2h(acc_rgb, m->frame_acc, sizeof(float) * 3 * npix, root.xstream);
            }
            for (int r = 0; r < n && r < 64; ++r) { float ms = 0; int32_t nl = 0; (void) jtx_mi_kernel_time(m->shards[r].scene, &ms, &nl); m->last_ms[r] = ms; }
            m->last_completed = done;
            if (prev >= 0) (void) hipSetDevice(prev);
            return cancelled ? JTX_MI_CANCELLED : 0;
        }
        for (int b = sb; b < se && !cancelled; b += tick) {
            const int e = b + tick < se ? b + tick : se;
            // ---- every device: its shard of this pass, pack, push to device 0 ----
            for (int r = 0; r < n; ++r) {
                Shard &s = m->shards[r];
                SetDev sd(s.device);
                jtx_mi_render_opts q = o;
                q.sample_begin = b; q.sample_end = e; q.tile_rank = r; q.tile_world = n;
                if (!s.slab_slots) continue;                                            // more shards than 32x32 tiles: this one owns nothing
                MCHK(jtx_mi_render_device(s.scene, cam, &q, s.acc, s.img, s.stream));
                const int nslots = (int) s.slab_slots;
                hipLaunchKernelGGL(k_pack_shard, dim3((nslots + 255) / 256), dim3(256), 0, s.stream, s.acc, img_rgb ? s.img : nullptr, s.slab_acc,
                                   s.slab_img, nslots, r, n, W, H);
                MHIPCHK(hipGetLastError());
                if (r > 0 && s.peer) {
                    MHIPCHK(hipMemcpyPeerAsync(s.recv_acc, root.device, s.slab_acc, s.device, s.slab_slots * 3 * sizeof(float), s.stream));
                    if (img_rgb) MHIPCHK(hipMemcpyPeerAsync(s.recv_img, root.device, s.slab_img, s.device, s.slab_slots * 3, s.stream));
                } else if (r > 0) {                                                     // peer access refused: through the library's page-locked stage
                    MHIPCHK(hipMemcpyAsync(s.stage_acc, s.slab_acc, s.slab_slots * 3 * sizeof(float), hipMemcpyDeviceToHost, s.stream));
                    if (img_rgb) MHIPCHK(hipMemcpyAsync(s.stage_img, s.slab_img, s.slab_slots * 3, hipMemcpyDeviceToHost, s.stream));
                    MHIPCHK(hipEventRecord(s.staged, s.stream));
                    SetDev sr(root.device);
                    MHIPCHK(hipStreamWaitEvent(root.stream, s.staged, 0));
                    MHIPCHK(hipMemcpyAsync(s.recv_acc, s.stage_acc, s.slab_slots * 3 * sizeof(float), hipMemcpyHostToDevice, root.stream));
                    if (img_rgb) MHIPCHK(hipMemcpyAsync(s.recv_img, s.stage_img, s.slab_slots * 3, hipMemcpyHostToDevice, root.stream));
                }
            }
            // ---- all shards in; a pass that any shard abandoned is void everywhere ----
            int wasCancelled = 0;
            for (int r = 0; r < n; ++r) { SetDev sd(m->shards[r].device); MHIPCHK(hipStreamSynchronize(m->shards[r].stream)); }
            for (int r = 0; r < n; ++r) { int32_t flag = 0; MCHK(jtx_mi_cancel_pending(m->shards[r].scene, &flag)); wasCancelled |= flag; }   // pending on any shard: the pass is void everywhere
            if (wasCancelled) { cancelled = true; break; }
            // ---- device 0: scatter the slabs into the frame, preview to the host ----
            {
                SetDev sd(root.device);
                for (int r = 0; r < n; ++r) {
                    Shard &s = m->shards[r];
                    if (!s.slab_slots) continue;
                    const int nslots = (int) s.slab_slots;
                    hipLaunchKernelGGL(k_scatter_shard, dim3((nslots + 255) / 256), dim3(256), 0, root.stream, r ? s.recv_acc : s.slab_acc,
                                       r ? s.recv_img : s.slab_img, m->frame_acc, img_rgb ? m->frame_img : nullptr, nslots, r, n, W, H);
                    MHIPCHK(hipGetLastError());
                }
                if (img_rgb) jtx_capi_d2h(img_rgb, m->frame_img, 3 * npix, root.stream);
                MHIPCHK(hipStreamSynchronize(root.stream));
            }
            done = e;
            if (cb && cb(done, spp, user)) { cancelled = true; break; }
        }
        {
            SetDev sd(root.device);
            jtx_capi_d2h(acc_rgb, m->frame_acc, sizeof(float) * 3 * npix, root.stream);
        }
        for (int r = 0; r < n && r < 64; ++r) { float ms = 0; int32_t nl = 0; (void) jtx_mi_kernel_time(m->shards[r].scene, &ms, &nl); m->last_ms[r] = ms; }
        m->last_completed = done;
        if (prev >= 0) (void) hipSetDevice(prev);
        return cancelled ? JTX_MI_CANCELLED : 0;
    } catch (const std::exception &e) {
        const std::string msg = e.what();
        if (prev >= 0) (void) hipSetDevice(prev);
        return jtx_capi_fail(msg);
    }
}

} // extern "C"
