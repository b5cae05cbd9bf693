"""Python mirror of the reference's Scene / StaticCamera interface over the C-ABI (include/jtx_mi.h).

Names and argument meaning follow the reference (src/scene.hpp:24-91, src/camera.hpp:20-188):
`Scene.buildBVH / rebuildBVH / destroy / closestHit / anyHit / bounds / getSceneRadius`,
`StaticCamera(width, height, cameraProperties, xPixelSamples, yPixelSamples, maxDepth)`,
`StaticCamera.render(scene)`, members `img_`, `currentSample_`, `getSpp()`, `terminateRender()`.
Everything that computes runs in the HIP library; nothing here falls back to the CPU.
"""
import ctypes as C
import os

import numpy as np

from . import _capi as capi
from ._capi import JtxMiError, check


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Scene:
    """Scene (src/scene.hpp:24-91) backed by a device-resident jtx_mi_scene."""

    def __init__(self, data):
        self.data = data                    # scenes.SceneData
        self.name = data.name
        self._h = None
        self._lib = capi.load()

    # -- Scene::buildBVH(maxPrimsInNode) scene.cpp:96-135: host SAH build + device upload
    def buildBVH(self, maxPrimsInNode=1):
        if self._h is not None:
            return
        self.data.max_prims_in_node = maxPrimsInNode
        desc = self.data.to_desc()
        h = C.c_void_p()
        check(self._lib.jtx_mi_scene_create(C.byref(desc), C.byref(h)))
        self._h = h

    def destroyBVH(self):
        if self._h is not None:
            self._lib.jtx_mi_scene_destroy(self._h)
            self._h = None

    def rebuildBVH(self, maxPrimsInNode=1):
        self.destroyBVH()
        self.buildBVH(maxPrimsInNode)

    def destroy(self):
        self.destroyBVH()

    def __del__(self):
        try:
            self.destroyBVH()
        except Exception:
            pass

    @property
    def handle(self):
        if self._h is None:
            raise JtxMiError("Scene.buildBVH() has not been called")
        return self._h

    def info(self):
        i = capi.SceneInfo()
        check(self._lib.jtx_mi_scene_get_info(self.handle, C.byref(i)))
        return dict(num_nodes=i.num_nodes, num_prims=i.num_prims, max_depth=i.max_depth,
                    lds_resident=bool(i.lds_resident), scene_radius=float(i.scene_radius),
                    auto_integrator=int(i.auto_integrator), wide_depth=int(i.wide_depth), wide_bytes=int(i.wide_bytes),
                    device_bytes=int(i.device_bytes), refitted=bool(i.refitted), device_built=bool(i.device_built),
                    num_cus=int(i.num_cus), resident_workgroups=int(i.resident_workgroups), workgroup_size=int(i.workgroup_size),
                    wide_bytes64=int(i.wide_bytes64), rebuild_spare_bytes=int(i.rebuild_spare_bytes), frame_slot_bytes=int(i.frame_slot_bytes))

    def releaseFrames(self):
        """jtx_mi_scene_release_frames: the frame slots' working memory (info()["frame_slot_bytes"]) back to the device."""
        check(self._lib.jtx_mi_scene_release_frames(self.handle))

    # -- transform edits (display.cpp:545-588): new Mesh::transform per mesh, then a device refit (topology kept)
    def setTransform(self, mesh, transform):
        m = np.ascontiguousarray(transform, np.float32).reshape(16)
        check(self._lib.jtx_mi_scene_set_transform(self.handle, int(mesh), m.ctypes.data_as(C.POINTER(C.c_float))))
        self.data.meshes[mesh]["transform"] = m.reshape(4, 4).copy()

    def refit(self):
        check(self._lib.jtx_mi_scene_refit(self.handle))

    def rebuildBVHOnDevice(self, maxPrimsInNode=1):
        """Scene::rebuildBVH after setTransform edits without leaving the device (jtx_mi_scene_rebuild): the reference's
        binned-SAH tree for the edited geometry, node for node and primitive for primitive (`device_built` records where it was built)."""
        check(self._lib.jtx_mi_scene_rebuild(self.handle, int(maxPrimsInNode)))

    def reserveRebuild(self):
        """allocate the second buffer set and the builder's scratch of rebuildBVHOnDevice now (jtx_mi_scene_reserve_rebuild), so that
        the first edit of a session costs what every later one does"""
        check(self._lib.jtx_mi_scene_reserve_rebuild(self.handle))

    def releaseRebuild(self):
        """free the second buffer set, the builder's scratch and its landing buffers again (jtx_mi_scene_release_rebuild;
        info()["rebuild_spare_bytes"] says how much they hold)"""
        check(self._lib.jtx_mi_scene_release_rebuild(self.handle))

    def bvh(self):
        i = self.info()
        nodes = (capi.BvhNode * max(1, i["num_nodes"]))()
        refs = (capi.TriRef * max(1, i["num_prims"]))()
        check(self._lib.jtx_mi_scene_get_bvh(self.handle, nodes, refs))
        return nodes_to_numpy(nodes, i["num_nodes"]), refs_to_numpy(refs, i["num_prims"])

    def wide(self):
        """the scene's 8-ary node set as it stands on the device (uint32 [granules, 4]; layout of wide_build_host)"""
        ng = C.c_int64()
        check(self._lib.jtx_mi_scene_get_wide(self.handle, None, 0, C.byref(ng)))
        out = np.zeros((max(1, ng.value), 4), np.uint32)
        if ng.value:
            check(self._lib.jtx_mi_scene_get_wide(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint32)), ng.value, C.byref(ng)))
        return out[: ng.value]

    def bounds(self):
        """AABB of the root node (scene.hpp:71-74)."""
        n, _ = self.bvh()
        return (n["pmin"][0], n["pmax"][0]) if len(n) else (None, None)

    def getSceneRadius(self):
        return self.info()["scene_radius"]

    def numPrimitives(self):
        return self.data.num_triangles

    # -- Scene::closestHit scene.cpp:10-55, batched
    TRAVERSAL_BINARY, TRAVERSAL_PRODUCTION, TRAVERSAL_SOURCE = 0, 1, 2          # jtx_mi.h: JTX_MI_TRAVERSAL_*
    SOURCE_NAMES = {0: "binary", 1: "lds", 2: "wide", 3: "leaf"}

    def closestHit(self, o, d, tmin=0.001, tmax=float("inf"), traversal=0):
        """traversal: 0 the binary records (the reference's node visits), 1 the structure the TIMED launch of this scene walks,
        2 + s a named source; `self.last_source` = the source that ran (SOURCE_NAMES)"""
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = len(o)
        out = dict(hit=np.zeros(n, np.int32), t=np.zeros(n, np.float32), prim=np.zeros(n, np.int32),
                   b1=np.zeros(n, np.float32), b2=np.zeros(n, np.float32), point=np.zeros((n, 3), np.float32),
                   normal=np.zeros((n, 3), np.float32), uv=np.zeros((n, 2), np.float32))
        src = C.c_int32(-1)
        check(self._lib.jtx_mi_closest_hit_batch_via(self.handle, traversal, n, _fp(o), _fp(d), tmin, tmax, _ip(out["hit"]), _fp(out["t"]),
                                                     _ip(out["prim"]), _fp(out["b1"]), _fp(out["b2"]), _fp(out["point"]),
                                                     _fp(out["normal"]), _fp(out["uv"]), C.byref(src)))
        self.last_source = src.value
        return out

    # -- Scene::anyHit scene.cpp:57-94, batched
    def anyHit(self, o, d, tmin, tmax, traversal=0):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = len(o)
        tmin = np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, np.float32), (n,)))
        tmax = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, np.float32), (n,)))
        hit = np.zeros(n, np.int32)
        src = C.c_int32(-1)
        check(self._lib.jtx_mi_any_hit_batch_via(self.handle, traversal, n, _fp(o), _fp(d), _fp(tmin), _fp(tmax), _ip(hit), C.byref(src)))
        self.last_source = src.value
        return hit

    # -- sampleBxdf / evalBxdf / pdfBxdf bsdf/bxdf.cpp:9,79,130, batched per material
    def sampleBxdf(self, material, normal, wo, uc, u2, uv=None):
        normal = np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
        wo = np.ascontiguousarray(wo, np.float32).reshape(-1, 3)
        uc = np.ascontiguousarray(uc, np.float32).reshape(-1)
        u2 = np.ascontiguousarray(u2, np.float32).reshape(-1, 2)
        n = len(normal)
        uvp = _fp(np.ascontiguousarray(uv, np.float32)) if uv is not None else None
        ok = np.zeros(n, np.int32); f = np.zeros((n, 3), np.float32); wi = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32)
        check(self._lib.jtx_mi_bxdf_sample_batch(self.handle, material, n, _fp(normal), uvp, _fp(wo), _fp(uc), _fp(u2),
                                                 _ip(ok), _fp(f), _fp(wi), _fp(pdf)))
        return dict(ok=ok, f=f, wi=wi, pdf=pdf)

    def evalBxdf(self, material, normal, wo, wi, uv=None):
        normal = np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
        wo = np.ascontiguousarray(wo, np.float32).reshape(-1, 3)
        wi = np.ascontiguousarray(wi, np.float32).reshape(-1, 3)
        n = len(normal)
        uvp = _fp(np.ascontiguousarray(uv, np.float32)) if uv is not None else None
        f = np.zeros((n, 3), np.float32)
        check(self._lib.jtx_mi_bxdf_eval_batch(self.handle, material, n, _fp(normal), uvp, _fp(wo), _fp(wi), _fp(f)))
        return f

    def pdfBxdf(self, material, normal, wo, wi, uv=None):
        normal = np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
        wo = np.ascontiguousarray(wo, np.float32).reshape(-1, 3)
        wi = np.ascontiguousarray(wi, np.float32).reshape(-1, 3)
        n = len(normal)
        uvp = _fp(np.ascontiguousarray(uv, np.float32)) if uv is not None else None
        pdf = np.zeros(n, np.float32)
        check(self._lib.jtx_mi_bxdf_pdf_batch(self.handle, material, n, _fp(normal), uvp, _fp(wo), _fp(wi), _fp(pdf)))
        return pdf


NODE_DTYPE = np.dtype([("pmin", np.float32, 3), ("pmax", np.float32, 3), ("offset", np.int32),
                       ("num_prims", np.uint16), ("axis", np.uint8), ("pad", np.uint8)])
REF_DTYPE = np.dtype([("index", np.int32), ("mesh_index", np.int32)])


def nodes_to_numpy(nodes, n):
    return np.frombuffer(bytes(memoryview(nodes))[: n * 32], NODE_DTYPE).copy()


def refs_to_numpy(refs, n):
    return np.frombuffer(bytes(memoryview(refs))[: n * 8], REF_DTYPE).copy()


def bvh_build_host(data):
    """Host-only Scene::buildBVH through the C-ABI (needs no GPU)."""
    lib = capi.load()
    desc = data.to_desc()
    n = max(1, data.num_triangles)
    nodes = (capi.BvhNode * (2 * n + 1))()
    refs = (capi.TriRef * n)()
    nn, md = C.c_int32(), C.c_int32()
    check(lib.jtx_mi_bvh_build(C.byref(desc), nodes, C.byref(nn), refs, C.byref(md)))
    return nodes_to_numpy(nodes, nn.value), refs_to_numpy(refs, data.num_triangles), md.value


def wide_build_host(nodes):
    """Host-only: the 8-ary quantised node set derived from flat BVH nodes (numpy NODE_DTYPE) ->
    (uint32 array [granules, 4], depth)."""
    lib = capi.load()
    nn = len(nodes)
    raw = (capi.BvhNode * max(1, nn)).from_buffer_copy(np.ascontiguousarray(nodes).tobytes() or bytes(32))
    ng, depth = C.c_int64(), C.c_int32()
    check(lib.jtx_mi_wide_build(raw, nn, None, 0, C.byref(ng), C.byref(depth)))
    out = np.zeros((max(1, ng.value), 4), np.uint32)
    check(lib.jtx_mi_wide_build(raw, nn, out.ctypes.data_as(C.POINTER(C.c_uint32)), ng.value, C.byref(ng), C.byref(depth)))
    return out[: ng.value], depth.value


class MultiScene:
    """N replicas of a scene on N devices behind one handle (jtx_mi_multi_*, csrc/jtx_multi.hip): what a C++ host gets
    from Scene::useDevices().  devices: list of device indices (a device may repeat: its shards share it)."""

    def __init__(self, data, devices, maxPrimsInNode=1):
        self._lib = capi.load()
        self.data = data
        data.max_prims_in_node = maxPrimsInNode
        desc = data.to_desc()
        devs = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        check(self._lib.jtx_mi_multi_create(C.byref(desc), devs, len(devices), C.byref(h)))
        self._h, self.n = h, len(devices)

    def render(self, cam, progress=None, samples_per_tick=0, max_record_mb=0):
        """jtx_mi_multi_render into cam.acc_ / cam.img_ (a StaticCamera); returns True when complete, False when cancelled."""
        o = capi.RenderOpts(); o.samples_per_tick = samples_per_tick; o.max_record_mb = max_record_mb
        cb = capi.PROGRESS_CB(lambda cur, tot, _u: 1 if (progress is not None and progress(cur, tot)) else 0)
        d = cam.desc()
        rc = self._lib.jtx_mi_multi_render(self._h, C.byref(d), C.byref(o), _fp(cam.acc_), cam.img_.ctypes.data_as(C.POINTER(C.c_uint8)), cb, None)
        if rc != capi.CANCELLED:
            check(rc)
        done = C.c_int32(0)
        check(self._lib.jtx_mi_multi_last_completed_sample(self._h, C.byref(done)))
        cam.currentSample_ = done.value
        return rc == 0

    def cancel(self):
        self._lib.jtx_mi_multi_cancel(self._h)

    def shard_ms(self):
        ms = (C.c_float * self.n)()
        check(self._lib.jtx_mi_multi_shard_time(self._h, ms, self.n))
        return [float(x) for x in ms]

    def destroy(self):
        if self._h is not None:
            self._lib.jtx_mi_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def _film_array(shape, dtype):
    """A zeroed array in an anonymous mapping of its own.  The cameras page-lock img_ / acc_ (jtx_mi_pin_host = hipHostRegister) so
    that the frame is DMA-written into them; page-locking works on whole pages, and an array from numpy's allocator shares its first
    and last page with whatever else malloc put there and hands them back to a heap that is trimmed and re-grown under the
    registration's feet.  A private mapping is page-aligned, shares nothing and stays mapped until the array is gone."""
    import mmap
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    m = mmap.mmap(-1, max(n, 1), flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)      # (private: the default is a SHARED mapping, i.e. shmem)
    return np.frombuffer(m, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


class StaticCamera:
    """StaticCamera (src/camera.hpp:179-188) -- one blocking render per call."""

    def __init__(self, width, height, cameraProperties, xPixelSamples, yPixelSamples, maxDepth, threadCount=0):
        self.width_, self.height_ = int(width), int(height)
        self.properties_ = dict(cameraProperties)
        self.xPixelSamples_, self.yPixelSamples_ = int(xPixelSamples), int(yPixelSamples)
        self.maxDepth_ = int(maxDepth)
        self.samplesPerPass_ = 1
        self.currentSample_ = 0
        self.img_ = _film_array((self.height_, self.width_, 3), np.uint8)      # RGB8Image, row 0 = bottom
        self.acc_ = _film_array((self.height_, self.width_, 3), np.float32)    # AccumulationBuffer
        self.stopRender_ = False
        self.counters = None
        self._lib = capi.load()

    def getSpp(self):
        return self.xPixelSamples_ * self.yPixelSamples_

    def terminateRender(self):
        """Camera::terminateRender (camera.hpp:77): also reaches the pass in flight (jtx_mi_cancel)."""
        self.stopRender_ = True
        sc = getattr(self, "_active_scene", None)
        if sc is not None and sc.handle:
            self._lib.jtx_mi_cancel(sc.handle)

    def resize(self, w, h):
        self._unpin()
        self.width_, self.height_ = int(w), int(h)
        self.img_ = _film_array((h, w, 3), np.uint8)
        self.acc_ = _film_array((h, w, 3), np.float32)

    # img_ / acc_ live as long as the camera (image.hpp:60-61,90-91): page-lock them once so that jtx_mi_render DMA-writes
    # them directly (jtx_mi_pin_host); a refusal only costs the library's staging copy
    def _pin(self):
        if getattr(self, "_pinned", None) is None:
            self._pinned = []
            if os.environ.get("JTX_PIN_CAMERA_BUFFERS", "1") == "0":
                return
            for a in (self.img_, self.acc_):
                if self._lib.jtx_mi_pin_host(a.ctypes.data_as(C.c_void_p), a.nbytes) == 0:
                    self._pinned.append(a)

    def _unpin(self):
        for a in getattr(self, "_pinned", None) or []:
            self._lib.jtx_mi_unpin_host(a.ctypes.data_as(C.c_void_p))
        self._pinned = None

    def __del__(self):
        try:
            self._unpin()
        except Exception:
            pass

    def clear(self):
        self.img_[...] = 0

    def desc(self):
        c = capi.CameraDesc()
        p = self.properties_
        c.center = capi.c_float3(*p["center"]); c.target = capi.c_float3(*p["target"]); c.up = capi.c_float3(*p["up"])
        c.yfov, c.defocus_angle, c.focus_distance = p["yfov"], p["defocus_angle"], p["focus_distance"]
        c.width, c.height = self.width_, self.height_
        c.x_pixel_samples, c.y_pixel_samples, c.max_depth = self.xPixelSamples_, self.yPixelSamples_, self.maxDepth_
        return c

    def render(self, scene, count_rays=False, progress=None, tile_rank=0, tile_world=1, sample_begin=0, sample_end=0,
               integrator=0, path_integrator=0, max_record_mb=0):
        """StaticCamera::render(const Scene&) (camera.cpp:45-128).  path_integrator: which Li (camera.cpp:104-106):
        0 integrateMIS, 1 integrate, 2 integrateBasic."""
        self.stopRender_ = False
        self.currentSample_ = 0
        o = capi.RenderOpts()
        o.count_rays = 1 if count_rays else 0
        o.tile_rank, o.tile_world = tile_rank, tile_world
        o.sample_begin, o.sample_end = sample_begin, sample_end
        o.integrator = integrator
        o.path_integrator = path_integrator
        o.max_record_mb = max_record_mb
        o.samples_per_tick = self.samplesPerPass_ if progress is not None else 0

        def _cb(cur, total, _user):
            self.currentSample_ = cur
            if progress is not None:
                progress(cur, total)
            return 1 if self.stopRender_ else 0

        cb = capi.PROGRESS_CB(_cb)
        cam = self.desc()
        self._pin()
        self._active_scene = scene
        try:
            rc = self._lib.jtx_mi_render(scene.handle, C.byref(cam), C.byref(o), _fp(self.acc_),
                                         self.img_.ctypes.data_as(C.POINTER(C.c_uint8)), cb, None)
        finally:
            self._active_scene = None
        if rc != capi.CANCELLED:
            check(rc)
        done = C.c_int32(0)
        check(self._lib.jtx_mi_last_completed_sample(scene.handle, C.byref(done)))
        self.currentSample_ = done.value
        if count_rays:
            c = capi.Counters()
            check(self._lib.jtx_mi_get_counters(scene.handle, C.byref(c)))
            self.counters = c.as_dict()
        return self.img_

    def save(self, path):
        """RGB8Image::save (image.cpp:11-25): rows flipped; a PNG when the path ends in .png (the same pixels as the
        reference's stbi_write_png file, not the same bytes), else a binary PPM."""
        rows = np.ascontiguousarray(self.img_[::-1])
        with open(path, "wb") as f:
            if str(path).lower().endswith(".png"):
                from . import gltf
                f.write(gltf.encode_png(rows.reshape(self.height_, self.width_, 3)))
            else:
                f.write(b"P6\n%d %d\n255\n" % (self.width_, self.height_))
                f.write(rows.tobytes())


class DynamicCamera(StaticCamera):
    """DynamicCamera (src/camera.hpp:198-256, camera.cpp:130-255): the UI's restartable progressive render.
    render(scene) returns at once; a worker thread accumulates samplesPerPass strata per pass into acc_ / img_
    and advances currentSample_ after each pass; render() again abandons the frame in flight after its
    current pass and starts over (camera moved / scene edited).  A pass is one GPU launch, so one host thread."""

    def __init__(self, width, height, cameraProperties, xPixelSamples, yPixelSamples, maxDepth, samplesPerPass=1, threadCount=4):
        super().__init__(width, height, cameraProperties, xPixelSamples, yPixelSamples, maxDepth, threadCount)
        import threading
        self.samplesPerPass_ = max(1, int(samplesPerPass))
        self._cv = threading.Condition()
        self._generation = self._running = 0
        self._busy = self._pending = self._stop = False
        self._scene = None
        self._error = None
        self._thread = threading.Thread(target=self._workerThread, daemon=True)
        self._thread.start()

    def _abandon(self):             # caller holds _cv: reach the pass in flight instead of waiting for its end
        if self._busy and self._scene is not None and self._scene.handle:
            self._lib.jtx_mi_cancel(self._scene.handle)

    def render(self, scene, **_unused):
        with self._cv:
            self._pending = False
            self._generation += 1
            self._abandon()
            self._cv.wait_for(lambda: not self._busy)
            self._scene = scene
            self.acc_[...] = 0; self.img_[...] = 0
            self.currentSample_ = 0
            self._error = None
            self._pending = True
            self._cv.notify_all()

    def resize(self, w, h):
        with self._cv:
            self._pending = False
            self._generation += 1
            self._abandon()
            self._cv.wait_for(lambda: not self._busy)
            super().resize(w, h)
            self._scene = None
            self.currentSample_ = 0

    def stopRender(self):
        with self._cv:
            self._stop = True
            self._generation += 1
            self._abandon()
            self._cv.notify_all()
        self._thread.join()

    def finished(self):
        return self.currentSample_ >= self.getSpp()

    def wait(self, timeout=None):
        with self._cv:
            ok = self._cv.wait_for(lambda: not self._busy and not self._pending and
                                   (self._scene is None or self.finished() or self._error), timeout)
            if self._error:
                raise RuntimeError(self._error)
            return ok

    def _workerThread(self):
        def tick(_cur, _total, _user):
            self.currentSample_ += self.samplesPerPass_                 # end-of-pass barrier, camera.cpp:141-147
            with self._cv:
                return 1 if (self._generation != self._running or self._stop) else 0

        cb = capi.PROGRESS_CB(tick)
        with self._cv:
            while True:
                self._cv.wait_for(lambda: self._stop or self._pending)
                if self._stop:
                    return
                self._pending = False
                self._running = self._generation
                self._busy = True
                scene, cam = self._scene, self.desc()
                o = capi.RenderOpts()
                o.samples_per_tick = self.samplesPerPass_
                self._cv.release()
                try:
                    rc = self._lib.jtx_mi_render(scene.handle, C.byref(cam), C.byref(o), _fp(self.acc_),
                                                 self.img_.ctypes.data_as(C.POINTER(C.c_uint8)), cb, None)
                    err = None if rc in (0, capi.CANCELLED) else self._lib.jtx_mi_last_error().decode()
                finally:
                    self._cv.acquire()
                self._error = err
                self._busy = False
                self._cv.notify_all()


def camera_rays(cam_desc, row, col, sample):
    lib = capi.load()
    row = np.ascontiguousarray(row, np.int32); col = np.ascontiguousarray(col, np.int32); sample = np.ascontiguousarray(sample, np.int32)
    n = len(row)
    o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32)
    check(lib.jtx_mi_camera_rays(C.byref(cam_desc), n, _ip(row), _ip(col), _ip(sample), _fp(o), _fp(d)))
    return o, d


def radiance_samples(scene, cam_desc, row, col, sample, path_integrator=None):
    lib = capi.load()
    row = np.ascontiguousarray(row, np.int32); col = np.ascontiguousarray(col, np.int32); sample = np.ascontiguousarray(sample, np.int32)
    n = len(row)
    rgb = np.zeros((n, 3), np.float32)
    if path_integrator is None:
        check(lib.jtx_mi_radiance_samples(scene.handle, C.byref(cam_desc), n, _ip(row), _ip(col), _ip(sample), _fp(rgb)))
    else:
        check(lib.jtx_mi_radiance_samples_li(scene.handle, C.byref(cam_desc), path_integrator, n, _ip(row), _ip(col), _ip(sample), _fp(rgb)))
    return rgb


def rng_stream(x, y, n, count):
    lib = capi.load()
    u = np.zeros(count, np.uint32); f = np.zeros(count, np.float32)
    check(lib.jtx_mi_rng_stream(x, y, n, count, u.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(f)))
    return u, f


def sincos(x):
    lib = capi.load()
    x = np.ascontiguousarray(x, np.float32)
    s = np.zeros_like(x); c = np.zeros_like(x)
    check(lib.jtx_mi_sincos_batch(_fp(x), len(x), _fp(s), _fp(c)))
    return s, c
