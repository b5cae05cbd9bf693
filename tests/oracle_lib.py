"""ctypes binding of the CPU oracle (oracle/jtx_oracle.h).  Test infrastructure only.

Builds oracle/_build/libjtx_oracle.so with `make -C oracle` on first use.  The oracle's struct layouts
mirror include/jtx_mi.h, so the product's ctypes classes (jtx_pathtracer_amd._capi) describe them too.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "libjtx_oracle.so")

import sys
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import jtx_pathtracer_amd as jtx  # noqa: E402
capi = jtx._capi

_lib = None
P = C.POINTER
_f, _i, _u8, _u32 = P(C.c_float), P(C.c_int32), P(C.c_uint8), P(C.c_uint32)


def build(fast=False):
    target = [] if not fast else ["fast"]
    r = subprocess.run(["make", "-C", ORACLE_DIR] + target, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)


def load():
    global _lib
    if _lib is not None:
        return _lib
    src_t = max(os.path.getmtime(os.path.join(ORACLE_DIR, f)) for f in ("jtx_oracle.cpp", "jtx_oracle.h"))
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < src_t:
        build()
    lib = C.CDLL(LIB)
    sig = {
        "ora_set_sincos_mode": (None, [C.c_int]),
        "ora_set_integrator": (None, [C.c_int]),
        "ora_fnv1a_3": (C.c_uint32, [C.c_uint32] * 3),
        "ora_rng_stream": (None, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _u32, _f]),
        "ora_rng_sample_range": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int]),
        "ora_sincos_batch": (None, [_f, C.c_int, _f, _f]),
        "ora_scene_create": (C.c_void_p, [P(capi.SceneDesc)]),
        "ora_scene_destroy": (None, [C.c_void_p]),
        "ora_scene_num_nodes": (C.c_int, [C.c_void_p]),
        "ora_scene_num_prims": (C.c_int, [C.c_void_p]),
        "ora_scene_max_depth": (C.c_int, [C.c_void_p]),
        "ora_scene_radius": (C.c_float, [C.c_void_p]),
        "ora_scene_get_bvh": (None, [C.c_void_p, P(capi.BvhNode), P(capi.TriRef)]),
        "ora_aabb_hit": (C.c_int, [_f, _f, _f, _f, C.c_float, C.c_float]),
        "ora_closest_hit_batch": (None, [C.c_void_p, C.c_int, _f, _f, C.c_float, C.c_float, _i, _f, _i, _f, _f, _f, _f, _f]),
        "ora_any_hit_batch": (None, [C.c_void_p, C.c_int, _f, _f, _f, _f, _i]),
        "ora_bxdf_sample_batch": (None, [C.c_void_p, C.c_int, C.c_int, _f, _f, _f, _f, _f, _i, _f, _f, _f]),
        "ora_bxdf_eval_batch": (None, [C.c_void_p, C.c_int, C.c_int, _f, _f, _f, _f, _f]),
        "ora_bxdf_pdf_batch": (None, [C.c_void_p, C.c_int, C.c_int, _f, _f, _f, _f, _f]),
        "ora_camera_rays": (None, [P(capi.CameraDesc), C.c_int, _i, _i, _i, _f, _f]),
        "ora_radiance_samples": (None, [C.c_void_p, P(capi.CameraDesc), C.c_int, _i, _i, _i, _f]),
        "ora_render": (None, [C.c_void_p, P(capi.CameraDesc), C.c_int, C.c_int, C.c_int, C.c_int, _f, _u8, P(capi.Counters)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _fp(a):
    return a.ctypes.data_as(_f)


def _ip(a):
    return a.ctypes.data_as(_i)


class OracleScene:
    def __init__(self, data):
        self.lib = load()
        self.data = data
        desc = data.to_desc()
        self.h = C.c_void_p(self.lib.ora_scene_create(C.byref(desc)))

    def __del__(self):
        try:
            if self.h:
                self.lib.ora_scene_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def info(self):
        return dict(num_nodes=self.lib.ora_scene_num_nodes(self.h), num_prims=self.lib.ora_scene_num_prims(self.h),
                    max_depth=self.lib.ora_scene_max_depth(self.h), scene_radius=float(self.lib.ora_scene_radius(self.h)))

    def bvh(self):
        i = self.info()
        nodes = (capi.BvhNode * max(1, i["num_nodes"]))()
        refs = (capi.TriRef * max(1, i["num_prims"]))()
        self.lib.ora_scene_get_bvh(self.h, nodes, refs)
        return jtx.api.nodes_to_numpy(nodes, i["num_nodes"]), jtx.api.refs_to_numpy(refs, i["num_prims"])

    def closestHit(self, o, d, tmin=0.001, tmax=float("inf")):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3); d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = len(o)
        out = dict(hit=np.zeros(n, np.int32), t=np.zeros(n, np.float32), prim=np.zeros(n, np.int32),
                   b1=np.zeros(n, np.float32), b2=np.zeros(n, np.float32), point=np.zeros((n, 3), np.float32),
                   normal=np.zeros((n, 3), np.float32), uv=np.zeros((n, 2), np.float32))
        self.lib.ora_closest_hit_batch(self.h, n, _fp(o), _fp(d), tmin, tmax, _ip(out["hit"]), _fp(out["t"]), _ip(out["prim"]),
                                       _fp(out["b1"]), _fp(out["b2"]), _fp(out["point"]), _fp(out["normal"]), _fp(out["uv"]))
        return out

    def anyHit(self, o, d, tmin, tmax):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3); d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = len(o)
        tmin = np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, np.float32), (n,)))
        tmax = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, np.float32), (n,)))
        hit = np.zeros(n, np.int32)
        self.lib.ora_any_hit_batch(self.h, n, _fp(o), _fp(d), _fp(tmin), _fp(tmax), _ip(hit))
        return hit

    def sampleBxdf(self, material, normal, wo, uc, u2, uv=None):
        normal = np.ascontiguousarray(normal, np.float32).reshape(-1, 3); wo = np.ascontiguousarray(wo, np.float32).reshape(-1, 3)
        uc = np.ascontiguousarray(uc, np.float32).reshape(-1); u2 = np.ascontiguousarray(u2, np.float32).reshape(-1, 2)
        n = len(normal)
        uvp = _fp(np.ascontiguousarray(uv, np.float32)) if uv is not None else None
        ok = np.zeros(n, np.int32); f = np.zeros((n, 3), np.float32); wi = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32)
        self.lib.ora_bxdf_sample_batch(self.h, material, n, _fp(normal), uvp, _fp(wo), _fp(uc), _fp(u2), _ip(ok), _fp(f), _fp(wi), _fp(pdf))
        return dict(ok=ok, f=f, wi=wi, pdf=pdf)

    def evalBxdf(self, material, normal, wo, wi, uv=None):
        normal = np.ascontiguousarray(normal, np.float32).reshape(-1, 3); wo = np.ascontiguousarray(wo, np.float32).reshape(-1, 3)
        wi = np.ascontiguousarray(wi, np.float32).reshape(-1, 3)
        n = len(normal)
        uvp = _fp(np.ascontiguousarray(uv, np.float32)) if uv is not None else None
        f = np.zeros((n, 3), np.float32)
        self.lib.ora_bxdf_eval_batch(self.h, material, n, _fp(normal), uvp, _fp(wo), _fp(wi), _fp(f))
        return f

    def pdfBxdf(self, material, normal, wo, wi, uv=None):
        normal = np.ascontiguousarray(normal, np.float32).reshape(-1, 3); wo = np.ascontiguousarray(wo, np.float32).reshape(-1, 3)
        wi = np.ascontiguousarray(wi, np.float32).reshape(-1, 3)
        n = len(normal)
        uvp = _fp(np.ascontiguousarray(uv, np.float32)) if uv is not None else None
        pdf = np.zeros(n, np.float32)
        self.lib.ora_bxdf_pdf_batch(self.h, material, n, _fp(normal), uvp, _fp(wo), _fp(wi), _fp(pdf))
        return pdf

    def radiance_samples(self, cam, row, col, sample, path_integrator=0):
        row = np.ascontiguousarray(row, np.int32); col = np.ascontiguousarray(col, np.int32); sample = np.ascontiguousarray(sample, np.int32)
        rgb = np.zeros((len(row), 3), np.float32)
        self.lib.ora_set_integrator(path_integrator)
        try:
            self.lib.ora_radiance_samples(self.h, C.byref(cam), len(row), _ip(row), _ip(col), _ip(sample), _fp(rgb))
        finally:
            self.lib.ora_set_integrator(0)
        return rgb

    def render(self, cam, threads=0, sample_begin=0, sample_end=0, reference_barriers=False, count=True, acc=None, path_integrator=0):
        H, W = cam.height, cam.width
        spp = cam.x_pixel_samples * cam.y_pixel_samples
        if sample_end <= 0:
            sample_end = spp
        if threads <= 0:
            # the checker does not need every core of a 256-core host: a test process that also holds torch's and OpenMP's pools
            # comes close to a container's thread limit with 256 more (bench.py's cpu_baseline passes its thread count itself)
            threads = max(1, min(os.cpu_count() or 1, 64))
        if acc is None:
            acc = np.zeros((H, W, 3), np.float32)
        img = np.zeros((H, W, 3), np.uint8)
        cnt = capi.Counters()
        self.lib.ora_set_integrator(path_integrator)
        try:
            self.lib.ora_render(self.h, C.byref(cam), threads, sample_begin, sample_end, 1 if reference_barriers else 0,
                                _fp(acc), img.ctypes.data_as(_u8), C.byref(cnt) if count else None)
        finally:
            self.lib.ora_set_integrator(0)
        return acc, img, (cnt.as_dict() if count else None)


def camera_rays(cam, row, col, sample):
    lib = load()
    row = np.ascontiguousarray(row, np.int32); col = np.ascontiguousarray(col, np.int32); sample = np.ascontiguousarray(sample, np.int32)
    o = np.zeros((len(row), 3), np.float32); d = np.zeros((len(row), 3), np.float32)
    lib.ora_camera_rays(C.byref(cam), len(row), _ip(row), _ip(col), _ip(sample), _fp(o), _fp(d))
    return o, d


def rng_stream(x, y, n, count):
    lib = load()
    u = np.zeros(count, np.uint32); f = np.zeros(count, np.float32)
    lib.ora_rng_stream(x, y, n, count, u.ctypes.data_as(_u32), _fp(f))
    return u, f


def sincos(x):
    lib = load()
    x = np.ascontiguousarray(x, np.float32)
    s = np.zeros_like(x); c = np.zeros_like(x)
    lib.ora_sincos_batch(_fp(x), len(x), _fp(s), _fp(c))
    return s, c
