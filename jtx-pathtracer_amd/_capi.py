"""ctypes binding of the C-ABI in include/jtx_mi.h (libjtx_mi.so, built in-tree by build.py).

There is no fallback: if the shared library is missing, `load()` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libjtx_mi.so")

c_float3 = C.c_float * 3


class BvhNode(C.Structure):
    _fields_ = [("pmin", c_float3), ("pmax", c_float3), ("offset", C.c_int32),
                ("num_prims", C.c_uint16), ("axis", C.c_uint8), ("pad", C.c_uint8)]


class TriRef(C.Structure):
    _fields_ = [("index", C.c_int32), ("mesh_index", C.c_int32)]


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("albedo", c_float3), ("ior", c_float3), ("k", c_float3),
                ("alpha_x", C.c_float), ("alpha_y", C.c_float), ("emission", c_float3),
                ("albedo_tex", C.c_int32), ("mr_tex", C.c_int32)]


class Light(C.Structure):
    _fields_ = [("type", C.c_int32), ("position", c_float3), ("intensity", c_float3),
                ("scale", C.c_float), ("scene_radius", C.c_float)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32),
                ("texels", C.POINTER(C.c_float))]


class Mesh(C.Structure):
    _fields_ = [("num_triangles", C.c_int32), ("num_vertices", C.c_int32),
                ("indices", C.POINTER(C.c_int32)), ("vertices", C.POINTER(C.c_float)),
                ("normals", C.POINTER(C.c_float)), ("uvs", C.POINTER(C.c_float)),
                ("material", C.c_int32), ("transform", C.c_float * 16)]


class SceneDesc(C.Structure):
    _fields_ = [("num_meshes", C.c_int32), ("meshes", C.POINTER(Mesh)),
                ("num_tri_refs", C.c_int32), ("tri_refs", C.POINTER(TriRef)),
                ("num_materials", C.c_int32), ("materials", C.POINTER(Material)),
                ("num_lights", C.c_int32), ("lights", C.POINTER(Light)),
                ("num_textures", C.c_int32), ("textures", C.POINTER(Texture)),
                ("sky_color", c_float3), ("max_prims_in_node", C.c_int32)]


class CameraDesc(C.Structure):
    _fields_ = [("center", c_float3), ("target", c_float3), ("up", c_float3),
                ("yfov", C.c_float), ("defocus_angle", C.c_float), ("focus_distance", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32),
                ("x_pixel_samples", C.c_int32), ("y_pixel_samples", C.c_int32), ("max_depth", C.c_int32)]


class RenderOpts(C.Structure):
    _fields_ = [("sample_begin", C.c_int32), ("sample_end", C.c_int32), ("tile_rank", C.c_int32),
                ("tile_world", C.c_int32), ("integrator", C.c_int32), ("count_rays", C.c_int32),
                ("samples_per_tick", C.c_int32), ("reserved", C.c_int32), ("path_integrator", C.c_int32), ("frame_slot", C.c_int32), ("sequence_end", C.c_int32), ("max_record_mb", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_camera", "n_closest", "n_any", "n_nodes_closest", "n_tri_closest",
                                          "n_accept", "n_nodes_any", "n_tri_any", "n_shade")] + \
               [("n_shade_class", C.c_uint64 * 8), ("n_eval_class", C.c_uint64 * 8)]

    def as_dict(self):
        return {n: (int(getattr(self, n)) if t is C.c_uint64 else [int(x) for x in getattr(self, n)]) for n, t in self._fields_}


class SceneInfo(C.Structure):
    _fields_ = [("num_nodes", C.c_int32), ("num_prims", C.c_int32), ("max_depth", C.c_int32),
                ("lds_resident", C.c_int32), ("scene_radius", C.c_float), ("auto_integrator", C.c_int32),
                ("device_bytes", C.c_uint64), ("wide_depth", C.c_int32), ("wide_bytes", C.c_int32), ("refitted", C.c_int32),
                ("num_cus", C.c_int32), ("resident_workgroups", C.c_int32), ("workgroup_size", C.c_int32), ("device_built", C.c_int32),
                ("wide_bytes64", C.c_uint64), ("rebuild_spare_bytes", C.c_uint64), ("frame_slot_bytes", C.c_uint64)]


CANCELLED = 2          # JTX_MI_CANCELLED
PROGRESS_CB = C.CFUNCTYPE(C.c_int, C.c_int32, C.c_int32, C.c_void_p)

P = C.POINTER
_f, _i, _u8, _u32 = P(C.c_float), P(C.c_int32), P(C.c_uint8), P(C.c_uint32)
_scene = C.c_void_p

# every symbol include/jtx_mi.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "jtx_mi_last_error": (C.c_char_p, []),
    "jtx_mi_version": (C.c_int, []),
    "jtx_mi_device_count": (C.c_int, [_i]),
    "jtx_mi_set_device": (C.c_int, [C.c_int32]),
    "jtx_mi_bvh_build": (C.c_int, [P(SceneDesc), P(BvhNode), _i, P(TriRef), _i]),
    "jtx_mi_wide_build": (C.c_int, [P(BvhNode), C.c_int32, _u32, C.c_int64, P(C.c_int64), _i]),
    "jtx_mi_scene_create": (C.c_int, [P(SceneDesc), P(_scene)]),
    "jtx_mi_scene_destroy": (None, [_scene]),
    "jtx_mi_scene_get_info": (C.c_int, [_scene, P(SceneInfo)]),
    "jtx_mi_scene_get_bvh": (C.c_int, [_scene, P(BvhNode), P(TriRef)]),
    "jtx_mi_scene_get_wide": (C.c_int, [_scene, _u32, C.c_int64, P(C.c_int64)]),
    "jtx_mi_render": (C.c_int, [_scene, P(CameraDesc), P(RenderOpts), _f, _u8, PROGRESS_CB, C.c_void_p]),
    "jtx_mi_decode_jpeg": (C.c_int, [_u8, C.c_int64, P(C.c_int32), P(C.c_int32), P(C.c_int32), _u8, C.c_int64]),
    "jtx_mi_decode_png": (C.c_int, [_u8, C.c_int64, P(C.c_int32), P(C.c_int32), P(C.c_int32), _u8, C.c_int64]),
    "jtx_mi_decode_exr": (C.c_int, [_u8, C.c_int64, P(C.c_int32), P(C.c_int32), P(C.c_float), C.c_int64]),
    "jtx_mi_scene_set_transform": (C.c_int, [_scene, C.c_int32, _f]),
    "jtx_mi_scene_refit": (C.c_int, [_scene]),
    "jtx_mi_scene_rebuild": (C.c_int, [_scene, C.c_int32]),
    "jtx_mi_scene_release_rebuild": (C.c_int, [_scene]),
    "jtx_mi_scene_reserve_rebuild": (C.c_int, [_scene]),
    "jtx_mi_scene_release_frames": (C.c_int, [_scene]),
    "jtx_mi_cancel": (C.c_int, [_scene]),
    "jtx_mi_pin_host": (C.c_int, [C.c_void_p, C.c_uint64]),
    "jtx_mi_unpin_host": (C.c_int, [C.c_void_p]),
    "jtx_mi_cancel_pending": (C.c_int, [_scene, P(C.c_int32)]),
    "jtx_mi_cancel_reset": (C.c_int, [_scene]),
    "jtx_mi_multi_create": (C.c_int, [P(SceneDesc), P(C.c_int32), C.c_int32, P(C.c_void_p)]),
    "jtx_mi_multi_destroy": (None, [C.c_void_p]),
    "jtx_mi_multi_render": (C.c_int, [C.c_void_p, P(CameraDesc), P(RenderOpts), _f, _u8, PROGRESS_CB, C.c_void_p]),
    "jtx_mi_multi_cancel": (C.c_int, [C.c_void_p]),
    "jtx_mi_multi_last_completed_sample": (C.c_int, [C.c_void_p, P(C.c_int32)]),
    "jtx_mi_multi_shard_time": (C.c_int, [C.c_void_p, _f, C.c_int32]),
    "jtx_mi_last_completed_sample": (C.c_int, [_scene, P(C.c_int32)]),
    "jtx_mi_render_device": (C.c_int, [_scene, P(CameraDesc), P(RenderOpts), C.c_void_p, C.c_void_p, C.c_void_p]),
    "jtx_mi_sync": (C.c_int, [_scene]),
    "jtx_mi_kernel_time": (C.c_int, [_scene, _f, _i]),
    "jtx_mi_kernel_time_by_kind": (C.c_int, [_scene, _f, _i]),
    "jtx_mi_get_counters": (C.c_int, [_scene, P(Counters)]),
    "jtx_mi_closest_hit_batch": (C.c_int, [_scene, C.c_int32, _f, _f, C.c_float, C.c_float, _i, _f, _i, _f, _f, _f, _f, _f]),
    "jtx_mi_any_hit_batch": (C.c_int, [_scene, C.c_int32, _f, _f, _f, _f, _i]),
    "jtx_mi_closest_hit_batch_via": (C.c_int, [_scene, C.c_int32, C.c_int32, _f, _f, C.c_float, C.c_float, _i, _f, _i, _f, _f, _f, _f, _f, _i]),
    "jtx_mi_any_hit_batch_via": (C.c_int, [_scene, C.c_int32, C.c_int32, _f, _f, _f, _f, _i, _i]),
    "jtx_mi_bxdf_sample_batch": (C.c_int, [_scene, C.c_int32, C.c_int32, _f, _f, _f, _f, _f, _i, _f, _f, _f]),
    "jtx_mi_bxdf_eval_batch": (C.c_int, [_scene, C.c_int32, C.c_int32, _f, _f, _f, _f, _f]),
    "jtx_mi_bxdf_pdf_batch": (C.c_int, [_scene, C.c_int32, C.c_int32, _f, _f, _f, _f, _f]),
    "jtx_mi_camera_rays": (C.c_int, [P(CameraDesc), C.c_int32, _i, _i, _i, _f, _f]),
    "jtx_mi_radiance_samples": (C.c_int, [_scene, P(CameraDesc), C.c_int32, _i, _i, _i, _f]),
    "jtx_mi_radiance_samples_li": (C.c_int, [_scene, P(CameraDesc), C.c_int32, C.c_int32, _i, _i, _i, _f]),
    "jtx_mi_rng_stream": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, _u32, _f]),
    "jtx_mi_sincos_batch": (C.c_int, [_f, C.c_int32, _f, _f]),
}

_lib = None


class JtxMiError(RuntimeError):
    pass


def load():
    """Load libjtx_mi.so (once) and type every exported symbol.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("JTX_MI_LIB", LIB_PATH)      # developer builds (tools/: diagnostic -D variants, A/B libraries)
    if not os.path.exists(path):
        raise JtxMiError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The jtx_mi core has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)            # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise JtxMiError(load().jtx_mi_last_error().decode("utf-8", "replace"))
