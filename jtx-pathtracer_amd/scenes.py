"""Scene descriptions for the BASELINE.json configs, as flat numpy arrays.

The reference has no scene files besides hard-coded factories (src/scene.cpp:137-298) and OBJ/GLB
assets it loads through Assimp (src/loader.cpp).  The benchmark scenes are therefore stated here:

* cornell()  -- configs C1/C2: the 32 triangles / 8 objects of the classic Cornell box (the data of
                src/assets/scenes/cornell_box.obj restated as constants), materials and light as in
                SURVEY.md section 8d.  One mesh per object, three vertices per face (Assimp without
                JoinIdenticalVertices, loader.cpp:21), no UVs.
* atrium()   -- configs C3/C4: deterministic procedural "Sponza-class" atrium (~262 k triangles).
* mixed()    -- config C5: Cornell room + spheres with all four material types (+ textures).

`SceneData.to_desc()` produces the ctypes scene description accepted by jtx_mi_scene_create.
"""
import ctypes as C
import math

import numpy as np

from . import _capi as capi

DIFFUSE, DIELECTRIC, CONDUCTOR, METALLIC_ROUGHNESS = 0, 1, 2, 3
THIN_DIELECTRIC = 4                        # ThinDielectricBxDF (dielectric.hpp:163-207); not a Material::Type of the reference
POINT, DISTANT = 0, 1

GOLD_IOR = (0.15557, 0.42415, 1.3831)      # src/scene.cpp:7
GOLD_K = (-3.6024, -2.4721, -1.9155)       # src/scene.cpp:8
SKY_BLUE = (0.529, 0.808, 0.922)           # src/util/color.hpp:10


def material(type=DIFFUSE, albedo=(0, 0, 0), ior=(0, 0, 0), k=(0, 0, 0), alpha_x=0.0, alpha_y=0.0,
             emission=(0, 0, 0), albedo_tex=-1, mr_tex=-1):
    return dict(type=type, albedo=albedo, ior=ior, k=k, alpha_x=alpha_x, alpha_y=alpha_y,
                emission=emission, albedo_tex=albedo_tex, mr_tex=mr_tex)


def light(type=POINT, position=(0, 0, 0), intensity=(1, 1, 1), scale=1.0):
    return dict(type=type, position=position, intensity=intensity, scale=scale, scene_radius=0.0)


class SceneData:
    """Flat scene: meshes (indices/vertices/normals/uvs/material/transform), tri refs, materials, lights, textures."""

    def __init__(self, name="scene"):
        self.name = name
        self.meshes = []
        self.materials = []
        self.lights = []
        self.textures = []           # (H, W, C) float32 arrays
        self.sky = (0.0, 0.0, 0.0)
        self.max_prims_in_node = 1
        self.camera = dict(center=(0, 0, 8), target=(0, 0, 0), up=(0, 1, 0), yfov=20.0,
                           defocus_angle=0.0, focus_distance=1.0)
        self._keep = None

    def add_mesh(self, indices, vertices, normals, material, uvs=None, transform=None, name=""):
        m = dict(indices=np.ascontiguousarray(indices, np.int32).reshape(-1, 3),
                 vertices=np.ascontiguousarray(vertices, np.float32).reshape(-1, 3),
                 normals=np.ascontiguousarray(normals, np.float32).reshape(-1, 3),
                 uvs=None if uvs is None else np.ascontiguousarray(uvs, np.float32).reshape(-1, 2),
                 material=int(material),
                 transform=np.ascontiguousarray(np.eye(4) if transform is None else transform, np.float32),
                 name=name)
        assert m["vertices"].shape == m["normals"].shape
        self.meshes.append(m)
        return len(self.meshes) - 1

    @property
    def num_triangles(self):
        return sum(len(m["indices"]) for m in self.meshes)

    def tri_refs(self):
        """Scene::triangles: one {index, meshIndex} per face, mesh by mesh (loader.cpp:216-222)."""
        refs = np.zeros((self.num_triangles, 2), np.int32)
        at = 0
        for mi, m in enumerate(self.meshes):
            n = len(m["indices"])
            refs[at:at + n, 0] = np.arange(n)
            refs[at:at + n, 1] = mi
            at += n
        return refs

    def to_desc(self):
        """Build a ctypes SceneDesc; all backing arrays are kept alive on `self`."""
        keep = []
        meshes = (capi.Mesh * len(self.meshes))()
        for i, m in enumerate(self.meshes):
            meshes[i].num_triangles = len(m["indices"])
            meshes[i].num_vertices = len(m["vertices"])
            meshes[i].indices = m["indices"].ctypes.data_as(C.POINTER(C.c_int32))
            meshes[i].vertices = m["vertices"].ctypes.data_as(C.POINTER(C.c_float))
            meshes[i].normals = m["normals"].ctypes.data_as(C.POINTER(C.c_float))
            meshes[i].uvs = m["uvs"].ctypes.data_as(C.POINTER(C.c_float)) if m["uvs"] is not None else None
            meshes[i].material = m["material"]
            meshes[i].transform = (C.c_float * 16)(*m["transform"].reshape(-1).tolist())
        refs_np = self.tri_refs()
        keep.append(refs_np)
        mats = (capi.Material * max(1, len(self.materials)))()
        for i, m in enumerate(self.materials):
            mats[i].type = m["type"]
            mats[i].albedo = capi.c_float3(*m["albedo"])
            mats[i].ior = capi.c_float3(*m["ior"])
            mats[i].k = capi.c_float3(*m["k"])
            mats[i].alpha_x, mats[i].alpha_y = m["alpha_x"], m["alpha_y"]
            mats[i].emission = capi.c_float3(*m["emission"])
            mats[i].albedo_tex, mats[i].mr_tex = m["albedo_tex"], m["mr_tex"]
        lights = (capi.Light * max(1, len(self.lights)))()
        for i, l in enumerate(self.lights):
            lights[i].type = l["type"]
            lights[i].position = capi.c_float3(*l["position"])
            lights[i].intensity = capi.c_float3(*l["intensity"])
            lights[i].scale = l["scale"]
            lights[i].scene_radius = l.get("scene_radius", 0.0)
        texs = (capi.Texture * max(1, len(self.textures)))()
        for i, t in enumerate(self.textures):
            t = np.ascontiguousarray(t, np.float32)
            keep.append(t)
            texs[i].height, texs[i].width, texs[i].channels = t.shape
            texs[i].texels = t.ctypes.data_as(C.POINTER(C.c_float))
        d = capi.SceneDesc()
        d.num_meshes, d.meshes = len(self.meshes), meshes
        d.num_tri_refs, d.tri_refs = len(refs_np), refs_np.ctypes.data_as(C.POINTER(capi.TriRef))
        d.num_materials, d.materials = len(self.materials), mats
        d.num_lights, d.lights = len(self.lights), lights
        d.num_textures, d.textures = len(self.textures), texs
        d.sky_color = capi.c_float3(*self.sky)
        d.max_prims_in_node = self.max_prims_in_node
        keep += [meshes, mats, lights, texs]
        self._keep = keep
        return d

    def camera_desc(self, width, height, xs, ys, max_depth):
        c = capi.CameraDesc()
        cam = self.camera
        c.center = capi.c_float3(*cam["center"])
        c.target = capi.c_float3(*cam["target"])
        c.up = capi.c_float3(*cam["up"])
        c.yfov, c.defocus_angle, c.focus_distance = cam["yfov"], cam["defocus_angle"], cam["focus_distance"]
        c.width, c.height = width, height
        c.x_pixel_samples, c.y_pixel_samples, c.max_depth = xs, ys, max_depth
        return c


# ------------------------------------------------------------------------------------------------
# geometry helpers
# ------------------------------------------------------------------------------------------------
def _faces_to_mesh(face_pos, face_nrm, face_uv=None):
    """Three fresh vertices per face (no vertex joining), indices 0..3F-1."""
    v = np.asarray(face_pos, np.float32).reshape(-1, 3)
    n = np.asarray(face_nrm, np.float32).reshape(-1, 3)
    idx = np.arange(len(v), dtype=np.int32).reshape(-1, 3)
    uv = None if face_uv is None else np.asarray(face_uv, np.float32).reshape(-1, 2)
    return idx, v, n, uv


def quad(p0, p1, p2, p3, normal, uv=False):
    """Two triangles (p0,p1,p2), (p0,p2,p3) with a constant normal."""
    pos = [p0, p1, p2, p0, p2, p3]
    nrm = [normal] * 6
    uvs = [(0, 0), (1, 0), (1, 1), (0, 0), (1, 1), (0, 1)] if uv else None
    return _faces_to_mesh(pos, nrm, uvs)


def uv_sphere(center, radius, nu=24, nv=12, uv=True):
    """Indexed latitude/longitude sphere with smooth normals (shared vertices)."""
    cx, cy, cz = center
    verts, nrms, uvs = [], [], []
    for j in range(nv + 1):
        th = math.pi * j / nv
        for i in range(nu + 1):
            ph = 2 * math.pi * i / nu
            n = (math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph))
            verts.append((cx + radius * n[0], cy + radius * n[1], cz + radius * n[2]))
            nrms.append(n)
            uvs.append((i / nu, j / nv))
    idx = []
    for j in range(nv):
        for i in range(nu):
            a = j * (nu + 1) + i
            b = a + nu + 1
            if j != 0:
                idx.append((a, b, a + 1))
            if j != nv - 1:
                idx.append((a + 1, b, b + 1))
    return (np.asarray(idx, np.int32), np.asarray(verts, np.float32), np.asarray(nrms, np.float32),
            np.asarray(uvs, np.float32) if uv else None)


# ------------------------------------------------------------------------------------------------
# C1 / C2: Cornell box
# ------------------------------------------------------------------------------------------------
# (object name, [(v0, v1, v2, v3)] quads as position tuples, [normal per triangle])  -- the data of
# src/assets/scenes/cornell_box.obj; every quad is the two faces (a b c), (a c d).
_CB = [
    ("back_wall", [((549.599976, 0.000084, 559.200012), (0.0, 0.000084, 559.200012), (0.0, 548.800110, 559.199890),
                    (556.0, 548.800110, 559.199890))], [(-0.0, -0.0, -1.0)]),
    ("ceiling", [((556.0, 548.799988, -0.000082), (556.0, 548.800110, 559.199890), (0.0, 548.800110, 559.199890),
                  (0.0, 548.799988, -0.000082))], [(-0.0, -1.0, -0.0)]),
    ("floor", [((552.799988, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.000084, 559.200012),
                (549.599976, 0.000084, 559.200012))], [(-0.0, 1.0, -0.0)]),
    ("left_wall", [((552.799988, 0.0, 0.0), (549.599976, 0.000084, 559.200012), (556.0, 548.800110, 559.199890),
                    (556.0, 548.799988, -0.000082))], [((-0.9999, 0.0117, -0.0057), (-1.0, 0.0058, -0.0))]),
    ("right_wall", [((0.0, 0.000084, 559.200012), (0.0, 0.0, 0.0), (0.0, 548.799988, -0.000082),
                     (0.0, 548.800110, 559.199890))], [(1.0, -0.0, -0.0)]),
    ("short_box", [
        ((130.0, 165.0, 64.999969), (82.0, 165.000031, 224.999969), (240.0, 165.000031, 272.0), (290.0, 165.000031, 113.999969)),
        ((290.0, 0.000018, 114.0), (290.0, 165.000031, 113.999969), (240.0, 165.000031, 272.0), (240.0, 0.000042, 272.0)),
        ((130.0, 0.000010, 65.0), (130.0, 165.0, 64.999969), (290.0, 165.000031, 113.999969), (290.0, 0.000018, 114.0)),
        ((82.0, 0.000034, 225.0), (82.0, 165.000031, 224.999969), (130.0, 165.0, 64.999969), (130.0, 0.000010, 65.0)),
        ((240.0, 0.000042, 272.0), (240.0, 165.000031, 272.0), (82.0, 165.000031, 224.999969), (82.0, 0.000034, 225.0)),
    ], [(-0.0, 1.0, -0.0), (0.9534, -0.0, 0.3017), (0.2928, -0.0, -0.9562), (-0.9578, -0.0, -0.2873), (-0.2851, -0.0, 0.9585)]),
    ("tall_block", [
        ((423.0, 330.000061, 246.999939), (265.0, 330.000061, 295.999939), (314.0, 330.000061, 455.999939), (472.0, 330.000061, 405.999939)),
        ((423.0, 0.000038, 247.0), (423.0, 330.000061, 246.999939), (472.0, 330.000061, 405.999939), (472.0, 0.000062, 406.0)),
        ((472.0, 0.000062, 406.0), (472.0, 330.000061, 405.999939), (314.0, 330.000061, 455.999939), (314.0, 0.000068, 456.0)),
        ((314.0, 0.000068, 456.0), (314.0, 330.000061, 455.999939), (265.0, 330.000061, 295.999939), (265.0, 0.000044, 296.0)),
        ((265.0, 0.000044, 296.0), (265.0, 330.000061, 295.999939), (423.0, 330.000061, 246.999939), (423.0, 0.000038, 247.0)),
    ], [(-0.0, 1.0, -0.0), (0.9556, -0.0, -0.2945), (0.3017, -0.0, 0.9534), (-0.9562, -0.0, 0.2928), (-0.2962, -0.0, -0.9551)]),
    ("area_light", [((353.418701, 547.799988, 203.747177), (353.418701, 547.800049, 355.452698),
                     (202.581299, 547.800049, 355.452698), (202.581299, 547.799988, 203.747177))], [(-0.0, -1.0, -0.0)]),
]


def _cornell_meshes(scene, mat_for):
    for name, quads, normals in _CB:
        pos, nrm = [], []
        for q, n in zip(quads, normals):
            n_a, n_b = (n if isinstance(n[0], tuple) else (n, n))
            a, b, c, d = q
            pos += [a, b, c, a, c, d]
            nrm += [n_a] * 3 + [n_b] * 3
        idx, v, nn, _ = _faces_to_mesh(pos, nrm)
        scene.add_mesh(idx, v, nn, mat_for(name), name=name)


def cornell():
    """Cornell box of configs C1/C2: all diffuse, one point light, black sky."""
    s = SceneData("cornell")
    s.materials = [material(DIFFUSE, (0.73, 0.73, 0.73)), material(DIFFUSE, (0.65, 0.05, 0.05)),
                   material(DIFFUSE, (0.12, 0.45, 0.15))]
    _cornell_meshes(s, lambda n: {"left_wall": 1, "right_wall": 2}.get(n, 0))
    s.lights = [light(POINT, (278.0, 500.0, 279.5), (1, 1, 1), 60000.0)]
    s.sky = (0.0, 0.0, 0.0)
    s.camera = dict(center=(278.0, 273.0, -800.0), target=(278.0, 273.0, 0.0), up=(0, 1, 0), yfov=39.3077,
                    defocus_angle=0.0, focus_distance=1.0)
    return s


# ------------------------------------------------------------------------------------------------
# C5: mixed materials
# ------------------------------------------------------------------------------------------------
def _procedural_textures(size=256):
    """An sRGB-ish albedo checker/gradient and a metallic-roughness map (roughness = .y, metallic = .z)."""
    y, x = np.mgrid[0:size, 0:size].astype(np.float32) / np.float32(size)
    checker = ((np.floor(x * 8) + np.floor(y * 8)) % 2).astype(np.float32)
    albedo = np.stack([0.15 + 0.8 * checker, 0.2 + 0.6 * x, 0.25 + 0.5 * y], -1).astype(np.float32)
    mr = np.stack([np.zeros_like(x), 0.08 + 0.8 * y, (x > 0.5).astype(np.float32)], -1).astype(np.float32)
    return albedo, mr


def mixed(sphere_res=(48, 24), textured=True):
    """Cornell room + spheres cycling all four material types, 1 point + 1 distant light (config C5)."""
    s = SceneData("mixed")
    s.materials = [material(DIFFUSE, (0.73, 0.73, 0.73)), material(DIFFUSE, (0.65, 0.05, 0.05)),
                   material(DIFFUSE, (0.12, 0.45, 0.15))]
    _cornell_meshes(s, lambda n: {"left_wall": 1, "right_wall": 2}.get(n, 0))
    if textured:
        albedo, mr = _procedural_textures()
        s.textures = [albedo, mr]
    specs = [
        material(METALLIC_ROUGHNESS, (0.9, 0.6, 0.2), alpha_x=1.0, alpha_y=0.3),
        material(METALLIC_ROUGHNESS, (0.2, 0.5, 0.9), alpha_x=0.0, alpha_y=0.05),
        material(METALLIC_ROUGHNESS, (0.8, 0.8, 0.8), alpha_x=0.0, alpha_y=0.8),
        material(DIELECTRIC, ior=(1.5, 1.5, 1.5), alpha_x=0.0, alpha_y=0.0),
        material(DIELECTRIC, ior=(1.5, 1.5, 1.5), alpha_x=0.3, alpha_y=0.3),
        material(CONDUCTOR, ior=GOLD_IOR, k=GOLD_K, alpha_x=0.05, alpha_y=0.05),
        material(CONDUCTOR, ior=GOLD_IOR, k=GOLD_K, alpha_x=0.0, alpha_y=0.0),
        material(DIFFUSE, (0.4, 0.7, 0.4)),
    ]
    if textured:
        specs.append(material(METALLIC_ROUGHNESS, (1, 1, 1), alpha_x=0.5, alpha_y=0.5, albedo_tex=0, mr_tex=1))
        specs.append(material(DIFFUSE, (1, 1, 1), albedo_tex=0))
    base = len(s.materials)
    s.materials += specs
    nu, nv = sphere_res
    rng = np.random.RandomState(7)
    k = 0
    for gy in range(3):
        for gx in range(4):
            if k >= len(specs) + 2:
                break
            c = (90.0 + gx * 125.0, 60.0 + gy * 150.0 + (gx % 2) * 40.0, 90.0 + ((gx + gy) % 3) * 150.0)
            r = 38.0 + 14.0 * rng.rand()
            idx, v, n, uv = uv_sphere(c, r, nu, nv)
            s.add_mesh(idx, v, n, base + (k % len(specs)), uvs=uv, name=f"sphere{k}")
            k += 1
    s.lights = [light(POINT, (278.0, 500.0, 279.5), (1, 1, 1), 60000.0),
                light(DISTANT, (0.3, -0.8, 0.52), (1, 0.95, 0.9), 1.5)]
    s.sky = SKY_BLUE
    s.camera = dict(center=(278.0, 273.0, -800.0), target=(278.0, 273.0, 0.0), up=(0, 1, 0), yfov=39.3077,
                    defocus_angle=0.0, focus_distance=1.0)
    return s


def emissive(sphere_res=(16, 8)):
    """Scene for the alternate integrators: the Cornell room whose area_light quad EMITS (cornell_box.obj:121-132),
    spheres of every BxDF incl. a smooth dielectric (the one BxDF that reports isSpecular), a rough one and two
    THIN_DIELECTRIC panes, one emissive sphere; a point light so that integrate has something to sample."""
    s = SceneData("emissive")
    s.materials = [material(DIFFUSE, (0.73, 0.73, 0.73)), material(DIFFUSE, (0.65, 0.05, 0.05)),
                   material(DIFFUSE, (0.12, 0.45, 0.15)), material(DIFFUSE, (0.78, 0.78, 0.78), emission=(17.0, 12.0, 4.0))]
    _cornell_meshes(s, lambda n: {"left_wall": 1, "right_wall": 2, "area_light": 3}.get(n, 0))
    specs = [
        material(DIELECTRIC, ior=(1.5, 1.5, 1.5), alpha_x=0.0, alpha_y=0.0),
        material(THIN_DIELECTRIC, ior=(1.5, 1.5, 1.5)),
        material(DIELECTRIC, ior=(1.5, 1.5, 1.5), alpha_x=0.3, alpha_y=0.3),
        material(CONDUCTOR, ior=GOLD_IOR, k=GOLD_K, alpha_x=0.0, alpha_y=0.0),
        material(METALLIC_ROUGHNESS, (0.9, 0.6, 0.2), alpha_x=1.0, alpha_y=0.3),
        material(DIFFUSE, (0.2, 0.2, 0.25), emission=(0.4, 0.9, 1.6)),
        material(THIN_DIELECTRIC, ior=(1.33, 1.33, 1.33)),
        material(DIFFUSE, (0.4, 0.7, 0.4)),
    ]
    base = len(s.materials)
    s.materials += specs
    nu, nv = sphere_res
    k = 0
    for gy in range(2):
        for gx in range(4):
            c = (90.0 + gx * 125.0, 90.0 + gy * 200.0 + (gx % 2) * 40.0, 120.0 + ((gx + gy) % 3) * 140.0)
            idx, v, n, uv = uv_sphere(c, 48.0, nu, nv)
            s.add_mesh(idx, v, n, base + k, uvs=uv, name=f"sphere{k}")
            k += 1
    s.lights = [light(POINT, (278.0, 500.0, 279.5), (1, 1, 1), 30000.0)]
    s.sky = (0.05, 0.06, 0.08)
    s.camera = dict(center=(278.0, 273.0, -800.0), target=(278.0, 273.0, 0.0), up=(0, 1, 0), yfov=39.3077,
                    defocus_angle=0.0, focus_distance=1.0)
    return s


# ------------------------------------------------------------------------------------------------
# C3 / C4: procedural "Sponza-class" atrium
# ------------------------------------------------------------------------------------------------
def _grid_quad(origin, eu, ev, nu, nv, normal):
    """nu x nv tessellated parallelogram, indexed, shared vertices, constant normal, uvs."""
    o = np.asarray(origin, np.float64)
    eu = np.asarray(eu, np.float64)
    ev = np.asarray(ev, np.float64)
    us, vs = np.meshgrid(np.arange(nu + 1) / nu, np.arange(nv + 1) / nv)
    P = o[None, None, :] + us[..., None] * eu + vs[..., None] * ev
    verts = P.reshape(-1, 3).astype(np.float32)
    nrm = np.tile(np.asarray(normal, np.float32), (len(verts), 1))
    uvs = np.stack([us, vs], -1).reshape(-1, 2).astype(np.float32)
    a = (np.arange(nv)[:, None] * (nu + 1) + np.arange(nu)[None, :]).reshape(-1)
    idx = np.concatenate([np.stack([a, a + 1, a + nu + 2], 1), np.stack([a, a + nu + 2, a + nu + 1], 1)]).astype(np.int32)
    return idx, verts, nrm, uvs


def _column(cx, cz, y0, y1, radius, seg, rings):
    """Tessellated cylinder: long thin triangles on purpose (deep, anisotropic BVH)."""
    ang = 2 * np.pi * np.arange(seg + 1) / seg
    ys = y0 + (y1 - y0) * np.arange(rings + 1) / rings
    A, Y = np.meshgrid(ang, ys)
    verts = np.stack([cx + radius * np.cos(A), Y, cz + radius * np.sin(A)], -1).reshape(-1, 3).astype(np.float32)
    nrm = np.stack([np.cos(A), np.zeros_like(A), np.sin(A)], -1).reshape(-1, 3).astype(np.float32)
    uvs = np.stack([A / (2 * np.pi), (Y - y0) / (y1 - y0)], -1).reshape(-1, 2).astype(np.float32)
    a = (np.arange(rings)[:, None] * (seg + 1) + np.arange(seg)[None, :]).reshape(-1)
    idx = np.concatenate([np.stack([a, a + seg + 2, a + 1], 1), np.stack([a, a + seg + 1, a + seg + 2], 1)]).astype(np.int32)
    return idx, verts, nrm, uvs


def _arch(x0, x1, y0, z, thickness, seg, depth_seg):
    """Half-ring arch between two columns, extruded along z."""
    r_out = (x1 - x0) / 2
    r_in = r_out - thickness
    cx = (x0 + x1) / 2
    ang = np.pi * np.arange(seg + 1) / seg
    zs = z + np.arange(depth_seg + 1) / depth_seg * thickness * 2 - thickness
    meshes = []
    for r, sgn in ((r_in, -1.0), (r_out, 1.0)):
        A, Z = np.meshgrid(ang, zs)
        verts = np.stack([cx + r * np.cos(A), y0 + r * np.sin(A), Z], -1).reshape(-1, 3).astype(np.float32)
        nrm = (sgn * np.stack([np.cos(A), np.sin(A), np.zeros_like(A)], -1)).reshape(-1, 3).astype(np.float32)
        uvs = np.stack([A / np.pi, (Z - zs[0]) / (zs[-1] - zs[0])], -1).reshape(-1, 2).astype(np.float32)
        a = (np.arange(depth_seg)[:, None] * (seg + 1) + np.arange(seg)[None, :]).reshape(-1)
        idx = np.concatenate([np.stack([a, a + 1, a + seg + 2], 1), np.stack([a, a + seg + 2, a + seg + 1], 1)]).astype(np.int32)
        meshes.append((idx, verts, nrm, uvs))
    return meshes


def atrium(target_tris=262144, seed=1):
    """Deterministic procedural atrium for configs C3/C4: floor/walls/ceiling ring, two storeys of
    colonnades (tessellated columns + arches) and hanging cloth-like wavy quads.  All DIFFUSE with 8
    albedos, one DISTANT light + sky blue, camera inside looking down the nave."""
    rng = np.random.RandomState(seed)
    s = SceneData("atrium")
    albedos = [(0.73, 0.71, 0.68), (0.62, 0.55, 0.45), (0.55, 0.22, 0.18), (0.25, 0.42, 0.30),
               (0.30, 0.35, 0.60), (0.80, 0.75, 0.55), (0.45, 0.45, 0.47), (0.70, 0.40, 0.25)]
    s.materials = [material(DIFFUSE, a) for a in albedos]
    L, W, H = 120.0, 40.0, 30.0          # nave length (x), width (z), height (y)
    # scale tessellation so that the total lands near target_tris
    f = max(0.05, math.sqrt(target_tris / 335360.0))      # 335360 = triangle count at f = 1

    def T(n):
        return max(1, int(round(n * f)))

    s.add_mesh(*_grid_quad((0, 0, 0), (L, 0, 0), (0, 0, W), T(96), T(32), (0, 1, 0))[:3], 0,
               uvs=_grid_quad((0, 0, 0), (L, 0, 0), (0, 0, W), T(96), T(32), (0, 1, 0))[3], name="floor")
    for (o, eu, ev, n, m, nm) in [
        ((0, 0, 0), (0, 0, W), (0, H, 0), (1, 0, 0), 1, "wall_x0"),
        ((L, 0, 0), (0, H, 0), (0, 0, W), (-1, 0, 0), 1, "wall_x1"),
        ((0, 0, 0), (0, H, 0), (L, 0, 0), (0, 0, 1), 6, "wall_z0"),
        ((0, 0, W), (L, 0, 0), (0, H, 0), (0, 0, -1), 6, "wall_z1"),
    ]:
        g = _grid_quad(o, eu, ev, T(64), T(24), n)
        s.add_mesh(g[0], g[1], g[2], m, uvs=g[3], name=nm)
    # open roof: a frame of four strips, so that sky light enters
    for (o, eu, ev) in [((0, H, 0), (L, 0, 0), (0, 0, W * 0.3)), ((0, H, W * 0.7), (L, 0, 0), (0, 0, W * 0.3))]:
        g = _grid_quad(o, ev, eu, T(16), T(64), (0, -1, 0))
        s.add_mesh(g[0], g[1], g[2], 0, uvs=g[3], name="roof")
    ncol = 14
    for storey in range(2):
        y0, y1 = storey * 13.0, storey * 13.0 + 10.0
        for side, z in enumerate((W * 0.22, W * 0.78)):
            xs = [6.0 + i * (L - 12.0) / (ncol - 1) for i in range(ncol)]
            for i, x in enumerate(xs):
                g = _column(x, z, y0, y1, 0.9, T(48), T(40))
                s.add_mesh(g[0], g[1], g[2], 2 + (i + side + storey) % 6, uvs=g[3], name=f"col{storey}{side}{i}")
                if i + 1 < ncol:
                    for g in _arch(x, xs[i + 1], y1, z, 0.8, T(40), T(6)):
                        s.add_mesh(g[0], g[1], g[2], 1, uvs=g[3], name="arch")
            # gallery floor slab between storeys
            g = _grid_quad((0, y1 + 3.0, z - 4.0 if side == 0 else z), (L, 0, 0), (0, 0, 4.0), T(96), T(4), (0, -1, 0))
            s.add_mesh(g[0], g[1], g[2], 5, uvs=g[3], name="slab")
    # hanging cloths: wavy quads across the nave
    for c in range(10):
        x = 10.0 + c * 11.0
        nu, nv = T(40), T(56)
        us, vs = np.meshgrid(np.arange(nu + 1) / nu, np.arange(nv + 1) / nv)
        ph = rng.rand() * 6.28
        X = x + 0.6 * np.sin(6.0 * vs + ph) * (0.2 + us)
        Y = H - 2.0 - 14.0 * vs
        Z = W * 0.3 + W * 0.4 * us + 0.3 * np.sin(9.0 * vs + ph)
        verts = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32)
        # smooth normals from the analytic-ish tangents (finite differences on the grid)
        P = np.stack([X, Y, Z], -1)
        du = np.gradient(P, axis=1)
        dv = np.gradient(P, axis=0)
        n = np.cross(du, dv)
        n /= np.linalg.norm(n, axis=-1, keepdims=True)
        a = (np.arange(nv)[:, None] * (nu + 1) + np.arange(nu)[None, :]).reshape(-1)
        idx = np.concatenate([np.stack([a, a + 1, a + nu + 2], 1), np.stack([a, a + nu + 2, a + nu + 1], 1)]).astype(np.int32)
        uvs = np.stack([us, vs], -1).reshape(-1, 2).astype(np.float32)
        s.add_mesh(idx, verts, n.reshape(-1, 3).astype(np.float32), 2 + c % 6, uvs=uvs, name=f"cloth{c}")
    s.lights = [light(DISTANT, (0.25, -0.9, 0.36), (1.0, 0.96, 0.9), 3.0)]
    s.sky = SKY_BLUE
    s.camera = dict(center=(8.0, 7.0, W * 0.5), target=(L, 9.0, W * 0.52), up=(0, 1, 0), yfov=55.0,
                    defocus_angle=0.0, focus_distance=1.0)
    return s


def quad_scene():
    """The two-triangle quad of createMeshScene (src/scene.cpp:137-174) with UVs and tex ids = -1."""
    s = SceneData("mesh")
    s.materials = [material(DIFFUSE, (1.0, 0.3, 0.5))]
    idx = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    v = np.array([[-1, -1, -1], [-1, 1, -1], [1, 1, -1], [1, -1, -1]], np.float32)
    n = np.tile(np.array([[0, 0, 1]], np.float32), (4, 1))
    uv = np.array([[0, 0], [0, 1], [1, 1], [1, 0]], np.float32)
    s.add_mesh(idx, v, n, 0, uvs=uv)
    s.sky = (0.7, 0.8, 1.0)
    s.camera = dict(center=(0, 0, 8), target=(0, 0, -1), up=(0, 1, 0), yfov=20.0, defocus_angle=0.0, focus_distance=3.4)
    return s


# ------------------------------------------------------------------------------------------------
# Wavefront OBJ ingestion: the geometry side of src/loader.cpp without Assimp
# ------------------------------------------------------------------------------------------------
def load_obj(path, default_material=None, materials_by_name=None):
    """Minimal OBJ reader with the semantics of the reference's Assimp import (loader.cpp:21):
    aiProcess_Triangulate (fan), no vertex joining (three fresh vertices per face corner, in face order),
    aiProcess_GenNormals when a face has no vn (flat face normal), aiProcess_FlipUVs (v -> 1 - v),
    PreTransformVertices with identity.  One mesh per `o` / `g` group and material (Assimp splits meshes
    by material); one Triangle ref per face (loader.cpp:216-222).  `materials_by_name` maps usemtl /
    object names to material dicts; everything else gets `default_material` (white Lambert 0.73).
    A `mtllib`'s `map_Kd` becomes the material's albedo texture, read as TextureImage::load reads it (loader.cpp:64-101, 114:
    .exr through the EXR reader, PNG / JPEG through the stb restatements; a file that cannot be read leaves the id at -1);
    such a material is otherwise a copy of `default_material` (the reference's is DIFFUSE white, loader.cpp:139-144)."""
    import os
    default_material = default_material or material(DIFFUSE, (0.73, 0.73, 0.73))
    materials_by_name = dict(materials_by_name or {})
    base = os.path.dirname(path)
    mtl_maps = {}                     # material name -> map_Kd path
    with open(path) as f:
        libs = [ln.split(None, 1)[1].strip() for ln in f if ln.startswith("mtllib") and len(ln.split()) > 1]
    def inside(rel):                   # an asset is untrusted input: its library / map names stay inside its own directory
        root = os.path.realpath(base or ".")
        full = os.path.realpath(os.path.join(root, rel))
        return full if not os.path.isabs(rel) and os.path.commonpath([root, full]) == root else None
    for lib in libs:
        try:
            if inside(lib) is None:
                continue
            with open(inside(lib)) as f:
                name = None
                for ln in f:
                    t = ln.split()
                    if len(t) >= 2 and t[0] == "newmtl":
                        name = t[1]
                    elif len(t) >= 2 and t[0] == "map_Kd" and name is not None:
                        mtl_maps[name] = t[-1]
        except OSError:
            pass
    tex_files, tex_store = {}, []
    for name, rel in mtl_maps.items():
        if name in materials_by_name:
            continue
        full = inside(rel)
        if full not in tex_files:
            from . import gltf
            px = gltf.load_texture_file(full) if full is not None else None
            tex_files[full] = -1
            if px is not None and px.shape[-1] >= 3:
                tex_store.append(px)
                tex_files[full] = len(tex_store) - 1
        m = dict(default_material)
        m["albedo_tex"] = tex_files[full]
        materials_by_name[name] = m
    V, VN, VT = [], [], []
    groups = []                       # (name, mtl, [face corners [(v, vt, vn), ...]])
    cur = None
    cur_name, cur_mtl = "default", None

    def start():
        nonlocal cur
        cur = (cur_name, cur_mtl, [])
        groups.append(cur)

    with open(path) as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "v":
                V.append(tuple(float(x) for x in t[1:4]))
            elif t[0] == "vn":
                VN.append(tuple(float(x) for x in t[1:4]))
            elif t[0] == "vt":
                VT.append((float(t[1]), float(t[2]) if len(t) > 2 else 0.0))
            elif t[0] in ("o", "g"):
                cur_name = t[1] if len(t) > 1 else "default"
                cur = None
            elif t[0] == "usemtl":
                cur_mtl = t[1] if len(t) > 1 else None
                cur = None
            elif t[0] == "f":
                if cur is None:
                    start()
                corners = []
                for c in t[1:]:
                    a = (c.split("/") + ["", ""])[:3]
                    vi = int(a[0]); vti = int(a[1]) if a[1] else 0; vni = int(a[2]) if a[2] else 0
                    corners.append((vi - 1 if vi > 0 else len(V) + vi,
                                    (vti - 1 if vti > 0 else len(VT) + vti) if vti else None,
                                    (vni - 1 if vni > 0 else len(VN) + vni) if vni else None))
                for k in range(1, len(corners) - 1):            # triangle fan
                    cur[2].append((corners[0], corners[k], corners[k + 1]))
    scene = SceneData(path.rsplit("/", 1)[-1])
    scene.textures = tex_store
    mat_index = {}

    def mat_id(name, mtl):
        key = mtl if mtl in materials_by_name else (name if name in materials_by_name else None)
        m = materials_by_name.get(key, default_material)
        k = id(m)
        if k not in mat_index:
            mat_index[k] = len(scene.materials)
            scene.materials.append(dict(m))
        return mat_index[k]

    for name, mtl, faces in groups:
        if not faces:
            continue
        pos, nrm, uvs = [], [], []
        has_uv = all(c[1] is not None for f_ in faces for c in f_)
        for f_ in faces:
            p = [np.asarray(V[c[0]], np.float32) for c in f_]
            if all(c[2] is not None for c in f_):
                n = [VN[c[2]] for c in f_]
            else:                                                # GenNormals: flat
                g = np.cross(p[1] - p[0], p[2] - p[0]).astype(np.float32)
                l = np.float32(np.sqrt(np.dot(g, g)))
                g = g / l if l > 0 else g
                n = [tuple(g)] * 3
            pos += [tuple(x) for x in p]
            nrm += n
            if has_uv:
                uvs += [(np.float32(VT[c[1]][0]), np.float32(1.0) - np.float32(VT[c[1]][1])) for c in f_]   # FlipUVs, in float as Assimp does it
        idx, v, nn, uv = _faces_to_mesh(pos, nrm, uvs if has_uv else None)
        scene.add_mesh(idx, v, nn, mat_id(name, mtl), uvs=uv, name=name)
    return scene


def write_obj(scene, path):
    """Inverse of load_obj for scenes built from three-vertices-per-face meshes (used by the tests)."""
    with open(path, "w") as f:
        vbase = 1
        for m in scene.meshes:
            f.write(f"o {m['name'] or 'mesh'}\n")
            for v in m["vertices"]:
                f.write("v %.9g %.9g %.9g\n" % tuple(float(x) for x in v))
            for n in m["normals"]:
                f.write("vn %.9g %.9g %.9g\n" % tuple(float(x) for x in n))
            for t in m["indices"]:
                f.write("f " + " ".join(f"{vbase + int(i)}//{vbase + int(i)}" for i in t) + "\n")
            vbase += len(m["vertices"])


# ------------------------------------------------------------------------------------------------
# The reference's file-scene factories (src/scene.cpp:176-298).  The asset is passed in: the reference reads
# "assets/scenes/..." next to its binary; nothing of it ships with this package.
# ------------------------------------------------------------------------------------------------
def create_scene(path, transform=None, background=SKY_BLUE):
    """createScene(path, Mat4, background) (scene.cpp:176-209): loadScene + the default camera (0,0,8) -> origin,
    yfov 20; `transform` (4x4) is applied to the vertices and normals of meshes[0] ONLY, as the reference does
    (scene.cpp:203-206).  Every material of an OBJ import is DIFFUSE WHITE (loader.cpp:139-144: Kd is ignored)."""
    white = material(DIFFUSE, (1.0, 1.0, 1.0))
    if path.lower().endswith((".glb", ".gltf")):
        from . import gltf
        s = gltf.load_gltf(path, background=background, allow_missing_textures=True)
    else:
        s = load_obj(path, default_material=white)
        # one Material per distinct usemtl name, as materialMap does (loader.cpp:105-149); load_obj shares one dict
        s.sky = tuple(background)
    s.name = "File scene"
    s.sky = tuple(background)
    s.camera = dict(center=(0, 0, 8), target=(0, 0, 0), up=(0, 1, 0), yfov=20.0, defocus_angle=0.0, focus_distance=1.0)
    if transform is not None and s.meshes:
        m = np.asarray(transform, np.float32)
        v = s.meshes[0]["vertices"]
        s.meshes[0]["vertices"] = (v @ m[:3, :3].T + m[:3, 3]).astype(np.float32)            # Transform::applyToPoint
        s.meshes[0]["normals"] = (s.meshes[0]["normals"] @ m[:3, :3].T).astype(np.float32)   # applyToNormal: upper 3x3
    return s


def _set_mesh_material(s, mesh_index, mat):
    s.materials.append(mat)
    s.meshes[mesh_index]["material"] = len(s.materials) - 1


def knob_scene(path):
    """createKnobScene (scene.cpp:272-298): knob.obj with mesh 0 dull yellow Lambert, meshes 1-2 gold, mesh 3 rough glass."""
    s = create_scene(path)
    if len(s.meshes) < 4:
        raise ValueError("createKnobScene addresses meshes[0..3]")
    s.camera.update(center=(0, 3, 8), target=(0, 0, 0), yfov=15.0)
    _set_mesh_material(s, 0, material(DIFFUSE, (0.3, 0.3, 0.0)))
    gold = material(CONDUCTOR, ior=GOLD_IOR, k=GOLD_K, alpha_x=0.05, alpha_y=0.05)
    _set_mesh_material(s, 1, gold)
    s.meshes[2]["material"] = s.meshes[1]["material"]
    _set_mesh_material(s, 3, material(DIELECTRIC, ior=(1.5, 1.5, 1.5), alpha_x=0.3, alpha_y=0.3))
    return s


def shaderball_scene(path, with_light=False):
    """createShaderBallScene / createShaderBallSceneWithLight (scene.cpp:211-270): mesh 3 gold; the lit variant adds
    one DISTANT light straight down (0,-1,0), scale 10 -- every shadow ray of that scene is axis-parallel."""
    s = create_scene(path)
    if len(s.meshes) < 4:
        raise ValueError("createShaderBallScene addresses meshes[3]")
    s.camera.update(center=(2.5, 16, 12), target=(0, 3, 0), yfov=40.0)
    s.sky = SKY_BLUE if with_light else (0.7, 0.8, 1.0)
    if with_light:
        s.lights.append(light(DISTANT, (0.0, -1.0, 0.0), (1, 1, 1), 10.0))
    _set_mesh_material(s, 3, material(CONDUCTOR, ior=GOLD_IOR, k=GOLD_K, alpha_x=0.05, alpha_y=0.05))
    return s
