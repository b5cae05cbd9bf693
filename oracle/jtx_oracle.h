/*
 * jtx_oracle.h -- C interface of the CPU ORACLE for the JTX path-tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (jtx-pathtracer_amd/, include/)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and there only as the checker / reported
 * baseline -- never as the thing shipped or measured as the product.
 *
 * PARITY UNPINNED: the reference (jebikoh/JTX-PathTracer) has no tests, golden
 * vectors or fixtures for this path, and it cannot be built in this image (its
 * math library ext/jtxlib is an un-vendored, un-pinned git submodule:
 * .gitmodules:1-3, and the rules forbid building it against stand-in headers).
 * This oracle is therefore a restatement of the reference's algorithm, function
 * by function with file:line citations, with the semantics of the missing
 * jtx:: math fixed as written in DESIGN.md ("jtx math spec").
 *
 * The struct layouts below intentionally mirror include/jtx_mi.h so the same
 * flat scene description can be fed to both sides; they are re-declared here so
 * that the oracle stays self-contained.
 */
#ifndef JTX_ORACLE_H
#define JTX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    float    pmin[3];
    float    pmax[3];
    int32_t  offset;      /* leaf: primitivesOffset; interior: secondChildOffset (bvh.hpp:9-12) */
    uint16_t num_prims;   /* >0 => leaf (bvh.hpp:13) */
    uint8_t  axis;        /* split axis of an interior node (bvh.hpp:14) */
    uint8_t  pad;
} ora_bvh_node;           /* 32 B, as LinearBVHNode bvh.hpp:7-15 */

typedef struct { int32_t index; int32_t mesh_index; } ora_tri_ref; /* mesh.hpp:202-204 */

typedef struct {
    int32_t type;          /* 0 DIFFUSE 1 DIELECTRIC 2 CONDUCTOR 3 METALLIC_ROUGHNESS (material.hpp:6-11) */
    float   albedo[3];
    float   ior[3];
    float   k[3];
    float   alpha_x, alpha_y;
    float   emission[3];
    int32_t albedo_tex;    /* -1 = none */
    int32_t mr_tex;        /* -1 = none */
} ora_material;

typedef struct {
    int32_t type;          /* 0 POINT 1 DISTANT (lights.hpp:25-28) */
    float   position[3];
    float   intensity[3];
    float   scale;
    float   scene_radius;  /* filled by the BVH build for DISTANT (scene.cpp:128-134) */
} ora_light;

typedef struct {
    int32_t width, height, channels;
    const float *texels;   /* height*width*channels floats */
} ora_texture;

typedef struct {
    int32_t        num_triangles;
    int32_t        num_vertices;
    const int32_t *indices;   /* 3 per triangle */
    const float   *vertices;  /* 3 per vertex */
    const float   *normals;   /* 3 per vertex */
    const float   *uvs;       /* 2 per vertex or NULL (=> uv 0,0) */
    int32_t        material;  /* index into materials */
    float          transform[16]; /* row-major 4x4, Mesh::transform mesh.hpp:27 */
} ora_mesh;

typedef struct {
    int32_t             num_meshes;
    const ora_mesh     *meshes;
    int32_t             num_tri_refs;
    const ora_tri_ref  *tri_refs;     /* Scene::triangles order (scene.hpp:33) */
    int32_t             num_materials;
    const ora_material *materials;
    int32_t             num_lights;
    const ora_light    *lights;
    int32_t             num_textures;
    const ora_texture  *textures;
    float               sky_color[3];
    int32_t             max_prims_in_node; /* Scene::buildBVH arg, default 1 (scene.hpp:55) */
} ora_scene_desc;

typedef struct {
    float   center[3], target[3], up[3];
    float   yfov, defocus_angle, focus_distance;   /* CameraProperties scene.hpp:15-22 */
    int32_t width, height;
    int32_t x_pixel_samples, y_pixel_samples;
    int32_t max_depth;
} ora_camera_desc;

typedef struct {
    uint64_t n_camera;        /* camera samples */
    uint64_t n_closest;       /* Scene::closestHit calls */
    uint64_t n_any;           /* Scene::anyHit calls */
    uint64_t n_nodes_closest; /* node visits inside closestHit */
    uint64_t n_tri_closest;   /* triangle tests inside closestHit */
    uint64_t n_accept;        /* accepted closer hits */
    uint64_t n_nodes_any;
    uint64_t n_tri_any;
    uint64_t n_shade;         /* shading events (sampleBxdf calls) */
    uint64_t n_shade_class[8]; /* ... by BxDF class: 0 DIFFUSE, 1 DIELECTRIC rough, 2 CONDUCTOR rough, 3 METALLIC_ROUGHNESS, 4 thin dielectric,
                                * 5 DIELECTRIC smooth / index-matched (dielectric.hpp:44), 6 CONDUCTOR smooth (conductor.hpp:33), 7 unused */
    uint64_t n_eval_class[8];  /* evalBxdf + pdfBxdf calls by BxDF class (integrator.cpp:94-101, 151-166) */
} ora_counters;

typedef struct ora_scene ora_scene;

/* sincos_mode: 0 = deterministic polynomial sin/cos of DESIGN.md (bit-matched by the HIP kernels),
 *              1 = host libm sinf/cosf (what the reference's jtx::sin/cos presumably forward to). */
void ora_set_sincos_mode(int mode);
/* which Li the render / radiance entry points evaluate (camera.cpp:104-106 picks by (un)commenting):
 * 0 integrateMIS (integrator.cpp:171-216), 1 integrate (:54-132), 2 integrateBasic (:12-52).  Material type 4 =
 * ThinDielectricBxDF (dielectric.hpp:163-207). */
void ora_set_integrator(int li);

/* ---- RNG (util/rand.hpp) ---- */
uint32_t ora_fnv1a_3(uint32_t x, uint32_t y, uint32_t n);
void     ora_rng_stream(uint32_t x, uint32_t y, uint32_t n, int count, uint32_t *out_u32, float *out_f32);
uint32_t ora_rng_sample_range(uint32_t x, uint32_t y, uint32_t n, int skip, int range_arg);

/* ---- deterministic sin/cos ---- */
void ora_sincos_batch(const float *x, int n, float *out_sin, float *out_cos);

/* ---- scene / BVH (scene.cpp:96-135, bvh.cpp) ---- */
ora_scene *ora_scene_create(const ora_scene_desc *desc);
void       ora_scene_destroy(ora_scene *s);
int        ora_scene_num_nodes(const ora_scene *s);
int        ora_scene_num_prims(const ora_scene *s);
int        ora_scene_max_depth(const ora_scene *s);
void       ora_scene_get_bvh(const ora_scene *s, ora_bvh_node *nodes_out, ora_tri_ref *ordered_refs_out);
float      ora_scene_radius(const ora_scene *s);

/* ---- traversal (scene.cpp:10-94) ----  rays: o[3n], d[3n]; prim = index into the BVH-ordered refs */
void ora_closest_hit_batch(const ora_scene *s, int n, const float *o, const float *d, float tmin, float tmax,
                           int32_t *hit, float *t, int32_t *prim, float *b1, float *b2,
                           float *point, float *normal, float *uv);
void ora_any_hit_batch(const ora_scene *s, int n, const float *o, const float *d, const float *tmin,
                       const float *tmax, int32_t *hit);
int  ora_aabb_hit(const float pmin[3], const float pmax[3], const float o[3], const float d[3], float t0, float t1);

/* ---- BxDF (bsdf/bxdf.cpp) ---- inputs per item: normal[3], uv[2], wo[3] (world), wi[3] (world), uc, u[2] */
void ora_bxdf_sample_batch(const ora_scene *s, int material, int n, const float *normal, const float *uv,
                           const float *wo, const float *uc, const float *u2,
                           int32_t *ok, float *f, float *wi, float *pdf);
void ora_bxdf_eval_batch(const ora_scene *s, int material, int n, const float *normal, const float *uv,
                         const float *wo, const float *wi, float *f);
void ora_bxdf_pdf_batch(const ora_scene *s, int material, int n, const float *normal, const float *uv,
                        const float *wo, const float *wi, float *pdf);

/* ---- camera + integrator + film (camera.cpp:45-128, integrator.cpp:171-216, image.hpp) ---- */
void ora_camera_rays(const ora_camera_desc *cam, int n, const int32_t *row, const int32_t *col,
                     const int32_t *sample, float *o, float *d);
/* per-sample radiance (after the <=1 clamp of camera.cpp:110-112) for listed (row,col,sample) triples */
void ora_radiance_samples(const ora_scene *s, const ora_camera_desc *cam, int n, const int32_t *row,
                          const int32_t *col, const int32_t *sample, float *rgb);
/* full frame: acc_rgb [H*W*3] float sums, img_rgb [H*W*3] u8; row 0 = bottom scan-line.
 * threads <= 0 => all cores. sample range [sample_begin, sample_end) of the xs*ys strata.
 * reference_barriers != 0 => one barrier per sample pass as StaticCamera::render (camera.cpp:68-74,120). */
void ora_render(const ora_scene *s, const ora_camera_desc *cam, int threads, int sample_begin, int sample_end,
                int reference_barriers, float *acc_rgb, uint8_t *img_rgb, ora_counters *counters);

/* diagnostic: per-node visit counts of one frame (closest + any), out[num_nodes] */
void ora_node_histogram(const ora_scene *s, const ora_camera_desc *cam, int threads, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif
