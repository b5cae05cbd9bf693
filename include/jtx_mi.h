/*
 * jtx_mi.h -- C-ABI of the MI355X-native path-tracing core for JTX-PathTracer.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference
 * has no FFI of its own -- the path is reached by direct C++ calls -- so each entry point below
 * names the reference interface it replaces.  Plain pointers and sizes only; no C++/torch types.
 * All functions return 0 on success, non-zero on failure (jtx_mi_last_error() has the text) and
 * never throw.  The library has NO CPU fallback: every compute entry point needs a gfx950 device.
 *
 * Buffer orientation (reference quirk Q9: camera.cpp:103, camera.hpp:127-139): row 0 of the
 * acc/img buffers is the BOTTOM scan-line, exactly as Camera::img_ / acc_ in the reference
 * (RGB8Image::save flips rows, image.cpp:14-22).
 */
#ifndef JTX_MI_H
#define JTX_MI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JTX_MI_VERSION 6
#define JTX_MI_FRAME_SLOTS 3
#define JTX_MI_CANCELLED 2   /* jtx_mi_render: stopped by the callback / jtx_mi_cancel; the film holds the completed passes */

/* LinearBVHNode, src/bvh.hpp:7-15 (32 B) */
typedef struct {
    float    pmin[3];
    float    pmax[3];
    int32_t  offset;      /* leaf: primitivesOffset; interior: secondChildOffset */
    uint16_t num_prims;   /* > 0 => leaf */
    uint8_t  axis;
    uint8_t  pad;
} jtx_mi_bvh_node;

/* Triangle{index, meshIndex}, src/mesh.hpp:202-204 */
typedef struct { int32_t index; int32_t mesh_index; } jtx_mi_tri_ref;

/* Material, src/material.hpp:5-23.  Texture ids: -1 = none (reference quirk Q4). */
typedef struct {
    int32_t type;          /* 0 DIFFUSE, 1 DIELECTRIC, 2 CONDUCTOR, 3 METALLIC_ROUGHNESS (material.hpp:6-11);
                            * 4 THIN_DIELECTRIC = ThinDielectricBxDF (dielectric.hpp:163-207; eta = ior[0]), which the reference's
                            * bxdf.cpp never dispatches */
    float   albedo[3];
    float   ior[3];
    float   k[3];
    float   alpha_x, alpha_y;   /* METALLIC_ROUGHNESS: alpha_x = metallic, alpha_y = roughness (loader.cpp:136-137) */
    float   emission[3];        /* Material::emission: added by integrate / integrateBasic (integrator.cpp:27,75), not by integrateMIS (:189-190) */
    int32_t albedo_tex;
    int32_t mr_tex;
} jtx_mi_material;

/* Light, src/lights/lights.hpp:24-34 */
typedef struct {
    int32_t type;          /* 0 POINT, 1 DISTANT */
    float   position[3];   /* DISTANT: direction the light travels */
    float   intensity[3];
    float   scale;
    float   scene_radius;  /* overwritten by scene_create for DISTANT (scene.cpp:128-134) */
} jtx_mi_light;

/* TextureImage, src/image.hpp:99-226: float texels, nearest lookup, wrap */
typedef struct {
    int32_t width, height, channels;
    const float *texels;
} jtx_mi_texture;

/* Mesh, src/mesh.hpp:10-69 */
typedef struct {
    int32_t        num_triangles;
    int32_t        num_vertices;
    const int32_t *indices;       /* 3 per triangle */
    const float   *vertices;      /* 3 per vertex */
    const float   *normals;       /* 3 per vertex */
    const float   *uvs;           /* 2 per vertex, or NULL => uv (0,0) (reference quirk Q3) */
    int32_t        material;      /* index into materials (replaces Material*) */
    float          transform[16]; /* row-major Mesh::transform, baked into the device copy at create */
} jtx_mi_mesh;

/* Scene, src/scene.hpp:24-91 */
typedef struct {
    int32_t                num_meshes;
    const jtx_mi_mesh     *meshes;
    int32_t                num_tri_refs;
    const jtx_mi_tri_ref  *tri_refs;      /* Scene::triangles */
    int32_t                num_materials;
    const jtx_mi_material *materials;
    int32_t                num_lights;
    const jtx_mi_light    *lights;
    int32_t                num_textures;
    const jtx_mi_texture  *textures;
    float                  sky_color[3];
    int32_t                max_prims_in_node;  /* Scene::buildBVH(maxPrimsInNode), default 1 */
} jtx_mi_scene_desc;

/* Camera ctor args + CameraProperties, src/camera.hpp:44-54, src/scene.hpp:15-22 */
typedef struct {
    float   center[3], target[3], up[3];
    float   yfov, defocus_angle, focus_distance;
    int32_t width, height;
    int32_t x_pixel_samples, y_pixel_samples;
    int32_t max_depth;
} jtx_mi_camera_desc;

typedef struct {
    int32_t sample_begin;     /* first stratum to render; 0 clears the accumulation buffer */
    int32_t sample_end;       /* one past the last stratum; <= 0 => xs*ys */
    int32_t tile_rank;        /* multi-GPU pixel-tile sharding: this process owns 32x32 tiles k with */
    int32_t tile_world;       /*   k % tile_world == tile_rank (0/0 or 0/1 => whole frame) */
    int32_t integrator;       /* 0 auto = the measured policy of DESIGN.md section 5 (1 in practice; env JTX_INTEGRATOR overrides);
                               * 1 pixel-persistent (the timed one), 2 HBM-queued wavefront.  Both give the same film bit for bit. */
    int32_t count_rays;       /* != 0: accumulate jtx_mi_counters on the device (slower); the counting kernels walk the
                               * reference's binary BVH node by node, so the counters are Scene::closestHit / anyHit's own.
                               * 0: same film bit for bit; scenes that do not fit LDS walk an 8-ary quantised BVH instead
                               * (same leaves, same order, same hits: DESIGN.md section 3) */
    int32_t samples_per_tick; /* progress callback granularity for jtx_mi_render; <= 0 => all */
    int32_t reserved;         /* bit 0: time every wavefront kernel with its own HIP events (jtx_mi_kernel_time_by_kind) */
    int32_t path_integrator;  /* which Li (camera.cpp:104-106 picks by (un)commenting a line): 0 integrateMIS (integrator.cpp:171-216,
                               * the timed path), 1 integrate (:54-132: NEE without MIS, emission), 2 integrateBasic (:12-52: emission,
                               * no light sampling).  1 and 2 (and scenes with a THIN_DIELECTRIC material) run in k_render_alt. */
    int32_t frame_slot;       /* 0 .. JTX_MI_FRAME_SLOTS - 1: which of the scene's sets of per-frame working memory (the per-path radiance
                               * records of the persistent path kernel) this render uses.  Renders of one scene in DIFFERENT slots may be
                               * in flight at the same time on different streams -- the workers of StaticCamera::render never wait for a
                               * frame boundary either (camera.cpp:53-64, 81-123): here the last chunks of frame i overlap the first chunks
                               * of frame i + 1.  Renders in the SAME slot are ordered
                               * by the library (an event per slot), whatever streams they come on; launches that need the scene's
                               * singletons (count_rays, integrator 2, path_integrator != 0) are ordered against all slots.  jtx_mi_render
                               * uses slot 0. */
    int32_t sequence_end;     /* != 0: nothing follows this frame that could fill the end of its launch (the last frame of a sequence): the
                               * launch is cut into many small chunks like a lone one, although other frames of the scene are in flight
                               * (with a successor in flight the library prefers few large chunks: fewer fetches, the tail does not matter) */
    int32_t max_record_mb;    /* cap, in MiB, on a frame slot's buffer of per-path radiance records (16 B per path of one launch of the
                               * persistent path kernel: C2's 64-spp frame 2.1 GB); <= 0: 8192 (env JTX_MAX_RAD_MB: the same for callers
                               * that pass no opts).  A range of strata that would need more goes in several launches of consecutive
                               * strata, the film continuing in order */
} jtx_mi_render_opts;

/* ray / traffic counters */
typedef struct {
    uint64_t n_camera, n_closest, n_any;
    uint64_t n_nodes_closest, n_tri_closest, n_accept;
    uint64_t n_nodes_any, n_tri_any, n_shade;
    /* by BxDF class = the code path a material's calls take: 0 DIFFUSE, 1 DIELECTRIC rough, 2 CONDUCTOR rough, 3 METALLIC_ROUGHNESS
     * (material.hpp:5-23), 4 ThinDielectricBxDF (dielectric.hpp:163-207), 5 DIELECTRIC smooth or index-matched (dielectric.hpp:44),
     * 6 CONDUCTOR smooth (conductor.hpp:33; smooth: max(alpha) < 1e-3, microfacet.hpp:23-25), 7 unused: sampleBxdf calls (they sum to
     * n_shade) and evalBxdf + pdfBxdf calls -- in integrateMIS the light samples that were not occluded (integrator.cpp:151-166), in
     * integrate every light sample with pdf > 0 (:94-101) */
    uint64_t n_shade_class[8], n_eval_class[8];
} jtx_mi_counters;

typedef struct {
    int32_t num_nodes, num_prims, max_depth, lds_resident;
    float   scene_radius;
    int32_t auto_integrator;  /* what opts.integrator == 0 resolves to for this scene */
    uint64_t device_bytes;
    int32_t wide_depth;       /* levels of the 8-ary quantised BVH the uncounted kernels walk (0 = not built) */
    int32_t wide_bytes;       /* its size in bytes */
    int32_t refitted;         /* != 0: boxes / triangles come from jtx_mi_scene_refit (topology of the last build), not from a build */
    int32_t num_cus;          /* compute units of the scene's device (hipDeviceAttributeMultiprocessorCount) */
    int32_t resident_workgroups;   /* workgroups of the persistent k_render_paths grid for this scene on this device (upper limit: a launch
                                    * never has more workgroups than chunks) */
    int32_t workgroup_size;   /* lanes per workgroup of that kernel: 256 when the scene is staged in LDS, 64 otherwise */
    int32_t device_built;     /* != 0: the tree comes from jtx_mi_scene_rebuild (the reference's tree, node for node and primitive for primitive) */
    uint64_t wide_bytes64;    /* wide_bytes without the 2 GiB clamp of the 32-bit field; 0 when the kernels have no 8-ary nodes to walk */
    uint64_t rebuild_spare_bytes;  /* device memory held by the edit loop's second set of structures + the builder's scratch (NOT part of
                                    * device_bytes): 0 before the first jtx_mi_scene_rebuild / reserve_rebuild and after release_rebuild */
    uint64_t frame_slot_bytes;     /* device memory held by the frame slots' working memory -- the per-path radiance records of the persistent
                                    * path kernel and its chunk counters, allocated by the first render that uses a slot and sized by the
                                    * largest launch since (NOT part of device_bytes; the reference's only per-frame state is acc_ / img_,
                                    * image.hpp:64-92): 0 after jtx_mi_scene_release_frames */
} jtx_mi_scene_info;

typedef struct jtx_mi_scene jtx_mi_scene;

/* return non-zero to abort (replaces Camera::terminateRender / stopRender_, camera.hpp:77) */
typedef int (*jtx_mi_progress_cb)(int32_t current_sample, int32_t total_samples, void *user);

const char *jtx_mi_last_error(void);
int         jtx_mi_version(void);
int         jtx_mi_device_count(int32_t *count);
int         jtx_mi_set_device(int32_t device);

/* Host-only: Scene::buildBVH (scene.cpp:96-135) = buildTree (bvh.cpp:9-133) + flattenBVH (bvh.cpp:135-149).
 * nodes_out needs room for 2*num_tri_refs nodes, refs_out for num_tri_refs.  No GPU needed. */
int jtx_mi_bvh_build(const jtx_mi_scene_desc *desc, jtx_mi_bvh_node *nodes_out, int32_t *num_nodes_out,
                     jtx_mi_tri_ref *refs_out, int32_t *max_depth_out);

/* Host-only: JPEG (baseline + progressive Huffman, 8 bit) -> interleaved 8-bit samples with the arithmetic of the decoder
 * the reference uses for textures (stbi_load_from_memory, image.cpp:97: its fixed-point IDCT, chroma filters and YCbCr
 * conversion), so that texel values are the reference's.  components: 3 for a colour image, 1 for greyscale.
 * out == NULL: only width / height / components are filled in.  No GPU needed. */
int jtx_mi_decode_jpeg(const uint8_t *bytes, int64_t num_bytes, int32_t *width, int32_t *height, int32_t *components,
                       uint8_t *out, int64_t capacity);

/* Host-only: PNG -> interleaved 8-bit samples as stbi_load_from_memory(.., req_comp = 0) of the reference's stb_image returns
 * them (image.cpp:70,97): components 1 (grey), 2 (grey + alpha), 3 (RGB / palette), 4 (RGBA / palette + tRNS); a tRNS colour
 * key adds the alpha channel; 1 - 16 bits, Adam7.  out == NULL: only width / height / components.  No GPU needed. */
int jtx_mi_decode_png(const uint8_t *bytes, int64_t num_bytes, int32_t *width, int32_t *height, int32_t *components,
                      uint8_t *out, int64_t capacity);

/* Host-only: OpenEXR -> RGBA float, rows top to bottom: what TextureImage::load gets from tinyexr's LoadEXR /
 * LoadEXRFromMemory (image.cpp:63-66, 81-95, 108-121; channels R, G, B, optional A else 1.0; a single channel goes to all
 * four outputs).  Single-part scan-line files, compression NONE / RLE / ZIPS / ZIP / PIZ (the reference's maps are ZIP), HALF /
 * FLOAT / UINT samples; tiled, deep, multi-part files and the lossy block types are refused.  rgba_out == NULL: only width /
 * height are filled in; capacity counts floats.  No GPU needed. */
int jtx_mi_decode_exr(const uint8_t *bytes, int64_t num_bytes, int32_t *width, int32_t *height, float *rgba_out, int64_t capacity);

/* Host-only: the 8-ary quantised node set the uncounted kernels walk for HBM-resident scenes, derived from the
 * flat nodes of jtx_mi_bvh_build (layout: DESIGN.md "Data layout", 16-byte granules = 4 uint32 each; 0 granules
 * when the tree is a single leaf or cannot be quantised).  granules_out may be NULL to query the size.  No GPU needed. */
int jtx_mi_wide_build(const jtx_mi_bvh_node *nodes, int32_t num_nodes, uint32_t *granules_out, int64_t capacity,
                      int64_t *num_granules_out, int32_t *depth_out);

/* Scene::buildBVH + upload: builds the BVH on the host, bakes transforms, lays nodes / triangles /
 * shading records out for the kernels and copies them to the current device. */
int  jtx_mi_scene_create(const jtx_mi_scene_desc *desc, jtx_mi_scene **out);
void jtx_mi_scene_destroy(jtx_mi_scene *scene);                      /* Scene::destroy */
/* Transform edits (Display::renderScene's edit loop: recalculateTransform -> rebuildBVH_, display.cpp:545-588, 902-905).
 * set_transform stores a mesh's new row-major Mesh::transform; refit recomputes, ON THE DEVICE and with the topology of the
 * last build kept, every position-dependent record -- triangles, shading normals, all node boxes (binary, the 8 stackless
 * orderings, the re-quantised 8-ary nodes), scene radius -- in a fraction of a host rebuild.  Boxes are bit for bit what a
 * build computes for the same primitive sets; the TOPOLOGY is not re-chosen, so the frame is a correct render of the edited
 * scene but is not claimed bit-identical to the reference's frame after Scene::rebuildBVH (scene_info.refitted says so;
 * jtx_mi_scene_create on the edited description gives the reference-identical tree). */
int  jtx_mi_scene_set_transform(jtx_mi_scene *scene, int32_t mesh, const float *m16);
int  jtx_mi_scene_refit(jtx_mi_scene *scene);
/* Scene::rebuildBVH(maxPrimsInNode) (scene.hpp:66-69) after transform edits, ON THE DEVICE: a NEW topology for the edited geometry.
 * The binned-SAH build of bvh.cpp:9-133 runs breadth first over the resident primitives (12 centroid buckets, 11 split costs, the
 * reference's fp32 arithmetic for every decision; jtx_build_dev.hip) and gives the reference's tree node for node: boxes, split
 * axes, child order, leaves, flattenBVH's depth-first numbering (bvh.cpp:135-149) -- and primitive for primitive: std::partition's
 * swaps are a function of the predicate flags alone and are replayed in parallel, so Scene::triangles_ comes out in the order the
 * host build (libstdc++) gives.  Everything derived from the tree follows on the device (the 8 stackless
 * orderings, the 8-ary quantised nodes with their surface-area cut, triangle / shading records, refit sources).
 * max_prims_in_node <= 0: 1. */
int  jtx_mi_scene_rebuild(jtx_mi_scene *scene, int32_t max_prims_in_node);
/* The edit loop's memory, ahead of the first edit (display.cpp:545-588 arms rebuildBVH_, :902-905 rebuilds): jtx_mi_scene_rebuild
 * writes a SECOND set of every structure it replaces and swaps the two when all of it stands (a failed rebuild leaves the scene as it
 * was); that set and the builder's scratch come into being with the first rebuild (~330 MB for 256 k triangles; device memory is
 * mapped at first touch and the builder's code object loaded at its first launch: 26-33 ms against 7 ms for every later rebuild)
 * -- or here, e.g. right after loading: a dry run of the rebuild that stops before the commit, so that the first edit costs what
 * every later one does.  The scene is not changed.  Optional; may be called again. */
int  jtx_mi_scene_reserve_rebuild(jtx_mi_scene *scene);
/* ... and back: frees that second set, the builder's scratch and the page-locked landing buffers (scene_info.rebuild_spare_bytes says
 * how much; about the geometry's own size again).  For a host that has finished editing, or holds many scenes on one device.  The next
 * rebuild allocates them again (and pays the first rebuild's price once more). */
int  jtx_mi_scene_release_rebuild(jtx_mi_scene *scene);
/* Frees the frame slots' working memory (scene_info.frame_slot_bytes; e.g. a host that drops from three frames in flight to one, or
 * holds many scenes on one device).  Waits for the device to go idle first.  The next render allocates what it needs again. */
int  jtx_mi_scene_release_frames(jtx_mi_scene *scene);
int  jtx_mi_scene_get_info(const jtx_mi_scene *scene, jtx_mi_scene_info *out);
int  jtx_mi_scene_get_bvh(const jtx_mi_scene *scene, jtx_mi_bvh_node *nodes_out, jtx_mi_tri_ref *refs_out);
/* The scene's 8-ary node set as it stands on the device -- after jtx_mi_scene_create, a device rebuild or a refit -- in the layout of
 * jtx_mi_wide_build (read back for inspection: tests walk it against the binary tree).  granules_out may be NULL to query the size;
 * 0 granules when the scene has none.  No counterpart in the reference (whose only acceleration structure is LinearBVHNode, bvh.hpp:7-15). */
int  jtx_mi_scene_get_wide(jtx_mi_scene *scene, uint32_t *granules_out, int64_t capacity, int64_t *num_granules_out);

/* StaticCamera::render(const Scene&) (camera.cpp:45-128), blocking.  acc_rgb: W*H*3 float sums
 * (AccumulationBuffer), img_rgb: W*H*3 u8 (RGB8Image); both HOST buffers owned by the caller.
 * With a callback, one pass = opts.samples_per_tick strata (samplesPerPass_, camera.hpp:181).  All passes of a frame go in ONE launch
 * of the persistent path kernel (integrateMIS; counting launches, the alternate Li and integrator 2 go pass by pass; a range whose records
 * exceed opts.max_record_mb, or of more than 32 767 passes, goes in several such launches, one after the other), a second kernel
 * beside it adding every finished pass to the film and the preview, in order.  cb(current_sample, total, user) runs once per pass, in
 * order, when that pass is in the film of EVERY pixel, while the launch goes on rendering; img_rgb then holds the preview as it
 * stood at that moment (what the UI uploads, display.cpp:702-703): every pixel shows at least current_sample strata, some already a
 * pass more -- the reference's UI reads img_ unsynchronised beside the tile workers as well.  The launch does not wait for the
 * callback: a slow one is told of several passes in a row.  acc_rgb is written once, before the call returns.
 * Cancellation (Camera::terminateRender, camera.hpp:77; the reference polls stopRender_ per pixel, camera.cpp:84-98):
 * a non-zero return of the callback, or jtx_mi_cancel() from ANY thread at ANY time -- the persistent waves poll the
 * flag when they fetch chunks of 64 x strata paths, so a running launch stops within microseconds of work per wave; a pass
 * that is unfinished then leaves no trace in the film.  Returns JTX_MI_CANCELLED then: acc_rgb / img_rgb hold EXACTLY the strata
 * [0, n) of every pixel -- what a render of sample_end = n gives, bit for bit -- and jtx_mi_last_completed_sample tells n (passes
 * that were already in flight when the stop arrived may be in it); 0 when the frame is complete.
 * Threads: calls on ONE scene are serialised by the library, but the callback runs with the scene unlocked (it may call
 * jtx_mi_cancel, jtx_mi_last_completed_sample, jtx_mi_scene_get_info): do not start another render or an edit of the same scene
 * from it or beside it -- one blocking render per scene at a time (jtx_mi_render_device with opts.frame_slot is the interface for
 * several frames of a scene in flight).  Different scenes are independent. */
int jtx_mi_render(jtx_mi_scene *scene, const jtx_mi_camera_desc *cam, const jtx_mi_render_opts *opts,
                  float *acc_rgb, uint8_t *img_rgb, jtx_mi_progress_cb cb, void *user);
int jtx_mi_cancel(jtx_mi_scene *scene);                                     /* Camera::terminateRender(); thread-safe */
/* Optional: page-lock a caller buffer (Camera::img_ / acc_, image.hpp:60-61,90-91) for as long as it lives, so that
 * jtx_mi_render DMA-writes it directly; pageable buffers work too (one extra host copy from the library's pinned staging).
 * Page-locking works on whole pages: give such a buffer pages of its own (posix_memalign to 4096 with the size rounded up to pages,
 * or an anonymous mmap) -- a plain malloc / std::vector block shares its first and last page with other heap objects -- and call
 * jtx_mi_unpin_host before it is freed.  Host memory the library is handed WITHOUT this call never reaches a GPU copy directly. */
int jtx_mi_pin_host(void *ptr, uint64_t bytes);
int jtx_mi_unpin_host(void *ptr);
int jtx_mi_last_completed_sample(const jtx_mi_scene *scene, int32_t *out);  /* currentSample_ after the last jtx_mi_render */

/* Same, DEVICE buffers, asynchronous on `stream` (a hipStream_t, NULL = the library's own stream).
 * d_acc_rgb must stay valid until the stream is synchronised.  Used by bench.py / multi-GPU.
 * Renders of one scene are ordered per opts.frame_slot (see there): JTX_MI_FRAME_SLOTS frames of a scene can be in flight, one per
 * slot, each on a stream and into film buffers of its own (launches that use the scene's singletons -- count_rays, integrator 2,
 * path_integrator != 0, JTX_DYNAMIC_PATHS=0 -- wait for every slot); renders of DIFFERENT scenes are independent.  jtx_mi_render uses the library's own
 * stream and slot 0.  (jtx_mi_kernel_time pairs are per launch: overlapping launches overlap in time, their durations do not add up
 * to the wall time.)
 * jtx_mi_cancel() also stops a device render in flight (the film then keeps its previous content). */
int jtx_mi_render_device(jtx_mi_scene *scene, const jtx_mi_camera_desc *cam, const jtx_mi_render_opts *opts,
                         void *d_acc_rgb, void *d_img_rgb, void *stream);
/* for jtx_mi_render_device users: the cancellation word of jtx_mi_cancel is sticky; jtx_mi_render clears it itself */
int jtx_mi_cancel_pending(jtx_mi_scene *scene, int32_t *out);   /* after a sync: 0 none, 1 pending (the last pass completed), 2 pending and the last pass was abandoned */
int jtx_mi_cancel_reset(jtx_mi_scene *scene);
int jtx_mi_sync(jtx_mi_scene *scene);
/* GPU time of the integrator kernel(s) of the last render call(s) since the previous query, from HIP events recorded on the launch
 * stream: sum in ms and number of launches.  (With several frames of a scene in flight -- opts.frame_slot -- a launch's pair also times
 * its wait for free wave slots: time single launches with one frame in flight.) */
int jtx_mi_kernel_time(jtx_mi_scene *scene, float *ms_total, int32_t *launches);
/* Wavefront renders with opts.reserved bit 0: summed GPU ms and launch count per kernel kind since the last
 * query: [0] generate, [1] trace closest, [2] shade, [3] trace any (shadow), [4] resolve. */
int jtx_mi_kernel_time_by_kind(jtx_mi_scene *scene, float *ms5, int32_t *n5);
int jtx_mi_get_counters(jtx_mi_scene *scene, jtx_mi_counters *out);  /* of the last count_rays render */

/* ---- one host process, N devices (replaces the thread pool of StaticCamera::render, camera.cpp:55-127) ----
 * jtx_mi_multi_create replicates the scene on every listed device (devices == NULL: 0..n-1; a device may be listed more
 * than once, its shards then share it).  jtx_mi_multi_render is jtx_mi_render over all of them: shard r renders the
 * 32x32 tiles k % n == r, pushes its own pixels to devices[0] over xGMI (one hipMemcpyPeerAsync pair per shard and pass)
 * and devices[0] assembles the frame -- the same bytes one device renders alone.  Same progress / cancellation contract
 * as jtx_mi_render (jtx_mi_multi_cancel from any thread); count_rays and sample_begin > 0 are not supported here.
 * With a callback every shard traces all passes in ONE progressive launch (as jtx_mi_render does); the frame's current_sample is the
 * minimum over the shards, previews are exchanged when it has advanced, the exact film once at the end; after a stop the shards that
 * ended behind the furthest one render on to it, so that every pixel of the frame holds the same strata [0, n). */
typedef struct jtx_mi_multi jtx_mi_multi;
int  jtx_mi_multi_create(const jtx_mi_scene_desc *desc, const int32_t *devices, int32_t n_devices, jtx_mi_multi **out);
void jtx_mi_multi_destroy(jtx_mi_multi *m);
int  jtx_mi_multi_render(jtx_mi_multi *m, const jtx_mi_camera_desc *cam, const jtx_mi_render_opts *opts,
                         float *acc_rgb, uint8_t *img_rgb, jtx_mi_progress_cb cb, void *user);
int  jtx_mi_multi_cancel(jtx_mi_multi *m);
int  jtx_mi_multi_last_completed_sample(const jtx_mi_multi *m, int32_t *out);
/* GPU ms of the integrator kernel per shard in the last jtx_mi_multi_render (n values) */
int  jtx_mi_multi_shard_time(jtx_mi_multi *m, float *ms_per_shard, int32_t n);

/* Fine-grained entry points for parity tests (HOST buffers; blocking). */
/* Scene::closestHit (scene.cpp:10-55): prim = index into the BVH-ordered refs, -1 on miss */
int jtx_mi_closest_hit_batch(jtx_mi_scene *scene, int32_t n, const float *o, const float *d, float tmin, float tmax,
                             int32_t *hit, float *t, int32_t *prim, float *b1, float *b2,
                             float *point, float *normal, float *uv);
/* Scene::anyHit (scene.cpp:57-94) */
int jtx_mi_any_hit_batch(jtx_mi_scene *scene, int32_t n, const float *o, const float *d, const float *tmin,
                         const float *tmax, int32_t *hit);
/* The same two calls through a NAMED traversal structure (Scene::closestHit / anyHit, scene.cpp:10-94; same results by construction --
 * which is what the parity tests check ray by ray, on the code the timed launches run):
 *   JTX_MI_TRAVERSAL_BINARY      the reference's 32-byte nodes as threaded records (what the two calls above and the counting launches walk)
 *   JTX_MI_TRAVERSAL_PRODUCTION  what the TIMED launch of this scene walks: the flat leaf list (<= 32 leaves), the LDS copy of the threaded
 *                                records, the 8-ary quantised nodes (scenes that live in HBM) or the binary records (scenes without wide nodes)
 *   JTX_MI_TRAVERSAL_SOURCE + s  source s (0 binary records in HBM / 1 their LDS copy / 2 8-ary nodes / 3 leaf list); an error if the
 *                                scene does not carry it
 * source_out (may be null): the source that ran (0..3 as above). */
#define JTX_MI_TRAVERSAL_BINARY 0
#define JTX_MI_TRAVERSAL_PRODUCTION 1
#define JTX_MI_TRAVERSAL_SOURCE 2
int jtx_mi_closest_hit_batch_via(jtx_mi_scene *scene, int32_t traversal, int32_t n, const float *o, const float *d, float tmin, float tmax,
                                 int32_t *hit, float *t, int32_t *prim, float *b1, float *b2,
                                 float *point, float *normal, float *uv, int32_t *source_out);
int jtx_mi_any_hit_batch_via(jtx_mi_scene *scene, int32_t traversal, int32_t n, const float *o, const float *d, const float *tmin,
                             const float *tmax, int32_t *hit, int32_t *source_out);
/* sampleBxdf / evalBxdf / pdfBxdf (bsdf/bxdf.cpp:9,79,130) for one material over n inputs */
int jtx_mi_bxdf_sample_batch(jtx_mi_scene *scene, int32_t material, int32_t n, const float *normal, const float *uv,
                             const float *wo, const float *uc, const float *u2,
                             int32_t *ok, float *f, float *wi, float *pdf);
int jtx_mi_bxdf_eval_batch(jtx_mi_scene *scene, int32_t material, int32_t n, const float *normal, const float *uv,
                           const float *wo, const float *wi, float *f);
int jtx_mi_bxdf_pdf_batch(jtx_mi_scene *scene, int32_t material, int32_t n, const float *normal, const float *uv,
                          const float *wo, const float *wi, float *pdf);
/* Camera::getRay (camera.hpp:127-139) seeded as camera.cpp:101 */
int jtx_mi_camera_rays(const jtx_mi_camera_desc *cam, int32_t n, const int32_t *row, const int32_t *col,
                       const int32_t *sample, float *o, float *d);
/* integrateMIS (integrator.cpp:171-216) + the <=1 clamp (camera.cpp:110-112) per listed (row,col,sample) */
int jtx_mi_radiance_samples(jtx_mi_scene *scene, const jtx_mi_camera_desc *cam, int32_t n, const int32_t *row,
                            const int32_t *col, const int32_t *sample, float *rgb);
/* the same for integrate (li = 1) / integrateBasic (li = 2) / integrateMIS with every BxDF incl. THIN_DIELECTRIC (li = 0) */
int jtx_mi_radiance_samples_li(jtx_mi_scene *scene, const jtx_mi_camera_desc *cam, int32_t li, int32_t n, const int32_t *row,
                               const int32_t *col, const int32_t *sample, float *rgb);
/* RNG (util/rand.hpp:42-107) streams and the deterministic sin/cos, for known-answer tests */
int jtx_mi_rng_stream(uint32_t x, uint32_t y, uint32_t n, int32_t count, uint32_t *out_u32, float *out_f32);
int jtx_mi_sincos_batch(const float *x, int32_t n, float *out_sin, float *out_cos);

/* ---- environment variables the library reads (each once, at first use; none is needed: defaults in brackets) ----------------------
 * Choice of code path (all paths give the same film bit for bit):
 *   JTX_INTEGRATOR=1|2           what opts.integrator == 0 resolves to [the measured policy: 1]
 *   JTX_DYNAMIC_PATHS=0          uncounted integrator-1 renders through the one-lane-per-pixel kernel instead of the persistent path kernel [1]
 *   JTX_LEAF_WALK=0              LDS-resident scenes with <= 32 leaves walk the LDS copy of the binary records instead of the flat leaf list [1]
 *   JTX_NO_WIDE=1                scenes that do not fit LDS walk the binary records instead of the 8-ary quantised nodes [unset]
 *   JTX_WIDE_SAH_CUT=0           8-ary nodes cut by depth instead of by summed box area [1]
 *   JTX_WF_SORT_SHADE=1          integrator 2: one shade launch per Material::type (the material-sorted queues) [0]
 *   JTX_PROGRESSIVE_LAUNCH=0     jtx_mi_render with a callback goes pass by pass (a launch per pass) instead of one launch for all passes [1]
 * Tuning (measured defaults; DESIGN.md / EXPERIMENTS.md say where they come from):
 *   JTX_STRATA_GROUPS=n          strata groups per 8x8 pixel block of a k_render_paths launch [by frame size: ~250 k chunks per lone launch]
 *   JTX_MAX_RAD_MB=n             opts.max_record_mb for callers that pass no opts [8192]
 *   JTX_WF_BATCH=n               integrator 2: strata per batch [by frame size]
 *   JTX_RESOLVER_WGS=n           workgroups of the progressive resolver, 2 .. 128 [64]
 *   JTX_PROG_MIN_STRATA=n        progressive launches group passes shorter than n strata into chunks of >= n [4]
 *   JTX_BVH_THREADS=n            host threads of jtx_mi_bvh_build / jtx_mi_scene_create's build [hardware threads, at most 32]
 * Developer aids (stderr / a file; no effect on results):
 *   JTX_TRACE_CREATE, JTX_TRACE_RENDER   host-side phase times of scene creation / of jtx_mi_render
 *   JTX_ABORT_LOG=<file>         a back trace into <file> when the process aborts or terminates on an uncaught exception
 * (libjtx_mi_testhooks.so -- test infrastructure, never the product -- also reads JTX_FAIL_REBUILD_BEFORE_COMMIT, JTX_TEST_REFUSE_PEER_ACCESS,
 *  JTX_TEST_PROGRESSIVE_ONE_STREAM, JTX_TEST_PROGRESSIVE_NO_PATH_KERNEL, JTX_TEST_PROGRESSIVE_DELAY_PATH_MS and JTX_TEST_RESOLVER_PATIENCE_MS.) */

#ifdef __cplusplus
}
#endif
#endif /* JTX_MI_H */
