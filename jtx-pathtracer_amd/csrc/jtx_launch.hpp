// jtx_launch.hpp -- kernel parameter blocks and launcher prototypes shared by jtx_kernels.hip and
// the C-ABI implementation (jtx_capi.hip).
#pragma once
#include "jtx_scene_dev.hpp"
#include <vector>

#ifndef JTX_RP_BLOCK
#define JTX_RP_BLOCK 256      // threads per workgroup of the render kernels (4 waves = 4 of the 16 8x8 pixel blocks of a tile)
#endif

namespace jtx {

struct RenderParams {
    DevScene scene;
    DCam     cam;
    int width, height, max_depth;
    int sample_begin, sample_end;
    int tile_rank, tile_world, tiles_x;
    float *acc;                    // W*H*3 float sums (AccumulationBuffer)
    unsigned char *img;            // W*H*3 u8 (RGB8Image) or null
    unsigned long long *counters;  // 9 x u64 or null
    // strata-split mode (small shards): blockIdx.y picks a group of `strata_per_group` strata, lanes write the
    // clamped per-sample radiance to rad[(s - sample_begin) * rad_stride + owned pixel slot]; k_resolve_samples
    // then adds them to the film in sample order
    float4 *rad; int strata_per_group; int rad_stride;
    // k_render_paths: persistent waves take (8x8 pixel block, strata group) chunks from this counter (zeroed per launch)
    // (Round 5 tried in-kernel launch clocks -- first wave in, last wave out, two atomics per wave -- to time launches that overlap:
    //  their mere presence moved the register allocation of the 8-ary instances: +9 % static instructions, C3 300 -> 308 ms.  Taken
    //  out again: launches are timed by event pairs, and the roofline's durations come from launches with one frame in flight.)
    unsigned *work; int num_subblocks; int num_groups;
    // cancellation (Camera::terminateRender / stopRender_, camera.hpp:77, polled per pixel camera.cpp:84-98): a host-mapped
    // word; persistent waves read it whenever they fetch a chunk and stop handing out paths, k_resolve_samples then leaves
    // the film as the last completed pass left it
    const unsigned *stop;
    // k_resolve_samples only (at the END of the block: the path kernels' argument offsets stay where they were):
    // prev_work: a pass split into several launches (radiance record cap) -- the chunk counter of the launch BEFORE this one, or null.  The
    //   resolve leaves the film alone, and marks its own launch abandoned, when that one was: strata enter the film in order or not at all.
    // abandoned: a host-mapped word of this launch; the resolve sets it when the launch was abandoned, so that the host learns it from its
    //   own memory after the stream has drained -- a 4-byte device-to-host copy is a blit KERNEL, and waits for a wave slot.
    unsigned *prev_work;
    unsigned *abandoned;
    // PROGRESSIVE launches (round 6: jtx_mi_render with a callback; k_render_paths<.., PROG = true> beside k_resolve_progressive): one
    // launch for all passes.  prog_slots: one word per persistent wave -- the oldest stratum the wave still has a path of, or is about
    // to start (0xffffffff: not started yet / gone): every record below the minimum over all waves is written.  prog_closed_at: after a
    // cancellation the number of chunks that were dealt (the counter's value when it was closed), else 0xffffffff.
    // All control words of a progressive launch are ONE allocation with the chunk counter (`work`) at its head, each on cache lines of its
    // own (PROG_CTL_*): the path kernel addresses them from `work` -- two pointers fewer to keep in scalar registers across its loop.
    unsigned *prog_slots;          // = work + PROG_CTL_SLOTS
    unsigned *prog_closed_at;      // = work + PROG_CTL_CLOSED_AT
    unsigned *prog_leader;         // the resolver's own: complete passes | bit 31 when final, from its watching workgroup to the others
    int prog_groups_per_pass;      // the resolver adds whole passes only: strata groups in multiples of this (1 unless a long pass is cut into several groups)
};

constexpr int PROG_CTL_CLOSED_AT = 32, PROG_CTL_LEADER = 64, PROG_CTL_SLOTS = 128;      // words from the chunk counter

// ---- wavefront integrator: slot-indexed SoA buffers in HBM (jtx_wavefront.hip) ----
enum { WF_LIVE = 1, WF_TYPE_SHIFT = 8, WF_TYPE_MASK = 0xf00 };   // flags[]: extension ray pending / hit ready;
                                                                // bits 8-11: 1 + Material::type of the hit (0 = miss)
enum { WF_SH_PENDING = 1, WF_SH_UNOCCLUDED = 2,         // sflags[]: shadow ray to trace / traced and unoccluded
       WF_SH_NF_SHIFT = 4, WF_SH_NF_MASK = 0x70,         //   bits 4-6: components of beta that were inf/NaN at that vertex
       WF_SH_TYPE_SHIFT = 8, WF_SH_TYPE_MASK = 0x700 };  //   bits 8-10: BxDF class of that vertex (bxdfClass: the per-class tally of evalBxdf calls)

struct WfBuffers {
    int   *flags;                                   // WF_LIVE
    int   *sflags;                                  // WF_SH_*
    float *rox, *roy, *roz, *rdx, *rdy, *rdz;       // extension ray (Ray, integrator.cpp:212)
    float4 *hit;                                    // {t, b1, b2, asfloat(prim or -1)}
    float *betax, *betay, *betaz;                   // path throughput
    float *radx, *rady, *radz;                      // path radiance
    unsigned *rng; int *depth;
    float *sox, *soy, *soz, *sdx, *sdy, *sdz, *stmax;   // shadow ray (integrator.cpp:146-150)
    float *pendx, *pendy, *pendz;                   // what the shadow ray adds when unoccluded
};

struct WfParams {
    DevScene scene;
    DCam     cam;
    WfBuffers b;
    int width, height, max_depth;
    int tile_rank, tile_world, tiles_x;
    int sort_shade;                // != 0: closest-hit tags slots with the hit material type, shade runs once per type
    int pixels;                    // owned 32x32 tiles * 1024 (padded)
    int num_slots;                 // pixels * strata per batch
    float *acc; unsigned char *img;
    unsigned long long *counters;
};

// ---- device refit after a transform edit (jtx_refit.hip) ----
struct RefitArgs {
    // geometry
    const float4 *prim_src;    // 5 float4 per BVH-ordered primitive: [v0.xyz v1.x][v1.yz v2.xy][v2.z n0.xyz][n1.xyz n2.x][n2.yz asfloat(mesh) -]
    const float  *mesh_xf;     // 16 floats per mesh, row-major Mesh::transform
    float4 *tris, *shade;      // the kernels' records (jtx_scene_dev.hpp)
    float4 *pbox;              // 2 float4 per primitive [min.xyz -][max.xyz -]
    // binary nodes
    float4 *nbox;              // 2 float4 per node [min.x max.x min.y max.y][min.z max.z asfloat(offset) asfloat(numPrims)]
    const int *leaf_nodes, *level_nodes;
    // derived structures
    float4 *tnodes; const int *rec_node;   // 8 * num_nodes threaded records and their binary nodes
    uint4 *wide; const int *wide_map;      // 16 ints per wide node (WideBuilder::fill)
    int *wide_fail;
    int num_prims, num_nodes, num_leaves, num_wide;
};

} // namespace jtx

// ---- device BVH build (jtx_build_dev.hip): Scene::rebuildBVH on the device ----
namespace jtx {
struct DevBuildBuffers {               // caller-allocated device buffers, sized for np primitives / 2 np nodes
    // in: the primitives in their current order
    const float4 *prim_src, *tris, *shade; const int *orig; const float *mesh_xf; int np, max_prims;
    // out: the primitives in the new leaf order, the nodes in depth-first order, lists for the refit / threading stages
    float4 *prim_src_out, *tris_out, *shade_out; int *orig_out;
    float4 *nbox; void *hnodes; int *order; int *leaf_nodes, *level_nodes; int *pos, *size;
    uint4 *wide; int *wide_map; size_t wide_cap;     // wide == nullptr: no 8-ary nodes wanted (the builder knows the area-optimal cut only:
                                                     // under JTX_WIDE_SAH_CUT=0 the caller builds them on the host, as scene_create does)
};
struct DevBuildArena { void *base = nullptr; size_t cap = 0; };   // the builder's temporaries: owned by the caller, grown on demand, never shrunk
struct DevBuildResult {
    int nn = 0, nleaves = 0, max_depth = 0, num_wide = 0, wide_depth = 0; size_t wide_granules = 0; bool wide_ok = false;
    std::vector<int> level_begin;      // interior nodes by depth: offsets into level_nodes
    const char *declined = nullptr;    // != nullptr: the device builder did not build, for this reason (the caller builds on the host)
};
} // namespace jtx
hipError_t jtx_device_build(const jtx::DevBuildBuffers &b, jtx::DevBuildArena &arena, jtx::DevBuildResult &r, hipStream_t st);
hipError_t jtx_device_build_reserve(int np, jtx::DevBuildArena &arena, hipStream_t st);      // the builder's scratch for np primitives, ahead of the first build
hipError_t jtx_launch_refit_prims(const jtx::RefitArgs &a, hipStream_t st);
hipError_t jtx_launch_build_threaded(const float4 *nbox, const int *pos, const int *size, int nn, float4 *tnodes, int *rec_node, hipStream_t st);
hipError_t jtx_launch_refit(const jtx::RefitArgs &a, const int *level_begin, int num_levels, hipStream_t st);
hipError_t jtx_wf_generate(const jtx::WfParams &p, int s0, int nstrata, hipStream_t st);
hipError_t jtx_wf_trace(const jtx::WfParams &p, int any, int grid, bool count, hipStream_t st);
hipError_t jtx_wf_shade(const jtx::WfParams &p, int grid, bool count, int typeCode, int matMask, hipStream_t st);
hipError_t jtx_wf_resolve(const jtx::WfParams &p, int s0, int nstrata, int write_img, hipStream_t st);

hipError_t jtx_launch_render_pixels(const jtx::RenderParams &p, int num_owned_tiles, bool count, hipStream_t stream);
hipError_t jtx_launch_render_paths(const jtx::RenderParams &p, int num_owned_tiles, int num_cus, hipStream_t stream, int share = 1, bool progressive = false,
                                   int leave_waves = 0);
int jtx_resolve_progressive_waves(int num_workgroups);    // wave slots the resolver takes
// the resolver of a progressive launch (beside k_render_paths<.., PROG>, on a stream of its own): num_path_waves words in p.prog_slots; every workgroup
// writes its word of started_host when it runs and of progress_host (epoch << 16 | groups in the film of its pixels) as it goes
// patience_ms: how long the resolver waits without a chunk fetched or a wave's word moving before it gives up (0: a minute) -- unless the host
// vouches for the path kernel: *keepalive_host == epoch (host-mapped; the resolver clears it, the host's polling loop rewrites it while the path
// kernel has not ended)
hipError_t jtx_launch_resolve_progressive(const jtx::RenderParams &p, int num_owned_tiles, int num_path_waves, int num_workgroups, unsigned *started_host,
                                          unsigned *progress_host, unsigned *keepalive_host, unsigned epoch, hipStream_t stream, unsigned patience_ms = 0);
int jtx_render_paths_waves(const jtx::RenderParams &p, int num_cus, int share, int leave_waves);   // waves jtx_launch_render_paths starts for p
int jtx_render_paths_grid(const jtx::DevScene &sc, int num_cus, int *block_size);   // workgroups the persistent grid holds (host only)
hipError_t jtx_launch_render_alt(const jtx::RenderParams &p, int num_owned_tiles, bool count, int li, hipStream_t stream);
hipError_t jtx_launch_radiance_samples_alt(const jtx::DevScene &sc, const jtx::DCam &cam, int maxDepth, int li, int n, const int *row,
                                           const int *col, const int *sample, float *rgb, hipStream_t stream);
hipError_t jtx_launch_resolve_samples(const jtx::RenderParams &p, int num_owned_tiles, hipStream_t stream);
int jtx_production_source(const jtx::DevScene &sc);      // SRC_* the timed launch of this scene walks
hipError_t jtx_launch_closest_batch(const jtx::DevScene &sc, int src, int n, const float *o, const float *d, float tmin, float tmax,
                                    int *hit, float *t, int *prim, float *b1, float *b2, float *point, float *normal,
                                    float *uv, hipStream_t stream);
hipError_t jtx_launch_any_batch(const jtx::DevScene &sc, int src, int n, const float *o, const float *d, const float *tmin,
                                const float *tmax, int *hit, hipStream_t stream);
hipError_t jtx_launch_bxdf_batch(const jtx::DevScene &sc, int mode, int material, int n, const float *normal, const float *uv,
                                 const float *wo, const float *wi_in, const float *uc, const float *u2, int *ok, float *f,
                                 float *wi_out, float *pdf, hipStream_t stream);
hipError_t jtx_launch_camera_rays(const jtx::DCam &cam, int n, const int *row, const int *col, const int *sample, float *o,
                                  float *d, hipStream_t stream);
hipError_t jtx_launch_radiance_samples(const jtx::DevScene &sc, const jtx::DCam &cam, int maxDepth, int n, const int *row,
                                       const int *col, const int *sample, float *rgb, hipStream_t stream);
hipError_t jtx_launch_rng_stream(uint32_t x, uint32_t y, uint32_t n, int count, uint32_t *out_u32, float *out_f32,
                                 hipStream_t stream);
hipError_t jtx_launch_sincos(const float *x, int n, float *s, float *c, hipStream_t stream);
