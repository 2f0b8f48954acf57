// prt_kernels.h — POD types shared by the HIP kernels (prt_kernels.hip) and the C-ABI host code
// (prt_api.cpp), plus the launcher prototypes.  No kernel syntax here.
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stdint.h>

#include "../../include/prt.h"

struct f3 {
    float x, y, z;
};

struct DevPrim {  // 28 dwords; mat/inv keep rows 0..2 of glm's column-major mat4 (cols 0..3)
    uint32_t shape_type;
    float p0, p1;
    uint32_t material;
    float mat[12];  // mat[c*3 + r]
    float inv[12];
};

struct DevCamera {  // the Camera members GetCameraRay reads (reference: src/core/camera.h:134-141)
    f3 pos, front, right, up;
    float W, H, tan_fov_y;
};

// One placed mesh copy (PrtInstance) as the kernels see it: 128 B.  mat / inv: rows 0..2 of the column-major mat4
// (like DevPrim).  root: its mesh's root in nodes8; slot_base: first triangle slot of its mesh in tris / tri_normals;
// prim_base: global primitive index of its first triangle (tie-break order); virt_base: first hit id of this copy
// minus n_prims (hit id = n_prims + virt_base + slot - slot_base).
struct DevInstance {
    float mat[12];
    float inv[12];
    uint32_t root, slot_base, prim_base, virt_base;
    uint32_t material, n_tris;
    float inv_scale;  // 1 / uniform scale of mat
    float extent;     // max |coordinate| of the mesh in its own space (culling pad)
};

struct DevScene {
    const DevPrim* prims;
    const float4* mat_rgbs;     // rgb + scalar
    const uint32_t* mat_type;
    const float4* nodes;        // 4 x float4 per BVH2 node (see bvh.h)
    const float4* nodes4;       // 8 x float4 per BVH4 node (see bvh.h)
    const uint4* nodes8;        // 5 x uint4 per compressed 8-wide node (see bvh.h); null if the tree has none
    const float4* tris;         // 3 x float4 per triangle, leaf order: {P0, prim}, {P1, material}, {P2, 0}
    const float4* tri_normals;  // 3 x float4 per triangle, leaf order
    // BVH over the ANALYTIC primitives' world boxes (scenes with many of them, e.g. the RANDOM_BALLS presets: the
    // reference scans all of them per ray, primitive.cpp:26): 4-wide nodes in the bvh.h layout, leaf slot -> primitive
    const float4* abvh_nodes;   // null: the producers scan the primitives linearly
    const uint32_t* abvh_order;
    const DevInstance* insts;   // placed mesh copies; with n_insts > 0 nodes8 starts with a top-level tree over them
    const uint32_t* tlas_inst;  // top-level leaf slot -> instance index
    uint32_t n_insts;
    uint32_t node_stride;       // uint4 per 8-wide node slot: 5 (packed, 80 B) or 8 (one node per 128-B line: big trees, see upload_scene)
    uint32_t depth8;            // levels of the 8-wide tree (two-level scenes: top level + deepest mesh tree)
    uint32_t n_prims;
    uint32_t n_nodes;
    uint32_t n_tris;
    float pad;     // culling pad coefficient (2^-18): pad_ray = pad * (|o|_1 + extent)
    // primitive walk only: + (abvh_q[0] * A + abvh_q[1]) * A + abvh_q[2] with A = |o|_1, an upper envelope of
    // K / R_i * (A + |c_i|_1)^2 over the scene's spheres (world radius R_i, centre c_i): how far OUTSIDE a sphere a ray may
    // pass and still be a hit in the reference's fp32 arithmetic (see prt_set_scene)
    float abvh_q[3];
    float extent;  // max |coordinate| of any mesh vertex
    float root_min[3], root_max[3];  // bounds of all triangles (BVH root box)
    float sky[3];
};

struct PrtTileMap {
    uint32_t W, H, tiles_x, tiles_y, rank, world;
    uint32_t n_tiles_local;  // tiles owned by this rank
    uint32_t n_pix_local;    // n_tiles_local * 64
    uint32_t stride;         // ceil(tiles_total / world) * 64: per-rank payload (float4 units), equal on all ranks
};

// Tunables of k_traverse_persistent (prt_set_param).
struct PrtTravTuning {
    uint32_t grid_blocks;  // resident 256-thread blocks of the persistent grid
    uint32_t chunk;        // rays a wave grabs per global atomic (multiple of 64)
    uint32_t refill_min;   // idle lanes of a wave that trigger a refill
    uint32_t exit_max;     // leave the node loop when at most this many lanes still search for a leaf (0xFFFFFFFF = per instance: 32 two-level, 16 otherwise)
    uint32_t xcd_affinity; // 4-wide / binary kernels, A/B: 1 = each XCD drains its own eighth of the ray buffer first (L2 locality), then steals
    uint32_t wide;         // 2: walk the compressed 8-wide tree (default), 1: the 4-wide tree, 0: the binary tree
    uint32_t tri_min;      // 8-wide kernel: start a triangle phase once this many lane-steps have queued triangles (0 = per tree: 12 for one-level trees with one node per 128-B line, 24 otherwise)
    uint32_t fuse;         // k_shade: 1 = shade one analytic-only segment in place per call (default), 0 = store every ray
    uint32_t stack_cap;    // test hook: the 8-wide kernel treats its stack as this many entries (0 = all of them)
    uint32_t stack_lds;    // selects the kernel instance (stack entries in LDS / waves per SIMD), see prt_launch_traverse
    uint32_t exact_grids;  // host: size k_shade's grid from the bounce's ray count read back during the traversal (big batches)
    uint32_t steal;        // 8-wide kernel at 5 waves/SIMD: a draining wave with at least this many idle lanes lets them take pending subtrees of its remaining rays (0 = off)
    uint32_t tail;         // 8-wide kernel: the last `tail` 64-ray granules per resident wave are handed out one at a time
    uint32_t probe_slot;   // instrumented instance only: this launch's timeline goes to stats[16 + 8 * probe_slot ..] (see PRT_TIMELINE)
    const uint32_t* perm;  // measurement aid (sort_rays): the 8-wide kernel takes ray perm[i] where it would take ray i (nullptr = identity)
    uint32_t path_kernel;  // host: 0 = off (default); 1 = a batch of ONE sample with at most path_max paths runs as one launch of the path instance of the 8-wide kernel (below); 2 = any batch of at most path_max paths
    uint32_t path_max;
    uint32_t big;          // 8-wide kernel: launches with >= big_min granules per resident wave hand out the front of their bulk `big` chunks per grab (1 = off)
    uint32_t static_small; // 8-wide kernel: launches of single granules only (< 8 per resident wave) deal them to the waves round-robin instead of through the cursor
    uint32_t big_min;      // (granules per resident wave)
    uint32_t big_keep;     // granules per resident wave at the end of the bulk that stay ordinary chunks
    uint32_t primary_hit;  // host: with compact primary rays, rebuild the primary hit's surface interaction once per pixel (k_primary_hit); 0 = per sample in k_shade (A/B)
};

// Timeline of one launch of the instrumented 8-wide kernel, in s_memrealtime ticks (100 MHz), 8 words per launch:
// (of XCD 0's waves:) [0] first wave start (min), [1] last wave end (max), [2] / [3] first / last wave to find the ray buffer exhausted,
// [4] sum over waves of (end - exhausted) = wave time spent draining, [5] sum over waves of (end - start), [6] waves,
// [7] node steps of the longest ray
#define PRT_TIMELINE_WORDS 8
// k_accumulate's per-depth ray counters exist PRT_RAY_STAT_SLOTS times ([slot][PRT_MAX_DEPTH], block b adds to slot
// b mod SLOTS); the host sums the slots when it reads them
#define PRT_RAY_STAT_SLOTS 256u

struct PrtRayBuf {
    float4* o;      // origin.xyz, path id
    float4* d;      // direction.xyz, rng state
    float4* t;      // throughput.rgb, -
    uint32_t* hit;  // closest-hit id so far (producer: analytic scan; traversal: final)
    float* hd2;     // its world distance^2
};

// Compact primary rays.  Without jitter the primary ray of a path, its RNG seed, throughput (1,1,1) and segment index
// (0) are functions of the path id alone, so k_raygen stores 12 B per path (path id, the analytic scan's hit id and
// distance) instead of 56 B plus ONE 16-B record per pixel (direction, pixel index), and the first bounce's traversal
// and k_shade rebuild the ray from those (C3: 372 M stored primary rays per 256-spp batch = 16 GB less written by
// k_raygen and 12 GB less read by the first k_shade; round 3: the analytic scan's hit of a primary ray is per pixel too, so a
// path's slot holds its id only).  Every primary ray is still traced on its own.  pid aliases the
// `t` array of the ray buffer (first 4 bytes per slot).
struct PrtPrimary {
    const uint32_t* pid;  // path id per ray slot
    const float4* pix;    // per local pixel: camera-ray direction, pixel index y * W + x (bits); then n_pix_local more records:
                          // what the pixel's paths deliver if they end with their primary ray (read by k_accumulate; for
                          // pixels whose paths go on: x = ray slot of the first stored sample); then 2 x n_pix_local more:
                          // the primary hit's surface interaction (k_primary_hit): {position, hit id}, {normal, material | front << 31}
    float origin[3];      // camera position
    uint32_t n_pix_local;
    float inv_n;          // 1 / n_pix_local (first guess of path id / n_pix_local, corrected exactly)
    uint32_t first_sample, seed;
};

// The PATH instance of k_traverse8_persistent (small batches: the reference's one sample per ProgressiveRender call): ONE
// persistent launch carries whole paths.  A lane generates its path's primary ray, walks it, shades the hit when the walk is
// over (advance_path, the code of k_shade) and goes on with the scattered ray, until the path ends (rad[path] written) and
// the lane takes the next path.  What the launch needs beyond the scene:
struct PrtPathArgs {
    DevCamera cam;
    PrtTileMap tm;
    PrtSampling sp;
    float4* rad;
    uint32_t first_sample, seed, max_depth, n_paths;
};

#define PRT_CNT_STRIDE 64u  // uint32 per bounce in the counter array: [0] front, [32] back, [16] finished-in-producer counts

void prt_launch_raygen(hipStream_t st, const DevScene& sc, const DevCamera& cam, const PrtTileMap& tm, uint32_t n_paths,
                       uint32_t first_sample, uint32_t seed, const PrtRayBuf& out, float4* rad, uint32_t* counts,
                       uint32_t* work, uint32_t max_depth, const PrtSampling& sp, float4* compact_pix = nullptr);
void prt_launch_scan_prims(hipStream_t st, const DevScene& sc, const PrtRayBuf& in, const uint32_t* count_ptr,
                           uint32_t* work, uint32_t max_rays, unsigned long long* stats);
void prt_launch_traverse(hipStream_t st, const DevScene& sc, const PrtRayBuf& in, const uint32_t* count_ptr,
                         uint32_t* work, uint32_t* spill, uint32_t max_rays, uint32_t tree_depth, uint32_t stack4,
                         const PrtTravTuning& tune, unsigned long long* stats, const PrtPrimary* primary = nullptr);
// one launch for a whole batch of paths (PrtPathArgs); `work` (cursors + error flags) must have been zeroed on the stream
void prt_launch_path(hipStream_t st, const DevScene& sc, const PrtPathArgs& pa, uint32_t* work, const PrtTravTuning& tune);
bool prt_path_kernel_applies(const DevScene& sc, const PrtTravTuning& tune);
// true if prt_launch_traverse would run the instance that can rebuild compact primary rays (and needs no overflow list)
bool prt_traverse_takes_primary(const DevScene& sc, const PrtTravTuning& tune);
int prt_traverse_occupancy(const DevScene& sc, const PrtTravTuning& tune, int* blocks_per_cu, int* vgprs, int* sgprs, int* lds_bytes);
// name of the traversal kernel instance prt_launch_traverse runs for this scene / these tunables (as tools/isa_count.py names them)
const char* prt_traverse_instance(const DevScene& sc, const PrtTravTuning& tune);
void prt_launch_intersect(hipStream_t st, const DevScene& sc, const PrtRayBuf& in, const uint32_t* count_ptr,
                          uint32_t max_rays, int stack_depth, int variant, unsigned long long* stats);
void prt_launch_shade(hipStream_t st, const DevScene& sc, const PrtRayBuf& in, const PrtRayBuf& out, float4* rad,
                      uint32_t* counts, uint32_t* work, uint32_t depth, uint32_t max_depth, uint32_t cap,
                      uint32_t fuse_max, const PrtSampling& sp, uint32_t n_rays_known, const PrtPrimary* primary = nullptr);
// diagnostic: per-wave material mix of what k_shade of bounce `iter` is about to shade (16 words per bounce in `out`)
void prt_launch_shade_divstats(hipStream_t st, const DevScene& sc, const PrtRayBuf& in, const uint32_t* counts, uint32_t iter,
                               uint32_t cap, unsigned long long* out, const PrtPrimary* primary = nullptr);
// compact primary rays: per-pixel surface interaction of the primary hit (records 2n.. and 3n.. of `pix`), between the
// first traversal and the first k_shade of a batch
void prt_launch_primary_hit(hipStream_t st, const DevScene& sc, const PrtPrimary& pr, const uint32_t* hit, float4* pix,
                            const uint32_t* counts);
void prt_launch_accumulate(hipStream_t st, const float4* rad, float4* film_local, const PrtTileMap& tm, uint32_t S,
                           uint32_t max_depth, bool update_film, unsigned long long* ray_stats,
                           const float4* pix_end = nullptr);
void prt_launch_resolve(hipStream_t st, const float4* gathered, uint32_t world, uint32_t stride, uint32_t W,
                        uint32_t H, float* rgb, float* weight);
void prt_launch_tonemap(hipStream_t st, const float* rgb, const float* weight, uint32_t n_pix, float exposure,
                        float inv_gamma, uint8_t* out);
void prt_launch_camera_rays(hipStream_t st, const DevCamera& cam, uint32_t n, const float* px, const float* py,
                            float* o, float* d);
void prt_launch_pack_rays(hipStream_t st, uint32_t n, const float* o, const float* d, const PrtRayBuf& out,
                          uint32_t* counts);
void prt_launch_hit_records(hipStream_t st, const DevScene& sc, uint32_t n, const PrtRayBuf& in, PrtHit* out);
void prt_launch_scatter_test(hipStream_t st, const DevScene& sc, uint32_t n, const float* in_d, const PrtHit* hits,
                             uint32_t* rng_io, uint32_t* scattered, float* atten, float* emitted, float* o_out,
                             float* d_out);
