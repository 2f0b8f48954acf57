/*
 * prt.h — C-ABI of the MI355X-native wavefront path tracer (libprt.so).
 *
 * This is the drop-in boundary for ONE hot path of Rickyeeeeee/ParallelRayTracing:
 *   camera-ray generation -> closest-hit (BVH traversal + shape intersection)
 *   -> material shade/scatter -> film accumulation.
 * It replaces what the reference's `class Renderer` backends do
 * (reference: src/core/renderer.h:8-16 — Init / ProgressiveRender / SetCamera) and the
 * Film accumulate/display entry points they call (src/core/film.h:10-47).
 *
 * Conventions
 *  - Plain C, no torch / C++ types.  Every call returns 0 on success, nonzero on error;
 *    prt_last_error() gives the message.  No exception crosses this boundary.
 *  - Matrices are column-major float[16], exactly glm::mat4's memory layout
 *    (reference: Transform::m_Mat / m_InvMat, src/core/geometry.h:129-130).
 *  - The film is row-major, row 0 = top of the image, interleaved RGB fp32 *sums* plus a
 *    per-pixel weight (reference: Film::m_Accum / m_Weights, src/core/film.h:54-60).
 *  - All data passed in is COPIED; the library keeps no pointer into caller memory
 *    (the reference backends keep raw non-owning pointers, src/backend/cpu/renderer.cpp:8-15).
 *  - Not thread-safe per context; one context drives one GPU.
 *  - There is NO CPU fallback.  Compute entry points fail with PRT_ERR_NO_DEVICE when no
 *    HIP device is usable.
 */
#ifndef PRT_H
#define PRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRT_VERSION 1

/* status codes */
enum {
    PRT_OK = 0,
    PRT_ERR_INVALID = 1,   /* bad argument / call order */
    PRT_ERR_NO_DEVICE = 2, /* no usable HIP device */
    PRT_ERR_HIP = 3,       /* a HIP runtime call failed */
    PRT_ERR_IO = 4,        /* file could not be read / parsed / written */
    PRT_ERR_NOMEM = 5
};

/* reference: enum class ShapeType, src/core/shape.h:10-15 */
enum { PRT_SHAPE_CIRCLE = 0, PRT_SHAPE_QUAD = 1, PRT_SHAPE_TRIANGLE = 2 };
/* reference: enum MatType, src/core/material_handle.h:13-19 */
enum { PRT_MAT_NONE = 0, PRT_MAT_LAMBERTIAN = 1, PRT_MAT_METAL = 2, PRT_MAT_DIELECTRIC = 3, PRT_MAT_EMISSIVE = 4 };
/* reference: enum class ScenePreset, src/core/scene.h:6-15 */
enum {
    PRT_PRESET_DEFAULT = 0, PRT_PRESET_LIGHT_TEST = 1, PRT_PRESET_MATERIAL_TEST = 2, PRT_PRESET_CORNELL = 3,
    PRT_PRESET_RANDOM_BALLS_SMALL = 4, PRT_PRESET_RANDOM_BALLS_MEDIUM = 5, PRT_PRESET_RANDOM_BALLS_LARGE = 6
};

/* One material.  rgb = albedo (Lambertian, Metal) or emission (Emissive); scalar = roughness
 * (Metal) or refraction index (Dielectric).
 * reference: LambertianMaterial/MetalMaterial/DielectricMaterial/EmissiveMaterial getters,
 * src/core/material.h:38,64-65,102,129. */
typedef struct PrtMaterial {
    uint32_t type;
    float rgb[3];
    float scalar;
} PrtMaterial;

/* One analytic primitive = Shape + Material + Transform.
 * shape_param: CIRCLE {radius, -}; QUAD {width, height}.
 * reference: struct Primitive, src/core/primitive.h:7-12; Circle::getRadius / Quad::GetWidth/GetHeight,
 * src/core/shape.h:25,39-40. */
typedef struct PrtPrimitive {
    uint32_t shape_type;
    float shape_param[2];
    uint32_t material_id;
    float mat[16];
    float inv[16];
} PrtPrimitive;

/* One triangle mesh in WORLD space (its Transform is the identity).  Semantically it is a run of
 * Triangle primitives (src/core/shape.h:50-81) appended to the primitive list after all analytic
 * primitives, in face order; vertex data as Mesh exposes it (src/core/mesh.h:12-14):
 * positions/normals are n_vertices*3 floats, indices n_triangles*3 uint32. normals must not be NULL. */
typedef struct PrtMesh {
    const float* positions;
    const float* normals;
    const uint32_t* indices;
    uint32_t n_vertices;
    uint32_t n_triangles;
    uint32_t material_id;
} PrtMesh;

/* One placed copy of a mesh.  Semantically a run of Triangle primitives that share one Transform:
 * struct Primitive {Shape = Triangle, Material, Transform{Mat, Inv}} (src/core/primitive.h:7-12), intersected exactly as
 * PrimitiveList::Intersect does it (src/core/primitive.cpp:29-43): local origin = Inv * o, local direction =
 * normalize(transpose(mat3(Mat)) * d), Triangle::Intersect in the mesh's own space, position back through Mat, normal
 * through Inv, distance measured in world space.  The transform must be rotation + uniform scale + translation
 * (prt_set_scene rejects anything else): only then is the reference's direction transform a ray transform, so that
 * an acceleration structure in the mesh's space returns what the reference's linear scan returns.
 * `mesh` indexes PrtSceneDesc.instanced_meshes (their own material_id is ignored). */
typedef struct PrtInstance {
    uint32_t mesh;
    uint32_t material_id;
    float mat[16];
    float inv[16];
} PrtInstance;

/* Primitive order (= tie-break order of the closest hit): analytic primitives, then the triangles of `meshes` in mesh
 * and face order, then the triangles of `instances` in instance and face order. */
typedef struct PrtSceneDesc {
    const PrtMaterial* materials;
    const PrtPrimitive* primitives;
    const PrtMesh* meshes;
    uint32_t n_materials;
    uint32_t n_primitives;
    uint32_t n_meshes;
    float sky[3]; /* reference literal (0.4,0.3,0.6): src/backend/cpu/renderer.h:31 */
    const PrtMesh* instanced_meshes; /* meshes that only exist through `instances` (may be NULL) */
    const PrtInstance* instances;
    uint32_t n_instanced_meshes;
    uint32_t n_instances;
} PrtSceneDesc;

/* Pinhole camera; right/up are derived exactly as Camera::Camera does (src/core/camera.h:10-16);
 * vertical FoV is fixed at 1 rad (src/core/camera.h:111). */
typedef struct PrtCameraDesc {
    float position[3];
    float front[3];
    float width;
    float height;
} PrtCameraDesc;

/* Optional sampling upgrades; all zero = the reference CPU backend's behaviour.
 *  jitter   1: the primary ray goes through (x + u1, y + u2), u1/u2 = the path's first two RNG draws (reference OptiX
 *           backend, src/backend/optix/device_programs.cu:172-173); 0: pixel centres (src/backend/cpu/renderer.cpp:45).
 *  rr_depth > 0: Russian roulette (reference roadmap, wavefront.md:98-100): a scatter that would start segment index
 *           >= rr_depth survives with p = clamp(max component of the new throughput, 0.05, 1), one RNG draw after the
 *           material's own; survivors carry throughput / p.
 *  clamp    > 0: every component of the radiance a path delivers is limited to it (wavefront.md:102-104). */
typedef struct PrtSampling {
    uint32_t jitter;
    uint32_t rr_depth;
    float clamp;
} PrtSampling;

/* Closest-hit record of one ray (what Scene::Intersect returns, src/core/surface_interaction.h:6-13,
 * plus the winning primitive index and the world distance^2 the reference minimises,
 * src/core/primitive.cpp:42-48).  prim < 0: miss. */
typedef struct PrtHit {
    int32_t prim;
    uint32_t front_face;
    uint32_t material_id;
    float d2;
    float position[3];
    float normal[3];
} PrtHit;

/* Counters and timings of the render calls since the last prt_reset_stats. */
#define PRT_MAX_DEPTH 64
typedef struct PrtStats {
    uint64_t rays_total;               /* ray segments for which a closest-hit query ran */
    uint64_t rays_per_depth[PRT_MAX_DEPTH];
    uint64_t samples;                  /* ProgressiveRender-equivalents done */
    uint64_t intersect_launches;       /* launches of the dominant (closest-hit) kernel */
    double intersect_ms;               /* summed HIP-event time of those launches (timing enabled) */
    double shade_ms;
    double raygen_ms;
    double accumulate_ms;
    uint64_t bvh_node_visits;          /* only from prt_measure_traversal */
    uint64_t bvh_tri_tests;
    uint64_t prim_tests;
    uint64_t node_lane_slots;          /* 64 x node-loop iterations of all waves: visits / slots = lane efficiency */
    double scan_ms;                    /* analytic-primitive scan kernel (function-level entry points only) */
    uint64_t rays_traversed;           /* prt_measure_traversal: rays that entered the BVH root box (walked the tree) */
    uint64_t tri_lane_slots;           /* 64 x triangle-loop iterations of all waves: tri tests / slots = lane efficiency */
    uint64_t max_stack_used;           /* deepest traversal stack any ray reached (prt_measure_traversal) */
    /* prt_measure_traversal, 8-wide kernel: shader cycles (s_memtime) summed over all waves, by section of the wave's
     * outer loop: finish + refill (+ level switches), node loop, triangle phase */
    uint64_t wave_cycles_refill;
    uint64_t wave_cycles_node;
    uint64_t wave_cycles_tri;
} PrtStats;

typedef struct PrtBvhInfo {
    uint32_t n_nodes;
    uint32_t n_triangles;
    uint32_t max_depth;
    uint32_t max_leaf_size;
    float sah_cost;
    float pad_abs;     /* absolute AABB padding applied (conservative culling) */
    uint64_t node_bytes;   /* bytes of the 4-wide tree the default kernel walks */
    uint64_t tri_bytes;
    uint32_t n_nodes4;     /* 4-wide nodes (128 B each) */
    uint32_t max_stack4;   /* worst-case traversal stack entries of the 4-wide tree */
    uint32_t n_nodes8;     /* compressed 8-wide nodes (80 B each) the default kernel walks; 0 = not available */
    uint32_t depth8;       /* levels of the 8-wide tree (the traversal stacks at most depth8 - 1 node groups) */
    float build_ms;        /* wall time of the BVH construction (device-side build: without the vertex upload) */
    uint32_t built_on_device; /* 1: built by the device-side builder (prt_set_param("gpu_build", 1)) */
    float refit_ms;        /* device time of the last prt_refit_meshes (records + boxes bottom-up + re-quantization), 0 = never */
    uint32_t refits;       /* prt_refit_meshes calls since prt_set_scene */
} PrtBvhInfo;

/* Static wavefront occupancy of the traversal kernel (the dominant kernel) for the current scene. */
typedef struct PrtOccupancy {
    uint32_t blocks_per_cu;        /* resident 256-thread blocks per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor) */
    uint32_t waves_per_cu;         /* = 4 x blocks_per_cu */
    uint32_t max_waves_per_cu;     /* hardware limit (32 on gfx950) */
    uint32_t vgprs;                /* per lane */
    uint32_t lds_bytes_per_block;
    uint32_t compute_units;
    uint32_t resident_grid_blocks; /* blocks of the persistent grid the library launches */
} PrtOccupancy;

typedef struct PrtContext PrtContext;

/* ---- lifetime ---------------------------------------------------------------------------------- */
/* device_id < 0: host-only context (scene/mesh/BVH utilities work, compute calls fail loudly).
 * Side effect of every call that touches the device: it makes the context's GPU the CALLING THREAD's current HIP device
 * (hipSetDevice) and leaves it so; a host that holds several devices (a torch process, say) restores its own afterwards
 * or calls from a thread of its own, as prt_group_* does. */
int prt_create(int device_id, PrtContext** out);
void prt_destroy(PrtContext* ctx);
const char* prt_last_error(const PrtContext* ctx);
int prt_version(void);
/* Launch on this hipStream_t (e.g. torch's current stream); NULL = the context's own stream. */
int prt_set_stream(PrtContext* ctx, void* hip_stream);
/* The hipStream_t the context launches on / its device id (-1: host-only context). */
int prt_get_stream(PrtContext* ctx, void** hip_stream);
int prt_get_device(const PrtContext* ctx);

/* ---- Renderer::Init / SetCamera (src/core/renderer.h:13-15) ---------------------------------- */
/* Flattens + uploads the scene and builds the BVH over all mesh triangles.  Replaces
 * BuildWavefrontSceneBuffers (src/backend/cuda_wavefront/soa.cpp:37-114). */
int prt_set_scene(PrtContext* ctx, const PrtSceneDesc* scene);
/* Replicates the scene `src` holds (flattened primitives, trees, triangle records) onto dst's device without building
 * anything again: the scene is read-only and every GPU of a multi-GPU render needs its own copy. */
int prt_clone_scene(PrtContext* dst, const PrtContext* src);
/* Deforming geometry (SURVEY.md §8f-3 "refit"; the reference rebuilds its OptiX acceleration structures from scratch,
 * src/backend/optix/renderer.cpp:736-871): the scene's world-space meshes again, with NEW positions / normals and the SAME
 * vertex counts, triangle counts and index buffers as at prt_set_scene.  The compressed 8-wide tree keeps its topology; its
 * triangle records are rewritten and its boxes refitted bottom-up and re-quantized on the device.  Scenes with placed
 * copies (PrtInstance) are not refitted (PRT_ERR_INVALID).  After a refit the A/B kernels over the binary / 4-wide
 * trees are unavailable (those trees are dropped).  Results equal a fresh prt_set_scene of the new geometry bit for bit
 * (the closest hit does not depend on the tree); traversal gets slower as the deformation grows. */
int prt_refit_meshes(PrtContext* ctx, const PrtMesh* meshes, uint32_t n_meshes);
int prt_set_camera(PrtContext* ctx, const PrtCameraDesc* cam);
/* Film::Resize + Clear (src/core/film.cu:11-35) and the image partition of this context:
 * 8x8-pixel tiles, tile t (row-major) belongs to rank t % world_size. */
int prt_set_film(PrtContext* ctx, uint32_t width, uint32_t height, uint32_t rank, uint32_t world_size);
int prt_film_clear(PrtContext* ctx);

/* ---- Renderer::ProgressiveRender (src/core/renderer.h:14) ------------------------------------- */
/* Adds `spp` samples per pixel (sample indices first_sample .. first_sample+spp-1) to the film.
 * spp = 1 is exactly one ProgressiveRender.  max_depth = max ray segments per path
 * (reference m_Depth = 20, src/backend/cpu/renderer.h:34).  Result is independent of batching and
 * of world_size because the RNG is keyed by (global pixel, sample, seed). Synchronous on return. */
int prt_render(PrtContext* ctx, uint32_t spp, uint32_t max_depth, uint32_t seed, uint32_t first_sample);
/* Same, but only enqueues on the stream (no host sync). */
int prt_render_async(PrtContext* ctx, uint32_t spp, uint32_t max_depth, uint32_t seed, uint32_t first_sample);
int prt_synchronize(PrtContext* ctx);
/* Sampling upgrades for the following prt_render calls (NULL = all off). */
int prt_set_sampling(PrtContext* ctx, const PrtSampling* sampling);
/* Samples kept in flight together (paths = local pixels * n); default 1. */
int prt_set_samples_in_flight(PrtContext* ctx, uint32_t n);

/* ---- Film read-back (Film::m_Accum / m_Weights; src/core/film.h:54-60) ------------------------ */
/* Whole film to host, row-major, top-left origin; only pixels owned by this rank are non-zero. */
int prt_film_read(PrtContext* ctx, float* rgb_sum, float* weight);
/* Device pointer + float count of this rank's tile-ordered accumulation ([local_tile][64][4] =
 * r,g,b,weight) — the payload of the per-frame RCCL gather. Padded so every rank has the same count. */
int prt_film_local(PrtContext* ctx, void** d_ptr, uint64_t* n_floats);
/* Un-tile a gathered buffer (world_size consecutive rank payloads) into film layout on the device:
 * d_rgb_sum [H*W*3], d_weight [H*W].  Equivalent of Film::AddSampleBufferGPU's target layout
 * (src/core/film.cu:79-99). */
int prt_film_resolve(PrtContext* ctx, const void* d_gathered, uint32_t world_size, void* d_rgb_sum, void* d_weight);
/* The same on a stream of the caller's choice (NULL = the context's stream): the per-frame gather and this un-tiling can
 * then run on a side stream from a snapshot of the payload while the context's stream already renders the next frame. */
int prt_film_resolve_on(PrtContext* ctx, void* hip_stream, const void* d_gathered, uint32_t world_size, void* d_rgb_sum,
                        void* d_weight);
/* Film::UpdateDisplayGPU (src/core/film.cu:101-132): mean -> Reinhard -> gamma -> RGBA8, from device
 * film buffers to a device RGBA8 buffer [H*W*4]. */
int prt_film_tonemap(PrtContext* ctx, const void* d_rgb_sum, const void* d_weight, float exposure, float gamma, void* d_rgba8);
/* Convenience: resolve this context's own film (world_size==1) and tonemap to host RGBA8. */
int prt_film_display(PrtContext* ctx, float exposure, float gamma, uint8_t* rgba8);

/* ---- function-level entry points (used by the parity tests; all go through the same kernels) -- */
/* Camera::GetCameraRay at pixel-space points (px,py)  (src/core/camera.h:103-132). Host in/out. */
int prt_camera_rays(PrtContext* ctx, uint32_t n, const float* px, const float* py, float* origins, float* dirs);
/* Scene::Intersect for n rays (src/core/scene.h:22-25).  Host in/out. */
int prt_closest_hit(PrtContext* ctx, uint32_t n, const float* origins, const float* dirs, PrtHit* hits);
/* MaterialHandle::Scatter + Emit for n (ray, hit, rng state) tuples (src/core/material.h:139-161).
 * rng_state is advanced in place. scattered[i] = 0/1. */
int prt_scatter(PrtContext* ctx, uint32_t n, const float* in_dirs, const PrtHit* hits, uint32_t* rng_state,
                uint32_t* scattered, float* attenuation, float* emitted, float* out_origins, float* out_dirs);

/* ---- measurement ----------------------------------------------------------------------------- */
int prt_enable_timing(PrtContext* ctx, int on); /* HIP events around every kernel launch */
int prt_get_stats(PrtContext* ctx, PrtStats* out);
int prt_reset_stats(PrtContext* ctx);
/* Runs ONE sample with the instrumented traversal kernel (film untouched) and fills
 * bvh_node_visits / bvh_tri_tests / prim_tests / rays_per_depth in `out`. */
int prt_measure_traversal(PrtContext* ctx, uint32_t max_depth, uint32_t seed, uint32_t sample, PrtStats* out);
/* Diagnostic (wavefront.md:92-93 "material-coherent queues"): runs one batch (film untouched) and reports, per
 * bounce d, what the waves of the shade kernel find in their 64 ray slots by the material of the hit; out[16 * d + ...]:
 * [0] waves, [1] lanes with a ray, [2 + t] lanes of material type t (0 = miss, 1..4 = PRT_MAT_*), [8 + t] waves holding type t,
 * [14] sum over waves of distinct scattering types present, [15] waves with a scattering lane.  out: 16 * max_depth words. */
int prt_measure_shade_divergence(PrtContext* ctx, uint32_t max_depth, uint32_t seed, uint32_t sample, uint64_t* out);
int prt_bvh_info(PrtContext* ctx, PrtBvhInfo* out);
int prt_kernel_occupancy(PrtContext* ctx, PrtOccupancy* out);
/* Name of the traversal kernel instance the current scene and tunables select (what prt_render launches and what
 * prt_kernel_occupancy describes): "lean8_5waves", "deep15_4waves", "inst12_4waves", "wide11_5waves", "bvh4", "bvh2".
 * Written NUL-terminated into name[capacity].  The same decision function drives the launch (csrc/prt_kernels.hip). */
int prt_kernel_instance(PrtContext* ctx, char* name, uint32_t capacity);
/* Copies the built BVH out (host arrays): nodes n_nodes*16 floats (layout: csrc/bvh.h), tris
 * n_triangles*12 floats in leaf order.  Either pointer may be NULL.  Works on host-only contexts. */
int prt_bvh_read(PrtContext* ctx, float* nodes, float* tris);
/* The 4-wide tree: n_nodes4*32 floats (layout: csrc/bvh.h). */
int prt_bvh_read4(PrtContext* ctx, float* nodes4);
/* The compressed 8-wide tree: n_nodes8*20 uint32 (layout: csrc/bvh.h). */
int prt_bvh_read8(PrtContext* ctx, uint32_t* nodes8);
/* Selects the traversal kernel variant (0 = default). For A/B benchmarking only. */
int prt_set_variant(PrtContext* ctx, int variant);
/* Tunables (A/B benchmarking; defaults = measured best on C3): "variant", "grid_blocks", "chunk", "refill_min",
 * "exit_max", "tri_min", "wide" (2 = compressed 8-wide tree, default; 1 = 4-wide; 0 = binary), "stack_lds" (kernel
 * instance), "xcd_affinity" (4-wide / binary kernels), "tail" (64-ray granules per resident wave handed out singly at
 * the end of the ray buffer), "steal" (idle lanes a draining wave needs before they take over pending subtrees; 0 = off),
 * "exact_grids" (0/1/2: k_shade grids from the ray counts read back during the traversal: never / big batches / always),
 * "fuse", "prim_bvh" (0: linear scan over the analytic primitives as in the reference), "measure_spp", "stack_cap"
 * (test hook), "gpu_build" (the next prt_set_scene builds the 8-wide tree(s) on the device, placed copies and the top level
 * included: 1 = PLOC + SAH top + optimal collapse, ~20 ms for 870 k triangles, traverses within 2-3 % of the host tree;
 * 2 = Morton octree, ~3 ms, ~20 % slower to traverse), "node_stride" (before prt_set_scene: 5 = 8-wide nodes packed at
 * 80 B, 8 = one node per 128-B line, 0 = by tree size, default), "pad_log2" (before prt_set_scene: the culling pad is 2^-n of
 * the coordinates' magnitude, default 18; A/B only).  Results never depend on a tunable.  Unknown names / bad values
 * return PRT_ERR_INVALID. */
int prt_set_param(PrtContext* ctx, const char* name, int value);

/* ---- several GPUs of one node behind one Renderer (the reference is single-GPU:
 *      cudaSetDevice(0), src/backend/optix/renderer.cpp:217) ------------------------------------------
 * A group owns one context per entry of device_ids and drives each from its own host thread.  The image is tiled over
 * the contexts (prt_set_film(w, h, rank, n): 8x8 tiles dealt round-robin), the scene is built once and cloned, every
 * rank renders all samples of its tiles (RNG keyed by global pixel and sample: the image does not depend on n), and ONE
 * gather of the per-tile radiance brings the payloads to rank 0's device, which un-tiles them into the Film layout:
 *   transport "rccl": ncclSend / ncclRecv inside one group call over xGMI (librccl.so, loaded on demand; needs distinct devices)
 *   transport "peer": hipMemcpyPeerAsync into rank 0's buffer (also the fallback, and what several ranks on ONE device use)
 * PRT_GROUP_TRANSPORT=rccl|peer overrides the choice (rccl with a single rank runs a 1-rank ncclAllGather: a hardware
 * smoke test of the RCCL path).  The same device may appear several times in device_ids (rehearsal / tests on one GPU).
 * All calls are synchronous and are made from one caller thread.
 * prt_group_create ALWAYS stores a group in *out, also when it returns an error (so that prt_group_last_error can say why):
 * the caller destroys it with prt_group_destroy in either case.  The "rccl" transport with n > 1 distinct devices has not
 * run on hardware yet (no multi-GPU box was available to the builder; prt_group_transport says which one a group uses). */
typedef struct PrtGroup PrtGroup;
int prt_group_create(const int* device_ids, uint32_t n, PrtGroup** out);
void prt_group_destroy(PrtGroup* g);
const char* prt_group_last_error(const PrtGroup* g);
uint32_t prt_group_size(const PrtGroup* g);
const char* prt_group_transport(const PrtGroup* g);          /* "rccl" | "peer" | "none" (one rank) */
PrtContext* prt_group_context(PrtGroup* g, uint32_t rank);   /* per-rank tunables / stats; owned by the group */
int prt_group_set_scene(PrtGroup* g, const PrtSceneDesc* scene);   /* built on rank 0, cloned to the others */
/* prt_refit_meshes on every rank (each refits the tree on its own device, in parallel; the meshes are read by all ranks). */
int prt_group_refit_meshes(PrtGroup* g, const PrtMesh* meshes, uint32_t n_meshes);
int prt_group_set_camera(PrtGroup* g, const PrtCameraDesc* cam);
int prt_group_set_film(PrtGroup* g, uint32_t width, uint32_t height);
int prt_group_film_clear(PrtGroup* g);
int prt_group_set_sampling(PrtGroup* g, const PrtSampling* s);
int prt_group_set_samples_in_flight(PrtGroup* g, uint32_t n);
int prt_group_set_param(PrtGroup* g, const char* name, int value);
/* Renderer::ProgressiveRender x spp on every rank's tiles, then the gather + un-tiling on rank 0's device. */
int prt_group_render(PrtGroup* g, uint32_t spp, uint32_t max_depth, uint32_t seed, uint32_t first_sample);
/* Whole film (all ranks' tiles) to host / tonemapped to host RGBA8, as prt_film_read / prt_film_display. */
int prt_group_film_read(PrtGroup* g, float* rgb_sum, float* weight);
int prt_group_film_display(PrtGroup* g, float exposure, float gamma, uint8_t* rgba8);
/* Ray counters summed over the ranks; the *_ms fields are the maxima over the ranks. */
int prt_group_get_stats(PrtGroup* g, PrtStats* out);

/* ---- host-side data formats either side of the path ------------------------------------------- */
/* PLY ingest with the subset the reference's Mesh asks tinyply for (src/core/mesh.cpp:79-97,113-144):
 * vertex x,y,z (+ optional nx,ny,nz), face vertex_indices list; ascii and binary_little_endian.
 * Missing normals are computed (area-weighted); polygons are fan-triangulated. */
typedef struct PrtMeshData PrtMeshData;
int prt_mesh_load_ply(const char* path, PrtMeshData** out, char* err, size_t err_len);
int prt_mesh_create(const float* positions, const float* normals, uint32_t n_vertices, const uint32_t* indices,
                    uint32_t n_triangles, PrtMeshData** out);
void prt_mesh_free(PrtMeshData* m);
uint32_t prt_mesh_vertex_count(const PrtMeshData* m);
uint32_t prt_mesh_triangle_count(const PrtMeshData* m);
const float* prt_mesh_positions(const PrtMeshData* m);
const float* prt_mesh_normals(const PrtMeshData* m);
const uint32_t* prt_mesh_indices(const PrtMeshData* m);
int prt_mesh_had_normals(const PrtMeshData* m);
/* Deterministic longest-edge bisection up to >= target_triangles (SURVEY §8d synthetic inputs). */
int prt_mesh_refine(PrtMeshData* m, uint32_t target_triangles);
/* positions = mat * positions, normals = normalize(mat3(inverse-transpose) * normals). */
int prt_mesh_transform(PrtMeshData* m, const float mat[16], const float inv[16]);
int prt_mesh_append(PrtMeshData* dst, const PrtMeshData* src);

/* Scene presets (src/core/scene.cpp:62-350) flattened into materials + primitives.
 * Pass NULL arrays to query counts. */
int prt_scene_preset(int preset, PrtMaterial* materials, uint32_t* n_materials, PrtPrimitive* primitives,
                     uint32_t* n_primitives);
/* Scene::MakeTransform (src/core/scene.cpp:9-17): T * eulerAngleXYZ(radians(deg)) * S and its inverse. */
void prt_make_transform(const float scale[3], const float euler_deg[3], const float translation[3], float mat[16],
                        float inv[16]);

/* Offline framebuffer dump (stands in for the GLFW/OpenGL viewer, src/main.cpp:504-527). */
int prt_write_ppm(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height);
int prt_write_pfm(const char* path, const float* rgb, uint32_t width, uint32_t height);

#ifdef __cplusplus
}
#endif
#endif /* PRT_H */
