/*
 * prt_oracle.h — C interface of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (libprt.so) never links, loads or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference (Rickyeeeeee/ParallelRayTracing) cannot be compiled in this
 * image (it needs glm, tinyply, CUDA/cuRAND headers that are absent, and it ships no tests,
 * golden vectors or fixtures), so this restatement is checked only against hand-derived
 * known-answer cases and its own internal invariants.  Every function cites the reference
 * file:line it restates.  glm (un-vendored submodule, no pinned commit: .gitmodules:7-9) operation
 * order is restated from its published source.
 */
#ifndef PRT_ORACLE_H
#define PRT_ORACLE_H

#include "../include/prt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcScene OrcScene;

/* RNG contract (the reference CPU backend uses unseeded std::rand, src/core/math.h:15; the contract
 * below is the reference's only deterministic generator, the PCG hash of
 * src/backend/optix/device_types.h:109-114, seeded like src/backend/optix/device_programs.cu:169). */
uint32_t orc_pcg_hash(uint32_t v);
uint32_t orc_path_seed(uint32_t pixel, uint32_t sample, uint32_t seed);
float orc_random(uint32_t* state);                       /* Random(), math.h:10-17 */
void orc_random_unit_vector(uint32_t* state, float out[3]); /* RandomUnitVector, math.h:26-36 */

/* Camera (src/core/camera.h:10-16, 103-132) */
void orc_camera_basis(const PrtCameraDesc* cam, float front[3], float right[3], float up[3]);
void orc_camera_rays(const PrtCameraDesc* cam, uint32_t n, const float* px, const float* py, float* origins,
                     float* dirs);

/* Shapes in LOCAL space (src/core/shape.h:157-203, 213-239, 262-303).
 * params: CIRCLE {r}; QUAD {w,h}; TRIANGLE {P0,P1,P2,N0,N1,N2} (18 floats). returns HasIntersection. */
int orc_shape_intersect(int shape_type, const float* params, const float o[3], const float d[3], float pos[3],
                        float normal[3], int* front);

/* Transform helpers (src/core/geometry.h:139-148; src/core/scene.cpp:9-17) */
void orc_transform_point(const float m[16], const float p[3], float out[3]);
void orc_transform_normal(const float m[16], const float n[3], float out[3]);
void orc_make_transform(const float scale[3], const float euler_deg[3], const float translation[3], float mat[16],
                        float inv[16]);

/* Scene presets (src/core/scene.cpp:62-350). NULL arrays: query counts. */
int orc_scene_preset(int preset, PrtMaterial* materials, uint32_t* n_materials, PrtPrimitive* primitives,
                     uint32_t* n_primitives);

/* Scene (copies everything) */
OrcScene* orc_scene_create(const PrtSceneDesc* desc);
void orc_scene_destroy(OrcScene* s);
uint32_t orc_scene_prim_count(const OrcScene* s); /* analytic + triangles */

/* Scene::Intersect / PrimitiveList::Intersect (src/core/primitive.cpp:21-59).
 * use_bvh = 0: the reference's linear scan.  use_bvh = 1: same result through the oracle's own
 * median-split BVH (used for large meshes and the CPU baseline). */
void orc_closest_hit(const OrcScene* s, uint32_t n, const float* origins, const float* dirs, PrtHit* hits,
                     int use_bvh, int n_threads);

/* MaterialHandle::Emit + Scatter (src/core/material.h:16-31, 48-57, 76-109, 114-161).
 * returns scattered (0/1). */
int orc_scatter(const PrtMaterial* m, const float in_dir[3], const PrtHit* hit, uint32_t* rng_state,
                float attenuation[3], float emitted[3], float out_origin[3], float out_dir[3]);

/* n hits at once: hit i uses materials[hits[i].material_id] and rng_state[i] (updated in place). */
void orc_scatter_batch(const PrtMaterial* materials, uint32_t n, const float* in_dirs, const PrtHit* hits,
                       uint32_t* rng_state, uint32_t* scattered, float* attenuation, float* emitted, float* out_origins,
                       float* out_dirs);

/* fresnelReflectance (src/core/material.h:105-109).  orc_fresnel: the contract (correctly rounded x^5 in double, the
 * same text as csrc/prt_device.h); orc_fresnel_libm: the literal std::pow((double)x, 5) of the host's libm. */
float orc_fresnel(float cosine, float ri);
float orc_fresnel_libm(float cosine, float ri);
void orc_fresnel_batch(uint32_t n, const float* cosine, const float* ri, float* out, float* out_libm);

/* CPURenderer::TraceRay (src/backend/cpu/renderer.cpp:59-103) when iterative = 0, or the
 * throughput form TraceRayGPU (src/backend/cuda_megakernel/renderer.cu:81-119) when iterative = 1. */
void orc_trace(const OrcScene* s, const float o[3], const float d[3], int max_depth, uint32_t* rng_state,
               int iterative, int use_bvh, float L[3], uint32_t* n_segments);

/* CPURenderer::ProgressiveRender x spp (src/backend/cpu/renderer.cpp:18-57) + Film::AddSample
 * (src/core/film.cu:37-55) over the pixel rectangle [x0,x1) x [y0,y1) of a W x H film.
 * accum: W*H*3, weights: W*H (must be zero-initialised by the caller or hold earlier samples). */
void orc_render(const OrcScene* s, const PrtCameraDesc* cam, uint32_t W, uint32_t H, uint32_t x0, uint32_t y0,
                uint32_t x1, uint32_t y1, uint32_t spp, uint32_t first_sample, int max_depth, uint32_t seed,
                int iterative, int use_bvh, int n_threads, float* accum, float* weights, uint64_t* rays);

/* The same with the optional sampling upgrades of PrtSampling (include/prt.h; NULL = orc_render). */
void orc_render_sampling(const OrcScene* s, const PrtCameraDesc* cam, uint32_t W, uint32_t H, uint32_t x0, uint32_t y0,
                         uint32_t x1, uint32_t y1, uint32_t spp, uint32_t first_sample, int max_depth, uint32_t seed,
                         int iterative, int use_bvh, int n_threads, const PrtSampling* sp, float* accum, float* weights,
                         uint64_t* rays);

/* Film::UpdateDisplay (src/core/film.cu:134-194) */
void orc_tonemap(const float* accum, const float* weights, uint32_t n_pixels, float exposure, float gamma,
                 uint8_t* rgba8);

/* AABB::IntersectP (src/core/geometry.h:170-192) */
int orc_aabb_intersect_p(const float bmin[3], const float bmax[3], const float o[3], const float d[3]);

#ifdef __cplusplus
}
#endif
#endif
