// bvh_gpu.h — device-side builder of the compressed 8-wide tree (see bvh_gpu.hip, bvh.h).
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stdint.h>

struct PrtGpuBvh {
    uint32_t* d_nodes8;  // n_nodes x 20 dwords (ownership passes to the caller, hipFree)
    float4* d_tris;      // 3 x float4 per triangle in the tree's slot order
    float4* d_nrms;
    uint32_t n_nodes;
    uint32_t depth;      // levels of the tree
};

// d_verts / d_norms: 9 floats per triangle (P0,P1,P2 / N0,N1,N2), d_tri_mat: material per triangle, all on the device.
// cmin / cmax: bounds of the triangle centroids.  Returns 0 on success (synchronous: the stream is drained).
int prt_gpu_bvh8_build(hipStream_t st, const float* d_verts, const float* d_norms, const uint32_t* d_tri_mat, uint32_t n_tris,
                       uint32_t n_prims, const float cmin[3], const float cmax[3], PrtGpuBvh* out);
// The quality builder (round 2): the same Morton sort, then PLOC (locally-ordered agglomerative clustering) for the binary
// topology and the SAH-optimal collapse to 8-wide nodes by dynamic programming, all on the device.  Same contract.
// leaf_cost: cost of testing one primitive of a leaf relative to one 8-wide node visit (0 = the builder's default for
// triangles; large for a top-level tree, whose "primitives" are whole instances that are entered without a box test of
// their own: every instance then gets a leaf to itself, except copies whose boxes coincide).
int prt_gpu_bvh8_build_ploc(hipStream_t st, const float* d_verts, const float* d_norms, const uint32_t* d_tri_mat, uint32_t n_tris,
                            uint32_t n_prims, const float cmin[3], const float cmax[3], PrtGpuBvh* out, float leaf_cost = 0.0f);

// Measurement aid (prt_set_param("sort_rays", 1 | 2)): idx2[0..n) = the order of rays 0..n-1 by (Morton cell of the origin
// in a 32^3 grid over [bmin, bmax], direction octant) (mode 1) or (octant, cell) (mode 2).  keys, keys2, idx, idx2: n
// uint32 each; temp: prt_sort_rays_temp_bytes(n) bytes.
size_t prt_sort_rays_temp_bytes(uint32_t n);
int prt_sort_rays(hipStream_t st, const float4* ro, const float4* rd, uint32_t n, const float bmin[3], const float bmax[3],
                  uint32_t mode, uint32_t* keys, uint32_t* keys2, uint32_t* idx, uint32_t* idx2, void* temp, size_t temp_bytes);

// Refit: new vertex positions over an existing 8-wide tree on the device.  d_nodes8: n_nodes nodes at a stride of
// stride_dwords (20 packed / 32 one node per 128-B line); level_nodes (HOST array, n_nodes entries): the node indices sorted
// by tree level, level l = level_nodes[level_start[l] .. level_start[l + 1]) (level_start: host array of n_levels + 1
// entries); d_verts / d_norms: 9 floats per triangle in INPUT order (d_norms may be null);
// d_tris / d_nrms: the scene's records in slot order, rewritten in place.  root_box[6] receives the new root bounds.
// Synchronous.  Returns 0, a hipError_t, or -6 if a box does not fit its node's quantization grid.
int prt_gpu_bvh8_refit(hipStream_t st, uint32_t* d_nodes8, uint32_t stride_dwords, uint32_t n_nodes, const uint32_t* level_nodes,
                       const uint32_t* level_start, uint32_t n_levels, const float* d_verts, const float* d_norms, uint32_t n_tris,
                       uint32_t n_prims, float4* d_tris, float4* d_nrms, float root_box[6]);
