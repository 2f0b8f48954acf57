// bvh.h — host-side BVH2 builder (binned SAH) for the traversal kernel in prt_kernels.hip.
//
// The reference has NO acceleration structure (closest hit is a linear scan,
// src/core/primitive.cpp:26-49; "add a simple BVH" is on its roadmap, wavefront.md:88-90); only the
// building blocks exist (AABB::Union / MaxExtent, src/core/geometry.h:155-205).  The BVH below is this
// project's own; results are BVH-independent by construction (see k_intersect).
//
// Device node = 64 B = 4 x float4:
//   q0 = (Lmin.x, Lmin.y, Lmin.z, Lmax.x)   q1 = (Lmax.y, Lmax.z, Rmin.x, Rmin.y)
//   q2 = (Rmin.z, Rmax.x, Rmax.y, Rmax.z)   q3 = (left, right, 0, 0) as int bits
// child ref >= 0: internal node index; < 0: leaf, ~ref = (first_triangle_slot << 4) | count (count 0..15).
// Boxes are the exact fp32 bounds of the vertices (padding is applied per ray by the kernel).
//
// Wide node (BVH4, what the default traversal kernel walks) = 128 B = 8 x float4, one cache line:
//   q0 = min.x of children 0..3   q1 = max.x   q2 = min.y   q3 = max.y   q4 = min.z   q5 = max.z
//   q6 = child refs 0..3 (same encoding)       q7 = unused
// built by collapsing the binary tree; an absent child has an all-+inf box and the empty-leaf ref ~0.
//
// Compressed 8-wide node (BVH8Q, what the default traversal kernel walks) = 80 B = 5 x uint4, after the
// compressed wide BVH of Ylitie, Karras and Laine (HPG 2017), re-derived here for wave64 / LDS stacks:
//   w0  = origin p.x, p.y, p.z (fp32: the node's own box minimum), {ex, ey, ez, imask} (one byte each)
//   w1  = child_base, tri_base, meta[0..3], meta[4..7]
//   w2  = qlo_x[0..7], qlo_y[0..7]      w3 = qlo_z[0..7], qhi_x[0..7]      w4 = qhi_y[0..7], qhi_z[0..7]
// Child i (its SLOT: children are placed so that visiting slots in descending (slot ^ (7 - octant)) order is
// roughly front to back for a ray of that direction octant) has the box
//   [p + qlo * 2^(e - 127), p + qhi * 2^(e - 127)]  per axis, a superset of the child's exact fp32 bounds.
// imask bit i: child i is an internal node; its index is child_base + popcount(imask & ((1 << i) - 1)), so the
// internal children of a node are contiguous.  meta[i]: 0 = empty slot; internal: (1 << 5) | (24 + i);
// leaf: (unary triangle count {1, 3, 7} << 5) | offset, triangles tri_base + offset .. (offsets < 24, <= 3 per leaf).
// The triangle slots of ALL trees follow the order this layout needs (the leaf children of one wide node are
// contiguous); the BVH2 / BVH4 leaf refs point into the same order.  Empty when a leaf has more than 3 triangles.
#pragma once
#include <stdint.h>

#include <vector>

struct BvhBuild {
    std::vector<float> nodes;     // 16 floats per binary node
    std::vector<float> nodes4;    // 32 floats per wide node
    uint32_t max_stack4 = 0;      // worst-case number of stacked refs when walking the wide tree
    std::vector<uint32_t> nodes8;  // 20 dwords per compressed 8-wide node (root = node 0); empty if unavailable
    uint32_t depth8 = 0;           // levels of 8-wide nodes; the traversal stacks at most depth8 - 1 node groups
    std::vector<uint32_t> order;  // leaf slot -> input triangle index
    uint32_t max_depth = 0;       // edges on the longest root-to-leaf path (+1 for the root node itself)
    uint32_t max_leaf = 0;
    float sah_cost = 0.0f;
};

// verts: n_tris * 9 floats (P0,P1,P2).  max_leaf_size <= 15.  n_threads <= 0: hardware concurrency.
// Returns false if the tree would be deeper than max_allowed_depth.
bool bvh_build(const float* verts, uint32_t n_tris, uint32_t max_leaf_size, int n_threads, uint32_t max_allowed_depth,
               BvhBuild* out);
