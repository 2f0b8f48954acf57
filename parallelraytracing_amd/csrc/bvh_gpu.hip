// bvh_gpu.hip — GPU-side builder of the compressed 8-wide tree (bvh.h "BVH8Q"), for fast (re)builds.
//
// SURVEY.md §8f-3 asks for a device-side builder next to the two-level BVH.  The host builder (bvh.cpp: binned SAH +
// SAH-optimal collapse) stays the default because its trees traverse faster; this one trades tree quality for build
// time and keeps everything on the device (prt_set_param("gpu_build", 1)).
//
// Design (MI355X-first, not a port of any reference code — the reference has no BVH at all):
//   1. 30-bit Morton codes of the triangle centroids (10 bits per axis), rocPRIM radix sort.
//   2. The 8-wide tree is built DIRECTLY from the sorted codes, one octree level per tree level: a node is a range of
//      sorted triangles that share a code prefix, its children are the sub-ranges by the next 3 bits (binary searches;
//      levels at which the whole range falls into one octant are skipped).  The Morton digit IS the child's slot, and
//      "visit slots in descending (slot ^ (7 - octant)) order" is exactly front-to-back for octants, so the
//      traversal kernel's ordering needs no placement step.  Ranges of <= 3 triangles become leaves.
//   3. Levels are processed breadth first (one launch per level; the nodes a level creates are a contiguous index
//      range), siblings get consecutive indices from one atomic, and every node packs the triangles of its leaf
//      children contiguously (one atomic on the triangle cursor), which is the layout the node format requires.
//   4. Boxes bottom-up (one launch per level), then quantization in double precision with the same containment
//      fix-ups as the host builder: every quantized box contains the child's exact fp32 bounds.
// Results are tree-independent by construction (closest hit = min world d^2, ties to the lowest primitive index), so
// the parity tests run unchanged against trees from this builder.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <stdint.h>

#include "bvh_gpu.h"

namespace {

#define GB_CHECK(x)                        \
    do {                                   \
        hipError_t e_ = (x);               \
        if (e_ != hipSuccess) return (int)e_; \
    } while (0)

__device__ __forceinline__ uint32_t expand10(uint32_t v) {  // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// code bit layout per level (most significant first): z y x, i.e. digit = (z << 2) | (y << 1) | x: bit a of a digit
// = upper half on axis a = the slot convention of the traversal kernel (bvh.h)
__global__ void k_morton(const float* __restrict__ verts, uint32_t n, float3 cmin, float3 cinv, uint32_t* __restrict__ codes,
                         uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float* v = verts + 9 * (size_t)i;
    float c[3];
    for (int a = 0; a < 3; ++a) {
        const float lo = fminf(v[a], fminf(v[3 + a], v[6 + a]));
        const float hi = fmaxf(v[a], fmaxf(v[3 + a], v[6 + a]));
        c[a] = 0.5f * lo + 0.5f * hi;
    }
    const float fx = fminf(fmaxf((c[0] - cmin.x) * cinv.x, 0.0f), 1023.0f);
    const float fy = fminf(fmaxf((c[1] - cmin.y) * cinv.y, 0.0f), 1023.0f);
    const float fz = fminf(fmaxf((c[2] - cmin.z) * cinv.z, 0.0f), 1023.0f);
    codes[i] = expand10((uint32_t)fx) | (expand10((uint32_t)fy) << 1) | (expand10((uint32_t)fz) << 2);
    idx[i] = i;
}

struct Range {
    uint32_t first, count, level;  // level = octree levels already consumed by the prefix (0..10)
};

__device__ __forceinline__ uint32_t lower_bound(const uint32_t* codes, uint32_t a, uint32_t b, uint32_t key) {
    while (a < b) {  // first position in [a, b) whose code is >= key
        const uint32_t m = (a + b) >> 1;
        if (codes[m] < key) a = m + 1; else b = m;
    }
    return a;
}

// One thread per node of the current level: split its range, create its children, pack its leaf triangles.
__global__ void k_split(const uint32_t* __restrict__ codes, const uint32_t* __restrict__ sorted_idx, Range* __restrict__ ranges,
                        uint32_t* __restrict__ nodes8, uint32_t* __restrict__ order, uint32_t* __restrict__ counters,
                        uint32_t node_begin, uint32_t node_end, uint32_t max_nodes) {
    const uint32_t nd = node_begin + blockIdx.x * 128u + threadIdx.x;
    if (nd >= node_end) return;
    const Range r = ranges[nd];
    uint32_t cf[8], cc[8];
    uint32_t level = r.level;
    int n_child = 0;
    for (;;) {
        if (r.count <= 3u) {  // only the root of a tiny mesh: one leaf child
            for (int c = 0; c < 8; ++c) cf[c] = r.first, cc[c] = 0;
            cc[0] = r.count;
            n_child = 1;
            break;
        }
        if (level >= 10u) {  // identical codes: split by index
            const uint32_t parts = r.count <= 24u ? (r.count + 2u) / 3u : 8u;
            for (uint32_t c = 0; c < 8u; ++c) {
                const uint32_t lo = (uint32_t)(((unsigned long long)r.count * c) / parts);
                const uint32_t hi = (uint32_t)(((unsigned long long)r.count * (c + 1u)) / parts);
                cf[c] = r.first + (c < parts ? lo : r.count);
                cc[c] = c < parts ? hi - lo : 0u;
            }
            n_child = (int)parts;
            break;
        }
        const uint32_t shift = 3u * (9u - level);
        const uint32_t prefix = codes[r.first] & ~((8u << shift) - 1u);
        uint32_t prev = r.first;
        n_child = 0;
        for (uint32_t c = 0; c < 8u; ++c) {
            const uint32_t nxt = c == 7u ? r.first + r.count : lower_bound(codes, prev, r.first + r.count, prefix | ((c + 1u) << shift));
            cf[c] = prev;
            cc[c] = nxt - prev;
            if (cc[c]) ++n_child;
            prev = nxt;
        }
        ++level;
        if (n_child >= 2) break;  // a level that does not split the range is skipped
    }
    uint32_t n_int = 0, n_leaf_tris = 0;
    for (int c = 0; c < 8; ++c) {
        if (cc[c] > 3u) ++n_int;
        else n_leaf_tris += cc[c];
    }
    uint32_t child_base = n_int ? atomicAdd(&counters[0], n_int) : 0u;
    const uint32_t tri_base = n_leaf_tris ? atomicAdd(&counters[1], n_leaf_tris) : 0u;
    if (n_int && child_base + n_int > max_nodes) {  // cannot happen (every node has >= 2 children); flagged, not written
        atomicOr(&counters[2], 1u);
        n_int = 0;
        child_base = 0;
        for (int c = 0; c < 8; ++c)
            if (cc[c] > 3u) cc[c] = 0;
    }
    uint32_t imask = 0, meta[8], rank = 0, off = 0;
    for (int c = 0; c < 8; ++c) {
        meta[c] = 0;
        if (cc[c] > 3u) {
            imask |= 1u << c;
            meta[c] = (1u << 5) | (24u + (uint32_t)c);
            ranges[child_base + rank] = Range{cf[c], cc[c], level};
            ++rank;
        } else if (cc[c]) {
            meta[c] = (((1u << cc[c]) - 1u) << 5) | off;
            for (uint32_t k = 0; k < cc[c]; ++k) order[tri_base + off + k] = sorted_idx[cf[c] + k];
            off += cc[c];
        }
    }
    uint32_t* w = nodes8 + 20 * (size_t)nd;
    w[3] = imask << 24;  // exponents are filled by k_quantize
    w[4] = child_base;
    w[5] = tri_base;
    w[6] = meta[0] | (meta[1] << 8) | (meta[2] << 16) | (meta[3] << 24);
    w[7] = meta[4] | (meta[5] << 8) | (meta[6] << 16) | (meta[7] << 24);
}

// Bottom-up boxes of one level: cbox[node][child][6] (exact fp32 bounds of every child), nbox[node][6] = their union.
__global__ void k_boxes(const float* __restrict__ verts, const uint32_t* __restrict__ nodes8, const uint32_t* __restrict__ order,
                        float* __restrict__ cbox, float* __restrict__ nbox, uint32_t node_begin, uint32_t node_end) {
    const uint32_t nd = node_begin + blockIdx.x * 128u + threadIdx.x;
    if (nd >= node_end) return;
    const uint32_t* w = nodes8 + 20 * (size_t)nd;
    const uint32_t imask = w[3] >> 24, child_base = w[4], tri_base = w[5];
    float nmn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    float nmx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    uint32_t rank = 0;
    for (int c = 0; c < 8; ++c) {
        const uint32_t meta = (w[6 + (c >> 2)] >> (8 * (c & 3))) & 0xFFu;
        float* cb = cbox + 48 * (size_t)nd + 6 * c;
        float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
        float mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
        if (meta) {
            if ((imask >> c) & 1u) {
                const float* cbx = nbox + 6 * (size_t)(child_base + rank);
                ++rank;
                for (int a = 0; a < 3; ++a) {
                    mn[a] = cbx[a];
                    mx[a] = cbx[3 + a];
                }
            } else {
                const uint32_t cnt = (uint32_t)__popc(meta >> 5), first = tri_base + (meta & 31u);
                for (uint32_t k = 0; k < cnt; ++k) {
                    const float* v = verts + 9 * (size_t)order[first + k];
                    for (int vv = 0; vv < 3; ++vv)
                        for (int a = 0; a < 3; ++a) {
                            mn[a] = fminf(mn[a], v[3 * vv + a]);
                            mx[a] = fmaxf(mx[a], v[3 * vv + a]);
                        }
                }
            }
            for (int a = 0; a < 3; ++a) {
                nmn[a] = fminf(nmn[a], mn[a]);
                nmx[a] = fmaxf(nmx[a], mx[a]);
            }
        }
        for (int a = 0; a < 3; ++a) {
            cb[a] = mn[a];
            cb[3 + a] = mx[a];
        }
    }
    for (int a = 0; a < 3; ++a) {
        nbox[6 * (size_t)nd + a] = nmn[a];
        nbox[6 * (size_t)nd + 3 + a] = nmx[a];
    }
}

// Quantization (bvh.h): origin = the node's own minimum, one power-of-two cell per axis, 8-bit planes; the same
// containment fix-ups as bvh.cpp, in double precision (p + q * cell is exact there).
__global__ void k_quantize(uint32_t* __restrict__ nodes8, const float* __restrict__ cbox, const float* __restrict__ nbox,
                           uint32_t n_nodes, uint32_t* __restrict__ counters, uint32_t stride_dwords = 20u) {
    const uint32_t nd = blockIdx.x * 128u + threadIdx.x;
    if (nd >= n_nodes) return;
    uint32_t* w = nodes8 + (size_t)stride_dwords * nd;
    const float* nb = nbox + 6 * (size_t)nd;
    uint32_t eb[3];
    double cell[3];
    for (int a = 0; a < 3; ++a) {
        const double p = nb[a], ext = (double)nb[3 + a] - p;
        const double big = fmax(fabs((double)nb[a]), fabs((double)nb[3 + a]));
        int e = ext > 0.0 ? (int)ceil(log2(ext / 255.0)) : -126;
        if (big > 0.0) e = max(e, (int)floor(log2(big)) - 30);
        e = max(e, -126);
        while (ceil(ext / ldexp(1.0, e)) > 255.0) ++e;
        if (e > 126) {
            atomicOr(&counters[2], 2u);
            e = 126;
        }
        eb[a] = (uint32_t)(e + 127);
        cell[a] = ldexp(1.0, e);
    }
    uint32_t q[6][8];
    for (int c = 0; c < 8; ++c) {
        const uint32_t meta = (w[6 + (c >> 2)] >> (8 * (c & 3))) & 0xFFu;
        const float* cb = cbox + 48 * (size_t)nd + 6 * c;
        for (int a = 0; a < 3; ++a) {
            uint32_t lo = 255u, hi = 0u;  // empty slot: inverted box
            if (meta) {
                const double p = nb[a];
                double l = floor(((double)cb[a] - p) / cell[a]);
                while (l > 0.0 && p + l * cell[a] > (double)cb[a]) l -= 1.0;
                if (l < 0.0) l = 0.0;
                double h = ceil(((double)cb[3 + a] - p) / cell[a]);
                while (p + h * cell[a] < (double)cb[3 + a]) h += 1.0;
                if (h > 255.0 || l > h) {
                    atomicOr(&counters[2], 4u);
                    h = 255.0;
                }
                lo = (uint32_t)l;
                hi = (uint32_t)h;
            }
            q[a][c] = lo;
            q[3 + a][c] = hi;
        }
    }
    w[0] = __float_as_uint(nb[0]);
    w[1] = __float_as_uint(nb[1]);
    w[2] = __float_as_uint(nb[2]);
    w[3] = (w[3] & 0xFF000000u) | eb[0] | (eb[1] << 8) | (eb[2] << 16);
    for (int pl = 0; pl < 6; ++pl) {
        w[8 + 2 * pl] = q[pl][0] | (q[pl][1] << 8) | (q[pl][2] << 16) | (q[pl][3] << 24);
        w[9 + 2 * pl] = q[pl][4] | (q[pl][5] << 8) | (q[pl][6] << 16) | (q[pl][7] << 24);
    }
}

// Triangle / normal records in the final slot order: {P0, global prim index}, {P1, material}, {P2, 0}.
__global__ void k_records(const float* __restrict__ verts, const float* __restrict__ norms, const uint32_t* __restrict__ tri_mat,
                          const uint32_t* __restrict__ order, uint32_t n, uint32_t n_prims, float4* __restrict__ tris,
                          float4* __restrict__ nrms) {
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= n) return;
    const uint32_t t = order[slot];
    const float* v = verts + 9 * (size_t)t;
    const float* q = norms + 9 * (size_t)t;
    tris[3 * (size_t)slot + 0] = make_float4(v[0], v[1], v[2], __uint_as_float(n_prims + t));
    tris[3 * (size_t)slot + 1] = make_float4(v[3], v[4], v[5], __uint_as_float(tri_mat[t]));
    tris[3 * (size_t)slot + 2] = make_float4(v[6], v[7], v[8], 0.0f);
    nrms[3 * (size_t)slot + 0] = make_float4(q[0], q[1], q[2], 0.0f);
    nrms[3 * (size_t)slot + 1] = make_float4(q[3], q[4], q[5], 0.0f);
    nrms[3 * (size_t)slot + 2] = make_float4(q[6], q[7], q[8], 0.0f);
}


// =========================================================================================================
// Quality builder (round 2): PLOC binary tree + SAH-optimal collapse to the 8-wide layout, all on the device.
//
// The Morton octree above builds in ~7 ms but its trees traverse ~20 % slower than the host's binned-SAH tree (median
// splits, 4.5 of 8 slots filled).  This builder keeps the Morton sort and replaces the topology:
//   1. PLOC (parallel locally-ordered clustering, Meister & Bittner 2018; re-derived here): the clusters, in Morton order,
//      each look RADIUS positions to both sides for the partner that minimises the surface area of the merged box;
//      mutual nearest neighbours merge into a new binary node; the cluster array is compacted (one 64-bit scan gives the
//      new node ids and the compacted positions) and the pass repeats until one cluster is left (~45 passes for 870 k).
//      A node's children always have smaller ids than the node, and every pass's nodes form one contiguous id range.
//   2. The same dynamic programme as the host's collapse (bvh.cpp; after Ylitie et al. 2017, section 3), pass range by
//      pass range bottom-up: cost[n][i] = least cost of representing the subtree of n in at most i + 1 slots of its wide
//      parent, cost = area of the wide nodes created + CI x triangles x area of the leaves; subtrees of <= 3 triangles may
//      become leaves (PLOC's tree goes down to single triangles, the leaves are part of the optimisation here).
//   3. Emission level by level, breadth first (one launch per level, as above): a wide node's children follow from the
//      DP's picks, go to octant-ordered slots by the host's greedy rule, internal children get consecutive indices from one
//      atomic, leaf triangles consecutive slots from another; quantization as in k_quantize (double precision, contained).
// =========================================================================================================
#define PLOC_RADIUS_MAX 64
static int g_ploc_radius = 24;   // search radius in Morton order (PRT_PLOC_RADIUS; tuned on C3, tools/build_compare.py)
static int g_ploc_top = 0;       // 1: the top of the tree by full-search clustering on the device (PRT_PLOC_TOP=device)
static float g_ploc_ci = 0.6f;   // cost of one triangle test relative to one 8-wide node visit (PRT_PLOC_CI)

__device__ __forceinline__ float half_area6(const float* lo, const float* hi) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// leaves of the binary tree = the triangles in Morton order: node i (< n) is sorted triangle i
__global__ void k_ploc_leaves(const float* __restrict__ verts, const uint32_t* __restrict__ sorted_idx, uint32_t n,
                              float* __restrict__ box, int* __restrict__ left, int* __restrict__ right, uint32_t* __restrict__ cnt,
                              uint32_t* __restrict__ cl) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float* v = verts + 9 * (size_t)sorted_idx[i];
    for (int a = 0; a < 3; ++a) {
        box[6 * (size_t)i + a] = fminf(v[a], fminf(v[3 + a], v[6 + a]));
        box[6 * (size_t)i + 3 + a] = fmaxf(v[a], fmaxf(v[3 + a], v[6 + a]));
    }
    left[i] = right[i] = -1;
    cnt[i] = 1u;
    cl[i] = i;
}

// nearest neighbour of every cluster within PLOC_RADIUS positions.  Candidates are ranked by a key that is SYMMETRIC in the
// pair: (merged-box area, distance in the array, parity of the lower position, lower position), so the globally best pair is
// the best pair of both its members: it is mutual and every pass merges at least one pair.  The tie terms matter for runs
// of clusters with identical boxes (coincident or duplicated triangles): "lowest position wins" made all of them point at
// the run's first cluster, one merge per pass and a pass per triangle; nearest-then-even-lower-position pairs the run up
// as (0,1), (2,3), ... and halves it per pass.
__global__ void __launch_bounds__(256) k_ploc_nn(const uint32_t* __restrict__ cl, uint32_t m, const float* __restrict__ box,
                                                 uint32_t* __restrict__ nn, int PLOC_RADIUS) {
    __shared__ float s_box[(256 + 2 * PLOC_RADIUS_MAX) * 6];
    const int base = (int)(blockIdx.x * 256u) - PLOC_RADIUS;
    for (int t = (int)threadIdx.x; t < 256 + 2 * PLOC_RADIUS; t += 256) {
        const int j = base + t;
        if (j >= 0 && j < (int)m) {
            const float* b = box + 6 * (size_t)cl[j];
            for (int a = 0; a < 6; ++a) s_box[6 * t + a] = b[a];
        }
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const float* bi = &s_box[6 * ((int)threadIdx.x + PLOC_RADIUS)];
    float best = 3.402823466e+38f;
    unsigned long long best_tie = ~0ull;
    uint32_t bj = i;
    const int j0 = (int)i - PLOC_RADIUS < 0 ? 0 : (int)i - PLOC_RADIUS;
    const int j1 = (int)i + PLOC_RADIUS >= (int)m ? (int)m - 1 : (int)i + PLOC_RADIUS;
    for (int j = j0; j <= j1; ++j) {
        if (j == (int)i) continue;
        const float* bq = &s_box[6 * (j - base)];
        float lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(bi[a], bq[a]);
            hi[a] = fmaxf(bi[3 + a], bq[3 + a]);
        }
        const float ar = half_area6(lo, hi);
        const uint32_t lowp = (uint32_t)j < i ? (uint32_t)j : i, dist = (uint32_t)j < i ? i - (uint32_t)j : (uint32_t)j - i;
        const unsigned long long tie = ((unsigned long long)dist << 33) | ((unsigned long long)(lowp & 1u) << 32) | lowp;
        if (ar < best || (ar == best && tie < best_tie)) {
            best = ar;
            best_tie = tie;
            bj = (uint32_t)j;
        }
    }
    nn[i] = bj;
}

// The same search over ALL m clusters (the top of the tree, m <= PLOC_TOP): clusters that are far apart in Morton order see
// each other, which the windowed search cannot offer and the upper levels need.  O(m^2) box unions = 16 M for m = 4096.
__global__ void __launch_bounds__(256) k_ploc_nn_full(const uint32_t* __restrict__ cl, uint32_t m, const float* __restrict__ box,
                                                      uint32_t* __restrict__ nn) {
    __shared__ float s_box[256 * 6];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    float bi[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < m) {
        const float* b = box + 6 * (size_t)cl[i];
        for (int a = 0; a < 6; ++a) bi[a] = b[a];
    }
    float best = 3.402823466e+38f;
    unsigned long long best_tie = ~0ull;
    uint32_t bj = i;
    for (uint32_t t0 = 0; t0 < m; t0 += 256u) {  // block-uniform trip count
        __syncthreads();
        if (t0 + threadIdx.x < m) {
            const float* b = box + 6 * (size_t)cl[t0 + threadIdx.x];
            for (int a = 0; a < 6; ++a) s_box[6 * threadIdx.x + a] = b[a];
        }
        __syncthreads();
        const uint32_t nt = m - t0 < 256u ? m - t0 : 256u;
        if (i < m)
            for (uint32_t t = 0; t < nt; ++t) {
                const uint32_t j = t0 + t;
                if (j == i) continue;
                const float* bq = &s_box[6 * t];
                float lo[3], hi[3];
                for (int a = 0; a < 3; ++a) {
                    lo[a] = fminf(bi[a], bq[a]);
                    hi[a] = fmaxf(bi[3 + a], bq[3 + a]);
                }
                const float ar = half_area6(lo, hi);
                const uint32_t lowp = j < i ? j : i, dist = j < i ? i - j : j - i;
                const unsigned long long tie = ((unsigned long long)dist << 33) | ((unsigned long long)(lowp & 1u) << 32) | lowp;
                if (ar < best || (ar == best && tie < best_tie)) {
                    best = ar;
                    best_tie = tie;
                    bj = j;
                }
            }
    }
    if (i < m) nn[i] = bj;
}

// flags for the scan: low word = 1 for the leader of a mutual pair (the lower position), high word = 1 for every
// cluster that stays in the array (everything except the higher position of a mutual pair)
__global__ void k_ploc_flags(const uint32_t* __restrict__ nn, uint32_t m, unsigned long long* __restrict__ flags) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const uint32_t j = nn[i];
    const bool mutual = j != i && nn[j] == i;
    const unsigned long long leader = (mutual && i < j) ? 1ull : 0ull, keep = (mutual && i > j) ? 0ull : 1ull;
    flags[i] = leader | (keep << 32);
}

__global__ void k_ploc_merge(const uint32_t* __restrict__ cl, const uint32_t* __restrict__ nn, const unsigned long long* __restrict__ scan,
                             uint32_t m, uint32_t node_base, uint32_t* __restrict__ cl_out, float* __restrict__ box,
                             int* __restrict__ left, int* __restrict__ right, uint32_t* __restrict__ cnt) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const uint32_t j = nn[i];
    const bool mutual = j != i && nn[j] == i;
    if (mutual && i > j) return;  // absorbed by its partner
    const unsigned long long sc = scan[i];
    uint32_t id = cl[i];
    if (mutual) {
        const uint32_t a = cl[i], b = cl[j];
        id = node_base + (uint32_t)(sc & 0xFFFFFFFFull);
        for (int k = 0; k < 3; ++k) {
            box[6 * (size_t)id + k] = fminf(box[6 * (size_t)a + k], box[6 * (size_t)b + k]);
            box[6 * (size_t)id + 3 + k] = fmaxf(box[6 * (size_t)a + 3 + k], box[6 * (size_t)b + 3 + k]);
        }
        left[id] = (int)a;
        right[id] = (int)b;
        cnt[id] = cnt[a] + cnt[b];
    }
    cl_out[(uint32_t)(sc >> 32)] = id;
}

// rows of the remaining clusters, contiguous, for the host's top builder
__global__ void k_ploc_gather(const uint32_t* __restrict__ cl, uint32_t m, const float* __restrict__ box, const uint32_t* __restrict__ cnt,
                              const float* __restrict__ cost, float* __restrict__ o_box, uint32_t* __restrict__ o_cnt, float* __restrict__ o_cost) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const uint32_t id = cl[i];
    for (int a = 0; a < 6; ++a) o_box[6 * (size_t)i + a] = box[6 * (size_t)id + a];
    for (int a = 0; a < 8; ++a) o_cost[8 * (size_t)i + a] = cost[8 * (size_t)id + a];
    o_cnt[i] = cnt[id];
}

#define PLOC_LEAF 0xFFu  // pick value: "this subtree is one leaf"
// The collapse DP for the binary nodes [nb, ne) of one PLOC pass (their children belong to earlier passes).
// cost[n][i], i = 0..6 <-> 1..7 slots of the wide parent; [n][7] is unused (pick[n][7] = split of n's own 8 slots).
__host__ __device__ inline float ploc_half_area(const float* b) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
}
// One node of the DP.  cl / cr: the children's cost rows (null for a single triangle), c: triangles below the node.
__host__ __device__ inline void ploc_dp_node(const float* bx, uint32_t c, const float* cl, const float* cr, float ci, float* cn, uint8_t* pn) {
    const float area = ploc_half_area(bx);
    if (!cl) {  // a single triangle: one slot, one test
        for (int i = 0; i < 8; ++i) {
            cn[i] = ci * area;
            pn[i] = PLOC_LEAF;
        }
        return;
    }
    float dist[9];
    uint8_t dk[9];
    for (int j = 2; j <= 8; ++j) {  // distribute j slots over the two children
        float bestc = 3.402823466e+38f;
        int bk = 1;
        for (int k = 1; k < j; ++k) {
            const int a = (k < 7 ? k : 7) - 1, b = (j - k < 7 ? j - k : 7) - 1;
            const float v = cl[a] + cr[b];
            if (v < bestc) {
                bestc = v;
                bk = k;
            }
        }
        dist[j] = bestc;
        dk[j] = (uint8_t)bk;
    }
    const float leaf_cost = c <= 3u ? ci * (float)c * area : 3.402823466e+38f;
    // one slot: a leaf, or a wide node of its own (its children take its 8 slots)
    const float own = dist[8] + area;
    cn[0] = leaf_cost <= own ? leaf_cost : own;
    pn[0] = leaf_cost <= own ? PLOC_LEAF : 0u;
    pn[7] = dk[8];
    cn[7] = dist[8];
    for (int i = 2; i <= 7; ++i) {  // i slots: split over the children, or do with i - 1 slots
        if (dist[i] < cn[i - 2]) {
            cn[i - 1] = dist[i];
            pn[i - 1] = dk[i];
        } else {
            cn[i - 1] = cn[i - 2];
            pn[i - 1] = pn[i - 2] == PLOC_LEAF ? PLOC_LEAF : 0u;
        }
    }
}
__global__ void k_ploc_dp(uint32_t nb, uint32_t ne, uint32_t n_leaves, const int* __restrict__ left, const int* __restrict__ right,
                          const float* __restrict__ box, const uint32_t* __restrict__ cnt, float* __restrict__ cost,
                          uint8_t* __restrict__ pick, float PLOC_CI) {
    const uint32_t nd = nb + blockIdx.x * 128u + threadIdx.x;
    if (nd >= ne) return;
    const bool tri = nd < n_leaves;
    ploc_dp_node(box + 6 * (size_t)nd, cnt[nd], tri ? nullptr : cost + 8 * (size_t)left[nd], tri ? nullptr : cost + 8 * (size_t)right[nd],
                 PLOC_CI, cost + 8 * (size_t)nd, pick + 8 * (size_t)nd);
}

// Emission of the wide nodes [wb, we) of one level.  wroot[w] = binary node whose subtree wide node w represents.
__global__ void k_ploc_emit(uint32_t wb, uint32_t we, uint32_t n_leaves, uint32_t max_nodes, uint32_t* __restrict__ wroot,
                            const int* __restrict__ left, const int* __restrict__ right, const float* __restrict__ box,
                            const uint32_t* __restrict__ cnt, const uint8_t* __restrict__ pick, const uint32_t* __restrict__ sorted_idx,
                            uint32_t* __restrict__ nodes8, uint32_t* __restrict__ order, uint32_t* __restrict__ counters) {
    const uint32_t w = wb + blockIdx.x * 64u + threadIdx.x;
    if (w >= we) return;
    const uint32_t R = wroot[w];
    // ---- children of the wide node: walk the binary tree below R handing out R's 8 slots by the DP's picks ----
    uint32_t kids[8];
    bool kid_leaf[8];
    int n = 0;
    if (R < n_leaves || (w == 0u && pick[8 * (size_t)R] == PLOC_LEAF)) {
        kids[n] = R;  // a mesh of <= 3 triangles: the root holds one leaf child
        kid_leaf[n++] = true;
    } else {
        uint32_t st_node[8];
        int st_slots[8], top = 0;
        const int k8 = pick[8 * (size_t)R + 7];
        st_node[top] = (uint32_t)right[R];
        st_slots[top++] = 8 - k8;
        st_node[top] = (uint32_t)left[R];
        st_slots[top++] = k8;
        while (top > 0) {
            --top;
            const uint32_t nd = st_node[top];
            const int slots = st_slots[top];
            if (nd < n_leaves) {
                kids[n] = nd;
                kid_leaf[n++] = true;
                continue;
            }
            int i = slots < 7 ? slots : 7;
            const uint8_t* pn = pick + 8 * (size_t)nd;
            while (i > 1 && pn[i - 1] == 0u) --i;  // "use fewer slots"
            if (pn[i - 1] == PLOC_LEAF) {  // the whole subtree (<= 3 triangles) is one leaf child
                kids[n] = nd;
                kid_leaf[n++] = true;
                continue;
            }
            if (i == 1) {  // stays an internal child: a wide node of its own
                kids[n] = nd;
                kid_leaf[n++] = false;
                continue;
            }
            const int k = pn[i - 1];
            st_node[top] = (uint32_t)right[nd];
            st_slots[top++] = i - k;
            st_node[top] = (uint32_t)left[nd];
            st_slots[top++] = k;
        }
    }
    const float* nbx = box + 6 * (size_t)R;
    // ---- slots (bvh.cpp): slot s is visited first by rays whose direction is negative exactly on the axes whose bit is
    // set in s: greedy on dot(child centre - node centre, that diagonal) ----
    int slot_of[8], child_in[8];
    float score[8][8];
    for (int k = 0; k < 8; ++k) slot_of[k] = child_in[k] = -1;
    for (int c = 0; c < n; ++c) {
        const float* cb = box + 6 * (size_t)kids[c];
        float d[3];
        for (int a = 0; a < 3; ++a) d[a] = (0.5f * cb[a] + 0.5f * cb[3 + a]) - (0.5f * nbx[a] + 0.5f * nbx[3 + a]);
        for (int sl = 0; sl < 8; ++sl)
            score[c][sl] = ((sl & 1) ? d[0] : -d[0]) + ((sl & 2) ? d[1] : -d[1]) + ((sl & 4) ? d[2] : -d[2]);
    }
    for (int round = 0; round < n; ++round) {
        int bc = -1, bs = -1;
        float best = -3.402823466e+38f;
        for (int c = 0; c < n; ++c) {
            if (slot_of[c] >= 0) continue;
            for (int sl = 0; sl < 8; ++sl)
                if (child_in[sl] < 0 && (bc < 0 || score[c][sl] > best)) {
                    best = score[c][sl];
                    bc = c;
                    bs = sl;
                }
        }
        slot_of[bc] = bs;
        child_in[bs] = bc;
    }
    // ---- indices: internal children consecutive from one atomic, leaf triangles consecutive from another ----
    uint32_t n_int = 0, n_leaf_tris = 0;
    for (int c = 0; c < n; ++c) {
        if (kid_leaf[c]) n_leaf_tris += cnt[kids[c]];
        else ++n_int;
    }
    uint32_t child_base = n_int ? atomicAdd(&counters[0], n_int) : 0u;
    const uint32_t tri_base = n_leaf_tris ? atomicAdd(&counters[1], n_leaf_tris) : 0u;
    if (n_int && child_base + n_int > max_nodes) {  // cannot happen (every wide node has >= 2 children); flagged
        atomicOr(&counters[2], 1u);
        return;
    }
    // ---- quantization grid (k_quantize) ----
    uint32_t eb[3];
    double cell[3];
    for (int a = 0; a < 3; ++a) {
        const double p = nbx[a], ext = (double)nbx[3 + a] - p;
        const double big = fmax(fabs((double)nbx[a]), fabs((double)nbx[3 + a]));
        int e = ext > 0.0 ? (int)ceil(log2(ext / 255.0)) : -126;
        if (big > 0.0) e = max(e, (int)floor(log2(big)) - 30);
        e = max(e, -126);
        while (ceil(ext / ldexp(1.0, e)) > 255.0) ++e;
        if (e > 126) {
            atomicOr(&counters[2], 2u);
            e = 126;
        }
        eb[a] = (uint32_t)(e + 127);
        cell[a] = ldexp(1.0, e);
    }
    uint32_t imask = 0, meta[8], q[6][8], rank = 0, off = 0;
    for (int sl = 0; sl < 8; ++sl) {
        meta[sl] = 0;
        for (int a = 0; a < 3; ++a) {
            q[a][sl] = 255u;  // empty slot: inverted box
            q[3 + a][sl] = 0u;
        }
        const int c = child_in[sl];
        if (c < 0) continue;
        const uint32_t kid = kids[c];
        const float* cb = box + 6 * (size_t)kid;
        for (int a = 0; a < 3; ++a) {
            const double p = nbx[a];
            double l = floor(((double)cb[a] - p) / cell[a]);
            while (l > 0.0 && p + l * cell[a] > (double)cb[a]) l -= 1.0;
            if (l < 0.0) l = 0.0;
            double h = ceil(((double)cb[3 + a] - p) / cell[a]);
            while (p + h * cell[a] < (double)cb[3 + a]) h += 1.0;
            if (h > 255.0 || l > h) {
                atomicOr(&counters[2], 4u);
                h = 255.0;
            }
            q[a][sl] = (uint32_t)l;
            q[3 + a][sl] = (uint32_t)h;
        }
        if (kid_leaf[c]) {
            const uint32_t kc = cnt[kid];
            meta[sl] = (((1u << kc) - 1u) << 5) | off;
            // the <= 3 triangles below kid, in Morton order
            uint32_t tl[3], nt = 0, stk[4];
            int tp = 0;
            stk[tp++] = kid;
            while (tp > 0) {
                const uint32_t x = stk[--tp];
                if (x < n_leaves) {
                    tl[nt++] = x;
                } else {
                    stk[tp++] = (uint32_t)right[x];
                    stk[tp++] = (uint32_t)left[x];
                }
            }
            for (uint32_t k = 0; k < nt; ++k) order[tri_base + off + k] = sorted_idx[tl[k]];
            off += kc;
        } else {
            imask |= 1u << sl;
            meta[sl] = (1u << 5) | (24u + (uint32_t)sl);
        }
    }
    for (int sl = 0; sl < 8; ++sl)  // the internal children get consecutive node indices in slot order
        if ((imask >> sl) & 1u) wroot[child_base + rank++] = kids[child_in[sl]];
    uint32_t* wd = nodes8 + 20 * (size_t)w;
    wd[0] = __float_as_uint(nbx[0]);
    wd[1] = __float_as_uint(nbx[1]);
    wd[2] = __float_as_uint(nbx[2]);
    wd[3] = eb[0] | (eb[1] << 8) | (eb[2] << 16) | (imask << 24);
    wd[4] = child_base;
    wd[5] = tri_base;
    wd[6] = meta[0] | (meta[1] << 8) | (meta[2] << 16) | (meta[3] << 24);
    wd[7] = meta[4] | (meta[5] << 8) | (meta[6] << 16) | (meta[7] << 24);
    for (int pl = 0; pl < 6; ++pl) {
        wd[8 + 2 * pl] = q[pl][0] | (q[pl][1] << 8) | (q[pl][2] << 16) | (q[pl][3] << 24);
        wd[9 + 2 * pl] = q[pl][4] | (q[pl][5] << 8) | (q[pl][6] << 16) | (q[pl][7] << 24);
    }
}

}  // namespace

int prt_gpu_bvh8_build(hipStream_t st, const float* d_verts, const float* d_norms, const uint32_t* d_tri_mat, uint32_t n_tris,
                       uint32_t n_prims, const float cmin[3], const float cmax[3], PrtGpuBvh* out) {
    out->n_nodes = 0;
    out->depth = 0;
    out->d_nodes8 = nullptr;
    out->d_tris = out->d_nrms = nullptr;
    if (n_tris == 0) return 0;
    const uint32_t n = n_tris;
    const uint32_t max_nodes = n + 16u;
    uint32_t *codes = nullptr, *codes2 = nullptr, *idx = nullptr, *idx2 = nullptr, *order = nullptr, *counters = nullptr;
    uint32_t* nodes8 = nullptr;
    Range* ranges = nullptr;
    float *cbox = nullptr, *nbox = nullptr;
    void* temp = nullptr;
    float4 *tris = nullptr, *nrms = nullptr;
    auto cleanup = [&](bool keep) {
        (void)hipFree(codes); (void)hipFree(codes2); (void)hipFree(idx); (void)hipFree(idx2); (void)hipFree(order);
        (void)hipFree(counters); (void)hipFree(ranges); (void)hipFree(cbox); (void)hipFree(nbox); (void)hipFree(temp);
        if (!keep) {
            (void)hipFree(nodes8); (void)hipFree(tris); (void)hipFree(nrms);
        }
    };
#define GB_TRY(x)                   \
    do {                            \
        hipError_t e_ = (x);        \
        if (e_ != hipSuccess) {     \
            cleanup(false);         \
            return (int)e_;         \
        }                           \
    } while (0)
    GB_TRY(hipMalloc((void**)&codes, 4 * (size_t)n));
    GB_TRY(hipMalloc((void**)&codes2, 4 * (size_t)n));
    GB_TRY(hipMalloc((void**)&idx, 4 * (size_t)n));
    GB_TRY(hipMalloc((void**)&idx2, 4 * (size_t)n));
    GB_TRY(hipMalloc((void**)&order, 4 * (size_t)n));
    GB_TRY(hipMalloc((void**)&counters, 16));
    GB_TRY(hipMalloc((void**)&ranges, sizeof(Range) * (size_t)max_nodes));
    GB_TRY(hipMalloc((void**)&nodes8, 80 * (size_t)max_nodes));
    GB_TRY(hipMalloc((void**)&cbox, 192 * (size_t)max_nodes));
    GB_TRY(hipMalloc((void**)&nbox, 24 * (size_t)max_nodes));
    GB_TRY(hipMalloc((void**)&tris, 48 * (size_t)n));
    GB_TRY(hipMalloc((void**)&nrms, 48 * (size_t)n));
    // 1. Morton codes + sort
    float3 mn = make_float3(cmin[0], cmin[1], cmin[2]);
    float3 inv;
    inv.x = cmax[0] > cmin[0] ? 1024.0f / (cmax[0] - cmin[0]) : 0.0f;
    inv.y = cmax[1] > cmin[1] ? 1024.0f / (cmax[1] - cmin[1]) : 0.0f;
    inv.z = cmax[2] > cmin[2] ? 1024.0f / (cmax[2] - cmin[2]) : 0.0f;
    hipLaunchKernelGGL(k_morton, dim3((n + 255u) / 256u), dim3(256), 0, st, d_verts, n, mn, inv, codes, idx);
    size_t temp_bytes = 0;
    GB_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, codes, codes2, idx, idx2, n, 0, 30, st));
    GB_TRY(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
    GB_TRY(rocprim::radix_sort_pairs(temp, temp_bytes, codes, codes2, idx, idx2, n, 0, 30, st));
    // 2./3. breadth-first construction: one launch per level
    const uint32_t init[4] = {1u, 0u, 0u, 0u};  // node count (the root exists), triangle cursor, error flags
    GB_TRY(hipMemcpyAsync(counters, init, sizeof(init), hipMemcpyHostToDevice, st));
    const Range root{0u, n, 0u};
    GB_TRY(hipMemcpyAsync(ranges, &root, sizeof(root), hipMemcpyHostToDevice, st));
    uint32_t level_begin[64];
    uint32_t n_levels = 0, begin = 0, end = 1;
    while (begin < end && n_levels < 63u) {
        level_begin[n_levels++] = begin;
        hipLaunchKernelGGL(k_split, dim3((end - begin + 127u) / 128u), dim3(128), 0, st, codes2, idx2, ranges, nodes8, order,
                           counters, begin, end, max_nodes);
        uint32_t cnt[4];
        GB_TRY(hipMemcpyAsync(cnt, counters, sizeof(cnt), hipMemcpyDeviceToHost, st));
        GB_TRY(hipStreamSynchronize(st));
        if (cnt[2]) {
            cleanup(false);
            return -2;
        }
        begin = end;
        end = cnt[0];
    }
    level_begin[n_levels] = end;
    const uint32_t n_nodes = end;
    {   // an unfinished tree (more than 63 levels) leaves `order` unfinished: stop before anything gathers through it
        uint32_t chk[4];
        GB_TRY(hipMemcpyAsync(chk, counters, sizeof(chk), hipMemcpyDeviceToHost, st));
        GB_TRY(hipStreamSynchronize(st));
        if (begin < end || chk[2] || chk[1] != n) {
            cleanup(false);
            return -3;
        }
    }
    // 4. boxes bottom-up, quantization, records
    for (uint32_t L = n_levels; L-- > 0;) {
        const uint32_t b = level_begin[L], e = level_begin[L + 1];
        hipLaunchKernelGGL(k_boxes, dim3((e - b + 127u) / 128u), dim3(128), 0, st, d_verts, nodes8, order, cbox, nbox, b, e);
    }
    hipLaunchKernelGGL(k_quantize, dim3((n_nodes + 127u) / 128u), dim3(128), 0, st, nodes8, cbox, nbox, n_nodes, counters, 20u);
    hipLaunchKernelGGL(k_records, dim3((n + 255u) / 256u), dim3(256), 0, st, d_verts, d_norms, d_tri_mat, order, n, n_prims, tris,
                       nrms);
    uint32_t cnt[4];
    GB_TRY(hipMemcpyAsync(cnt, counters, sizeof(cnt), hipMemcpyDeviceToHost, st));
    GB_TRY(hipStreamSynchronize(st));
    GB_TRY(hipGetLastError());
    if (cnt[2] || cnt[1] != n) {
        cleanup(false);
        return -3;
    }
    cleanup(true);
    out->d_nodes8 = nodes8;
    out->d_tris = tris;
    out->d_nrms = nrms;
    out->n_nodes = n_nodes;
    out->depth = n_levels;
    return 0;
#undef GB_TRY
}


// PLOC's weak spot is the TOP of the tree: big clusters that are far apart in Morton order never see each other, and the
// top levels are what every ray visits.  The passes therefore stop when at most PLOC_TOP clusters are left, and the tree
// above them is built on the host by a full-sweep SAH over those few thousand boxes (weights = their triangle counts):
// microseconds of work, and the upper levels get the splits a top-down builder would choose.
#define PLOC_TOP 4096u
namespace {
struct TopBuilder {
    const float* cbox;       // [m][6] boxes of the remaining clusters
    const uint32_t* ccnt;    // [m] triangles per cluster
    const uint32_t* cid;     // [m] their binary node ids
    std::vector<int> left, right;  // new nodes, in creation (post) order
    std::vector<float> box;
    std::vector<uint32_t> cnt;
    uint32_t next_id;
    // builds the subtree over items[lo, hi) (indices into the cluster arrays); returns the binary node id of its root
    uint32_t build(std::vector<uint32_t>& items, size_t lo, size_t hi) {
        if (hi - lo == 1) return cid[items[lo]];
        float bb[6] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
        for (size_t i = lo; i < hi; ++i)
            for (int a = 0; a < 3; ++a) {
                bb[a] = std::min(bb[a], cbox[6 * (size_t)items[i] + a]);
                bb[3 + a] = std::max(bb[3 + a], cbox[6 * (size_t)items[i] + 3 + a]);
            }
        const size_t n = hi - lo;
        double best = 1e300;
        int best_axis = 0;
        size_t best_split = n / 2;
        std::vector<uint32_t> sorted(items.begin() + (long)lo, items.begin() + (long)hi);
        std::vector<double> right_cost(n);
        auto by_axis = [&](int axis) {
            std::sort(sorted.begin(), sorted.end(), [&](uint32_t x, uint32_t y) {
                const float cx = cbox[6 * (size_t)x + axis] + cbox[6 * (size_t)x + 3 + axis], cy = cbox[6 * (size_t)y + axis] + cbox[6 * (size_t)y + 3 + axis];
                return cx < cy || (cx == cy && x < y);
            });
        };
        for (int axis = 0; axis < 3; ++axis) {
            by_axis(axis);
            float rb[6] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
            double rc = 0.0;
            for (size_t i = n; i-- > 1;) {  // right part = sorted[i .. n)
                for (int a = 0; a < 3; ++a) {
                    rb[a] = std::min(rb[a], cbox[6 * (size_t)sorted[i] + a]);
                    rb[3 + a] = std::max(rb[3 + a], cbox[6 * (size_t)sorted[i] + 3 + a]);
                }
                rc += ccnt[sorted[i]];
                right_cost[i] = (double)ploc_half_area(rb) * rc;
            }
            float lb[6] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
            double lc = 0.0;
            for (size_t i = 0; i + 1 < n; ++i) {  // left part = sorted[0 .. i]
                for (int a = 0; a < 3; ++a) {
                    lb[a] = std::min(lb[a], cbox[6 * (size_t)sorted[i] + a]);
                    lb[3 + a] = std::max(lb[3 + a], cbox[6 * (size_t)sorted[i] + 3 + a]);
                }
                lc += ccnt[sorted[i]];
                const double c = (double)ploc_half_area(lb) * lc + right_cost[i + 1];
                // ties go to the split nearest the middle: clusters with IDENTICAL boxes (coincident triangles) make every
                // split cost the same, and "first split wins" peeled one cluster off per level: a chain thousands of levels
                // deep, far beyond the 63 levels the emission allows
                const size_t mid = n / 2, di = i + 1 > mid ? i + 1 - mid : mid - (i + 1), db = best_split > mid ? best_split - mid : mid - best_split;
                if (c < best || (c == best && di < db)) {
                    best = c;
                    best_axis = axis;
                    best_split = i + 1;
                }
            }
        }
        by_axis(best_axis);  // (the comparator is a strict total order: the same permutation as when the split was found)
        std::copy(sorted.begin(), sorted.end(), items.begin() + (long)lo);
        const uint32_t l = build(items, lo, lo + best_split);
        const uint32_t r = build(items, lo + best_split, hi);
        uint32_t c = 0;
        for (size_t i = lo; i < hi; ++i) c += ccnt[items[i]];
        left.push_back((int)l);
        right.push_back((int)r);
        box.insert(box.end(), bb, bb + 6);
        cnt.push_back(c);
        return next_id++;  // children were created first: their ids are smaller
    }
};
}  // namespace

// The quality builder: Morton sort + PLOC + optimal collapse (see the block comment above k_ploc_leaves).  Same contract
// as prt_gpu_bvh8_build.
int prt_gpu_bvh8_build_ploc(hipStream_t st, const float* d_verts, const float* d_norms, const uint32_t* d_tri_mat, uint32_t n_tris,
                            uint32_t n_prims, const float cmin[3], const float cmax[3], PrtGpuBvh* out, float leaf_cost) {
    out->n_nodes = 0;
    out->depth = 0;
    out->d_nodes8 = nullptr;
    out->d_tris = out->d_nrms = nullptr;
    if (n_tris == 0) return 0;
    if (const char* e = getenv("PRT_PLOC_RADIUS")) g_ploc_radius = std::min(PLOC_RADIUS_MAX, std::max(1, atoi(e)));
    if (const char* e = getenv("PRT_PLOC_CI")) g_ploc_ci = (float)atof(e);
    const float ploc_ci = leaf_cost > 0.0f ? leaf_cost : g_ploc_ci;
    const uint32_t n = n_tris;
    const uint32_t n_bin = 2u * n;  // binary nodes: n leaves + n - 1 internal
    const uint32_t max_nodes = n + 16u;
    uint32_t *codes = nullptr, *codes2 = nullptr, *idx = nullptr, *idx2 = nullptr, *order = nullptr, *counters = nullptr;
    uint32_t *cl_a = nullptr, *cl_b = nullptr, *nn = nullptr, *cnt = nullptr, *wroot = nullptr, *nodes8 = nullptr;
    unsigned long long *flags = nullptr, *scan = nullptr;
    int *left = nullptr, *right = nullptr;
    float *box = nullptr, *cost = nullptr;
    uint8_t* pick = nullptr;
    void *temp = nullptr, *temp2 = nullptr;
    float4 *tris = nullptr, *nrms = nullptr;
    auto cleanup = [&](bool keep) {
        for (void* p : {(void*)codes, (void*)codes2, (void*)idx, (void*)idx2, (void*)order, (void*)counters, (void*)cl_a, (void*)cl_b,
                        (void*)nn, (void*)cnt, (void*)wroot, (void*)flags, (void*)scan, (void*)left, (void*)right, (void*)box,
                        (void*)cost, (void*)pick, temp, temp2})
            (void)hipFree(p);
        if (!keep) {
            (void)hipFree(nodes8);
            (void)hipFree(tris);
            (void)hipFree(nrms);
        }
    };
#define GB_TRY(x)                   \
    do {                            \
        hipError_t e_ = (x);        \
        if (e_ != hipSuccess) {     \
            cleanup(false);         \
            return (int)e_;         \
        }                           \
    } while (0)
    GB_TRY(hipMalloc((void**)&codes, 4 * (size_t)n));
    GB_TRY(hipMalloc((void**)&codes2, 4 * (size_t)n));
    GB_TRY(hipMalloc((void**)&idx, 4 * (size_t)n));
    GB_TRY(hipMalloc((void**)&idx2, 4 * (size_t)n));
    GB_TRY(hipMalloc((void**)&order, 4 * (size_t)n));
    GB_TRY(hipMalloc((void**)&counters, 16));
    GB_TRY(hipMalloc((void**)&cl_a, 4 * (size_t)n));
    GB_TRY(hipMalloc((void**)&cl_b, 4 * (size_t)n));
    GB_TRY(hipMalloc((void**)&nn, std::max<size_t>(4 * (size_t)n, 4 * (size_t)PLOC_TOP)));
    GB_TRY(hipMalloc((void**)&flags, std::max<size_t>(8 * (size_t)n, 32 * (size_t)PLOC_TOP)));  // (also the top builder's gather buffers)
    GB_TRY(hipMalloc((void**)&scan, std::max<size_t>(8 * (size_t)n, 32 * (size_t)PLOC_TOP)));
    GB_TRY(hipMalloc((void**)&cnt, 4 * (size_t)n_bin));
    GB_TRY(hipMalloc((void**)&left, 4 * (size_t)n_bin));
    GB_TRY(hipMalloc((void**)&right, 4 * (size_t)n_bin));
    GB_TRY(hipMalloc((void**)&box, 24 * (size_t)n_bin));
    GB_TRY(hipMalloc((void**)&cost, 32 * (size_t)n_bin));
    GB_TRY(hipMalloc((void**)&pick, 8 * (size_t)n_bin));
    GB_TRY(hipMalloc((void**)&wroot, 4 * (size_t)max_nodes));
    GB_TRY(hipMalloc((void**)&nodes8, 80 * (size_t)max_nodes));
    GB_TRY(hipMalloc((void**)&tris, 48 * (size_t)n));
    GB_TRY(hipMalloc((void**)&nrms, 48 * (size_t)n));
    // 1. Morton codes + sort
    float3 mn = make_float3(cmin[0], cmin[1], cmin[2]);
    float3 inv;
    inv.x = cmax[0] > cmin[0] ? 1024.0f / (cmax[0] - cmin[0]) : 0.0f;
    inv.y = cmax[1] > cmin[1] ? 1024.0f / (cmax[1] - cmin[1]) : 0.0f;
    inv.z = cmax[2] > cmin[2] ? 1024.0f / (cmax[2] - cmin[2]) : 0.0f;
    hipLaunchKernelGGL(k_morton, dim3((n + 255u) / 256u), dim3(256), 0, st, d_verts, n, mn, inv, codes, idx);
    size_t temp_bytes = 0;
    GB_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, codes, codes2, idx, idx2, n, 0, 30, st));
    GB_TRY(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
    GB_TRY(rocprim::radix_sort_pairs(temp, temp_bytes, codes, codes2, idx, idx2, n, 0, 30, st));
    // 2. PLOC
    hipLaunchKernelGGL(k_ploc_leaves, dim3((n + 255u) / 256u), dim3(256), 0, st, d_verts, idx2, n, box, left, right, cnt, cl_a);
    size_t scan_bytes = 0;
    GB_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, flags, scan, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), st));
    GB_TRY(hipMalloc(&temp2, scan_bytes ? scan_bytes : 16));
    std::vector<uint32_t> pass_begin;  // binary node id ranges created per pass
    uint32_t m = n, next_node = n;
    uint32_t* cl_in = cl_a;
    uint32_t* cl_out = cl_b;
    pass_begin.push_back(n);
    // g_ploc_top = 1 (PRT_PLOC_TOP=device): the passes go on below PLOC_TOP clusters with the FULL search, down to the root,
    // and no part of the build runs on the host; 0 (default): the host's SAH sweep builds the top (2b below)
    if (const char* e = getenv("PRT_PLOC_TOP")) g_ploc_top = !strcmp(e, "device") ? 1 : 0;
    const uint32_t stop_at = g_ploc_top ? 1u : PLOC_TOP;
    for (uint32_t pass = 0; m > stop_at; ++pass) {
        if (pass > 4096u) {  // every pass merges at least the globally best pair: cannot happen
            cleanup(false);
            return -4;
        }
        const dim3 g((m + 255u) / 256u), b(256);
        if (m > PLOC_TOP)
            hipLaunchKernelGGL(k_ploc_nn, g, b, 0, st, cl_in, m, box, nn, g_ploc_radius);
        else
            hipLaunchKernelGGL(k_ploc_nn_full, g, b, 0, st, cl_in, m, box, nn);
        hipLaunchKernelGGL(k_ploc_flags, g, b, 0, st, nn, m, flags);
        size_t sb = scan_bytes;
        GB_TRY(rocprim::exclusive_scan(temp2, sb, flags, scan, 0ull, (size_t)m, rocprim::plus<unsigned long long>(), st));
        hipLaunchKernelGGL(k_ploc_merge, g, b, 0, st, cl_in, nn, scan, m, next_node, cl_out, box, left, right, cnt);
        unsigned long long last_scan = 0, last_flag = 0;
        GB_TRY(hipMemcpyAsync(&last_scan, scan + (m - 1u), 8, hipMemcpyDeviceToHost, st));
        GB_TRY(hipMemcpyAsync(&last_flag, flags + (m - 1u), 8, hipMemcpyDeviceToHost, st));
        GB_TRY(hipStreamSynchronize(st));
        const unsigned long long tot = last_scan + last_flag;
        const uint32_t merged = (uint32_t)(tot & 0xFFFFFFFFull), kept = (uint32_t)(tot >> 32);
        if (merged == 0u || kept + merged != m) {
            cleanup(false);
            return -5;
        }
        next_node += merged;
        pass_begin.push_back(next_node);
        m = kept;
        std::swap(cl_in, cl_out);
    }
    // 3. collapse DP, bottom-up: the leaves, then pass by pass
    hipLaunchKernelGGL(k_ploc_dp, dim3((n + 127u) / 128u), dim3(128), 0, st, 0u, n, n, left, right, box, cnt, cost, pick, ploc_ci);
    for (size_t p = 0; p + 1 < pass_begin.size(); ++p) {
        const uint32_t nb = pass_begin[p], ne = pass_begin[p + 1];
        if (ne > nb) hipLaunchKernelGGL(k_ploc_dp, dim3((ne - nb + 127u) / 128u), dim3(128), 0, st, nb, ne, n, left, right, box, cnt, cost, pick, ploc_ci);
    }
    // 2b. the top of the tree over the m remaining clusters: full-sweep SAH on the host, its DP rows as well
    uint32_t root_bin = 0u;
    if (m == 1u) {  // the passes went all the way (g_ploc_top): the last cluster is the root
        GB_TRY(hipMemcpyAsync(&root_bin, cl_in, 4, hipMemcpyDeviceToHost, st));
        GB_TRY(hipStreamSynchronize(st));
    } else {
        std::vector<uint32_t> h_cl(m), h_cnt(m);
        std::vector<float> h_box(6 * (size_t)m), h_cost(8 * (size_t)m);
        // (the clusters' rows are scattered over the node arrays: gathered on the device into the scan / flag buffers, which
        // are free now: 8 n bytes each, m <= 4096 rows of 56 B)
        float* g_box = (float*)flags;
        float* g_cost = (float*)scan;
        uint32_t* g_cnt = nn;
        hipLaunchKernelGGL(k_ploc_gather, dim3((m + 255u) / 256u), dim3(256), 0, st, cl_in, m, box, cnt, cost, g_box, g_cnt, g_cost);
        GB_TRY(hipMemcpyAsync(h_cl.data(), cl_in, 4 * (size_t)m, hipMemcpyDeviceToHost, st));
        GB_TRY(hipMemcpyAsync(h_box.data(), g_box, 24 * (size_t)m, hipMemcpyDeviceToHost, st));
        GB_TRY(hipMemcpyAsync(h_cnt.data(), g_cnt, 4 * (size_t)m, hipMemcpyDeviceToHost, st));
        GB_TRY(hipMemcpyAsync(h_cost.data(), g_cost, 32 * (size_t)m, hipMemcpyDeviceToHost, st));
        GB_TRY(hipStreamSynchronize(st));
        TopBuilder tb;
        tb.cbox = h_box.data();
        tb.ccnt = h_cnt.data();
        tb.cid = h_cl.data();
        tb.next_id = next_node;
        std::vector<uint32_t> items(m);
        for (uint32_t i = 0; i < m; ++i) items[i] = i;
        root_bin = tb.build(items, 0, m);
        const uint32_t n_top = tb.next_id - next_node;
        if (n_top != m - 1u || tb.next_id != (n > 1u ? 2u * n - 1u : 1u)) {
            cleanup(false);
            return -6;
        }
        if (n_top) {
            // DP rows of the top nodes, in creation order (children first); a child is a cluster (row copied above) or
            // an earlier top node
            std::vector<float> t_cost(8 * (size_t)n_top);
            std::vector<uint8_t> t_pick(8 * (size_t)n_top);
            std::vector<uint32_t> slot_of_id;  // cluster node id -> index into h_cost: via a sorted lookup
            std::vector<std::pair<uint32_t, uint32_t>> lut(m);
            for (uint32_t i = 0; i < m; ++i) lut[i] = {h_cl[i], i};
            std::sort(lut.begin(), lut.end());
            auto row = [&](int id) -> const float* {
                if ((uint32_t)id >= next_node) return &t_cost[8 * (size_t)((uint32_t)id - next_node)];
                const auto it = std::lower_bound(lut.begin(), lut.end(), std::make_pair((uint32_t)id, 0u));
                return &h_cost[8 * (size_t)it->second];
            };
            for (uint32_t k = 0; k < n_top; ++k)
                ploc_dp_node(&tb.box[6 * (size_t)k], tb.cnt[k], row(tb.left[k]), row(tb.right[k]), ploc_ci, &t_cost[8 * (size_t)k], &t_pick[8 * (size_t)k]);
            GB_TRY(hipMemcpyAsync(left + next_node, tb.left.data(), 4 * (size_t)n_top, hipMemcpyHostToDevice, st));
            GB_TRY(hipMemcpyAsync(right + next_node, tb.right.data(), 4 * (size_t)n_top, hipMemcpyHostToDevice, st));
            GB_TRY(hipMemcpyAsync(box + 6 * (size_t)next_node, tb.box.data(), 24 * (size_t)n_top, hipMemcpyHostToDevice, st));
            GB_TRY(hipMemcpyAsync(cnt + next_node, tb.cnt.data(), 4 * (size_t)n_top, hipMemcpyHostToDevice, st));
            GB_TRY(hipMemcpyAsync(cost + 8 * (size_t)next_node, t_cost.data(), 32 * (size_t)n_top, hipMemcpyHostToDevice, st));
            GB_TRY(hipMemcpyAsync(pick + 8 * (size_t)next_node, t_pick.data(), 8 * (size_t)n_top, hipMemcpyHostToDevice, st));
            GB_TRY(hipStreamSynchronize(st));  // (the host vectors go out of scope below)
        }
    }
    // 4. emission, breadth first
    const uint32_t init[4] = {1u, 0u, 0u, 0u};  // wide node count (the root exists), triangle cursor, error flags
    GB_TRY(hipMemcpyAsync(counters, init, sizeof(init), hipMemcpyHostToDevice, st));
    GB_TRY(hipMemcpyAsync(wroot, &root_bin, 4, hipMemcpyHostToDevice, st));
    uint32_t n_levels = 0, begin = 0, end = 1;
    while (begin < end && n_levels < 63u) {
        ++n_levels;
        hipLaunchKernelGGL(k_ploc_emit, dim3((end - begin + 63u) / 64u), dim3(64), 0, st, begin, end, n, max_nodes, wroot, left, right, box,
                           cnt, pick, idx2, nodes8, order, counters);
        uint32_t c4[4];
        GB_TRY(hipMemcpyAsync(c4, counters, sizeof(c4), hipMemcpyDeviceToHost, st));
        GB_TRY(hipStreamSynchronize(st));
        if (c4[2]) {
            cleanup(false);
            return -2;
        }
        begin = end;
        end = c4[0];
    }
    const uint32_t n_nodes = end;
    uint32_t c4[4];
    GB_TRY(hipMemcpyAsync(c4, counters, sizeof(c4), hipMemcpyDeviceToHost, st));
    GB_TRY(hipStreamSynchronize(st));
    GB_TRY(hipGetLastError());
    // a tree deeper than the 63 levels emitted above leaves `order` unfinished: stop BEFORE k_records gathers vertices
    // through it (uninitialised entries would be wild indices: a memory fault, not an error code)
    if (begin < end || c4[2] || c4[1] != n) {
        cleanup(false);
        return -3;
    }
    hipLaunchKernelGGL(k_records, dim3((n + 255u) / 256u), dim3(256), 0, st, d_verts, d_norms, d_tri_mat, order, n, n_prims, tris, nrms);
    GB_TRY(hipStreamSynchronize(st));
    GB_TRY(hipGetLastError());
    cleanup(true);
    out->d_nodes8 = nodes8;
    out->d_tris = tris;
    out->d_nrms = nrms;
    out->n_nodes = n_nodes;
    out->depth = n_levels;
    return 0;
#undef GB_TRY
}


// ---------------------------------------------------------------------------------------------------------------------
// Refit ("SAH refit"): new vertex positions over the EXISTING 8-wide topology (deforming geometry; the
// reference rebuilds its OptiX structures from scratch, optix/renderer.cpp:736-871).  Three steps, all on the device:
// the triangle / normal records are rewritten in place in their slot order (a record's w = global primitive index names
// the input triangle), the children's exact fp32 boxes are recomputed bottom-up, level by level (the host hands over
// the nodes of every level as a list: the builders' node orders differ; the same pass the builder runs, reading the
// records instead of an order array), and every node is
// re-quantized by k_quantize with its containment fix-ups.  No sort, no new topology: a refitted tree is as good as the
// deformation is small (tools/build_compare.py measures the traversal penalty).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__global__ void k_refit_records(const float* __restrict__ verts, const float* __restrict__ norms, uint32_t n, uint32_t n_prims,
                                float4* __restrict__ tris, float4* __restrict__ nrms) {
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= n) return;
    const float4 a = tris[3 * (size_t)slot + 0], b = tris[3 * (size_t)slot + 1];
    const uint32_t t = __float_as_uint(a.w) - n_prims;
    if (t >= n) return;  // (cannot happen for records the builders wrote)
    const float* v = verts + 9 * (size_t)t;
    tris[3 * (size_t)slot + 0] = make_float4(v[0], v[1], v[2], a.w);
    tris[3 * (size_t)slot + 1] = make_float4(v[3], v[4], v[5], b.w);
    tris[3 * (size_t)slot + 2] = make_float4(v[6], v[7], v[8], 0.0f);
    if (norms) {
        const float* q = norms + 9 * (size_t)t;
        nrms[3 * (size_t)slot + 0] = make_float4(q[0], q[1], q[2], 0.0f);
        nrms[3 * (size_t)slot + 1] = make_float4(q[3], q[4], q[5], 0.0f);
        nrms[3 * (size_t)slot + 2] = make_float4(q[6], q[7], q[8], 0.0f);
    }
}

__global__ void k_refit_boxes(const float4* __restrict__ tris, const uint32_t* __restrict__ nodes8, uint32_t stride_dwords,
                              float* __restrict__ cbox, float* __restrict__ nbox, const uint32_t* __restrict__ level_nodes,
                              uint32_t list_begin, uint32_t list_end) {
    const uint32_t li = list_begin + blockIdx.x * 128u + threadIdx.x;
    if (li >= list_end) return;
    const uint32_t nd = level_nodes[li];
    const uint32_t* w = nodes8 + (size_t)stride_dwords * nd;
    const uint32_t imask = w[3] >> 24, child_base = w[4], tri_base = w[5];
    float nmn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    float nmx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    uint32_t rank = 0;
    for (int c = 0; c < 8; ++c) {
        const uint32_t meta = (w[6 + (c >> 2)] >> (8 * (c & 3))) & 0xFFu;
        float* cb = cbox + 48 * (size_t)nd + 6 * c;
        float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
        float mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
        if (meta) {
            if ((imask >> c) & 1u) {
                const float* cbx = nbox + 6 * (size_t)(child_base + rank);
                ++rank;
                for (int a = 0; a < 3; ++a) {
                    mn[a] = cbx[a];
                    mx[a] = cbx[3 + a];
                }
            } else {
                const uint32_t cnt = (uint32_t)__popc(meta >> 5), first = tri_base + (meta & 31u);
                for (uint32_t k = 0; k < cnt; ++k)
                    for (int vv = 0; vv < 3; ++vv) {
                        const float4 v = tris[3 * (size_t)(first + k) + vv];
                        mn[0] = fminf(mn[0], v.x); mx[0] = fmaxf(mx[0], v.x);
                        mn[1] = fminf(mn[1], v.y); mx[1] = fmaxf(mx[1], v.y);
                        mn[2] = fminf(mn[2], v.z); mx[2] = fmaxf(mx[2], v.z);
                    }
            }
            for (int a = 0; a < 3; ++a) {
                nmn[a] = fminf(nmn[a], mn[a]);
                nmx[a] = fmaxf(nmx[a], mx[a]);
            }
        }
        for (int a = 0; a < 3; ++a) {
            cb[a] = mn[a];
            cb[3 + a] = mx[a];
        }
    }
    for (int a = 0; a < 3; ++a) {
        nbox[6 * (size_t)nd + a] = nmn[a];
        nbox[6 * (size_t)nd + 3 + a] = nmx[a];
    }
}
}  // namespace

int prt_gpu_bvh8_refit(hipStream_t st, uint32_t* d_nodes8, uint32_t stride_dwords, uint32_t n_nodes, const uint32_t* level_nodes,
                       const uint32_t* level_start, uint32_t n_levels, const float* d_verts, const float* d_norms, uint32_t n_tris,
                       uint32_t n_prims, float4* d_tris, float4* d_nrms, float root_box[6]) {
    if (n_nodes == 0 || n_tris == 0 || n_levels == 0) return 0;
    float *cbox = nullptr, *nbox = nullptr;
    uint32_t *counters = nullptr, *d_list = nullptr;
    auto drop = [&]() {
        (void)hipFree(cbox);
        (void)hipFree(nbox);
        (void)hipFree(counters);
        (void)hipFree(d_list);
    };
    hipError_t e = hipMalloc((void**)&cbox, 48 * sizeof(float) * (size_t)n_nodes);
    if (e == hipSuccess) e = hipMalloc((void**)&nbox, 6 * sizeof(float) * (size_t)n_nodes);
    if (e == hipSuccess) e = hipMalloc((void**)&counters, 4 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&d_list, sizeof(uint32_t) * (size_t)n_nodes);
    if (e == hipSuccess) e = hipMemcpyAsync(d_list, level_nodes, sizeof(uint32_t) * (size_t)n_nodes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(counters, 0, 4 * sizeof(uint32_t), st);
    if (e != hipSuccess) {
        drop();
        return (int)e;
    }
    hipLaunchKernelGGL(k_refit_records, dim3((n_tris + 255u) / 256u), dim3(256), 0, st, d_verts, d_norms, n_tris, n_prims, d_tris, d_nrms);
    for (uint32_t l = n_levels; l-- > 0;) {  // deepest level first: a node's internal children are on the next level
        const uint32_t b = level_start[l], en = level_start[l + 1];
        if (en > b)
            hipLaunchKernelGGL(k_refit_boxes, dim3((en - b + 127u) / 128u), dim3(128), 0, st, (const float4*)d_tris, (const uint32_t*)d_nodes8,
                               stride_dwords, cbox, nbox, (const uint32_t*)d_list, b, en);
    }
    hipLaunchKernelGGL(k_quantize, dim3((n_nodes + 127u) / 128u), dim3(128), 0, st, d_nodes8, (const float*)cbox, (const float*)nbox, n_nodes,
                       counters, stride_dwords);
    uint32_t flags[4] = {0, 0, 0, 0};
    e = hipMemcpyAsync(flags, counters, sizeof(flags), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(root_box, nbox, 6 * sizeof(float), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    drop();
    if (e != hipSuccess) return (int)e;
    return flags[2] ? -6 : 0;  // a box that does not fit its node's 8-bit grid (non-finite or absurdly large coordinates)
}

// ---------------------------------------------------------------------------------------------------------------------
// Measurement aid, not on the default path (prt_set_param("sort_rays", 1 | 2), tools/sort_ab.py; the round-2 review asked
// for ray binning to be measured on the memory-bound 10 M-triangle config): order of a bounce's front rays by the cell of
// their origin in a 32^3 grid over the BVH's root box (Morton order) and their direction octant.  The traversal kernel
// then takes its rays through the permutation; nothing is moved.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__global__ void k_ray_keys(const float4* __restrict__ ro, const float4* __restrict__ rd, uint32_t n, float3 bmin, float3 binv,
                           uint32_t mode, uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 O = ro[i], D = rd[i];
    const float c[3] = {(O.x - bmin.x) * binv.x, (O.y - bmin.y) * binv.y, (O.z - bmin.z) * binv.z};
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) q[a] = (uint32_t)fminf(fmaxf(c[a] * 32.0f, 0.0f), 31.0f);
    const uint32_t cell = expand10(q[0]) | (expand10(q[1]) << 1) | (expand10(q[2]) << 2);  // 15 bits
    const uint32_t oct = (D.x < 0.0f ? 1u : 0u) | (D.y < 0.0f ? 2u : 0u) | (D.z < 0.0f ? 4u : 0u);
    keys[i] = mode == 2u ? (oct << 15) | cell : (cell << 3) | oct;
    idx[i] = i;
}
}  // namespace

size_t prt_sort_rays_temp_bytes(uint32_t n) {
    size_t temp_bytes = 0;
    uint32_t* p = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, temp_bytes, p, p, p, p, n, 0, 18, (hipStream_t)0) != hipSuccess) return 0;
    return temp_bytes;
}

int prt_sort_rays(hipStream_t st, const float4* ro, const float4* rd, uint32_t n, const float bmin[3], const float bmax[3],
                  uint32_t mode, uint32_t* keys, uint32_t* keys2, uint32_t* idx, uint32_t* idx2, void* temp, size_t temp_bytes) {
    if (n == 0) return 0;
    const float3 mn = make_float3(bmin[0], bmin[1], bmin[2]);
    const float3 inv = make_float3(1.0f / fmaxf(bmax[0] - bmin[0], 1e-20f), 1.0f / fmaxf(bmax[1] - bmin[1], 1e-20f),
                                   1.0f / fmaxf(bmax[2] - bmin[2], 1e-20f));
    hipLaunchKernelGGL(k_ray_keys, dim3((n + 255u) / 256u), dim3(256), 0, st, ro, rd, n, mn, inv, mode, keys, idx);
    GB_CHECK(rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys2, idx, idx2, n, 0, 18, st));
    return 0;
}
