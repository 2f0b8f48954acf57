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
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
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
                           uint32_t n_nodes, uint32_t* __restrict__ counters) {
    const uint32_t nd = blockIdx.x * 128u + threadIdx.x;
    if (nd >= n_nodes) return;
    uint32_t* w = nodes8 + 20 * (size_t)nd;
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
    // 4. boxes bottom-up, quantization, records
    for (uint32_t L = n_levels; L-- > 0;) {
        const uint32_t b = level_begin[L], e = level_begin[L + 1];
        hipLaunchKernelGGL(k_boxes, dim3((e - b + 127u) / 128u), dim3(128), 0, st, d_verts, nodes8, order, cbox, nbox, b, e);
    }
    hipLaunchKernelGGL(k_quantize, dim3((n_nodes + 127u) / 128u), dim3(128), 0, st, nodes8, cbox, nbox, n_nodes, counters);
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
}
