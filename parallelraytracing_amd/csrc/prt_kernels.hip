// prt_kernels.hip — hand-written HIP kernels of the wavefront path tracer for gfx950 (MI355X).
//
// Pipeline per sample batch (one batch = `S` samples per local pixel in flight):
//   k_raygen -> [ k_intersect -> k_shade ] x max_depth -> k_accumulate
// replacing GenerateCameraRaysKernel / IntersectClosestKernel / ShadeHitsKernel / BlitRadianceKernel +
// addBufferGPU of the reference (src/backend/cuda_wavefront/renderer.cu:186-348, src/core/film.cu:79-88).
//
// Layout decisions:
//  * ray records are DENSE per bounce (structure of float4 arrays, slot k of bounce d), so every
//    kernel reads coalesced 16-B/lane streams; the reference gathers per-pixel state through a queue of
//    pixel indices instead (renderer.cu:226-228).
//  * the hit buffer is one u32 per ray slot (primitive id); the shade kernel re-derives position/normal
//    with the same device function the traversal used, so nothing else crosses HBM.
//  * wave64 ballot compaction into the next bounce's buffer: one atomic per 1024-thread block.
#include "prt_kernels.h"

#include "prt_device.h"

#define HIT_MISS 0xFFFFFFFFu
#define HIT_DEAD 0xFFFFFFFEu

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an XCD's L2).  Give each XCD a
// contiguous run of logical blocks so neighbouring rays (= neighbouring BVH subtrees) share an L2.
// Bijective for every n.  Speed only; correctness never depends on placement.
PRT_DEV uint32_t xcd_remap(uint32_t b, uint32_t n) {
    const uint32_t q = n >> 3, r = n & 7u;
    const uint32_t xcd = b & 7u, i = b >> 3;
    const uint32_t start = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return start + i;
}

// ---------------------------------------------------------------------------------------------------------
// Two-sided ray buffers.  Every producer (k_raygen, k_shade) classifies the ray it emits: rays that can still
// hit a triangle (they enter the BVH root box before their analytic hit) are packed from the FRONT of the next
// buffer, all others from the BACK.  The traversal kernel only sees the front part; the shade kernel sees both.
// Per bounce d: counts[d*64] = front count A_d, counts[d*64 + 32] = back count B_d (separate 128-B lines).
// ---------------------------------------------------------------------------------------------------------
#define CNT_STRIDE PRT_CNT_STRIDE
#define CNT_A(c, d) (c)[(d) * CNT_STRIDE]
#define CNT_B(c, d) (c)[(d) * CNT_STRIDE + 32u]
#define CNT_C(c, d) (c)[(d) * CNT_STRIDE + 16u]
#define PRODUCER_BLOCK 1024  // k_raygen
#ifndef SHADE_BLOCK
#define SHADE_BLOCK 512      // k_shade: its variants with 82 SGPRs / 78 VGPRs keep 6 waves per SIMD this way (1024: 4)
#endif

// Slot reservation for a BLOCK-thread block with ONE atomic per side (a counter word sustains only ~88
// returning atomics/us, MI355X_MICROARCH.md "dequeue"; the per-wave form — the wave64 equivalent of the
// reference's warp-aggregated AllocateSlot, renderer.cu:43-67 — costs ~0.9 ms per 8 M rays).
// Returns the buffer slot for this thread, or 0xFFFFFFFF if it emits nothing.  All threads must call.
// A kernel that calls this MORE THAN ONCE per block must put a __syncthreads() between two calls: the per-wave counts
// s_a/s_b/s_c are read by the other waves after the second barrier below, and a wave that ran ahead into the next call
// would overwrite its own entries under them (k_raygen's jitter loop does; k_shade calls once).
// `mult` (block-uniform) reserves that many slots per emitting thread.  With *stride (the block's emitting threads) copy j
// of a thread's ray can go to the returned slot + j * *stride (front) or - j * *stride (back), so that every copy index
// forms one contiguous run; with *base (where the block's reservation starts) the caller can instead give each thread
// `mult` consecutive slots: base + (slot - base) * mult + j (k_raygen's pixel-major order).
template <int BLOCK>
PRT_DEV uint32_t block_alloc2(bool front, bool back, bool done, uint32_t* cntA, uint32_t* cntB, uint32_t* cntC,
                              uint32_t cap, uint32_t mult = 1u, uint32_t* stride = nullptr, uint32_t* base = nullptr) {
    __shared__ uint32_t s_a[BLOCK / 64], s_b[BLOCK / 64], s_c[BLOCK / 64];
    __shared__ uint32_t s_base_a, s_base_b, s_tot_a, s_tot_b;
    const unsigned long long ma = __ballot(front), mb = __ballot(back), mc = __ballot(done);
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_a[wave] = (uint32_t)__popcll(ma);
        s_b[wave] = (uint32_t)__popcll(mb);
        s_c[wave] = (uint32_t)__popcll(mc);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ta = 0, tb = 0, tc = 0;
        for (uint32_t w = 0; w < BLOCK / 64; ++w) {
            ta += s_a[w];
            tb += s_b[w];
            tc += s_c[w];
        }
        s_base_a = ta ? atomicAdd(cntA, ta * mult) : 0u;
        s_base_b = tb ? atomicAdd(cntB, tb * mult) : 0u;
        if (tc) atomicAdd(cntC, tc);  // ray segments finished inside the producer: counted, never stored
        s_tot_a = ta;
        s_tot_b = tb;
    }
    __syncthreads();
    uint32_t slot = 0xFFFFFFFFu;
    if (front) {
        uint32_t j = s_base_a + (uint32_t)__popcll(ma & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < wave; ++w) j += s_a[w];
        slot = j;
        if (stride) *stride = s_tot_a;
        if (base) *base = s_base_a;
    } else if (back) {
        uint32_t j = s_base_b + (uint32_t)__popcll(mb & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < wave; ++w) j += s_b[w];
        slot = cap - 1u - j;
        if (stride) *stride = s_tot_b;
        if (base) *base = s_base_b;
    }
    return slot;
}

// The same reservation for K rays per thread with ONE atomic per side per block (k_raygen with jittered primary rays:
// a counter word executes ~87 returning atomics per microsecond whatever the block size, tools/atomic_rate.hip, so a
// kernel cannot retire more than 87 reserving blocks per microsecond; K rays per reservation divide that floor and the
// two barriers by K).  Within a wave ray k of every lane comes before ray k + 1.  Same contract as block_alloc2.
template <int BLOCK, int K>
PRT_DEV void block_alloc2k(const bool (&front)[K], const bool (&back)[K], uint32_t* cntA, uint32_t* cntB, uint32_t cap,
                           uint32_t (&slot)[K]) {
    __shared__ uint32_t s_a[BLOCK / 64], s_b[BLOCK / 64];
    __shared__ uint32_t s_base_a, s_base_b;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t ja[K], jb[K], na = 0, nb = 0;  // offsets of the lane's rays within the wave's reservation
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const unsigned long long ma = __ballot(front[k]), mb = __ballot(back[k]);
        ja[k] = na + (uint32_t)__popcll(ma & below);
        jb[k] = nb + (uint32_t)__popcll(mb & below);
        na += (uint32_t)__popcll(ma);
        nb += (uint32_t)__popcll(mb);
    }
    if (lane == 0) {
        s_a[wave] = na;
        s_b[wave] = nb;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ta = 0, tb = 0;
        for (uint32_t w = 0; w < BLOCK / 64; ++w) {
            ta += s_a[w];
            tb += s_b[w];
        }
        s_base_a = ta ? atomicAdd(cntA, ta) : 0u;
        s_base_b = tb ? atomicAdd(cntB, tb) : 0u;
    }
    __syncthreads();
    uint32_t wa = s_base_a, wb = s_base_b;
    for (uint32_t w = 0; w < wave; ++w) {
        wa += s_a[w];
        wb += s_b[w];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) slot[k] = front[k] ? wa + ja[k] : back[k] ? cap - 1u - (wb + jb[k]) : 0xFFFFFFFFu;
}

// ---------------------------------------------------------------------------------------------------------
// Tile map: 8x8-pixel tiles, tile t (row-major) belongs to rank t % world.
// ---------------------------------------------------------------------------------------------------------
PRT_DEV bool tile_pixel(const PrtTileMap& tm, uint32_t pl, uint32_t& x, uint32_t& y) {
    const uint32_t lt = pl >> 6, lane = pl & 63u;
    const uint32_t gt = lt * tm.world + tm.rank;
    const uint32_t tx = gt % tm.tiles_x, ty = gt / tm.tiles_x;
    x = tx * 8u + (lane & 7u);
    y = ty * 8u + (lane >> 3);
    return gt < tm.tiles_x * tm.tiles_y && x < tm.W && y < tm.H;
}

// The primary ray of path i (no jitter: the pixel centre, cpu/renderer.cpp:45) as k_raygen computed it for the path's
// pixel (PrtPrimary, prt_kernels.h): path id -> (sample, local pixel) -> the pixel's record.
PRT_DEV void primary_ray(const PrtPrimary& pr, uint32_t i, f3& o, f3& d, uint32_t& pixel, uint32_t& sample, uint32_t* local_pixel = nullptr) {
    uint32_t q = (uint32_t)((float)i * pr.inv_n);  // within one of i / n_pix_local
    uint32_t r = i - q * pr.n_pix_local;
    if ((int32_t)r < 0) {
        --q;
        r += pr.n_pix_local;
    }
    if (r >= pr.n_pix_local) {
        ++q;
        r -= pr.n_pix_local;
    }
    sample = q;
    if (local_pixel) *local_pixel = r;
    const float4 P = pr.pix[r];
    o = mk3(pr.origin[0], pr.origin[1], pr.origin[2]);
    d = mk3(P.x, P.y, P.z);
    pixel = __float_as_uint(P.w);
}

// ---------------------------------------------------------------------------------------------------------
// Ray generation (GenerateCameraRaysKernel, renderer.cu:186-204; pixel centres, cpu/renderer.cpp:45)
// ---------------------------------------------------------------------------------------------------------
template <bool ABVH, int BLOCK>
__device__ bool classify_ray(const DevScene& sc, f3 o, f3 d, uint32_t& id0, float& d2_0);
__device__ bool ends_here(const DevScene& sc, bool front, uint32_t id0, f3 thr, f3& L);

// Fused analytic segments.  A ray that cannot hit a triangle (it does not enter the BVH root box before its closest
// analytic hit) is fully determined by the producer's own scan, so the producer shades it in place and goes on with
// the scattered ray, FUSE segments per kernel call: such rays are never stored or re-read (C3: 62 % of all segments).
// FUSE is a compile-time 0 or 1 and the steps are straight-line code: as a run-time loop the same body needs 94 instead
// of 63 VGPRs (loop-invariant operand copies stay live across the back edge), which halves the occupancy of these
// memory-bound kernels and costs more than the fusion saves (measured); one fused segment already gives all of the
// gain (a second one changes nothing).  k_shade fuses (512-thread blocks: 78 VGPRs / 82 SGPRs = 6 waves per SIMD;
// shade 76 -> 57 ms per 1024 spp on C3); k_raygen does not (its 8-samples-per-thread loop around the shading body needs
// 119 VGPRs and ran 2.4x slower fused): a primary ray that cannot hit a triangle is stored on the back side and
// continued by the first k_shade.  Paths therefore advance by different numbers of segments per bounce iteration; every stored ray
// carries its own segment index (depth) in throughput.w, and every path records the index of its LAST segment in
// rad.w when it ends: k_accumulate derives the per-depth ray counts from those (segments at depth d = paths whose
// last segment index is >= d), so the producers count nothing.
//   returns 0: the path ended here (rad written)   1: (o, d) must be traversed (front)
//           2: budget used up, (o, d) is stored with its known analytic hit id0 (back)
// ONE loop serves both producers: on entry `id` is the final closest hit of segment `depth` = (o, d) over all
// primitives (from the traversal kernel in k_shade, from the primary ray's classification in k_raygen); `budget` is
// the number of segments this call may shade (k_shade: 1 + FUSE, k_raygen: FUSE).
// Radiance a path delivers, with the optional firefly clamp (PrtSampling.clamp; 0 = off).
PRT_DEV float4 path_result(f3 L, float clamp, uint32_t depth) {
    if (clamp > 0.0f) {
        L.x = L.x > clamp ? clamp : L.x;
        L.y = L.y > clamp ? clamp : L.y;
        L.z = L.z > clamp ? clamp : L.z;
    }
    return make_float4(L.x, L.y, L.z, __uint_as_float(depth));
}

// PRE: (pre_a, pre_b) = the hit record k_primary_hit computed for this path's pixel: {position, hit id}, {normal, material
// | front face << 31}.  It replaces world_hit_from_id for the first segment when it was computed for the same hit id (it
// always was: all samples of a pixel trace the same primary ray; the comparison keeps the kernel correct by itself).
template <int BUDGET, bool INST, bool ABVH, int BLOCK, bool PRE = false>
PRT_DEV int advance_path(const DevScene& sc, uint32_t id, f3& o, f3& d, f3& thr, uint32_t& rng, uint32_t& depth,
                         uint32_t max_depth, const PrtSampling& sp, float4* __restrict__ rad_slot, uint32_t& id0,
                         float& d2_0, float4 pre_a = float4{0.f, 0.f, 0.f, 0.f}, float4 pre_b = float4{0.f, 0.f, 0.f, 0.f}) {
#pragma unroll
    for (int it = 0; it <= BUDGET; ++it) {
        if (id == HIT_MISS) {  // the miss branch of IntersectClosestKernel, renderer.cu:263-271
            st_stream(rad_slot, path_result(thr * mk3(sc.sky[0], sc.sky[1], sc.sky[2]), sp.clamp, depth));
            return 0;
        }
        if (it == BUDGET) {  // only reached with an analytic id (the ray was classified "cannot hit a triangle")
            const uint32_t m = sc.prims[id].material;
            if (sc.mat_type[m] == 4u) {  // emissive: never scatters (material.h:119-122): the path stops here
                const float4 e = sc.mat_rgbs[m];
                st_stream(rad_slot, path_result(thr * mk3(e.x, e.y, e.z), sp.clamp, depth));
                return 0;
            }
            id0 = id;
            return 2;
        }
        WorldHit w;
        if (PRE && it == 0 && __float_as_uint(pre_a.w) == id) {
            w.pos = mk3(pre_a.x, pre_a.y, pre_a.z);
            w.normal = mk3(pre_b.x, pre_b.y, pre_b.z);
            w.material = __float_as_uint(pre_b.w) & 0x7FFFFFFFu;
            w.front = (__float_as_uint(pre_b.w) >> 31) != 0u;
        } else {
            world_hit_from_id<INST>(sc, id, o, d, w);
        }
        const uint32_t type = sc.mat_type[w.material];
        const float4 rgbs = sc.mat_rgbs[w.material];
        f3 atten, emitted, so, sd;
        bool scattered = false;
        if (depth + 1u >= max_depth) {
            emitted = (type == 4u) ? mk3(rgbs.x, rgbs.y, rgbs.z) : mk3(0.f, 0.f, 0.f);
        } else {
            scattered = material_scatter(type, rgbs, d, w.pos, w.normal, w.front, rng, atten, emitted, so, sd);
        }
        if (!scattered) {
            st_stream(rad_slot, path_result(thr * emitted, sp.clamp, depth));
            return 0;
        }
        thr = thr * atten;
        o = so;
        d = normalize3(sd);  // scatteredRay.Normalize(), cpu/renderer.cpp:84
        if (sp.rr_depth != 0u && depth + 1u >= sp.rr_depth) {  // Russian roulette (PrtSampling, include/prt.h)
            float p = thr.x > thr.y ? thr.x : thr.y;
            p = p > thr.z ? p : thr.z;
            p = p > 1.0f ? 1.0f : p;
            p = p < 0.05f ? 0.05f : p;
            if (!(rnd01(rng) < p)) {
                st_stream(rad_slot, path_result(mk3(0.f, 0.f, 0.f), 0.0f, depth));
                return 0;
            }
            thr = mk3(thr.x / p, thr.y / p, thr.z / p);
        }
        ++depth;
        if (classify_ray<ABVH, BLOCK>(sc, o, d, id0, d2_0)) return 1;
        id = id0;
    }
    return 2;  // not reached
}

// All S samples of a pixel start with the same pixel-centre ray (no jitter: cpu/renderer.cpp:45), so one thread
// computes the camera ray and its classification once and emits it for RAYGEN_GROUP samples, each with its own RNG
// seed and path id; every sample's primary ray is still traced on its own by the traversal kernel.
#define RAYGEN_GROUP 8
#ifndef RAYGEN_ALLOC
#define RAYGEN_ALLOC 4  // jittered rays per slot reservation (divides RAYGEN_GROUP)
#endif
#define RAYGEN_GROUP_NOJITTER 64u  // without jitter: one wave's worth of samples per pixel and block (pixel-major slots)
// SAMPLING = false compiles the Russian-roulette / clamp code out: with it in, k_shade needs 82 instead of 74 SGPRs,
// which costs a wave per SIMD, i.e. with 1024-thread blocks one of the two blocks per CU (measured: shade 50 % slower).
template <bool JITTER, bool SAMPLING, bool ABVH, bool COMPACT = false>
__global__ void __launch_bounds__(PRODUCER_BLOCK) k_raygen(DevScene sc, DevCamera cam, PrtTileMap tm, uint32_t S,
                                                            uint32_t first_sample, uint32_t seed,
                                                            float4* __restrict__ ro, float4* __restrict__ rd,
                                                            float4* __restrict__ rt, uint32_t* __restrict__ hit,
                                                            float* __restrict__ hd2, float4* __restrict__ rad,
                                                            uint32_t* __restrict__ counts, uint32_t* __restrict__ work,
                                                            uint32_t max_depth, PrtSampling sp_arg, float4* __restrict__ pix) {
    const PrtSampling sp = SAMPLING ? sp_arg : PrtSampling{0u, 0u, 0.0f};
    const uint32_t pl = blockIdx.x * (uint32_t)PRODUCER_BLOCK + threadIdx.x;
    if (blockIdx.y == 0 && pl < 8u) work[32u * pl] = 0u;  // chunk cursors of the traversal kernel that follows
    if (blockIdx.y == 0 && pl == 8u) work[512] = 0u;      // its overflow-list counter
    const uint32_t n_paths = S * tm.n_pix_local;
    const bool in_range = pl < tm.n_pix_local;
    bool valid = false, front0 = false;
    f3 o0 = mk3(0.f, 0.f, 0.f), d0 = mk3(0.f, 0.f, 1.f);
    uint32_t id00 = HIT_MISS, pixel = 0, px = 0, py = 0;
    float d2_00 = 3.402823466e+38f;
    if (in_range) {
        valid = tile_pixel(tm, pl, px, py);
        if (valid) {
            pixel = py * tm.W + px;
            if (!JITTER) {  // pixel centre, the same ray for every sample (cpu/renderer.cpp:45)
                camera_ray(cam, (float)px + 0.5f, (float)py + 0.5f, o0, d0);
                if (COMPACT && blockIdx.y == 0) pix[pl] = make_float4(d0.x, d0.y, d0.z, __uint_as_float(pixel));
                front0 = classify_ray<ABVH, PRODUCER_BLOCK>(sc, o0, d0, id00, d2_00);
            }
        }
    }
    const uint32_t group = JITTER ? RAYGEN_GROUP : RAYGEN_GROUP_NOJITTER;
    const uint32_t s0 = blockIdx.y * group;
    const uint32_t s1 = (s0 + group < S) ? s0 + group : S;
    if (!JITTER) {
        // Without jitter whether the pixel's primary ray is stored (front / back) or ends right here does not depend
        // on the sample: decide once, reserve the slots of all samples of this group with ONE atomic per side per
        // block, and only the RNG seed and the path id differ per copy.
        f3 o = o0, d = d0, thr = mk3(1.f, 1.f, 1.f);
        uint32_t rng = 0, depth = 0, id0 = id00;
        float d2_0 = d2_00;
        float4 L0 = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));  // outside the image: no path
        bool front = false, back = false;
        if (valid) {
            front = front0;
            if (!front) {
                const int r = advance_path<0, false, ABVH, PRODUCER_BLOCK>(sc, id00, o, d, thr, rng, depth, max_depth, sp, &L0, id0, d2_0);
                back = r == 2;
            }
        }
        uint32_t stride = 0, base = 0;
        const uint32_t slot0 = block_alloc2<PRODUCER_BLOCK>(front, back, false, &CNT_A(counts, 0), &CNT_B(counts, 0),
                                                            &CNT_C(counts, 0), n_paths, s1 - s0, &stride, &base);
        uint32_t first_slot = slot0;  // slot of this pixel's sample s0
        if (s1 - s0 < 8u) {
            // few samples per pixel in this batch (interactive use: ProgressiveRender adds ONE sample per call): SAMPLE-major
            // slots, every thread stores its own pixel's copies, coalesced across the pixels of a wave
            if (in_range && slot0 != 0xFFFFFFFFu) {
                for (uint32_t sl = s0; sl < s1; ++sl) {
                    const uint32_t i = sl * tm.n_pix_local + pl;  // path id
                    const uint32_t slot = front ? slot0 + (sl - s0) * stride : slot0 - (sl - s0) * stride;
                    if (COMPACT) {
                        ((uint32_t*)rt)[slot] = i;
                    } else {
                        ro[slot] = make_float4(o.x, o.y, o.z, __uint_as_float(i));
                        rd[slot] = make_float4(d.x, d.y, d.z, __uint_as_float(path_seed(pixel, first_sample + sl, seed)));
                        rt[slot] = make_float4(1.f, 1.f, 1.f, __uint_as_float(0u));
                    }
                    if (!COMPACT) {  // (compact: the analytic scan's result is the pixel's, in its end record)
                        hit[slot] = id0;
                        hd2[slot] = d2_0;
                    }
                }
            }
        } else {
            // PIXEL-major slots: the (up to 64) samples of a pixel that this block handles sit next to each other, so a
            // wave of the first bounce's traversal / k_shade works on IDENTICAL rays: no divergence in the node loop (and,
            // with placed copies, level switches in lockstep), one cache line per node for the whole wave, one triangle /
            // one material per wave in k_shade.  Each of them is still traced on its own.  Written wave-transposed (pixel
            // by pixel, one sample per lane): 256-B stores of path ids (COMPACT) or 1-KB stores of full records.
            const uint32_t mult = s1 - s0;
            const bool stored = slot0 != 0xFFFFFFFFu;
            // slot of the pixel's first sample: rank among the block's stored pixels x mult, from the block's base
            uint32_t first = 0u;
            if (front) first = base + (slot0 - base) * mult;
            if (back) first = n_paths - 1u - (base + ((n_paths - 1u - slot0) - base) * mult);
            first_slot = first;
            const uint32_t lane = lane_id();
            for (unsigned long long m = __ballot(stored); m; m &= m - 1ull) {
                const int p = __builtin_ctzll(m);
                const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)first, p);
                const bool fr = __builtin_amdgcn_readlane((int)front, p) != 0;
                const uint32_t plp = (uint32_t)__builtin_amdgcn_readlane((int)pl, p);
                const uint32_t idp = (uint32_t)__builtin_amdgcn_readlane((int)id0, p);
                const float d2p = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(d2_0), p));
                float4 Op = make_float4(0.f, 0.f, 0.f, 0.f), Dp = Op;
                uint32_t pixp = 0u;
                if (!COMPACT) {
#define PRT_RL(X) __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(X), p))
                    Op = make_float4(PRT_RL(o.x), PRT_RL(o.y), PRT_RL(o.z), 0.f);
                    Dp = make_float4(PRT_RL(d.x), PRT_RL(d.y), PRT_RL(d.z), 0.f);
#undef PRT_RL
                    pixp = (uint32_t)__builtin_amdgcn_readlane((int)pixel, p);
                }
                if (lane < mult) {
                    const uint32_t slot = fr ? f + lane : f - lane;
                    const uint32_t i = (s0 + lane) * tm.n_pix_local + plp;  // path id
                    if (COMPACT) {  // 12 B per path: bounce 0 rebuilds the rest from the path id (PrtPrimary)
                        ((uint32_t*)rt)[slot] = i;
                    } else {
                        ro[slot] = make_float4(Op.x, Op.y, Op.z, __uint_as_float(i));
                        rd[slot] = make_float4(Dp.x, Dp.y, Dp.z, __uint_as_float(path_seed(pixp, first_sample + s0 + lane, seed)));
                        rt[slot] = make_float4(1.f, 1.f, 1.f, __uint_as_float(0u));
                    }
                    if (!COMPACT) {
                        hit[slot] = idp;
                        hd2[slot] = d2p;
                    }
                }
            }
        }
        {
            const bool stored = slot0 != 0xFFFFFFFFu;
            if (COMPACT) {
                // whether the pixel's paths end with their primary ray, and with what, is the same for all its samples: one
                // record per pixel (w = 0xFFFFFFFE: they go on, look in rad[]) instead of S copies in rad[]
                // (a "they go on" record: x = the ray slot of the pixel's first stored sample, for k_primary_hit; y, z = the
                // analytic scan's closest hit and its distance^2, the same for every sample of the pixel: the paths' slots
                // carry the path id only, 4 B instead of 12)
                if (in_range && blockIdx.y == 0)
                    pix[tm.n_pix_local + pl] = stored ? make_float4(__uint_as_float(first_slot), __uint_as_float(id0), d2_0, __uint_as_float(0xFFFFFFFEu)) : L0;
            } else if (in_range && !stored) {
                // the path ended with its primary ray (sky / light seen directly), or there is none
                for (uint32_t sl = s0; sl < s1; ++sl) rad[sl * tm.n_pix_local + pl] = L0;
            }
        }
        return;
    }
    // Jittered primary rays, RAYGEN_ALLOC samples of the pixel per slot reservation.  What is stored per ray is its
    // direction, RNG state and the analytic scan's result: with no shading budget advance_path never scatters, so the origin
    // stays the camera position, the throughput 1 and the segment index 0 (paths that end right here write rad[] inside).
    constexpr int KA = ABVH ? 2 : RAYGEN_ALLOC;  // (the primitive-BVH walk needs the registers: 4 rays would spill)
    for (uint32_t sl = s0; sl < s1; sl += KA) {  // block-uniform trip count
        f3 dk[KA];
        uint32_t rngk[KA], idk[KA], slotk[KA];
        float d2k[KA];
        bool frontk[KA], backk[KA];
#pragma unroll
        for (int kk = 0; kk < KA; ++kk) {
            const uint32_t sk = sl + (uint32_t)kk;
            const uint32_t i = sk * tm.n_pix_local + pl;  // path id
            f3 o = o0, d = d0, thr = mk3(1.f, 1.f, 1.f);
            uint32_t rng = 0, depth = 0, id0 = id00;
            float d2_0 = d2_00;
            bool front = false, back = false;
            if (sk < s1 && valid) {
                rng = path_seed(pixel, first_sample + sk, seed);
                {  // (x + u1, y + u2): the path's first two draws (optix/device_programs.cu:172-173)
                    const float u1 = rnd01(rng);
                    const float u2 = rnd01(rng);
                    camera_ray(cam, (float)px + u1, (float)py + u2, o, d);
                    front = classify_ray<ABVH, PRODUCER_BLOCK>(sc, o, d, id0, d2_0);
                }
                if (!front) {
                    const int r = advance_path<0, false, ABVH, PRODUCER_BLOCK>(sc, id0, o, d, thr, rng, depth, max_depth, sp, &rad[i], id0, d2_0);
                    front = r == 1;
                    back = r == 2;
                }
            } else if (sk < s1 && in_range) {
                rad[i] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));  // partial tiles outside the image: no path
            }
            dk[kk] = d;
            rngk[kk] = rng;
            idk[kk] = id0;
            d2k[kk] = d2_0;
            frontk[kk] = front;
            backk[kk] = back;
        }
        block_alloc2k<PRODUCER_BLOCK, KA>(frontk, backk, &CNT_A(counts, 0), &CNT_B(counts, 0), n_paths, slotk);
#pragma unroll
        for (int kk = 0; kk < KA; ++kk) {
            const uint32_t slot = slotk[kk];
            if (slot != 0xFFFFFFFFu) {
                const uint32_t i = (sl + (uint32_t)kk) * tm.n_pix_local + pl;
                ro[slot] = make_float4(cam.pos.x, cam.pos.y, cam.pos.z, __uint_as_float(i));
                rd[slot] = make_float4(dk[kk].x, dk[kk].y, dk[kk].z, __uint_as_float(rngk[kk]));
                rt[slot] = make_float4(1.f, 1.f, 1.f, __uint_as_float(0u));
                hit[slot] = idk[kk];
                hd2[slot] = d2k[kk];
            }
        }
        __syncthreads();  // block_alloc2k's LDS counts are reused by the next trip (see block_alloc2's contract)
    }
}

// ---------------------------------------------------------------------------------------------------------
// Closest hit (IntersectClosestKernel, renderer.cu:206-272; semantics of PrimitiveList::Intersect,
// src/core/primitive.cpp:21-59): linear scan over the analytic primitives + BVH2 traversal over all
// mesh triangles.  The result equals the reference's linear scan over every primitive: the winner is the
// smallest world distance^2, ties to the lowest primitive index, independent of visiting order.
// ---------------------------------------------------------------------------------------------------------
struct Closest {
    float d2;
    uint32_t id;    // hit id (analytic index, or n_prims + leaf slot)
    uint32_t prim;  // global primitive index (tie-break key)
};

PRT_DEV float limit_from_d2(float d2, float pad) {
    // Upper bound on the local ray parameter of anything that could still win (d2' <= d2).
    return (d2 < 3.0e38f) ? __builtin_sqrtf(d2) * 1.0000153f + 4.0f * pad : 3.4e38f;
}

#define ABVH_STACK 20  // entries of the primitive walk's stack (4-wide tree: at most 3 net pushes per level)
// Closest hit over the analytic primitives.  ABVH = false: the reference's linear scan (primitive.cpp:26-49).
// ABVH = true: a per-thread walk of a 4-wide BVH over the primitives' WORLD boxes (valid because every primitive of
// the scene has a rotation + uniform scale + translation transform, checked on the host: only then is the reference's
// local ray, primitive.cpp:29-30, the geometric ray and the hit lies inside the shape's world box).  Each primitive
// the walk reaches is tested with the reference's own arithmetic and candidates compete on (d2, primitive index), so
// the result equals the linear scan's bit for bit whatever the visiting order.
template <bool ABVH, int BLOCK = 1>
PRT_DEV void scan_analytic(const DevScene& sc, f3 o, f3 d, Closest& best, uint32_t& n_tests) {
    if (!ABVH) {
        for (uint32_t i = 0; i < sc.n_prims; ++i) {
            WorldHit w;
            analytic_hit(sc.prims[i], o, d, w);
            ++n_tests;
            if (w.has && w.d2 < best.d2) {  // strict <, first wins (primitive.cpp:44)
                best.d2 = w.d2;
                best.id = i;
                best.prim = i;
            }
        }
        return;
    }
    const f3 ld = normalize3(d);
    // The per-ray pad has a term that grows with the SQUARE of the distance: Circle::Intersect's discriminant
    // b*b - 4*a*c (shape.h:160-163) cancels catastrophically for a far origin, so a ray that passes up to
    // ~2e-7 * dist^2 / R outside a sphere can still be a hit of the reference's arithmetic (measured; 4.8e-7 is the
    // worst-case bound), which a world box padded in proportion to the distance alone would cull.
    const float A1 = __builtin_fabsf(o.x) + __builtin_fabsf(o.y) + __builtin_fabsf(o.z);
    const float pad = sc.pad * (A1 + sc.extent) + ((sc.abvh_q[0] * A1 + sc.abvh_q[1]) * A1 + sc.abvh_q[2]);
    const float ix = 1.0f / (__builtin_fabsf(ld.x) < 1e-30f ? __builtin_copysignf(1e-30f, ld.x) : ld.x);
    const float iy = 1.0f / (__builtin_fabsf(ld.y) < 1e-30f ? __builtin_copysignf(1e-30f, ld.y) : ld.y);
    const float iz = 1.0f / (__builtin_fabsf(ld.z) < 1e-30f ? __builtin_copysignf(1e-30f, ld.z) : ld.z);
    const float ax = (o.x + pad) * ix, ay = (o.y + pad) * iy, az = (o.z + pad) * iz;
    const float bx = (o.x - pad) * ix, by = (o.y - pad) * iy, bz = (o.z - pad) * iz;
    float tlimit = limit_from_d2(best.d2, pad);
    // per-thread stack in LDS, [entry][thread] (bank = lane: conflict-free): as a register array it costs 60 VGPRs
    __shared__ uint32_t s_astk[ABVH ? ABVH_STACK * BLOCK : 1];
    uint32_t* const stk = &s_astk[threadIdx.x];
    const uint32_t sstride = (uint32_t)BLOCK;
    int sp = 0;
    uint32_t node = 0;
    for (;;) {
        const float4* nb = sc.abvh_nodes + 8 * (size_t)node;
        const float4 mnx = nb[0], mxx = nb[1], mny = nb[2], mxy = nb[3], mnz = nb[4], mxz = nb[5], rf = nb[6];
#define ABVH_CHILD(C)                                                                                              \
    {                                                                                                              \
        const float x0 = __builtin_fmaf(mnx.C, ix, -ax), x1 = __builtin_fmaf(mxx.C, ix, -bx);                      \
        const float y0 = __builtin_fmaf(mny.C, iy, -ay), y1 = __builtin_fmaf(mxy.C, iy, -by);                      \
        const float z0 = __builtin_fmaf(mnz.C, iz, -az), z1 = __builtin_fmaf(mxz.C, iz, -bz);                      \
        const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)),        \
                                         __builtin_fmaxf(__builtin_fminf(z0, z1), 0.0f));                          \
        const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1)),        \
                                         __builtin_fminf(__builtin_fmaxf(z0, z1), tlimit));                        \
        if (tn <= tf * 1.0000005f) {                                                                               \
            const int ref = __float_as_int(rf.C);                                                                  \
            if (ref >= 0) {                                                                                        \
                stk[(sp++) * sstride] = (uint32_t)ref;                                                             \
            } else {                                                                                               \
                const uint32_t lr = ~(uint32_t)ref, first = lr >> 4, cnt = lr & 15u;                               \
                for (uint32_t k = 0; k < cnt; ++k) {                                                               \
                    const uint32_t i = sc.abvh_order[first + k];                                                   \
                    WorldHit w;                                                                                    \
                    analytic_hit(sc.prims[i], o, d, w);                                                            \
                    ++n_tests;                                                                                     \
                    if (w.has && (w.d2 < best.d2 || (w.d2 == best.d2 && best.id != HIT_MISS && i < best.prim))) {  \
                        best.d2 = w.d2;                                                                            \
                        best.id = i;                                                                               \
                        best.prim = i;                                                                             \
                        tlimit = limit_from_d2(best.d2, pad);                                                      \
                    }                                                                                              \
                }                                                                                                  \
            }                                                                                                      \
        }                                                                                                          \
    }
        if (sp + 4 > ABVH_STACK) {  // cannot happen for the trees the host builds; stay correct anyway
            best.d2 = 3.402823466e+38f;
            best.id = HIT_MISS;
            best.prim = 0xFFFFFFFFu;
            scan_analytic<false, 1>(sc, o, d, best, n_tests);
            return;
        }
        ABVH_CHILD(x)
        ABVH_CHILD(y)
        ABVH_CHILD(z)
        ABVH_CHILD(w)
#undef ABVH_CHILD
        if (sp == 0) break;
        node = stk[(--sp) * sstride];
    }
}


// What every producer does for the ray it emits: the linear scan over the analytic primitives (its result is the
// initial "best" of the traversal) and the decision whether the BVH has to be walked at all: only if the ray enters
// the (per-ray padded) root box before the analytic hit, by the same conservative test the traversal applies.
template <bool ABVH, int BLOCK>
__device__ __forceinline__ bool classify_ray(const DevScene& sc, f3 o, f3 d, uint32_t& id0, float& d2_0) {
    Closest best;
    best.d2 = 3.402823466e+38f;  // FLT_MAX (primitive.cpp:23)
    best.id = HIT_MISS;
    best.prim = 0xFFFFFFFFu;
    uint32_t n_tests = 0;
    scan_analytic<ABVH, BLOCK>(sc, o, d, best, n_tests);
    id0 = best.id;
    d2_0 = best.d2;
    if (sc.n_nodes == 0u) return false;
    // This box test only decides whether the traversal kernel has to look at the ray at all; it is conservative by the
    // per-ray pad (2^-18 of the coordinates' magnitude) and by the 1.0000153 factor in limit_from_d2.  The producers'
    // directions are unit vectors up to rounding (1e-7), and v_rcp_f32 is within 1 ulp (2^-23): both far inside those
    // margins, so neither the reference's re-normalisation of the direction nor IEEE divisions are spent here (the
    // producers are VALU-bound: SQ_ACTIVE_INST_VALU x waves per SIMD > 1 for k_shade).
    const f3 ld = d;
    const float pad = sc.pad * (__builtin_fabsf(o.x) + __builtin_fabsf(o.y) + __builtin_fabsf(o.z) + sc.extent);
    const float ix = __builtin_amdgcn_rcpf(__builtin_fabsf(ld.x) < 1e-30f ? __builtin_copysignf(1e-30f, ld.x) : ld.x);
    const float iy = __builtin_amdgcn_rcpf(__builtin_fabsf(ld.y) < 1e-30f ? __builtin_copysignf(1e-30f, ld.y) : ld.y);
    const float iz = __builtin_amdgcn_rcpf(__builtin_fabsf(ld.z) < 1e-30f ? __builtin_copysignf(1e-30f, ld.z) : ld.z);
    const float x0 = __builtin_fmaf(sc.root_min[0], ix, -(o.x + pad) * ix), x1 = __builtin_fmaf(sc.root_max[0], ix, -(o.x - pad) * ix);
    const float y0 = __builtin_fmaf(sc.root_min[1], iy, -(o.y + pad) * iy), y1 = __builtin_fmaf(sc.root_max[1], iy, -(o.y - pad) * iy);
    const float z0 = __builtin_fmaf(sc.root_min[2], iz, -(o.z + pad) * iz), z1 = __builtin_fmaf(sc.root_max[2], iz, -(o.z - pad) * iz);
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)),
                                     __builtin_fmaxf(__builtin_fminf(z0, z1), 0.0f));
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1)),
                                     __builtin_fminf(__builtin_fmaxf(z0, z1), limit_from_d2(d2_0, pad)));
    return tn <= tf * 1.0000005f;
}

// A classified ray that cannot hit a triangle AND whose analytic result ends the path (it misses everything, or it
// hits an emissive primitive, which never scatters: material.h:119-122) needs no slot in the next buffer: the
// value the next k_shade would write for it is already known.  Returns true and that radiance in that case.
__device__ __forceinline__ bool ends_here(const DevScene& sc, bool front, uint32_t id0, f3 thr, f3& L) {
    if (front) return false;
    if (id0 == HIT_MISS) {
        L = thr * mk3(sc.sky[0], sc.sky[1], sc.sky[2]);  // the miss branch of k_shade
        return true;
    }
    const uint32_t m = sc.prims[id0].material;
    if (sc.mat_type[m] == 4u) {
        const float4 e = sc.mat_rgbs[m];
        L = thr * mk3(e.x, e.y, e.z);  // throughput * emitted, then the path stops
        return true;
    }
    return false;
}

// Variant 1 (kept for A/B runs): one loop, each iteration is either a node step or a leaf, per lane.
template <int STACK, bool STATS>
PRT_DEV void traverse_ifif(const DevScene& sc, f3 o, f3 d, Closest& best, uint32_t* stk, uint32_t& n_nodes,
                           uint32_t& n_tris) {
    // Triangles carry the identity Transform: local origin = o, local direction = normalize(d)
    // (TransformNormal, primitive.cpp:30).
    const f3 ld = normalize3(d);
    // Culling only (never changes a result): per-ray conservative padding of every box by `pad`.
    const float pad = sc.pad * (__builtin_fabsf(o.x) + __builtin_fabsf(o.y) + __builtin_fabsf(o.z) + sc.extent);
    const float ix = 1.0f / (__builtin_fabsf(ld.x) < 1e-30f ? __builtin_copysignf(1e-30f, ld.x) : ld.x);
    const float iy = 1.0f / (__builtin_fabsf(ld.y) < 1e-30f ? __builtin_copysignf(1e-30f, ld.y) : ld.y);
    const float iz = 1.0f / (__builtin_fabsf(ld.z) < 1e-30f ? __builtin_copysignf(1e-30f, ld.z) : ld.z);
    // plane "min" is moved by -pad, plane "max" by +pad:  t_min = min*inv - (o+pad)*inv, t_max = max*inv - (o-pad)*inv
    const float ax = (o.x + pad) * ix, ay = (o.y + pad) * iy, az = (o.z + pad) * iz;
    const float bx = (o.x - pad) * ix, by = (o.y - pad) * iy, bz = (o.z - pad) * iz;
    float tlimit = limit_from_d2(best.d2, pad);
    int sp = 0;
    int node = 0;
    while (true) {
        if (node >= 0) {
            const float4 q0 = sc.nodes[4 * (size_t)node + 0];
            const float4 q1 = sc.nodes[4 * (size_t)node + 1];
            const float4 q2 = sc.nodes[4 * (size_t)node + 2];
            const float4 q3 = sc.nodes[4 * (size_t)node + 3];
            if (STATS) ++n_nodes;
            // left: min (q0.x q0.y q0.z) max (q0.w q1.x q1.y); right: min (q1.z q1.w q2.x) max (q2.y q2.z q2.w)
            const float l0x = __builtin_fmaf(q0.x, ix, -ax), l1x = __builtin_fmaf(q0.w, ix, -bx);
            const float l0y = __builtin_fmaf(q0.y, iy, -ay), l1y = __builtin_fmaf(q1.x, iy, -by);
            const float l0z = __builtin_fmaf(q0.z, iz, -az), l1z = __builtin_fmaf(q1.y, iz, -bz);
            const float r0x = __builtin_fmaf(q1.z, ix, -ax), r1x = __builtin_fmaf(q2.y, ix, -bx);
            const float r0y = __builtin_fmaf(q1.w, iy, -ay), r1y = __builtin_fmaf(q2.z, iy, -by);
            const float r0z = __builtin_fmaf(q2.x, iz, -az), r1z = __builtin_fmaf(q2.w, iz, -bz);
            const float tnL = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(l0x, l1x), __builtin_fminf(l0y, l1y)),
                                              __builtin_fmaxf(__builtin_fminf(l0z, l1z), 0.0f));
            const float tfL = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(l0x, l1x), __builtin_fmaxf(l0y, l1y)),
                                              __builtin_fminf(__builtin_fmaxf(l0z, l1z), tlimit));
            const float tnR = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(r0x, r1x), __builtin_fminf(r0y, r1y)),
                                              __builtin_fmaxf(__builtin_fminf(r0z, r1z), 0.0f));
            const float tfR = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(r0x, r1x), __builtin_fmaxf(r0y, r1y)),
                                              __builtin_fminf(__builtin_fmaxf(r0z, r1z), tlimit));
            const bool hL = tnL <= tfL * 1.0000005f;
            const bool hR = tnR <= tfR * 1.0000005f;
            const int left = __float_as_int(q3.x), right = __float_as_int(q3.y);
            if (hL && hR) {
                const bool lfirst = tnL <= tnR;
                const int nearc = lfirst ? left : right;
                const int farc = lfirst ? right : left;
                if (sp < STACK) stk[sp * 256] = (uint32_t)farc;
                ++sp;  // host guarantees STACK >= tree depth; the guard only prevents LDS corruption
                node = nearc;
                continue;
            }
            if (hL) {
                node = left;
                continue;
            }
            if (hR) {
                node = right;
                continue;
            }
        } else {
            const uint32_t ref = ~(uint32_t)node;
            const uint32_t first = ref >> 4, cnt = ref & 15u;
            for (uint32_t t = 0; t < cnt; ++t) {
                const uint32_t slot = first + t;
                const float4 a = sc.tris[3 * (size_t)slot + 0];
                const float4 b = sc.tris[3 * (size_t)slot + 1];
                const float4 c = sc.tris[3 * (size_t)slot + 2];
                if (STATS) ++n_tris;
                f3 pos;
                float b1, b2;
                if (triangle_hit_pos(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), o, ld, pos, b1, b2)) {
                    const float d2 = dist2(o, pos);
                    const uint32_t prim = __float_as_uint(a.w);
                    if (d2 < best.d2 || (d2 == best.d2 && best.id != HIT_MISS && prim < best.prim)) {
                        best.d2 = d2;
                        best.id = sc.n_prims + slot;
                        best.prim = prim;
                        tlimit = limit_from_d2(d2, pad);
                    }
                }
            }
        }
        if (sp == 0) break;
        --sp;
        node = (sp < STACK) ? (int)stk[sp * 256] : -1;  // -1 == ~0 == empty leaf
    }
}

// Variant 0 (default): speculative while-while traversal (Aila & Laine) on wave64.  Each lane walks internal
// nodes until it holds a leaf; the first leaf found is POSTPONED and the lane keeps walking, so the wave
// leaves the node loop only when no lane is still searching (one ballot per step), and then all lanes test
// their leaves together.  Node steps and triangle tests are therefore executed with far fewer idle lanes
// than the one-loop form, where the two code paths alternate within a wave.
typedef float v2f __attribute__((ext_vector_type(2)));
#define NODE_DONE 0x7FFFFFFF
#define LEAF_NONE 0x7FFFFFFE
#define NEED_POP 0x7FFFFFFD
#define PRT_OVF_CAP (1u << 20)  // entries of the stack-overflow list (work[513..])
template <int STACK, bool STATS>
PRT_DEV void traverse_ww(const DevScene& sc, f3 o, f3 d, Closest& best, uint32_t* stk, uint32_t& n_nodes,
                         uint32_t& n_tris) {
    const f3 ld = normalize3(d);  // TransformNormal(identity, d), primitive.cpp:30
    const float pad = sc.pad * (__builtin_fabsf(o.x) + __builtin_fabsf(o.y) + __builtin_fabsf(o.z) + sc.extent);
    const float ix = 1.0f / (__builtin_fabsf(ld.x) < 1e-30f ? __builtin_copysignf(1e-30f, ld.x) : ld.x);
    const float iy = 1.0f / (__builtin_fabsf(ld.y) < 1e-30f ? __builtin_copysignf(1e-30f, ld.y) : ld.y);
    const float iz = 1.0f / (__builtin_fabsf(ld.z) < 1e-30f ? __builtin_copysignf(1e-30f, ld.z) : ld.z);
    const float ax = (o.x + pad) * ix, ay = (o.y + pad) * iy, az = (o.z + pad) * iz;
    const float bx = (o.x - pad) * ix, by = (o.y - pad) * iy, bz = (o.z - pad) * iz;
    float tlimit = limit_from_d2(best.d2, pad);
    int sp = 0;
    int node = 0;
    int leaf = LEAF_NONE;
    while (node != NODE_DONE) {
        // ---- phase 1: internal nodes ----
        while ((unsigned)node < 0x40000000u) {
            const float4 q0 = sc.nodes[4 * (size_t)node + 0];
            const float4 q1 = sc.nodes[4 * (size_t)node + 1];
            const float4 q2 = sc.nodes[4 * (size_t)node + 2];
            const float4 q3 = sc.nodes[4 * (size_t)node + 3];
            if (STATS) ++n_nodes;
            const float l0x = __builtin_fmaf(q0.x, ix, -ax), l1x = __builtin_fmaf(q0.w, ix, -bx);
            const float l0y = __builtin_fmaf(q0.y, iy, -ay), l1y = __builtin_fmaf(q1.x, iy, -by);
            const float l0z = __builtin_fmaf(q0.z, iz, -az), l1z = __builtin_fmaf(q1.y, iz, -bz);
            const float r0x = __builtin_fmaf(q1.z, ix, -ax), r1x = __builtin_fmaf(q2.y, ix, -bx);
            const float r0y = __builtin_fmaf(q1.w, iy, -ay), r1y = __builtin_fmaf(q2.z, iy, -by);
            const float r0z = __builtin_fmaf(q2.x, iz, -az), r1z = __builtin_fmaf(q2.w, iz, -bz);
            const float tnL = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(l0x, l1x), __builtin_fminf(l0y, l1y)),
                                              __builtin_fmaxf(__builtin_fminf(l0z, l1z), 0.0f));
            const float tfL = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(l0x, l1x), __builtin_fmaxf(l0y, l1y)),
                                              __builtin_fminf(__builtin_fmaxf(l0z, l1z), tlimit));
            const float tnR = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(r0x, r1x), __builtin_fminf(r0y, r1y)),
                                              __builtin_fmaxf(__builtin_fminf(r0z, r1z), 0.0f));
            const float tfR = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(r0x, r1x), __builtin_fmaxf(r0y, r1y)),
                                              __builtin_fminf(__builtin_fmaxf(r0z, r1z), tlimit));
            const bool hL = tnL <= tfL * 1.0000005f;
            const bool hR = tnR <= tfR * 1.0000005f;
            const int left = __float_as_int(q3.x), right = __float_as_int(q3.y);
            if (hL && hR) {
                const bool lfirst = tnL <= tnR;
                node = lfirst ? left : right;
                const int farc = lfirst ? right : left;
                if (sp < STACK) stk[sp * 256] = (uint32_t)farc;  // host guarantees STACK >= tree depth
                ++sp;
            } else if (hL) {
                node = left;
            } else if (hR) {
                node = right;
            } else if (sp > 0) {
                --sp;
                node = (sp < STACK) ? (int)stk[sp * 256] : -1;
            } else {
                node = NODE_DONE;
            }
            if (node < 0 && leaf == LEAF_NONE) {  // first leaf: postpone it, keep walking
                leaf = node;
                if (sp > 0) {
                    --sp;
                    node = (sp < STACK) ? (int)stk[sp * 256] : -1;
                } else {
                    node = NODE_DONE;
                }
            }
            if (__ballot(leaf == LEAF_NONE && node != NODE_DONE) == 0ull) break;  // nobody is still searching
        }
        // ---- phase 2: leaves ----
        while (leaf != LEAF_NONE) {
            const uint32_t ref = ~(uint32_t)leaf;
            const uint32_t first = ref >> 4, cnt = ref & 15u;
            for (uint32_t t = 0; t < cnt; ++t) {
                const uint32_t slot = first + t;
                const float4 a = sc.tris[3 * (size_t)slot + 0];
                const float4 b = sc.tris[3 * (size_t)slot + 1];
                const float4 c = sc.tris[3 * (size_t)slot + 2];
                if (STATS) ++n_tris;
                f3 pos;
                float b1, b2;
                if (triangle_hit_pos(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), o, ld, pos, b1, b2)) {
                    const float d2 = dist2(o, pos);
                    const uint32_t prim = __float_as_uint(a.w);
                    if (d2 < best.d2 || (d2 == best.d2 && best.id != HIT_MISS && prim < best.prim)) {
                        best.d2 = d2;
                        best.id = sc.n_prims + slot;
                        best.prim = prim;
                        tlimit = limit_from_d2(d2, pad);
                    }
                }
            }
            leaf = LEAF_NONE;
            if (node < 0) {  // the walk stopped on a second leaf: take it now
                leaf = node;
                if (sp > 0) {
                    --sp;
                    node = (sp < STACK) ? (int)stk[sp * 256] : -1;
                } else {
                    node = NODE_DONE;
                }
            }
        }
    }
}

template <int STACK, bool STATS, int VARIANT>
PRT_DEV void traverse(const DevScene& sc, f3 o, f3 d, Closest& best, uint32_t* stk, uint32_t& n_nodes,
                      uint32_t& n_tris) {
    if (VARIANT == 1)
        traverse_ifif<STACK, STATS>(sc, o, d, best, stk, n_nodes, n_tris);
    else
        traverse_ww<STACK, STATS>(sc, o, d, best, stk, n_nodes, n_tris);
}

template <int STACK, bool STATS, int VARIANT>
__global__ void __launch_bounds__(256) k_intersect(DevScene sc, const float4* __restrict__ ro,
                                                   const float4* __restrict__ rd, uint32_t* __restrict__ hit,
                                                   const uint32_t* __restrict__ count_ptr,
                                                   unsigned long long* __restrict__ stats) {
    __shared__ uint32_t s_stack[STACK * 256];
    const uint32_t count = *count_ptr;
    const uint32_t nb_live = (count + 255u) >> 8;
    if (blockIdx.x >= nb_live) return;
    const uint32_t k = xcd_remap(blockIdx.x, nb_live) * 256u + threadIdx.x;
    uint32_t n_nodes = 0, n_tris = 0, n_ptests = 0;
    if (k < count) {
        const float4 O = ro[k];
        const float4 D = rd[k];
        if (D.x == 0.0f && D.y == 0.0f && D.z == 0.0f) {
            hit[k] = HIT_DEAD;
        } else {
            const f3 o = mk3(O.x, O.y, O.z), d = mk3(D.x, D.y, D.z);
            Closest best;
            best.d2 = 3.402823466e+38f;  // FLT_MAX (primitive.cpp:23)
            best.id = HIT_MISS;
            best.prim = 0xFFFFFFFFu;
            scan_analytic<false>(sc, o, d, best, n_ptests);
            if (sc.n_nodes) traverse<STACK, STATS, VARIANT>(sc, o, d, best, &s_stack[threadIdx.x], n_nodes, n_tris);
            hit[k] = best.id;
        }
    }
    if (STATS) {
        atomicAdd(&stats[0], (unsigned long long)n_nodes);
        atomicAdd(&stats[1], (unsigned long long)n_tris);
        atomicAdd(&stats[2], (unsigned long long)n_ptests);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Default closest-hit pipeline (variant 0): k_scan_prims (coherent linear scan over the analytic primitives,
// one thread per ray) followed by k_traverse_persistent over the triangles' BVH.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_scan_prims(DevScene sc, const float4* __restrict__ ro,
                                                    const float4* __restrict__ rd, uint32_t* __restrict__ hit,
                                                    float* __restrict__ hd2, const uint32_t* __restrict__ count_ptr,
                                                    uint32_t* __restrict__ work,
                                                    unsigned long long* __restrict__ stats) {
    const uint32_t count = *count_ptr;
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k < 8u) work[32u * k] = 0u;  // chunk cursors of the traversal kernel that follows on the stream (work[256]: watchdog flag)
    if (k == 8u) work[512] = 0u;     // its overflow-list counter
    if (k >= count) return;
    const float4 O = ro[k];
    const float4 D = rd[k];
    Closest best;
    best.d2 = 3.402823466e+38f;  // FLT_MAX (primitive.cpp:23)
    best.id = HIT_MISS;
    best.prim = 0xFFFFFFFFu;
    uint32_t n_ptests = 0;
    if (D.x == 0.0f && D.y == 0.0f && D.z == 0.0f) {
        best.id = HIT_DEAD;
    } else {
        if (sc.abvh_nodes)
            scan_analytic<true, 256>(sc, mk3(O.x, O.y, O.z), mk3(D.x, D.y, D.z), best, n_ptests);
        else
            scan_analytic<false>(sc, mk3(O.x, O.y, O.z), mk3(D.x, D.y, D.z), best, n_ptests);
    }
    hit[k] = best.id;
    hd2[k] = best.d2;
    if (stats) atomicAdd(&stats[2], (unsigned long long)n_ptests);
}

// Persistent-wavefront traversal with ray replacement.  A fixed grid of waves stays resident; every wave
// grabs chunks of the bounce's ray buffer with one global atomic per `chunk` rays (a single counter word only
// sustains ~88 returning atomics/us, so rays are never fetched one wave-load at a time) and hands the rays
// to its lanes.  A lane whose ray is finished writes its hit id and goes idle; when at least tune.refill_min
// lanes of the wave are idle they are re-filled together, so the 64 lanes stay busy although incoherent rays
// need very different numbers of steps.  Traversal is the speculative while-while of traverse_ww; the per-lane
// stack keeps its first STACK_L entries in LDS ([entry][lane]: bank = lane, conflict-free) and spills deeper
// entries to a global buffer ([entry][thread], coalesced).
// XCD id of the executing wave (HW_REG_XCC_ID, bits 3:0).  Used for L2 affinity only, never for correctness.
PRT_DEV uint32_t xcc_id() { return (uint32_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11)) & 7u; }

// Chunk cursor(s) of the persistent traversal kernels: work[32 * x] (own 128-B line each) counts the chunks
// taken from XCD x's contiguous eighth of the ray buffer.  Neighbouring rays start in neighbouring parts of the
// scene, so keeping an eighth of the buffer on one XCD keeps the deep BVH levels it touches in that XCD's 4 MB L2.
// A wave first drains its own XCD's range, then steals from the others, so the grid still balances.
PRT_DEV uint32_t grab_chunk(uint32_t* work, uint32_t n_chunks, uint32_t my_xcd, bool affinity) {
    if (!affinity) {
        const uint32_t c = atomicAdd(work, 1u);
        return c < n_chunks ? c : 0xFFFFFFFFu;
    }
    for (uint32_t i = 0; i < 8u; ++i) {
        const uint32_t x = (my_xcd + i) & 7u;
        const uint32_t lo = (uint32_t)(((unsigned long long)n_chunks * x) >> 3);
        const uint32_t hi = (uint32_t)(((unsigned long long)n_chunks * (x + 1u)) >> 3);
        if (work[32u * x] >= hi - lo) continue;  // already drained (plain read: a stale value only costs one atomic)
        const uint32_t c = atomicAdd(&work[32u * x], 1u);
        if (c < hi - lo) return lo + c;
    }
    return 0xFFFFFFFFu;
}

// MODE 0: all entries in LDS, the host guarantees the tree never needs more than STACK_L.
// MODE 1: LDS + global spill behind entry STACK_L (any depth; slower: the compiler merges the two address spaces
//         into flat accesses).
// MODE 2: all entries in LDS with an overflow check: a push beyond STACK_L sets `overflow` instead of writing; the
//         kernel then hands that ray to the MODE-1 instance through the overflow list (never seen in practice: the
//         host's bound is a worst case over all paths with every child hit).
// Wave64 inclusive scans with DPP row shifts / row broadcasts (no LDS round trips): the Hillis-Steele steps
// 1, 2, 4, 8 inside each 16-lane row, then lane 15 of rows 0/2 into rows 1/3 and lane 31 into rows 2/3.  Lanes that
// receive nothing keep the identity (0).  All 64 lanes must be active.
#define PRT_DPP(X, CTRL, ROWMASK) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(X), CTRL, ROWMASK, 0xF, false)
PRT_DEV uint32_t wave_scan_add(uint32_t x) {
    x += PRT_DPP(x, 0x111, 0xF);  // row_shr:1
    x += PRT_DPP(x, 0x112, 0xF);  // row_shr:2
    x += PRT_DPP(x, 0x114, 0xF);  // row_shr:4
    x += PRT_DPP(x, 0x118, 0xF);  // row_shr:8
    x += PRT_DPP(x, 0x142, 0xA);  // row_bcast:15 -> rows 1, 3
    x += PRT_DPP(x, 0x143, 0xC);  // row_bcast:31 -> rows 2, 3
    return x;
}
PRT_DEV uint32_t wave_scan_or(uint32_t x) {
    x |= PRT_DPP(x, 0x111, 0xF);
    x |= PRT_DPP(x, 0x112, 0xF);
    x |= PRT_DPP(x, 0x114, 0xF);
    x |= PRT_DPP(x, 0x118, 0xF);
    x |= PRT_DPP(x, 0x142, 0xA);
    x |= PRT_DPP(x, 0x143, 0xC);
    return x;
}
PRT_DEV uint32_t wave_scan_max(uint32_t x) {
    uint32_t t;
    t = PRT_DPP(x, 0x111, 0xF); x = x > t ? x : t;
    t = PRT_DPP(x, 0x112, 0xF); x = x > t ? x : t;
    t = PRT_DPP(x, 0x114, 0xF); x = x > t ? x : t;
    t = PRT_DPP(x, 0x118, 0xF); x = x > t ? x : t;
    t = PRT_DPP(x, 0x142, 0xA); x = x > t ? x : t;
    t = PRT_DPP(x, 0x143, 0xC); x = x > t ? x : t;
    return x;
}

template <int STACK_L, int MODE>
struct LaneStack {
    uint32_t* lds;      // &s_stack[threadIdx.x], stride 256
    uint32_t* spill;    // &spill[global thread], stride n_threads (MODE 1 only)
    uint32_t stride;
    int sp;
    bool overflow;
    PRT_DEV void push(uint32_t v) {
        if (MODE == 1) {
            if (sp < STACK_L)
                lds[sp * 256] = v;
            else
                spill[(size_t)(sp - STACK_L) * stride] = v;
            ++sp;
        } else if (MODE == 2) {
            if (sp < STACK_L) {
                lds[sp * 256] = v;
                ++sp;
            } else {
                overflow = true;
            }
        } else {
            lds[sp * 256] = v;
            ++sp;
        }
    }
    // precondition: sp > 0
    PRT_DEV int pop() {
        --sp;
        if (MODE != 1 || sp < STACK_L) return (int)lds[sp * 256];
        return (int)spill[(size_t)(sp - STACK_L) * stride];
    }
};

template <int STACK_L, int WAVES, int MODE, bool STATS>
__global__ void __launch_bounds__(256, WAVES) k_traverse_persistent(DevScene sc, const float4* __restrict__ ro,
                                                                    const float4* __restrict__ rd,
                                                                    uint32_t* __restrict__ hit,
                                                                    const float* __restrict__ hd2,
                                                                    const uint32_t* __restrict__ count_ptr,
                                                                    uint32_t* __restrict__ work,
                                                                    uint32_t* __restrict__ spill,
                                                                    const uint32_t* __restrict__ index_list,
                                                                    uint32_t* __restrict__ ovf, PrtTravTuning tune,
                                                                    unsigned long long* __restrict__ stats) {
    __shared__ uint32_t s_stack[STACK_L * 256];
    __shared__ uint32_t s_iters[4];
    uint32_t count = *count_ptr;
    if (index_list && count > PRT_OVF_CAP) count = PRT_OVF_CAP;
    const uint32_t chunk = tune.chunk;
    const uint32_t n_chunks = (count + chunk - 1u) / chunk;
    const uint32_t my_xcd = xcc_id();
    if (STATS) {
        if (threadIdx.x < 4) s_iters[threadIdx.x] = 0;
        __syncthreads();
    }
    LaneStack<STACK_L, MODE> st;
    st.overflow = false;
    st.lds = &s_stack[threadIdx.x];
    st.stride = gridDim.x * 256u;
    st.spill = spill + (blockIdx.x * 256u + threadIdx.x);
    st.sp = 0;
    const uint32_t lane = lane_id();
    uint32_t k = 0xFFFFFFFFu;
    int node = NODE_DONE, leaf = LEAF_NONE;
    f3 o = mk3(0.f, 0.f, 0.f), ld = mk3(0.f, 0.f, 1.f);
    float ix = 0.f, iy = 0.f, iz = 0.f, ax = 0.f, ay = 0.f, az = 0.f, bx = 0.f, by = 0.f, bz = 0.f, pad = 0.f, tlimit = 0.f;
    Closest best;
    best.d2 = 3.402823466e+38f;
    best.id = HIT_MISS;
    best.prim = 0xFFFFFFFFu;
    uint32_t n_nodes = 0, n_tris = 0;
    uint32_t cur = 0, cur_end = 0;  // wave-uniform: this wave's current chunk [cur, cur_end)
    bool exhausted = false;         // wave-uniform
    // Exit condition every wave reaches: the loop ends when the ray buffer is exhausted and the wave's lanes are
    // idle; the iteration cap is a watchdog (a wave handles ~count/waves rays x ~100 steps, orders of magnitude
    // below it) that turns a would-be hang into an error flag the host reports.
    for (uint32_t guard = 0;; ++guard) {
        if (guard > (1u << 22)) {
            if (lane == 0) atomicOr(work + 256, 1u);
            break;
        }
        const bool idle = (node == NODE_DONE) && (leaf == LEAF_NONE);
        if (idle && k != 0xFFFFFFFFu) {
            if (MODE == 2 && st.overflow) {
                const uint32_t j = atomicAdd(ovf, 1u);  // re-done from scratch by the spill-capable instance
                if (j < PRT_OVF_CAP)
                    ovf[1u + j] = k;
                else
                    atomicOr(work + 256, 2u);  // list full: reported by prt_synchronize like the watchdog
                st.overflow = false;
            } else {
                hit[k] = best.id;
            }
            k = 0xFFFFFFFFu;
        }
        const unsigned long long idle_mask = __ballot(idle);
        const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
        if (!exhausted && n_idle >= tune.refill_min) {
            if (cur == cur_end) {  // grab the next chunk (one global atomic per `chunk` rays)
                uint32_t c = 0xFFFFFFFFu;
                if (lane == 0) c = grab_chunk(work, n_chunks, my_xcd, tune.xcd_affinity != 0u);
                c = (uint32_t)__shfl((int)c, 0, 64);
                if (c == 0xFFFFFFFFu) {
                    exhausted = true;
                } else {
                    cur = c * chunk;
                    cur_end = (cur + chunk < count) ? cur + chunk : count;
                }
            }
            if (!exhausted) {
                const uint32_t qi = cur + (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
                if (idle && qi < cur_end) {
                    const uint32_t kk = index_list ? index_list[qi] : qi;
                    const uint32_t hid = hit[kk];
                    if (hid != HIT_DEAD) {
                        const float4 O = ro[kk];
                        const float4 D = rd[kk];
                        o = mk3(O.x, O.y, O.z);
                        ld = normalize3(mk3(D.x, D.y, D.z));  // TransformNormal(identity, d), primitive.cpp:30
                        pad = sc.pad * (__builtin_fabsf(o.x) + __builtin_fabsf(o.y) + __builtin_fabsf(o.z) + sc.extent);
                        ix = 1.0f / (__builtin_fabsf(ld.x) < 1e-30f ? __builtin_copysignf(1e-30f, ld.x) : ld.x);
                        iy = 1.0f / (__builtin_fabsf(ld.y) < 1e-30f ? __builtin_copysignf(1e-30f, ld.y) : ld.y);
                        iz = 1.0f / (__builtin_fabsf(ld.z) < 1e-30f ? __builtin_copysignf(1e-30f, ld.z) : ld.z);
                        ax = (o.x + pad) * ix; ay = (o.y + pad) * iy; az = (o.z + pad) * iz;
                        bx = (o.x - pad) * ix; by = (o.y - pad) * iy; bz = (o.z - pad) * iz;
                        best.id = hid;
                        best.prim = hid;  // analytic index, or 0xFFFFFFFF for a miss
                        best.d2 = hd2[kk];
                        tlimit = limit_from_d2(best.d2, pad);
                        k = kk;
                        node = 0;
                        leaf = LEAF_NONE;
                        st.sp = 0;
                    }
                }
                cur = (cur + n_idle < cur_end) ? cur + n_idle : cur_end;
            }
        }
        if (__ballot(k != 0xFFFFFFFFu) == 0ull) {
            if (exhausted) break;
            continue;
        }
        // ---- phase 1: internal nodes (leave when at most exit_max lanes are still looking for a leaf) ----
        while ((unsigned)node < 0x40000000u) {
            const float4 q0 = sc.nodes[4 * (size_t)node + 0];
            const float4 q1 = sc.nodes[4 * (size_t)node + 1];
            const float4 q2 = sc.nodes[4 * (size_t)node + 2];
            const float4 q3 = sc.nodes[4 * (size_t)node + 3];
            if (STATS) {
                ++n_nodes;
                if ((int)lane == __ffsll((long long)__ballot(true)) - 1) ++s_iters[threadIdx.x >> 6];
            }
            const float l0x = __builtin_fmaf(q0.x, ix, -ax), l1x = __builtin_fmaf(q0.w, ix, -bx);
            const float l0y = __builtin_fmaf(q0.y, iy, -ay), l1y = __builtin_fmaf(q1.x, iy, -by);
            const float l0z = __builtin_fmaf(q0.z, iz, -az), l1z = __builtin_fmaf(q1.y, iz, -bz);
            const float r0x = __builtin_fmaf(q1.z, ix, -ax), r1x = __builtin_fmaf(q2.y, ix, -bx);
            const float r0y = __builtin_fmaf(q1.w, iy, -ay), r1y = __builtin_fmaf(q2.z, iy, -by);
            const float r0z = __builtin_fmaf(q2.x, iz, -az), r1z = __builtin_fmaf(q2.w, iz, -bz);
            const float tnL = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(l0x, l1x), __builtin_fminf(l0y, l1y)),
                                              __builtin_fmaxf(__builtin_fminf(l0z, l1z), 0.0f));
            const float tfL = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(l0x, l1x), __builtin_fmaxf(l0y, l1y)),
                                              __builtin_fminf(__builtin_fmaxf(l0z, l1z), tlimit));
            const float tnR = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(r0x, r1x), __builtin_fminf(r0y, r1y)),
                                              __builtin_fmaxf(__builtin_fminf(r0z, r1z), 0.0f));
            const float tfR = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(r0x, r1x), __builtin_fmaxf(r0y, r1y)),
                                              __builtin_fminf(__builtin_fmaxf(r0z, r1z), tlimit));
            const bool hL = tnL <= tfL * 1.0000005f;
            const bool hR = tnR <= tfR * 1.0000005f;
            const int left = __float_as_int(q3.x), right = __float_as_int(q3.y);
            // select chain instead of a 4-way branch: candidate = nearer hit child, or NEED_POP
            const bool both = hL && hR;
            const bool lfirst = tnL <= tnR;
            const int nearc = lfirst ? left : right;
            const int farc = lfirst ? right : left;
            const int cand = both ? nearc : (hL ? left : (hR ? right : NEED_POP));
            if (both) st.push((uint32_t)farc);
            node = cand;
            if (cand == NEED_POP) node = (st.sp > 0) ? st.pop() : NODE_DONE;
            if (node < 0 && leaf == LEAF_NONE) {  // first leaf (a child or a popped entry): postpone it, keep walking
                leaf = node;
                node = (st.sp > 0) ? st.pop() : NODE_DONE;
            }
            if (MODE == 2 && st.overflow) {  // give the ray up; it is re-traversed by the MODE-1 instance
                node = NODE_DONE;
                leaf = LEAF_NONE;
                st.sp = 0;
            }
            if ((uint32_t)__popcll(__ballot(leaf == LEAF_NONE && node != NODE_DONE)) <= tune.exit_max) break;
        }
        // ---- phase 2: leaves ----
        while (leaf != LEAF_NONE) {
            const uint32_t ref = ~(uint32_t)leaf;
            const uint32_t first = ref >> 4, cnt = ref & 15u;
            for (uint32_t t = 0; t < cnt; ++t) {
                const uint32_t slot = first + t;
                const float4 a = sc.tris[3 * (size_t)slot + 0];
                const float4 b = sc.tris[3 * (size_t)slot + 1];
                const float4 c = sc.tris[3 * (size_t)slot + 2];
                if (STATS) ++n_tris;
                f3 pos;
                float b1, b2;
                if (triangle_hit_pos(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), o, ld, pos, b1, b2)) {
                    const float d2 = dist2(o, pos);
                    const uint32_t prim = __float_as_uint(a.w);
                    if (d2 < best.d2 || (d2 == best.d2 && best.id != HIT_MISS && prim < best.prim)) {
                        best.d2 = d2;
                        best.id = sc.n_prims + slot;
                        best.prim = prim;
                        tlimit = limit_from_d2(d2, pad);
                    }
                }
            }
            leaf = LEAF_NONE;
            if (node < 0) {  // the walk stopped on a second leaf: take it now
                leaf = node;
                node = (st.sp > 0) ? st.pop() : NODE_DONE;
            }
        }
    }
    if (STATS) {
        atomicAdd(&stats[0], (unsigned long long)n_nodes);
        atomicAdd(&stats[1], (unsigned long long)n_tris);
        __syncthreads();
        if (threadIdx.x < 4) atomicAdd(&stats[3], 64ull * s_iters[threadIdx.x]);
    }
}

// The same kernel over the 4-wide tree (default): half as many dependent node fetches per ray, one 128-B cache
// line per visit, children visited in entry-distance order (5-comparator sorting network).
template <int STACK_L, int WAVES, int MODE, bool STATS>
__global__ void __launch_bounds__(256, WAVES) k_traverse4_persistent(DevScene sc, const float4* __restrict__ ro,
                                                                    const float4* __restrict__ rd,
                                                                    uint32_t* __restrict__ hit,
                                                                    const float* __restrict__ hd2,
                                                                    const uint32_t* __restrict__ count_ptr,
                                                                    uint32_t* __restrict__ work,
                                                                    uint32_t* __restrict__ spill,
                                                                    const uint32_t* __restrict__ index_list,
                                                                    uint32_t* __restrict__ ovf, PrtTravTuning tune,
                                                                    unsigned long long* __restrict__ stats) {
    __shared__ uint32_t s_stack[(STACK_L + (MODE == 3 ? 3 : 0)) * 256];  // MODE 3 writes up to 3 rows past the top
    __shared__ unsigned long long s_key[256];  // per lane: best (d2 bits << 32 | prim) of the cooperative triangle tests
    __shared__ uint32_t s_slot[256];           // per lane: leaf-order slot of that best
    __shared__ uint32_t s_mark[256];           // per item position of a round: owner lane + 1 where an owner's range starts
    __shared__ uint32_t s_iters[8];  // [0..3] node-loop, [4..7] triangle-loop iterations per wave (STATS)
    uint32_t count = *count_ptr;
    if (index_list && count > PRT_OVF_CAP) count = PRT_OVF_CAP;
    const uint32_t chunk = tune.chunk;
    const uint32_t n_chunks = (count + chunk - 1u) / chunk;
    const uint32_t my_xcd = xcc_id();
    if (STATS) {
        if (threadIdx.x < 8) s_iters[threadIdx.x] = 0;
        __syncthreads();
    }
    LaneStack<STACK_L, MODE> st;
    st.overflow = false;
    st.lds = &s_stack[threadIdx.x];
    st.stride = gridDim.x * 256u;
    st.spill = spill + (blockIdx.x * 256u + threadIdx.x);
    st.sp = 0;
    const uint32_t lane = lane_id();
    uint32_t k = 0xFFFFFFFFu;
    int node = NODE_DONE, leaf = LEAF_NONE;
    f3 o = mk3(0.f, 0.f, 0.f), ld = mk3(0.f, 0.f, 1.f);
    float ix = 0.f, iy = 0.f, iz = 0.f, ax = 0.f, ay = 0.f, az = 0.f, bx = 0.f, by = 0.f, bz = 0.f, pad = 0.f, tlimit = 0.f;
    Closest best;
    best.d2 = 3.402823466e+38f;
    best.id = HIT_MISS;
    best.prim = 0xFFFFFFFFu;
    uint32_t n_nodes = 0, n_tris = 0, max_sp = 0;
    uint32_t cur = 0, cur_end = 0;  // wave-uniform: this wave's current chunk [cur, cur_end)
    bool exhausted = false;         // wave-uniform
    // Exit condition every wave reaches: the loop ends when the ray buffer is exhausted and the wave's lanes are
    // idle; the iteration cap is a watchdog (a wave handles ~count/waves rays x ~100 steps, orders of magnitude
    // below it) that turns a would-be hang into an error flag the host reports.
    for (uint32_t guard = 0;; ++guard) {
        if (guard > (1u << 22)) {
            if (lane == 0) atomicOr(work + 256, 1u);
            break;
        }
        const bool idle = (node == NODE_DONE) && (leaf == LEAF_NONE);
        if (idle && k != 0xFFFFFFFFu) {
            if ((MODE == 2 || MODE == 3) && st.overflow) {
                const uint32_t j = atomicAdd(ovf, 1u);  // re-done from scratch by the spill-capable instance
                if (j < PRT_OVF_CAP)
                    ovf[1u + j] = k;
                else
                    atomicOr(work + 256, 2u);  // list full: reported by prt_synchronize like the watchdog
                st.overflow = false;
            } else {
                hit[k] = best.id;
            }
            k = 0xFFFFFFFFu;
        }
        const unsigned long long idle_mask = __ballot(idle);
        const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
        if (!exhausted && n_idle >= tune.refill_min) {
            if (cur == cur_end) {  // grab the next chunk (one global atomic per `chunk` rays)
                uint32_t c = 0xFFFFFFFFu;
                if (lane == 0) c = grab_chunk(work, n_chunks, my_xcd, tune.xcd_affinity != 0u);
                c = (uint32_t)__shfl((int)c, 0, 64);
                if (c == 0xFFFFFFFFu) {
                    exhausted = true;
                } else {
                    cur = c * chunk;
                    cur_end = (cur + chunk < count) ? cur + chunk : count;
                }
            }
            if (!exhausted) {
                const uint32_t qi = cur + (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
                if (idle && qi < cur_end) {
                    const uint32_t kk = index_list ? index_list[qi] : qi;
                    const uint32_t hid = hit[kk];
                    if (hid != HIT_DEAD) {
                        const float4 O = ro[kk];
                        const float4 D = rd[kk];
                        o = mk3(O.x, O.y, O.z);
                        ld = normalize3(mk3(D.x, D.y, D.z));  // TransformNormal(identity, d), primitive.cpp:30
                        pad = sc.pad * (__builtin_fabsf(o.x) + __builtin_fabsf(o.y) + __builtin_fabsf(o.z) + sc.extent);
                        ix = 1.0f / (__builtin_fabsf(ld.x) < 1e-30f ? __builtin_copysignf(1e-30f, ld.x) : ld.x);
                        iy = 1.0f / (__builtin_fabsf(ld.y) < 1e-30f ? __builtin_copysignf(1e-30f, ld.y) : ld.y);
                        iz = 1.0f / (__builtin_fabsf(ld.z) < 1e-30f ? __builtin_copysignf(1e-30f, ld.z) : ld.z);
                        ax = (o.x + pad) * ix; ay = (o.y + pad) * iy; az = (o.z + pad) * iz;
                        bx = (o.x - pad) * ix; by = (o.y - pad) * iy; bz = (o.z - pad) * iz;
                        best.id = hid;
                        best.prim = hid;  // analytic index, or 0xFFFFFFFF for a miss
                        best.d2 = hd2[kk];
                        tlimit = limit_from_d2(best.d2, pad);
                        k = kk;
                        node = 0;
                        leaf = LEAF_NONE;
                        st.sp = 0;
                    }
                }
                cur = (cur + n_idle < cur_end) ? cur + n_idle : cur_end;
            }
        }
        if (__ballot(k != 0xFFFFFFFFu) == 0ull) {
            if (exhausted) break;
            continue;
        }
        // ---- phase 1: internal nodes (leave when at most exit_max lanes are still looking for a leaf) ----
        while ((unsigned)node < 0x40000000u) {
            const int sp0 = st.sp;
            int t1 = 0, t2 = 0;
            if (MODE == 3) {  // the two entries under the top, read before anything is written this step
                t1 = (int)st.lds[(sp0 > 0 ? sp0 - 1 : 0) * 256];
                t2 = (int)st.lds[(sp0 > 1 ? sp0 - 2 : 0) * 256];
            }
            const float4* nb = sc.nodes4 + 8 * (size_t)node;
            const float4 mnx = nb[0], mxx = nb[1], mny = nb[2], mxy = nb[3], mnz = nb[4], mxz = nb[5], rf = nb[6];
            if (STATS) {
                ++n_nodes;
                if ((int)lane == __ffsll((long long)__ballot(true)) - 1) ++s_iters[threadIdx.x >> 6];
            }
            const float kInf = __builtin_inff();
            float k0, k1, k2, k3;
            // children are tested two at a time with packed FMAs (v_pk_fma_f32: two independent fp32 FMAs, the same
            // rounding as the scalar form); min/max have no packed fp32 form
#define PRT_PAIR(LO, HI, KA, KB)                                                                                    \
    {                                                                                                               \
        const v2f x0 = __builtin_elementwise_fma((v2f){mnx.LO, mnx.HI}, (v2f){ix, ix}, (v2f){-ax, -ax});            \
        const v2f x1 = __builtin_elementwise_fma((v2f){mxx.LO, mxx.HI}, (v2f){ix, ix}, (v2f){-bx, -bx});            \
        const v2f y0 = __builtin_elementwise_fma((v2f){mny.LO, mny.HI}, (v2f){iy, iy}, (v2f){-ay, -ay});            \
        const v2f y1 = __builtin_elementwise_fma((v2f){mxy.LO, mxy.HI}, (v2f){iy, iy}, (v2f){-by, -by});            \
        const v2f z0 = __builtin_elementwise_fma((v2f){mnz.LO, mnz.HI}, (v2f){iz, iz}, (v2f){-az, -az});            \
        const v2f z1 = __builtin_elementwise_fma((v2f){mxz.LO, mxz.HI}, (v2f){iz, iz}, (v2f){-bz, -bz});            \
        const float tna = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0.x, x1.x), __builtin_fminf(y0.x, y1.x)), \
                                          __builtin_fmaxf(__builtin_fminf(z0.x, z1.x), 0.0f));                       \
        const float tfa = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0.x, x1.x), __builtin_fmaxf(y0.x, y1.x)), \
                                          __builtin_fminf(__builtin_fmaxf(z0.x, z1.x), tlimit));                     \
        const float tnb = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0.y, x1.y), __builtin_fminf(y0.y, y1.y)), \
                                          __builtin_fmaxf(__builtin_fminf(z0.y, z1.y), 0.0f));                       \
        const float tfb = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0.y, x1.y), __builtin_fmaxf(y0.y, y1.y)), \
                                          __builtin_fminf(__builtin_fmaxf(z0.y, z1.y), tlimit));                     \
        const v2f tfs = (v2f){tfa, tfb} * (v2f){1.0000005f, 1.0000005f};                                             \
        KA = (tna <= tfs.x) ? tna : kInf;                                                                           \
        KB = (tnb <= tfs.y) ? tnb : kInf;                                                                           \
    }
            PRT_PAIR(x, y, k0, k1)
            PRT_PAIR(z, w, k2, k3)
#undef PRT_PAIR
            int r0 = __float_as_int(rf.x), r1 = __float_as_int(rf.y), r2 = __float_as_int(rf.z), r3 = __float_as_int(rf.w);
            // sort the four (entry distance, ref) pairs ascending: 5-comparator network; misses (+inf) end up last
#define PRT_CSWAP(KA, RA, KB, RB)            \
    {                                        \
        const bool sw = KB < KA;             \
        const float kt = sw ? KB : KA;       \
        KB = sw ? KA : KB;                   \
        KA = kt;                             \
        const int rt = sw ? RB : RA;         \
        RB = sw ? RA : RB;                   \
        RA = rt;                             \
    }
            PRT_CSWAP(k0, r0, k1, r1)
            PRT_CSWAP(k2, r2, k3, r3)
            PRT_CSWAP(k0, r0, k2, r2)
            PRT_CSWAP(k1, r1, k3, r3)
            PRT_CSWAP(k1, r1, k2, r2)
#undef PRT_CSWAP
            if (MODE == 3) {
                // Branch-free step.  The three far children are stored unconditionally above the top (farthest
                // first) and the stack pointer advances only past the ones that were hit; the two entries below
                // the old top were read into t1/t2 at the head of the step, so "next node" and "postpone a leaf"
                // are pure selects.  A step that would leave more than STACK_L entries gives the ray up (it is
                // re-traversed by the spill-capable instance through the overflow list).
                const int h0 = k0 < kInf, h1 = k1 < kInf, h2 = k2 < kInf, h3 = k3 < kInf;
                st.lds[sp0 * 256] = (uint32_t)r3;
                const int s1 = sp0 + h3;
                st.lds[s1 * 256] = (uint32_t)r2;
                const int s2 = s1 + h2;
                st.lds[s2 * 256] = (uint32_t)r1;
                const int sp1 = s2 + h1;
                const int below0 = sp0 > 0 ? t1 : NODE_DONE;  // top of the stack before this step
                const int below1 = sp0 > 1 ? t2 : NODE_DONE;  // the entry under it
                const int nxt = h0 ? r0 : below0;
                const int spn = h0 ? sp1 : (sp0 > 0 ? sp0 - 1 : 0);
                const bool take = (nxt < 0) && (leaf == LEAF_NONE);  // first leaf: postpone it, keep walking
                const int under = h0 ? (h1 ? r1 : below0) : below1;
                const int spu = h0 ? (h1 ? sp1 - 1 : (sp0 > 0 ? sp0 - 1 : 0)) : (sp0 > 1 ? sp0 - 2 : 0);
                leaf = take ? nxt : leaf;
                node = take ? under : nxt;
                st.sp = take ? spu : spn;
                if (sp1 > STACK_L) st.overflow = true;
            } else {
            // farthest first, so that the nearest pending child is popped first
            if (k3 < kInf) st.push((uint32_t)r3);
            if (k2 < kInf) st.push((uint32_t)r2);
            if (k1 < kInf) st.push((uint32_t)r1);
            node = (k0 < kInf) ? r0 : NEED_POP;
            if (node == NEED_POP) node = (st.sp > 0) ? st.pop() : NODE_DONE;
            if (node < 0 && leaf == LEAF_NONE) {  // first leaf (a child or a popped entry): postpone it, keep walking
                leaf = node;
                node = (st.sp > 0) ? st.pop() : NODE_DONE;
            }
            }
            if ((MODE == 2 || MODE == 3) && st.overflow) {  // give the ray up; it is re-traversed by the MODE-1 instance
                node = NODE_DONE;
                leaf = LEAF_NONE;
                st.sp = 0;
            }
            if (STATS && (uint32_t)st.sp > max_sp) max_sp = (uint32_t)st.sp;
            if ((uint32_t)__popcll(__ballot(leaf == LEAF_NONE && node != NODE_DONE)) <= tune.exit_max) break;
        }
        // ---- phase 2: leaves, tested COOPERATIVELY by the wave ----
        // A lane holds 0..2 pending leaves (the postponed one and, if the walk stopped on a second leaf, that one)
        // = 0..8 triangles.  Testing them lane by lane leaves 86 % of the lanes idle (measured), so the wave pools
        // all pending (ray, triangle) pairs and every lane tests ONE pair per round: the pair's owner lane is found
        // by a 6-step search over the running offsets, the owner's ray comes over ds_bpermute, and the results are
        // merged per owner with a 64-bit LDS atomicMin on (d2 bits << 32 | primitive index) -- exactly the
        // reference's rule "smaller d2 wins, ties to the lower primitive index" (primitive.cpp:42-48), since d2 >= 0
        // makes the float order equal to the integer order of its bits.
        {
            const bool second = node < 0;  // NODE_DONE / internal ids are >= 0
            uint32_t f1 = 0, c1 = 0, f2 = 0, c2 = 0;
            if (leaf != LEAF_NONE) {
                const uint32_t ref = ~(uint32_t)leaf;
                f1 = ref >> 4;
                c1 = ref & 15u;
            }
            if (second) {
                const uint32_t ref = ~(uint32_t)node;
                f2 = ref >> 4;
                c2 = ref & 15u;
            }
            const uint32_t total = c1 + c2;
            if (__ballot(total != 0u) != 0ull) {  // wave-uniform
                const uint32_t incl = wave_scan_add(total);
                const uint32_t off = incl - total;
                const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                // a miss is encoded with prim 0 so that a candidate with d2 == FLT_MAX can never win (primitive.cpp:44)
                const unsigned long long key_best =
                    ((unsigned long long)__float_as_uint(best.d2) << 32) | (best.id == HIT_MISS ? 0u : best.prim);
                const uint32_t my = threadIdx.x;             // this lane's LDS cell
                const uint32_t wbase = threadIdx.x & ~63u;  // first cell of this wave
                s_key[my] = key_best;
                uint32_t carry = 0;  // wave-uniform
                for (uint32_t base = 0; base < T; base += 64u) {  // wave-uniform trip count
                    const uint32_t kq = base + lane;
                    // owner of item kq: every owner whose range starts inside this round marks its first position,
                    // a max-scan spreads the marks to the right; positions before the first mark continue the last
                    // owner of the previous round
                    s_mark[my] = 0u;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    if (total != 0u && off >= base && off < base + 64u) s_mark[wbase + (off - base)] = lane + 1u;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    const uint32_t mk = wave_scan_max(((volatile uint32_t*)s_mark)[my]);
                    const uint32_t owner = mk ? mk - 1u : carry;
                    carry = (uint32_t)__builtin_amdgcn_readlane((int)owner, 63);
                    const uint32_t ooff = (uint32_t)__shfl((int)off, (int)owner, 64);
                    const uint32_t oc1 = (uint32_t)__shfl((int)c1, (int)owner, 64);
                    const uint32_t of1 = (uint32_t)__shfl((int)f1, (int)owner, 64);
                    const uint32_t of2 = (uint32_t)__shfl((int)f2, (int)owner, 64);
                    const f3 qo = mk3(__shfl(o.x, (int)owner, 64), __shfl(o.y, (int)owner, 64), __shfl(o.z, (int)owner, 64));
                    const f3 qd = mk3(__shfl(ld.x, (int)owner, 64), __shfl(ld.y, (int)owner, 64), __shfl(ld.z, (int)owner, 64));
                    bool cand = false;
                    unsigned long long key = 0ull;
                    uint32_t slot = 0;
                    if (kq < T) {
                        const uint32_t j = kq - ooff;
                        slot = j < oc1 ? of1 + j : of2 + (j - oc1);
                        const float4 a = sc.tris[3 * (size_t)slot + 0];
                        const float4 b = sc.tris[3 * (size_t)slot + 1];
                        const float4 c = sc.tris[3 * (size_t)slot + 2];
                        if (STATS) ++n_tris;
                        f3 pos;
                        float b1, b2;
                        if (triangle_hit_pos(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), qo, qd, pos, b1, b2)) {
                            const float d2 = dist2(qo, pos);
                            key = ((unsigned long long)__float_as_uint(d2) << 32) | __float_as_uint(a.w);
                            // NaN / inf d2 have bit patterns above FLT_MAX's: they can never win, as in the reference
                            cand = true;
                            atomicMin(&s_key[wbase + owner], key);
                        }
                    }
                    if (STATS && lane == 0) ++s_iters[4 + (threadIdx.x >> 6)];
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    if (cand && ((volatile unsigned long long*)s_key)[wbase + owner] == key) s_slot[wbase + owner] = slot;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                }
                const unsigned long long won = ((volatile unsigned long long*)s_key)[my];
                if (won != key_best) {
                    best.d2 = __uint_as_float((uint32_t)(won >> 32));
                    best.prim = (uint32_t)won;
                    best.id = sc.n_prims + ((volatile uint32_t*)s_slot)[my];
                    tlimit = limit_from_d2(best.d2, pad);
                }
                leaf = LEAF_NONE;
                if (second) node = (st.sp > 0) ? st.pop() : NODE_DONE;
                if (node < 0) {  // a popped leaf: postpone it (and a second one stays in `node`)
                    leaf = node;
                    node = (st.sp > 0) ? st.pop() : NODE_DONE;
                }
            }
        }
    }
    if (STATS) {
        atomicAdd(&stats[0], (unsigned long long)n_nodes);
        atomicAdd(&stats[1], (unsigned long long)n_tris);
        __syncthreads();
        if (threadIdx.x < 4) atomicAdd(&stats[3], 64ull * s_iters[threadIdx.x]);
        if (threadIdx.x < 4) atomicAdd(&stats[4], 64ull * s_iters[4 + threadIdx.x]);
        atomicMax(&stats[5], (unsigned long long)max_sp);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Default traversal: persistent waves over the COMPRESSED 8-WIDE tree (BVH8Q, layout in bvh.h).
//
// Why: the 4-wide kernel above is bound by dependent node fetches that miss the 4 MB per-XCD L2 (C3: 37 MB of
// 128-B nodes, L2 hit 46 %, 3.4 TB/s leaving the L2s, waves parked on s_waitcnt 64 % of their cycles).  An 8-wide
// node with 8-bit quantized child boxes is 80 B for 8 children: the whole C3 node array is ~5 MB, a ray needs
// ~0.6x as many dependent fetches, and a step needs no sorting network: the children sit in octant-ordered slots, a
// step produces a hit MASK, and the stack holds one (child base, pending hit mask) "node group" per level
// (after Ylitie, Karras, Laine: "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", 2017;
// this formulation is for wave64 with the stack in LDS and cooperative triangle tests).
//
// Per lane: G = (child_base, pending internal hits in bits 24..31 by visiting priority | imask in bits 0..7) and
// up to two pending TRIANGLE groups (tri_base, 24-bit mask): the first one found is postponed while the lane keeps
// walking, a second one stops the lane until the wave's next triangle phase.  There all pending (ray, triangle)
// pairs of the wave are written to an LDS work queue and tested one pair per lane per round; results are merged per
// owner with a 64-bit LDS atomicMin on (d2 bits << 32 | primitive index) = the reference's rule (primitive.cpp:42-48).
// Culling stays conservative: the quantized boxes contain the exact ones (bvh.cpp verifies it in exact arithmetic),
// near planes move out by -pad and far planes by +pad exactly as in the 4-wide kernel, so results are bit-identical.
// A ray that would need more than STACK_L stacked groups goes to the overflow list (re-traversed by the 4-wide
// spill-capable instance); the host sizes STACK_L from the tree's depth, so that never happens in practice.
// ---------------------------------------------------------------------------------------------------------
#define T8_QCAP 256u    // (ray, triangle) pairs one pass of a triangle phase tests (4 B each)
#define T8_QGROUPS 128u // triangle groups a wave can queue between two phases (8 B each, the same memory)

// INST = true (scenes with placed mesh copies, PrtInstance): nodes8 starts with a TOP-LEVEL tree whose "triangles" are
// instances.  A lane that finds instance hits there saves its top-level group, pushes a sentinel and restarts in the
// instance's tree with the reference's per-primitive local ray (local origin = Inv * o, local direction =
// normalize(transpose(mat3(Mat)) * d), primitive.cpp:29-30); popping the sentinel restores the world ray.  All
// triangles live in instance space (world-space meshes form one identity instance, which reproduces the
// non-instanced results bit for bit); a candidate's key is its WORLD distance^2 |o - Mat * pos|^2 and its global
// primitive index, as in the reference.  A lane changes level only when none of its items is left in the queue.
// ---- the box tests of one 80-B node (w0..w4) against the lane's ray: expands to `eim` and `hitmask` --------------------
// Quantized planes, children 0..3 (a) and 4..7 (b): w2 = {lox a, lox b, loy a, loy b}, w3 = {loz a, loz b, hix a, hix b},
// w4 = {hiy a, hiy b, hiz a, hiz b}.  The near plane is the hi plane on an axis the ray travels along negatively; selected
// with v_bfi_b32 under per-axis sign masks, not v_cndmask_b32 (the compiler puts some of the twelve selects on vcc; with
// the bit selects the 5-wave instance needs no scratch).  The per-ray pad (2^-18 (|o|_1 + extent) in position space)
// exceeds the rounding of the plane distances (a few 2^-24 of the same magnitude) by a factor > 10, so no relative slack
// is needed on top of it.  Scalar FMAs: v_pk_fma_f32 measured 3.5 % slower here (12.95 -> 12.68 Grays/s on C3).
#ifndef PRT_T8_BFI
#define PRT_T8_BFI 1  // 0: v_cndmask_b32 selects (A/B builds: make EXTRA=-DPRT_T8_BFI=0)
#endif
#if PRT_T8_BFI
#define T8_SEL(M, A, B) (((A) & (M)) | ((B) & ~(M)))
#else
#define T8_SEL(M, A, B) ((M) ? (A) : (B))
#endif
#define T8_CHILD(J, NX, FX, NY, FY, NZ, FZ)                                                                          \
    {                                                                                                                \
        const float tnx = __builtin_fmaf((float)(((NX) >> (8 * J)) & 0xFFu), Ax, Bnx);                               \
        const float tny = __builtin_fmaf((float)(((NY) >> (8 * J)) & 0xFFu), Ay, Bny);                               \
        const float tnz = __builtin_fmaf((float)(((NZ) >> (8 * J)) & 0xFFu), Az, Bnz);                               \
        const float tfx = __builtin_fmaf((float)(((FX) >> (8 * J)) & 0xFFu), Ax, Bfx);                               \
        const float tfy = __builtin_fmaf((float)(((FY) >> (8 * J)) & 0xFFu), Ay, Bfy);                               \
        const float tfz = __builtin_fmaf((float)(((FZ) >> (8 * J)) & 0xFFu), Az, Bfz);                               \
        const float tn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, 0.0f));                     \
        const float tf = __builtin_fminf(__builtin_fminf(tfx, tfy), __builtin_fminf(tfz, tlimit));                   \
        const uint32_t cb = ((bits4 >> (8 * J)) & 0xFFu) << ((idx4 >> (8 * J)) & 0xFFu);                             \
        hitmask |= (tn <= tf) ? cb : 0u;                                                                             \
    }
#define T8_HALF(META4, NX, FX, NY, FY, NZ, FZ)                                                                       \
    {                                                                                                                \
        const uint32_t meta4 = (META4);                                                                              \
        const uint32_t in4 = ((meta4 & (meta4 << 1)) & 0x10101010u) >> 4; /* 1 in the bytes of internal children */  \
        const uint32_t idx4 = (meta4 ^ (octinv4 & ((in4 << 3) - in4))) & 0x1F1F1F1Fu;                                \
        const uint32_t bits4 = (meta4 >> 5) & 0x07070707u;                                                           \
        T8_CHILD(0, NX, FX, NY, FY, NZ, FZ)                                                                          \
        T8_CHILD(1, NX, FX, NY, FY, NZ, FZ)                                                                          \
        T8_CHILD(2, NX, FX, NY, FY, NZ, FZ)                                                                          \
        T8_CHILD(3, NX, FX, NY, FY, NZ, FZ)                                                                          \
    }
#define T8_BOXTEST()                                                                                                 \
    const uint32_t eim = w0.w;                                                                                       \
    const float Ax = __uint_as_float((eim & 0xFFu) << 23) * ix;                                                      \
    const float Ay = __uint_as_float(((eim >> 8) & 0xFFu) << 23) * iy;                                               \
    const float Az = __uint_as_float(((eim >> 16) & 0xFFu) << 23) * iz;                                              \
    const float px = __uint_as_float(w0.x), py = __uint_as_float(w0.y), pz = __uint_as_float(w0.z);                  \
    const float Bnx = __builtin_fmaf(px, ix, -anx), Bny = __builtin_fmaf(py, iy, -any), Bnz = __builtin_fmaf(pz, iz, -anz); \
    const float Bfx = __builtin_fmaf(px, ix, -afx), Bfy = __builtin_fmaf(py, iy, -afy), Bfz = __builtin_fmaf(pz, iz, -afz); \
    uint32_t mx = (uint32_t)((int32_t)__float_as_uint(ix) >> 31), my = (uint32_t)((int32_t)__float_as_uint(iy) >> 31), \
             mz = (uint32_t)((int32_t)__float_as_uint(iz) >> 31);                                                    \
    asm("" : "+v"(mx), "+v"(my), "+v"(mz)); /* opaque: keeps the and/or form from being folded back into selects */  \
    const uint32_t nxa = T8_SEL(mx, w3.z, w2.x), nxb = T8_SEL(mx, w3.w, w2.y), fxa = T8_SEL(mx, w2.x, w3.z), fxb = T8_SEL(mx, w2.y, w3.w); \
    const uint32_t nya = T8_SEL(my, w4.x, w2.z), nyb = T8_SEL(my, w4.y, w2.w), fya = T8_SEL(my, w2.z, w4.x), fyb = T8_SEL(my, w2.w, w4.y); \
    const uint32_t nza = T8_SEL(mz, w4.z, w3.x), nzb = T8_SEL(mz, w4.w, w3.y), fza = T8_SEL(mz, w3.x, w4.z), fzb = T8_SEL(mz, w3.y, w4.w); \
    uint32_t hitmask = 0u;                                                                                           \
    T8_HALF(w1.z, nxa, fxa, nya, fya, nza, fza)                                                                      \
    T8_HALF(w1.w, nxb, fxb, nyb, fyb, nzb, fzb)

#define T8_SENTINEL 0xFFFFFFFFu
template <int STACK_L, int WAVES, bool STATS, bool INST, bool LEAN = false, bool PRIM = false, bool PATH = false>
__global__ void __launch_bounds__(256, WAVES) k_traverse8_persistent(DevScene sc, const float4* __restrict__ ro,
                                                                    const float4* __restrict__ rd,
                                                                    uint32_t* __restrict__ hit,
                                                                    const float* __restrict__ hd2,
                                                                    const uint32_t* __restrict__ count_ptr,
                                                                    uint32_t* __restrict__ work,
                                                                    uint32_t* __restrict__ ovf, PrtTravTuning tune,
                                                                    unsigned long long* __restrict__ stats, PrtPrimary pr,
                                                                    PrtPathArgs pa) {
    // PRIM: the rays are compact primary rays (PrtPrimary): origin and direction are rebuilt from the path id
    // PATH (PrtPathArgs, prt_kernels.h): the work items are whole PATHS, not rays.  A lane whose walk is over is not
    // released: it waits (need_shade) until refill_min lanes of the wave wait or are idle, then those lanes run the shade
    // step together (advance_path: the body of k_shade, same draws in the same order) and walk the scattered ray next; a
    // segment that cannot hit a triangle is shaded on the spot, as the producers' classification decides it; idle lanes
    // take new paths from the launch's cursor and generate their primary ray (k_raygen's arithmetic).  One launch per
    // batch instead of 2 x max_depth + 1: a small batch's launches are dominated by their ramp and their longest rays
    // (C3, one sample: 5 traversal launches of ~150 us for ~30 us of work each), which this pays once.
    __shared__ uint2 s_stack[(STACK_L + 1) * 256];  // [entry][thread]; one row of slack above the top
    __shared__ unsigned long long s_key[256];       // per lane: best (d2 bits << 32 | prim) of the cooperative triangle tests
    __shared__ uint32_t s_slot[256];                // per lane: leaf-order slot of that best
    // per wave, between two triangle phases: up to T8_QGROUPS 8-byte GROUP items {first triangle slot, owner lane << 24 |
    // 24-bit triangle mask} appended by the node loop; the triangle phase takes them into registers and expands them IN
    // PLACE into up to T8_QCAP 4-byte PAIR items (owner lane << 26) | triangle slot, one per lane per round
    __shared__ uint32_t s_queue[4 * T8_QCAP];
    __shared__ uint32_t s_qn[8];                    // per wave: [w] group items appended since the last triangle phase
    __shared__ uint32_t s_iters[8];                 // [0..3] node-loop, [4..7] triangle-loop iterations per wave (STATS)
    __shared__ float s_wray[INST ? 6 * 256 : 1];    // INST: the lane's WORLD ray (origin, raw direction), [component][thread]
    // LEAN (5 waves per SIMD: <= 96 VGPRs): what only the triangle phase needs lives in LDS instead of registers: the
    // lane's ray (origin, unit direction) here, its best key in s_key and its best hit id in s_slot for its whole life
    __shared__ float s_lray[LEAN ? 6 * 256 : 1];
    // STEAL (LEAN instance, only while a wave drains, i.e. after the ray buffer is exhausted).  A launch ends with its
    // longest rays: C3's take ~130 node steps against a mean of 10, and a launch of 1 k rays lasts as long as one of
    // 260 k (tools/launch_floor.py: 170 vs 320 us), ~250 us per bounce during which the chip is almost idle.  Idle
    // lanes of a draining wave therefore take over pending subtrees of the wave's remaining rays: a thief copies the
    // ray's constants out of the donor's registers, takes the donor's BOTTOM stack entry (the oldest, usually largest
    // pending node group) and walks it as a helper of the ray's ROOT lane: its triangle tests resolve into the root's
    // key / slot (the queue item names the root as owner), so helper and root share one culling bound and no merge is
    // needed; the root writes its hit once its own walk is over and no helper of it is left.  The closest hit does
    // not depend on the order of the tests, so results stay bit-identical.
    // (No LDS for the bookkeeping: one more KB would cost the fifth block per CU.  Which roots still have helpers is a
    // wave-uniform 64-bit mask rebuilt from the helpers' k every outer iteration of a draining wave.)
    constexpr bool STEAL = LEAN && !INST;
    static_assert(!PATH || (!LEAN && !INST && !PRIM && !STATS), "the path instance is the plain one-level kernel");
    const uint32_t count = PATH ? pa.n_paths : *count_ptr;
    // PATH: the path a busy lane carries (k = its path id): direction as stored (the walk normalises it again, as the
    // reference's TransformNormal does), throughput, RNG state, segment index; need_shade: the walk of the current
    // segment is over, best.id is its closest hit
    f3 pd = mk3(0.f, 0.f, 1.f), pthr = mk3(0.f, 0.f, 0.f);
    uint32_t prng = 0u, pdepth = 0u;
    bool need_shade = false;
    // (tune.stack_cap: test hook that makes the stack look shorter, to exercise the overflow path)
    const int stack_cap = (tune.stack_cap != 0u && tune.stack_cap < (uint32_t)STACK_L) ? (int)tune.stack_cap : STACK_L;
    // Guided hand-out of the ray buffer, in granules of 64 rays.  The bulk goes out in chunks of tune.chunk rays (one
    // global atomic per chunk; coherent neighbours stay together), but the last tune.tail granules per resident wave go out
    // one at a time: otherwise a wave that grabs a full chunk just before the buffer runs dry works four more rounds
    // (~160 us on C3) while the rest of the chip idles, a fixed cost per launch that hurts most when the frame is split
    // over 8 GPUs.  Small launches consist of single granules only.  A wave's FIRST grab needs no atomic (grab number =
    // its index in the grid; 5120 simultaneous atomics on one word take ~60 us); the cursor counts the grabs after it.
    const uint32_t gran_per_chunk = tune.chunk >> 6;
    const uint32_t n_wv = gridDim.x * 4u;
    const uint32_t n_gran = (count + 63u) >> 6;
    // A launch with fewer than 8 granules per resident wave has no bulk at all: a quarter of its waves would start with a
    // four-granule chunk and still be on it when the rest of the buffer is gone (one-sample calls at 1080p, 2 M + 1.2 M
    // rays for 5,120 waves: traversal 764 -> 642 us per call; from ~12 granules per wave on chunks win again, and
    // single granules throughout cost 20-30 % at 4-64 samples per call: one atomic per 64 rays; TUNING.md.  Thinner
    // granules for launches smaller than the grid, 8-32 rays in every wave instead of 64 in a few, change nothing: such a
    // launch lasts as long as its longest ray, whatever shares the wave with it)
    const uint32_t tail_want = (n_gran < n_wv * 8u || n_gran < n_wv * tune.static_small) ? n_gran : n_wv * tune.tail;
    const uint32_t n_bulk = (n_gran - (tail_want < n_gran ? tail_want : n_gran)) / gran_per_chunk;  // chunks
    // Big launches (>= tune.big_min = 96 granules per resident wave; e.g. the first two bounces of a 256-sample batch at 1080p,
    // 190 M and 318 M rays) hand out the FRONT of the bulk in chunks of tune.big chunks each: at 256 rays per grab the 318 M-ray launch
    // asks the cursor 1.24 M times in 13.7 ms = 90 grabs per us, which is all a counter word executes (~87 / us,
    // tools/atomic_rate.hip); any launch that runs at > 20 G rays/s is there.  The last 32 granules per wave stay ordinary chunks,
    // so the end of the launch is balanced as before.  C3 +1.7 % at 256 samples per batch, +1.2 % at 64, +1 % at 32; C4 +2.3 %.
    const uint32_t big = (tune.big > 1u && n_gran >= n_wv * tune.big_min) ? tune.big : 1u;
    const uint32_t mid_keep = n_wv * tune.big_keep / gran_per_chunk;  // ordinary chunks kept for the end of the bulk (32 granules per wave)
    const uint32_t n_big = big > 1u ? (n_bulk - (mid_keep < n_bulk ? mid_keep : n_bulk)) / big : 0u;  // grabs of `big` chunks each
    const uint32_t n_grabs = n_big + (n_bulk - n_big * big) + (n_gran - n_bulk * gran_per_chunk);
    bool first_grab = true;  // wave-uniform
    uint32_t my_grabs = 0u;  // wave-uniform
    // Launches of single granules only (fewer than tune.static_small = 8 per resident wave) do not use the cursor at all: wave w
    // takes granules w, w + n_wv, w + 2 n_wv, ...  A grab through the cursor stalls the whole wave for the atomic's round trip,
    // and a one-sample call's two big launches made 32 k + 19 k of them in ~170 us each (one-sample calls 1,350 -> 1,510 per
    // second, 2 / 4 samples per call +11 / +6 %, 8-32 samples +1 to +3 % through their late bounces; with a threshold of 32 and
    // more the dealt-out launches of 16+ samples lose 2-5 % to imbalance; TUNING.md)
    const bool static_small = n_gran < n_wv * tune.static_small;
    const uint32_t my_xcd = xcc_id();
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = lane_id();
    const uint32_t wv = tid >> 6;
    const uint32_t wbase = tid & ~63u;
    if (tid < 8) {
        s_iters[tid] = 0;
        s_qn[tid] = 0u;
    }
    __syncthreads();
    uint32_t* const queue = &s_queue[wv * T8_QCAP];
    uint32_t k = 0xFFFFFFFFu;
    f3 o = mk3(0.f, 0.f, 0.f), ld = mk3(0.f, 0.f, 1.f);
    float ix = 1.f, iy = 1.f, iz = 1.f, anx = 0.f, any = 0.f, anz = 0.f, afx = 0.f, afy = 0.f, afz = 0.f, pad = 0.f, tlimit = 0.f;
    uint32_t octinv4 = 0x07070707u;  // (7 - direction octant) in every byte
    uint32_t gx = 0u, gy = 0u;    // current node group
    bool pending = false;         // this lane has items in the wave's queue
    // INST only: level (0 = top-level tree, world ray; 1 = inside an instance, local ray), the instance, the pending
    // instance hits of the current top-level node, the culling-bound conversion (world distance -> local parameter)
    bool in_blas = false, stall = false;
    uint32_t inst = 0u, ipb = 0u, ipm = 0u;
    float lscale = 1.0f, padw4 = 0.0f;
    int sp = 0;
    bool overflow = false;
    Closest best;
    best.d2 = 3.402823466e+38f;
    best.id = HIT_MISS;
    best.prim = 0xFFFFFFFFu;
    uint32_t n_nodes = 0, n_tris = 0, max_sp = 0;
    uint32_t cur = 0, cur_end = 0;  // wave-uniform: this wave's current chunk [cur, cur_end)
    bool exhausted = false;         // wave-uniform
    // a triangle phase starts once this many triangle groups are queued; at most 63 more arrive with the step that reaches it
    const uint32_t tri_min = tune.tri_min < T8_QGROUPS - 64u ? tune.tri_min : T8_QGROUPS - 64u;
    // STATS: shader cycles (s_memtime) this wave spent in the three sections of an outer iteration
    unsigned long long cyc_refill = 0, cyc_node = 0, cyc_tri = 0, t_mark = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
    // launch timeline (PRT_TIMELINE_WORDS): s_memrealtime is the constant 100 MHz clock all XCDs share (s_memtime, used
    // for the phase split, counts each XCD's own shader clock)
    const unsigned long long t_start = STATS ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned long long t_exh = 0ull;  // STATS: when this wave found the ray buffer exhausted
    uint32_t ray_steps = 0u, ray_steps_max = 0u;  // STATS: node steps of the lane's current ray / of its longest ray
    // Exit condition every wave reaches: the loop ends when the ray buffer is exhausted and the wave's lanes are
    // idle; every outer iteration makes progress (a node step, a triangle phase or a refill); the iteration cap is
    // a watchdog that turns a would-be hang into an error flag the host reports.
    for (uint32_t guard = 0;; ++guard) {
        if (guard > (1u << 22)) {
            if (lane == 0) atomicOr(work + 256, 1u);
            break;
        }
        unsigned long long helped = 0ull;  // STEAL: root lanes that have helpers at work (wave-uniform)
        if (STEAL && exhausted) {  // (all lanes are active here; two OR scans instead of one readlane per helper: a draining
            // wave has up to 63 of them and pays this in every outer iteration)
            const bool helper = k >= 0xFFFFFF00u && k != 0xFFFFFFFFu;
            if (__ballot(helper) != 0ull) {
                const uint32_t r = k & 63u;
                const uint32_t lo = wave_scan_or((helper && r < 32u) ? 1u << r : 0u);
                const uint32_t hi = wave_scan_or((helper && r >= 32u) ? 1u << (r - 32u) : 0u);
                helped = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)hi, 63) << 32) |
                         (uint32_t)__builtin_amdgcn_readlane((int)lo, 63);
            }
        }
        // a lane is released only when nothing of its ray is left in the queue (the testers read the owner's ray)
        if (k != 0xFFFFFFFFu && !pending && !(INST && (in_blas || ipm != 0u))) {
            if (overflow) {
                if (PATH) {  // (the host only uses this instance for trees its stack holds: an error prt_synchronize reports)
                    atomicOr(work + 256, 2u);
                    need_shade = true;
                } else if (INST) {
                    atomicOr(work + 256, 2u);  // no fallback for two-level scenes: prt_synchronize reports it
                } else {
                    const uint32_t j = atomicAdd(ovf, 1u);  // re-done from scratch by the spill-capable 4-wide instance
                    if (j < PRT_OVF_CAP)
                        ovf[1u + j] = k;
                    else
                        atomicOr(work + 256, 2u);
                }
                overflow = false;
                if (!PATH) k = 0xFFFFFFFFu;
            } else if (!(gy > 0x00FFFFFFu) && sp == 0) {
                if (PATH) {
                    need_shade = true;  // the lane keeps its path; the shade step below goes on with it
                } else if (STEAL && k >= 0xFFFFFF00u) {  // a helper (k = 0xFFFFFF00 | root lane): its subtree is done
                    k = 0xFFFFFFFFu;
                } else if (!STEAL || ((helped >> lane) & 1ull) == 0ull) {
                    st_stream(&hit[k], LEAN ? ((volatile uint32_t*)s_slot)[tid] : best.id);
#ifdef PRT_PROBE_REBOUND  // diagnostic build: leave the final distance where the next launch takes its initial bound from
                    ((float*)hd2)[k] = LEAN ? __uint_as_float((uint32_t)(((volatile unsigned long long*)s_key)[tid] >> 32)) : best.d2;
#endif
                    k = 0xFFFFFFFFu;
                }
            }
        }
        // (up to three rounds of pairing per outer iteration: a thief that received a group with several pending children, or a
        // donor with more stack entries, gives again at once; one-sample calls 1,303 -> 1,360 per second, no more from 4 or 6)
#ifndef PRT_STEAL_PASSES
#define PRT_STEAL_PASSES 3
#endif
        if (STEAL && exhausted && tune.steal != 0u)  // wave-uniform
        for (int steal_pass = 0; steal_pass < PRT_STEAL_PASSES; ++steal_pass) {
            // a donor gives its bottom stack entry, or, with an empty stack, the far half of its current group's pending children
            const uint32_t gh = gy >> 24;
            const bool can_give = k != 0xFFFFFFFFu && (sp > 0 || (gh & (gh - 1u)) != 0u);
            const unsigned long long free_m = __ballot(k == 0xFFFFFFFFu);
            const unsigned long long give_m = __ballot(can_give);
            const uint32_t n_free = (uint32_t)__popcll(free_m), n_give = (uint32_t)__popcll(give_m);
            if (n_free >= tune.steal && n_give != 0u) {
                // the queue is empty here (every triangle phase tests all of it): its first words pair thieves and donors
                const uint32_t n_pair = n_free < n_give ? n_free : n_give;
                const unsigned long long below = (1ull << lane) - 1ull;
                const uint32_t give_rank = (uint32_t)__popcll(give_m & below), free_rank = (uint32_t)__popcll(free_m & below);
                const bool donor = can_give && give_rank < n_pair;
                const bool thief = k == 0xFFFFFFFFu && free_rank < n_pair;
                if (donor) {
                    queue[give_rank] = lane;
                    if (sp == 0) {  // split the current group: the lowest-priority half of its hits goes through stack row 0
                        uint32_t far = 0u, rest = gh;
                        for (uint32_t i = (uint32_t)__popc(gh) >> 1; i != 0u; --i) {
                            far |= rest & (0u - rest);
                            rest &= rest - 1u;
                        }
                        s_stack[tid] = make_uint2(gx, (far << 24) | (gy & 0x00FFFFFFu));
                        gy = (rest << 24) | (gy & 0x00FFFFFFu);
                        sp = 1;  // (taken off again right below)
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                const uint32_t src = thief ? ((volatile uint32_t*)queue)[free_rank] : lane;
                const unsigned long long e64 = ((volatile unsigned long long*)s_stack)[wbase + src];  // the donor's bottom entry
                const uint2 e = make_uint2((uint32_t)e64, (uint32_t)(e64 >> 32));
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (donor) {
                    for (int i = 1; i < sp; ++i) s_stack[(i - 1) * 256 + tid] = s_stack[i * 256 + tid];
                    --sp;
                }
                // the ray's constants come out of the donor's registers (every lane takes part in the shuffles)
                const uint32_t k_src = (uint32_t)__shfl((int)k, (int)src, 64);
                const float t_ix = __shfl(ix, (int)src, 64), t_iy = __shfl(iy, (int)src, 64), t_iz = __shfl(iz, (int)src, 64);
                const float t_anx = __shfl(anx, (int)src, 64), t_any = __shfl(any, (int)src, 64), t_anz = __shfl(anz, (int)src, 64);
                const float t_afx = __shfl(afx, (int)src, 64), t_afy = __shfl(afy, (int)src, 64), t_afz = __shfl(afz, (int)src, 64);
                const float t_pad = __shfl(pad, (int)src, 64);
                const uint32_t t_oct = (uint32_t)__shfl((int)octinv4, (int)src, 64);
                if (thief) {
                    const uint32_t root = k_src >= 0xFFFFFF00u ? (k_src & 63u) : src;
                    k = 0xFFFFFF00u | root;
                    ix = t_ix; iy = t_iy; iz = t_iz;
                    anx = t_anx; any = t_any; anz = t_anz;
                    afx = t_afx; afy = t_afy; afz = t_afz;
                    pad = t_pad;
                    octinv4 = t_oct;
                    gx = e.x;
                    gy = e.y;
                    sp = 0;
                    tlimit = limit_from_d2(__uint_as_float((uint32_t)(((volatile unsigned long long*)s_key)[wbase + root] >> 32)), pad);
                }
            } else {
                break;
            }
        }
        const bool idle = k == 0xFFFFFFFFu;
        const unsigned long long idle_mask = __ballot(idle);
        const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
        bool do_sw = false;  // wave-uniform
        if (INST) {
            // Level switches (enter the next instance hit of a top-level node / leave an exhausted instance) cost ~150
            // instructions plus the instance record's gather.  Lanes that need one wait (stall) and the wave does them
            // together, under the same rule as the refill: when refill_min lanes are idle or waiting.  No lane has
            // items in the queue here (the triangle phase of the previous iteration emptied it), so the queued
            // triangles never see a ray other than the one they were found with.
            const uint32_t n_sw = (uint32_t)__popcll(__ballot(stall));
            do_sw = n_sw != 0u && n_idle + n_sw >= tune.refill_min;
            if (do_sw && stall) {
                const int spr = sp > 0 ? sp - 1 : 0;
                const bool has = gy > 0x00FFFFFFu;
                const bool enter = !in_blas;
                f3 dw = mk3(s_wray[3 * 256 + tid], s_wray[4 * 256 + tid], s_wray[5 * 256 + tid]);
                o = mk3(s_wray[0 * 256 + tid], s_wray[1 * 256 + tid], s_wray[2 * 256 + tid]);
                const float pad_w = sc.pad * (__builtin_fabsf(o.x) + __builtin_fabsf(o.y) + __builtin_fabsf(o.z) + sc.extent);
                // leaving an instance while the same top-level node has more instance hits goes straight into the
                // next one (the sentinel stays): no world-ray round trip in between
                const bool next = ipm != 0u;
                if (next) {
                    inst = sc.tlas_inst[ipb + (uint32_t)__builtin_ctz(ipm)];
                    ipm &= ipm - 1u;
                    if (enter) {
                        if (has) {  // the remaining top-level siblings wait below the sentinel
                            s_stack[sp * 256 + tid] = make_uint2(gx, gy);
                            ++sp;
                        }
                        s_stack[(sp <= STACK_L ? sp : STACK_L) * 256 + tid] = make_uint2(T8_SENTINEL, 0xFF000000u);
                        ++sp;
                    }
                    const DevInstance& I = sc.insts[inst];
                    o = transform_point(I.inv, o);      // primitive.cpp:29
                    ld = transform_normal(I.mat, dw);   // primitive.cpp:30
                    pad = sc.pad * (__builtin_fabsf(o.x) + __builtin_fabsf(o.y) + __builtin_fabsf(o.z) + I.extent);
                    lscale = I.inv_scale * 1.000001f;
                    padw4 = 4.0f * pad_w;
                    gx = I.root;
                    in_blas = true;
                } else {
                    sp = spr;  // pop the sentinel
                    ld = normalize3(dw);
                    pad = pad_w;
                    lscale = 1.0f;
                    padw4 = 0.0f;
                    gx = 0u;
                    in_blas = false;
                }
                ix = __builtin_amdgcn_rcpf(__builtin_fabsf(ld.x) < 1e-30f ? __builtin_copysignf(1e-30f, ld.x) : ld.x);
                iy = __builtin_amdgcn_rcpf(__builtin_fabsf(ld.y) < 1e-30f ? __builtin_copysignf(1e-30f, ld.y) : ld.y);
                iz = __builtin_amdgcn_rcpf(__builtin_fabsf(ld.z) < 1e-30f ? __builtin_copysignf(1e-30f, ld.z) : ld.z);
                const bool nx = ix < 0.0f, ny = iy < 0.0f, nz = iz < 0.0f;
                anx = (nx ? o.x - pad : o.x + pad) * ix; afx = (nx ? o.x + pad : o.x - pad) * ix;
                any = (ny ? o.y - pad : o.y + pad) * iy; afy = (ny ? o.y + pad : o.y - pad) * iy;
                anz = (nz ? o.z - pad : o.z + pad) * iz; afz = (nz ? o.z + pad : o.z - pad) * iz;
                octinv4 = (7u - ((nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u))) * 0x01010101u;
                tlimit = (limit_from_d2(best.d2, 0.0f) + padw4) * lscale + 4.0f * pad;
                gy = next ? (1u << (24u + (octinv4 & 7u))) : 0u;  // the instance's root "group" / nothing pending
                if (sp > STACK_L) {  // no room left: give the ray up (an error the host reports)
                    overflow = true;
                    gy = 0u;
                    sp = 0;
                    ipm = 0u;
                    in_blas = false;
                }
                stall = false;
            }
        }
        if (PATH) {
            // The shade step of the waiting lanes and new paths for the idle ones, together: when refill_min lanes wait or
            // are idle, or when no lane of the wave has anything left to walk or to test.
            const unsigned long long shade_m = __ballot(need_shade);
            const uint32_t n_wait = n_idle + (uint32_t)__popcll(shade_m);
            const bool others_busy = __ballot(k != 0xFFFFFFFFu && !need_shade) != 0ull;
            if ((shade_m != 0ull || (!exhausted && n_idle != 0u)) && (n_wait >= tune.refill_min || !others_busy)) {
                bool fresh = false;
                if (!exhausted && n_idle != 0u) {
                    if (cur == cur_end) {  // grab the next chunk / granule of path ids (as the ray hand-out above)
                        uint32_t c = blockIdx.x * 4u + wv;
                        if (!first_grab) {
                            if (n_grabs <= n_wv) c = n_grabs;  // (every grab went out with the waves' first ones)
                            else if (lane == 0) c = atomicAdd(work, 1u) + n_wv;
                            c = (uint32_t)__shfl((int)c, 0, 64);
                        }
                        first_grab = false;
                        if (c >= n_grabs) {
                            exhausted = true;
                        } else {
                            // grab c: [0, n_big) big grabs, then the ordinary chunks, then the single granules
                            const uint32_t cb = c - n_big + n_big * big;  // (c >= n_big) the grab's index counted in ordinary chunks
                            const uint32_t g0 = c < n_big ? c * big * gran_per_chunk : cb < n_bulk ? cb * gran_per_chunk : n_bulk * gran_per_chunk + (cb - n_bulk);
                            const uint32_t g1 = c < n_big ? g0 + big * gran_per_chunk : cb < n_bulk ? g0 + gran_per_chunk : g0 + 1u;
                            cur = g0 << 6;
                            cur_end = (g1 << 6) < count ? (g1 << 6) : count;
                        }
                    }
                    if (!exhausted) {
                        const uint32_t qi = cur + (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
                        fresh = idle && qi < cur_end;
                        if (fresh) k = qi;
                        cur = (cur + n_idle < cur_end) ? cur + n_idle : cur_end;
                    }
                }
                // r: 0 = the path has ended (rad written), 1 = walk (o, pd) from its analytic hit (id0, d2_0),
                //    2 = shade hit `id` of the segment (o, pd) next
                int r = 0;
                uint32_t id = HIT_MISS, id0 = HIT_MISS;
                float d2_0 = 3.402823466e+38f;
                if (fresh) {  // k_raygen's arithmetic for path k = sample * n_pix_local + local pixel
                    const uint32_t smp = k / pa.tm.n_pix_local, pl = k - smp * pa.tm.n_pix_local;
                    uint32_t px, py;
                    if (!tile_pixel(pa.tm, pl, px, py)) {  // a partial tile's pixel outside the image: no path
                        pa.rad[k] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));
                    } else {
                        prng = path_seed(py * pa.tm.W + px, pa.first_sample + smp, pa.seed);
                        float fx = (float)px + 0.5f, fy = (float)py + 0.5f;  // pixel centre (cpu/renderer.cpp:45)
                        if (pa.sp.jitter) {  // (x + u1, y + u2): the path's first two draws (optix/device_programs.cu:172-173)
                            const float u1 = rnd01(prng);
                            const float u2 = rnd01(prng);
                            fx = (float)px + u1;
                            fy = (float)py + u2;
                        }
                        camera_ray(pa.cam, fx, fy, o, pd);
                        pthr = mk3(1.f, 1.f, 1.f);
                        pdepth = 0u;
                        r = classify_ray<false, 256>(sc, o, pd, id0, d2_0) ? 1 : 2;
                        id = id0;
                    }
                }
                if (need_shade) {
                    r = 2;
                    id = best.id;
                    need_shade = false;
                }
                while (r == 2) {  // (a segment that cannot hit a triangle is shaded right away: its hit is the analytic one)
                    r = advance_path<1, false, false, 256>(sc, id, o, pd, pthr, prng, pdepth, pa.max_depth, pa.sp, &pa.rad[k], id0, d2_0);
                    id = id0;
                }
                if (r == 1) {  // start the walk of (o, pd): the set-up of a refill
                    ld = normalize3(pd);  // TransformNormal(identity, d), primitive.cpp:30
                    pad = sc.pad * (__builtin_fabsf(o.x) + __builtin_fabsf(o.y) + __builtin_fabsf(o.z) + sc.extent);
                    ix = __builtin_amdgcn_rcpf(__builtin_fabsf(ld.x) < 1e-30f ? __builtin_copysignf(1e-30f, ld.x) : ld.x);
                    iy = __builtin_amdgcn_rcpf(__builtin_fabsf(ld.y) < 1e-30f ? __builtin_copysignf(1e-30f, ld.y) : ld.y);
                    iz = __builtin_amdgcn_rcpf(__builtin_fabsf(ld.z) < 1e-30f ? __builtin_copysignf(1e-30f, ld.z) : ld.z);
                    const bool nx = ix < 0.0f, ny = iy < 0.0f, nz = iz < 0.0f;
                    anx = (nx ? o.x - pad : o.x + pad) * ix; afx = (nx ? o.x + pad : o.x - pad) * ix;
                    any = (ny ? o.y - pad : o.y + pad) * iy; afy = (ny ? o.y + pad : o.y - pad) * iy;
                    anz = (nz ? o.z - pad : o.z + pad) * iz; afz = (nz ? o.z + pad : o.z - pad) * iz;
                    octinv4 = (7u - ((nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u))) * 0x01010101u;
                    best.id = id0;
                    best.prim = id0;  // analytic index, or 0xFFFFFFFF for a miss
                    best.d2 = d2_0;
                    tlimit = limit_from_d2(best.d2, pad);
                    gx = 0u;  // the root "group": node 0, one pending hit that decodes to slot 0
                    gy = 1u << (24u + (octinv4 & 7u));
                    sp = 0;
                } else if ((fresh || ((shade_m >> lane) & 1ull)) && r == 0) {
                    k = 0xFFFFFFFFu;  // the path is over: the lane is idle again
                }
            }
        }
        if (!PATH && !exhausted && (n_idle >= tune.refill_min || (INST && do_sw && n_idle != 0u))) {
            if (cur == cur_end) {  // grab the next chunk / granule
                uint32_t c = blockIdx.x * 4u + wv;
                if (!first_grab) {
                    if (n_grabs <= n_wv) c = n_grabs;  // (every grab went out with the waves' first ones)
                    else if (static_small) c += my_grabs * n_wv;  // (a small launch: wave w takes granules w, w + n_wv, ...: no cursor)
                    else if (lane == 0) c = atomicAdd(work, 1u) + n_wv;
                    c = (uint32_t)__shfl((int)c, 0, 64);
                }
                first_grab = false;
                ++my_grabs;
                if (c >= n_grabs) {
                    exhausted = true;
                    if (STATS) t_exh = __builtin_amdgcn_s_memrealtime();
                } else {
                    // grab c: [0, n_big) big grabs, then the ordinary chunks, then the single granules
                    const uint32_t cb = c - n_big + n_big * big;  // (c >= n_big) the grab's index counted in ordinary chunks
                    const uint32_t g0 = c < n_big ? c * big * gran_per_chunk : cb < n_bulk ? cb * gran_per_chunk : n_bulk * gran_per_chunk + (cb - n_bulk);
                    const uint32_t g1 = c < n_big ? g0 + big * gran_per_chunk : cb < n_bulk ? g0 + gran_per_chunk : g0 + 1u;
                    cur = g0 << 6;
                    cur_end = (g1 << 6) < count ? (g1 << 6) : count;
                }
            }
            if (!exhausted) {
                const uint32_t qi_seq = cur + (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
                if (idle && qi_seq < cur_end) {
                    const uint32_t qi = (!PRIM && tune.perm) ? tune.perm[qi_seq] : qi_seq;  // (sort_rays: a measurement aid)
                    uint32_t hid = PRIM ? HIT_MISS : ld_stream(&hit[qi]);
                    if (hid != HIT_DEAD) {
                        float4 O, D;
                        float hd2_0 = 0.0f;
                        if (PRIM) {
                            f3 po, pd;
                            uint32_t pixel, sample, lp;
                            primary_ray(pr, ld_stream(&pr.pid[qi]), po, pd, pixel, sample, &lp);
                            const float4 E = pr.pix[pr.n_pix_local + lp];  // the pixel's analytic hit (k_raygen)
                            hid = __float_as_uint(E.y);
                            hd2_0 = E.z;
                            // every primary ray starts at the camera, a kernel argument: left visible, the compiler hoists
                            // everything that depends on the origin alone (o +- pad, |o|_1, ...) out of the persistent loop
                            // and keeps it in VGPRs for the kernel's life, 8 of which it then spills (36 B of scratch, 8
                            // scratch loads per refill); opaque, the same values cost 8 VALU instructions per refill
#ifndef PRT_NO_OPAQUE_ORIGIN  // (A/B builds)
                            asm volatile("" : "+v"(po.x), "+v"(po.y), "+v"(po.z));
#endif
                            O = make_float4(po.x, po.y, po.z, 0.f);
                            D = make_float4(pd.x, pd.y, pd.z, 0.f);
                        } else {
                            O = ld_stream(&ro[qi]);
                            D = ld_stream(&rd[qi]);
                        }
                        o = mk3(O.x, O.y, O.z);
                        ld = normalize3(mk3(D.x, D.y, D.z));  // TransformNormal(identity, d), primitive.cpp:30
                        pad = sc.pad * (__builtin_fabsf(o.x) + __builtin_fabsf(o.y) + __builtin_fabsf(o.z) + sc.extent);
                        ix = __builtin_amdgcn_rcpf(__builtin_fabsf(ld.x) < 1e-30f ? __builtin_copysignf(1e-30f, ld.x) : ld.x);
                        iy = __builtin_amdgcn_rcpf(__builtin_fabsf(ld.y) < 1e-30f ? __builtin_copysignf(1e-30f, ld.y) : ld.y);
                        iz = __builtin_amdgcn_rcpf(__builtin_fabsf(ld.z) < 1e-30f ? __builtin_copysignf(1e-30f, ld.z) : ld.z);
                        // (v_rcp_f32, 1 ulp, instead of IEEE divisions: these reciprocals only feed the box tests, whose
                        // pad of 2^-18 of the coordinates' magnitude is 32x the 2^-23 that costs; the triangle tests use the
                        // ray itself)
                        // near planes move out by -pad, far planes by +pad (the near plane is the box's max plane
                        // on an axis the ray travels along negatively):  t = plane * inv - (o +- pad) * inv
                        const bool nx = ix < 0.0f, ny = iy < 0.0f, nz = iz < 0.0f;
                        anx = (nx ? o.x - pad : o.x + pad) * ix; afx = (nx ? o.x + pad : o.x - pad) * ix;
                        any = (ny ? o.y - pad : o.y + pad) * iy; afy = (ny ? o.y + pad : o.y - pad) * iy;
                        anz = (nz ? o.z - pad : o.z + pad) * iz; afz = (nz ? o.z + pad : o.z - pad) * iz;
                        octinv4 = (7u - ((nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u))) * 0x01010101u;
                        best.id = hid;
                        best.prim = hid;  // analytic index, or 0xFFFFFFFF for a miss
                        best.d2 = PRIM ? hd2_0 : ld_stream(&hd2[qi]);
                        tlimit = limit_from_d2(best.d2, pad);
                        if (LEAN) {
                            s_lray[0 * 256 + tid] = o.x; s_lray[1 * 256 + tid] = o.y; s_lray[2 * 256 + tid] = o.z;
                            s_lray[3 * 256 + tid] = ld.x; s_lray[4 * 256 + tid] = ld.y; s_lray[5 * 256 + tid] = ld.z;
                            s_key[tid] = ((unsigned long long)__float_as_uint(best.d2) << 32) | (hid == HIT_MISS ? 0u : hid);
                            s_slot[tid] = hid;
                        }
                        k = qi;
                        if (STATS) ray_steps = 0u;
                        gx = 0u;  // the root "group": node 0, one pending hit that decodes to slot 0
                        gy = 1u << (24u + (octinv4 & 7u));
                        sp = 0;
                        if (INST) {
                            in_blas = false;
                            stall = false;
                            ipm = 0u;
                            lscale = 1.0f;
                            padw4 = 0.0f;
                            s_wray[0 * 256 + tid] = O.x; s_wray[1 * 256 + tid] = O.y; s_wray[2 * 256 + tid] = O.z;
                            s_wray[3 * 256 + tid] = D.x; s_wray[4 * 256 + tid] = D.y; s_wray[5 * 256 + tid] = D.z;
                        }
                    }
                }
                cur = (cur + n_idle < cur_end) ? cur + n_idle : cur_end;
            }
        }
        if (__ballot(k != 0xFFFFFFFFu) == 0ull) {
            if (exhausted) break;
            continue;
        }
        if (STATS) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            cyc_refill += t - t_mark;
            t_mark = t;
        }
        // ---- phase 1: node steps.  Triangles found go straight to the wave's queue and the lane keeps walking;
        // leave when at most exit_max lanes can still walk or enough work for a triangle phase has piled up ----
        bool walk = (k != 0xFFFFFFFFu) && ((gy > 0x00FFFFFFu) || sp > 0 || (INST && ipm != 0u)) && !(INST && stall);
        // Triangle groups queued by this run of the node loop.  The queue is empty on entry (a triangle phase tests all of
        // it), every lane still in the loop adds the same ballot counts, and a lane that has left the loop does not come
        // back before the next phase: the lanes in the loop agree on it, so a slot is base + rank in the ballot, without
        // an atomic and without a per-triangle loop in the node step.
        uint32_t q_groups = 0u;
        while (walk) {
            uint32_t tm = 0u, tb = 0u;
            {
                // current group: G while it has pending internal hits, else the top of the stack (read unconditionally)
                const int spr = sp > 0 ? sp - 1 : 0;
                const uint2 top = s_stack[spr * 256 + tid];
                const bool has = gy > 0x00FFFFFFu;
                bool level_switch = false;
                if (INST) {
                    const bool enter = !in_blas && ipm != 0u;               // instance hits of the last top-level node first
                    const bool leave = in_blas && !has && top.x == T8_SENTINEL;  // the instance's tree is exhausted
                    if (enter || leave) {  // level switches are done by the whole wave together, next to the refill
                        stall = true;
                        walk = false;
                        level_switch = true;
                    }
                }
                if (!level_switch) {
                    const uint32_t cx = has ? gx : top.x;
                    uint32_t cy = has ? gy : top.y;
                    sp = has ? sp : spr;
                    const uint32_t bit = 31u - (uint32_t)__builtin_clz(cy);  // highest pending hit: bits 24..31
                    cy &= ~(1u << bit);
                    s_stack[sp * 256 + tid] = make_uint2(cx, cy);  // the remaining siblings (kept only if there are any)
                    sp += (cy > 0x00FFFFFFu) ? 1 : 0;
                    const uint32_t slot = (bit - 24u) ^ (octinv4 & 7u);
                    const uint32_t idx = cx + (uint32_t)__popc(cy & ((1u << slot) - 1u));  // bits 0..7 of cy: imask
                    const uint4* nb = sc.nodes8 + (size_t)sc.node_stride * idx;
                    const uint4 w0 = nb[0], w1 = nb[1], w2 = nb[2], w3 = nb[3], w4 = nb[4];
                    if (STATS) {
                        ++n_nodes;
                        ++ray_steps;
                        if (ray_steps > ray_steps_max) ray_steps_max = ray_steps;
                        if ((int)lane == __ffsll((long long)__ballot(true)) - 1) ++s_iters[wv];
                        if ((uint32_t)sp > max_sp) max_sp = (uint32_t)sp;
                    }
                    T8_BOXTEST()
                    gx = w1.x;
                    gy = (hitmask & 0xFF000000u) | (eim >> 24);
                    tm = hitmask & 0x00FFFFFFu;
                    tb = w1.y;
                    if (sp > stack_cap) {  // give the ray up; it is re-traversed through the overflow list
                        overflow = true;
                        gy = 0u;
                        sp = 0;
                        tm = 0u;
                        if (INST) {
                            ipm = 0u;
                            in_blas = false;
                        }
                    } else if (INST && !in_blas) {  // top-level node: its "triangles" are instances, entered one by one
                        ipb = tb;
                        ipm = tm;
                        tm = 0u;
                    }
                    walk = (gy > 0x00FFFFFFu) || sp > 0 || (INST && ipm != 0u);
                }
            }
            const unsigned long long qm = __ballot(tm != 0u);
            if (tm != 0u) {
                const uint32_t own = (STEAL && k >= 0xFFFFFF00u) ? (k & 63u) : lane;  // a helper's tests belong to its root
                const uint32_t pos = q_groups + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(qm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)qm, 0u));
                ((uint2*)queue)[pos] = make_uint2(tb, (own << 24) | tm);
                s_qn[wv] = q_groups + (uint32_t)__popcll(qm);  // (the same value from every queueing lane)
                pending = true;
            }
            q_groups += (uint32_t)__popcll(qm);
            if ((uint32_t)__popcll(__ballot(walk)) <= tune.exit_max || q_groups >= tri_min) break;
        }
        if (STATS) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            cyc_node += t - t_mark;
            t_mark = t;
        }
        // ---- phase 2: the queued triangle groups become (ray, triangle) pairs, one pair per lane per round ----
        {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            const uint32_t G = ((volatile uint32_t*)s_qn)[wv];  // group items in the queue (wave-uniform)
            if (G != 0u) {
                // a miss is encoded with prim 0 so that a candidate with d2 == FLT_MAX can never win (primitive.cpp:44)
                const unsigned long long key_best =
                    LEAN ? 0ull : ((unsigned long long)__float_as_uint(best.d2) << 32) | (best.id == HIT_MISS ? 0u : best.prim);
                if (!LEAN) s_key[tid] = key_best;
                if (lane == 0) s_qn[wv] = 0u;
                // Sets of 64 group items, one per lane, taken into registers; after that the queue memory holds pair items.
                // While a second set waits (G > 64: rare, the node loop stops at tri_min groups) the pairs of the first one
                // stay in the lower half of the queue's memory, below the second set's items.
                for (uint32_t g0 = 0u; g0 < G; g0 += 64u) {  // wave-uniform, at most two trips
                const uint32_t cap = (G - g0 > 64u) ? T8_QCAP / 2u : T8_QCAP;
                const unsigned long long ga = g0 + lane < G ? ((volatile unsigned long long*)queue)[g0 + lane] : 0ull;
                const uint32_t base_a = (uint32_t)ga, own_a = (uint32_t)(ga >> 56);
                uint32_t ma = (uint32_t)(ga >> 32) & 0x00FFFFFFu;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                // passes of at most `cap` pairs (one pass unless the set holds more: 2-3 triangles per group on average);
                // groups are expanded in queue order, whole groups only (a group has at most 24 triangles: progress)
                while (__ballot(ma != 0u) != 0ull) {  // wave-uniform
                    const uint32_t ca = (uint32_t)__popc(ma);
                    const uint32_t ia = wave_scan_add(ca);
                    const bool fit_a = ia <= cap;  // true for a prefix of the lanes
                    const uint32_t nfa = (uint32_t)__popcll(__ballot(fit_a));
                    const uint32_t T = (uint32_t)__shfl((int)ia, (int)(nfa - 1u), 64);  // (nfa >= 1: the first group fits)
                    if (fit_a && ca) {
                        uint32_t qp = ia - ca;
                        for (uint32_t m = ma; m; m &= m - 1u) queue[qp++] = (own_a << 26) | (base_a + (uint32_t)__builtin_ctz(m));
                        ma = 0u;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    for (uint32_t base = 0; base < T; base += 64u) {  // wave-uniform trip count
                        const uint32_t kq = base + lane;
                        const bool act = kq < T;
                        const uint32_t item = ((volatile uint32_t*)queue)[act ? kq : 0u];
                        const uint32_t owner = act ? (item >> 26) : lane;
                        const uint32_t slot = item & 0x03FFFFFFu;
                        f3 qo, qd;
                        if (LEAN) {
                            const uint32_t oc = wbase + owner;
                            qo = mk3(s_lray[0 * 256 + oc], s_lray[1 * 256 + oc], s_lray[2 * 256 + oc]);
                            qd = mk3(s_lray[3 * 256 + oc], s_lray[4 * 256 + oc], s_lray[5 * 256 + oc]);
                        } else {
                            qo = mk3(__shfl(o.x, (int)owner, 64), __shfl(o.y, (int)owner, 64), __shfl(o.z, (int)owner, 64));
                            qd = mk3(__shfl(ld.x, (int)owner, 64), __shfl(ld.y, (int)owner, 64), __shfl(ld.z, (int)owner, 64));
                        }
                        uint32_t qinst = 0u, win = slot;
                        if (INST) qinst = (uint32_t)__shfl((int)inst, (int)owner, 64);
                        bool cand = false;
                        unsigned long long key = 0ull;
                        if (act) {
                            const float4 a = sc.tris[3 * (size_t)slot + 0];
                            const float4 b = sc.tris[3 * (size_t)slot + 1];
                            const float4 c = sc.tris[3 * (size_t)slot + 2];
                            if (STATS) ++n_tris;
                            f3 pos;
                            float b1, b2;
                            if (triangle_hit_pos(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), qo, qd, pos, b1, b2)) {
                                float d2;
                                uint32_t prim;
                                if (INST) {  // position back through Mat, distance in world space (primitive.cpp:38-43)
                                    const DevInstance& I = sc.insts[qinst];
                                    const uint32_t oc = wbase + owner;
                                    d2 = dist2(mk3(s_wray[0 * 256 + oc], s_wray[1 * 256 + oc], s_wray[2 * 256 + oc]), transform_point(I.mat, pos));
                                    prim = I.prim_base + __float_as_uint(a.w);
                                    win = I.virt_base + (slot - I.slot_base);
                                } else {
                                    d2 = dist2(qo, pos);
                                    prim = __float_as_uint(a.w);
                                }
                                key = ((unsigned long long)__float_as_uint(d2) << 32) | prim;
                                // NaN / inf d2 have bit patterns above FLT_MAX's: they can never win, as in the reference
                                cand = true;
                                atomicMin(&s_key[wbase + owner], key);
                            }
                        }
                        if (STATS && lane == 0) ++s_iters[4 + wv];
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                        if (cand && ((volatile unsigned long long*)s_key)[wbase + owner] == key) s_slot[wbase + owner] = LEAN ? sc.n_prims + win : win;
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    }
                }
                }
                const unsigned long long won =
                    ((volatile unsigned long long*)s_key)[(STEAL && k >= 0xFFFFFF00u && k != 0xFFFFFFFFu) ? wbase + (k & 63u) : tid];
                if (LEAN) {
                    tlimit = limit_from_d2(__uint_as_float((uint32_t)(won >> 32)), pad);  // same value if nothing changed
                } else if (won != key_best) {
                    best.d2 = __uint_as_float((uint32_t)(won >> 32));
                    best.prim = (uint32_t)won;
                    best.id = sc.n_prims + ((volatile uint32_t*)s_slot)[tid];
                    tlimit = INST ? (limit_from_d2(best.d2, 0.0f) + padw4) * lscale + 4.0f * pad : limit_from_d2(best.d2, pad);
                }
                pending = false;  // everything that was queued has been tested
            }
        }
        if (STATS) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            cyc_tri += t - t_mark;
            t_mark = t;
        }
    }
    if (STATS) {
        if (lane == 0) {
            atomicAdd(&stats[6], cyc_refill);
            atomicAdd(&stats[7], cyc_node);
            atomicAdd(&stats[8], cyc_tri);
            unsigned long long* tl = stats + 16 + PRT_TIMELINE_WORDS * tune.probe_slot;
            const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
            if (t_exh == 0ull) t_exh = t_end;
            if (my_xcd == 0u) {  // the XCDs' clocks are offset against each other: launch-level marks from XCD 0's waves only
                atomicMin(&tl[0], t_start);
                atomicMax(&tl[1], t_end);
                atomicMin(&tl[2], t_exh);
                atomicMax(&tl[3], t_exh);
            }
            atomicAdd(&tl[4], t_end - t_exh);
            atomicAdd(&tl[5], t_end - t_start);
            atomicAdd(&tl[6], 1ull);
        }
        // one atomic per wave and counter (1.3 M lanes adding to the same words used to take milliseconds)
        uint32_t w_nodes = n_nodes, w_tris = n_tris, w_steps = ray_steps_max, w_sp = max_sp;
        for (int off = 32; off > 0; off >>= 1) {
            w_nodes += (uint32_t)__shfl_xor((int)w_nodes, off, 64);
            w_tris += (uint32_t)__shfl_xor((int)w_tris, off, 64);
            const uint32_t a = (uint32_t)__shfl_xor((int)w_steps, off, 64), b = (uint32_t)__shfl_xor((int)w_sp, off, 64);
            w_steps = a > w_steps ? a : w_steps;
            w_sp = b > w_sp ? b : w_sp;
        }
        if (lane == 0) {
            atomicMax(&stats[16 + PRT_TIMELINE_WORDS * tune.probe_slot + 7], (unsigned long long)w_steps);
            atomicAdd(&stats[0], (unsigned long long)w_nodes);
            atomicAdd(&stats[1], (unsigned long long)w_tris);
            atomicMax(&stats[5], (unsigned long long)w_sp);
        }
        __syncthreads();
        if (threadIdx.x < 4) atomicAdd(&stats[3], 64ull * s_iters[threadIdx.x]);
        if (threadIdx.x < 4) atomicAdd(&stats[4], 64ull * s_iters[4 + threadIdx.x]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Shade + scatter + compaction (ShadeHitsKernel, renderer.cu:274-335; the miss branch of
// IntersectClosestKernel, renderer.cu:263-271; path logic of TraceRayGPU, cuda_megakernel/renderer.cu:81-119).
// Radiance can only be non-zero at the event that ends a path (emissive materials never scatter,
// material.h:119-122), so the path carries throughput only and writes rad[path] once, when it ends.
// ---------------------------------------------------------------------------------------------------------
template <int FUSE, bool SAMPLING, bool INST, bool ABVH, bool PRIM = false>
__global__ void __launch_bounds__(SHADE_BLOCK) k_shade(DevScene sc, const float4* __restrict__ ro,
                                                      const float4* __restrict__ rd, const float4* __restrict__ rt,
                                                      const uint32_t* __restrict__ hit, float4* __restrict__ no,
                                                      float4* __restrict__ nd, float4* __restrict__ nt,
                                                      uint32_t* __restrict__ nhit, float* __restrict__ nhd2,
                                                      float4* __restrict__ rad, uint32_t* __restrict__ counts,
                                                      uint32_t* __restrict__ work, uint32_t iter, uint32_t max_depth,
                                                      uint32_t cap, PrtSampling sp_arg, PrtPrimary pr) {
    // PRIM: the first k_shade of a batch whose k_raygen stored compact primary rays (PrtPrimary): ray, RNG seed,
    // throughput (1,1,1) and segment index (0) follow from the path id
    const PrtSampling sp = SAMPLING ? sp_arg : PrtSampling{0u, 0u, 0.0f};
    const uint32_t nA = CNT_A(counts, iter), nB = CNT_B(counts, iter);
    const uint32_t count = nA + nB;
    if (blockIdx.x * (uint32_t)SHADE_BLOCK >= count) return;  // whole block exits together
    const uint32_t k = blockIdx.x * (uint32_t)SHADE_BLOCK + threadIdx.x;
    if (k < 8u) work[32u * k] = 0u;  // chunk cursors of the next bounce's traversal kernel
    if (k == 8u) work[512] = 0u;     // its overflow-list counter
    bool front = false, back = false;
    f3 o = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 1.f), thr = mk3(0.f, 0.f, 0.f);
    uint32_t id0 = HIT_MISS, rng = 0, depth = 0, pid = 0;
    float d2_0 = 3.402823466e+38f;
    float4 pre_a = make_float4(0.f, 0.f, 0.f, 0.f), pre_b = pre_a;
    if (k < count) {
        const uint32_t src = k < nA ? k : cap - 1u - (k - nA);  // front part, then back part
        uint32_t id = (PRIM && k >= nA) ? HIT_MISS : ld_stream(&hit[src]);
        if (PRIM) {
            pid = ld_stream(&pr.pid[src]);
            uint32_t pixel, sample, lp;
            primary_ray(pr, pid, o, d, pixel, sample, &lp);
            // a back-side primary ray was never walked: its closest hit is the analytic scan's, kept per pixel (k_raygen)
            if (k >= nA) id = __float_as_uint(pr.pix[pr.n_pix_local + lp].y);
            rng = path_seed(pixel, pr.first_sample + sample, pr.seed);
            depth = 0u;
            thr = mk3(1.f, 1.f, 1.f);
            pre_a = pr.pix[2u * pr.n_pix_local + lp];
            pre_b = pr.pix[3u * pr.n_pix_local + lp];
        } else {
            const float4 O = ld_stream(&ro[src]);
            const float4 D = ld_stream(&rd[src]);
            const float4 T = ld_stream(&rt[src]);
            pid = __float_as_uint(O.w);
            rng = __float_as_uint(D.w);
            depth = __float_as_uint(T.w);  // segment index of this ray (paths advance at different rates, see advance_path)
            thr = mk3(T.x, T.y, T.z);
            o = mk3(O.x, O.y, O.z);
            d = mk3(D.x, D.y, D.z);
        }
        if (id != HIT_DEAD) {
            const int r = advance_path<1 + FUSE, INST, ABVH, SHADE_BLOCK, PRIM>(sc, id, o, d, thr, rng, depth, max_depth, sp, &rad[pid], id0, d2_0, pre_a, pre_b);
            front = r == 1;
            back = r == 2;
        }
    }
    const uint32_t slot = block_alloc2<SHADE_BLOCK>(front, back, false, &CNT_A(counts, iter + 1u), &CNT_B(counts, iter + 1u),
                                                    &CNT_C(counts, iter + 1u), cap);
    if (slot != 0xFFFFFFFFu) {
        st_stream(&no[slot], make_float4(o.x, o.y, o.z, __uint_as_float(pid)));
        st_stream(&nd[slot], make_float4(d.x, d.y, d.z, __uint_as_float(rng)));
        st_stream(&nt[slot], make_float4(thr.x, thr.y, thr.z, __uint_as_float(depth)));
        st_stream(&nhit[slot], id0);
        st_stream(&nhd2[slot], d2_0);
    }
}

// Compact primary rays: the surface interaction of a pixel's primary hit, ONCE per pixel.  Without jitter every sample of
// a pixel traces the same pixel-centre ray (cpu/renderer.cpp:45) -- each of them through the traversal kernel on its own --
// and finds the same closest hit, so the first k_shade of a batch used to rebuild the same position / normal / material
// (Triangle::Intersect or the analytic shape again: ~150 wave instructions) once per SAMPLE; a wave there holds 64 samples
// of one pixel.  This kernel rebuilds it per PIXEL from the hit id of the pixel's first stored sample, and k_shade<PRIM>
// uses the record for every sample whose own hit id equals the record's (otherwise it computes the hit itself).
__global__ void __launch_bounds__(256) k_primary_hit(DevScene sc, PrtPrimary pr, const uint32_t* __restrict__ hit,
                                                     float4* __restrict__ pix, const uint32_t* __restrict__ counts) {
    const uint32_t pl = blockIdx.x * 256u + threadIdx.x;
    if (pl >= pr.n_pix_local) return;
    const float4 E = pix[pr.n_pix_local + pl];
    float4 a = make_float4(0.f, 0.f, 0.f, __uint_as_float(HIT_DEAD)), b = make_float4(0.f, 0.f, 0.f, 0.f);  // no record
    if (__float_as_uint(E.w) == 0xFFFFFFFEu) {  // the pixel's paths were stored
        // front-side slots (below the front count) were walked: their hit is the traversal's; back-side paths keep the
        // analytic scan's hit of the pixel's record
        const uint32_t first = __float_as_uint(E.x);
        const uint32_t id = first < CNT_A(counts, 0) ? hit[first] : __float_as_uint(E.y);
        if (id != HIT_MISS && id != HIT_DEAD) {
            const float4 P = pix[pl];
            WorldHit w;
            world_hit_from_id<false>(sc, id, mk3(pr.origin[0], pr.origin[1], pr.origin[2]), mk3(P.x, P.y, P.z), w);
            if (w.has) {
                a = make_float4(w.pos.x, w.pos.y, w.pos.z, __uint_as_float(id));
                b = make_float4(w.normal.x, w.normal.y, w.normal.z, __uint_as_float(w.material | (w.front ? 0x80000000u : 0u)));
            }
        }
    }
    pix[2u * pr.n_pix_local + pl] = a;
    pix[3u * pr.n_pix_local + pl] = b;
}

// Diagnostic (prt_measure_shade_divergence, tools/shade_divergence.py; SURVEY 8f-4 "material-coherent work queues",
// wavefront.md:92-93): what a wave of k_shade finds in its 64 ray slots, by the material of the hit.  Per bounce, 16 words:
// [0] waves, [1] lanes with a ray, [2 + t] lanes whose hit has material type t (0 = miss / none, 1 Lambertian, 2 Metal,
// 3 Dielectric, 4 Emissive), [8 + t] waves in which type t occurs, [14] sum over waves of the number of distinct SCATTERING
// types (1..3) present, [15] waves with at least one scattering lane.  A wave executes the scatter code of every type it
// holds, so [14] / [15] is the factor by which material divergence multiplies the scatter work.
__global__ void __launch_bounds__(SHADE_BLOCK) k_shade_divstats(DevScene sc, const uint32_t* __restrict__ hit,
                                                                 const uint32_t* __restrict__ counts, uint32_t iter, uint32_t cap,
                                                                 unsigned long long* __restrict__ out, PrtPrimary pr, int compact) {
    const uint32_t nA = CNT_A(counts, iter), nB = CNT_B(counts, iter);
    const uint32_t count = nA + nB;
    if (blockIdx.x * (uint32_t)SHADE_BLOCK >= count) return;
    const uint32_t k = blockIdx.x * (uint32_t)SHADE_BLOCK + threadIdx.x;
    uint32_t type = 5u;  // no ray
    if (k < count) {
        const uint32_t src = k < nA ? k : cap - 1u - (k - nA);
        uint32_t id;
        if (compact && k >= nA) {  // compact primary rays: a back-side path's hit is its pixel's analytic one (k_shade<PRIM>)
            f3 o_, d_;
            uint32_t pixel, sample, lp;
            primary_ray(pr, pr.pid[src], o_, d_, pixel, sample, &lp);
            id = __float_as_uint(pr.pix[pr.n_pix_local + lp].y);
        } else {
            id = hit[src];
        }
        type = 0u;
        if (id != HIT_MISS && id != HIT_DEAD) {
            uint32_t m;
            if (id < sc.n_prims) {
                m = sc.prims[id].material;
            } else if (sc.n_insts) {
                const uint32_t v = id - sc.n_prims;
                uint32_t lo = 0, hi = sc.n_insts;
                while (hi - lo > 1u) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (sc.insts[mid].virt_base <= v) lo = mid; else hi = mid;
                }
                const DevInstance& I = sc.insts[lo];
                m = I.material == 0xFFFFFFFFu ? __float_as_uint(sc.tris[3 * (size_t)(I.slot_base + (v - I.virt_base)) + 1].w) : I.material;
            } else {
                m = __float_as_uint(sc.tris[3 * (size_t)(id - sc.n_prims) + 1].w);
            }
            type = sc.mat_type[m] <= 4u ? sc.mat_type[m] : 0u;
        }
    }
    unsigned long long* o = out + 16u * iter;
    const uint32_t lane = lane_id();
    uint32_t distinct = 0u;
    for (uint32_t t = 0; t < 5u; ++t) {
        const unsigned long long mk = __ballot(type == t);
        if (mk != 0ull && lane == 0u) {
            atomicAdd(&o[2 + t], (unsigned long long)__popcll(mk));
            atomicAdd(&o[8 + t], 1ull);
        }
        if (t >= 1u && t <= 3u && mk != 0ull) ++distinct;
    }
    const unsigned long long any = __ballot(type != 5u);
    if (lane == 0u && any != 0ull) {
        atomicAdd(&o[0], 1ull);
        atomicAdd(&o[1], (unsigned long long)__popcll(any));
        if (distinct) {
            atomicAdd(&o[14], (unsigned long long)distinct);
            atomicAdd(&o[15], 1ull);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Film accumulate (Film::AddSample, src/core/film.cu:37-55; BlitRadianceKernel + addBufferGPU,
// renderer.cu:337-348, film.cu:79-88): samples are added in sample order, weight 1 each.
// film_local is tile-ordered [n_pix_local] float4 {r_sum, g_sum, b_sum, weight}.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_accumulate(const float4* __restrict__ rad, float4* __restrict__ film_local,
                                                    PrtTileMap tm, uint32_t S, uint32_t max_depth, int update_film,
                                                    unsigned long long* __restrict__ ray_stats,
                                                    const float4* __restrict__ pix_end) {
    // pix_end (compact primary rays, optional): per local pixel what its paths deliver if they all ended with their
    // primary ray (w != 0xFFFFFFFE); those S identical values are not in rad[] and are added from the record.
    // rad[path] = {radiance, index of the path's last segment}.  Ray segments at depth d = paths whose last segment
    // index is >= d: a per-block histogram of the last indices (wave ballots -> LDS) gives the per-depth counts.
    // A block takes several groups of 256 pixels (grid-stride): its per-depth counts reach the global counters with ONE
    // atomic per depth per block, and a counter word only executes ~87 atomics per microsecond (tools/atomic_rate.hip): with
    // one block per 256 pixels a 1080p frame was 8,100 atomics per word = 93 us, most of a one-sample k_accumulate.
    __shared__ uint32_t s_ends[PRT_MAX_DEPTH];
    if (threadIdx.x < PRT_MAX_DEPTH) s_ends[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t pl0 = blockIdx.x * 256u; pl0 < tm.n_pix_local; pl0 += gridDim.x * 256u) {  // block-uniform trip count
    const uint32_t pl = pl0 + threadIdx.x;
    uint32_t x, y;
    const bool valid = pl < tm.n_pix_local && tile_pixel(tm, pl, x, y);
    float4 f = valid ? film_local[pl] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 E = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFEu));
    if (valid && pix_end) E = pix_end[pl];
    const bool ended = __float_as_uint(E.w) != 0xFFFFFFFEu;
    const float weight = 1.0f;
    const uint32_t lane = lane_id();
    // Samples are added in sample order (that fixes the fp32 sum), but their loads do not depend on each other: eight
    // at a time are in flight (a rank of an 8-GPU run has only four blocks per CU here, too few to hide the latency of
    // one load per iteration)
    for (uint32_t s0 = 0; s0 < S; s0 += 8u) {  // block-uniform trip counts
        float4 r[8];
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j)
            r[j] = (valid && !ended && s0 + j < S) ? ld_stream(&rad[(size_t)(s0 + j) * tm.n_pix_local + pl]) : E;
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) {
            if (s0 + j >= S) break;
            uint32_t e = 0xFFFFFFFFu;
            if (valid) {
                f.x += r[j].x * weight;
                f.y += r[j].y * weight;
                f.z += r[j].z * weight;
                f.w += weight;
                e = __float_as_uint(r[j].w);
            }
            for (uint32_t dd = 0; dd < max_depth; ++dd) {
                const unsigned long long mk = __ballot(e == dd);
                if (mk != 0ull && lane == 0) atomicAdd(&s_ends[dd], (uint32_t)__popcll(mk));
            }
        }
    }
    if (valid && update_film) film_local[pl] = f;
    }
    __syncthreads();
    if (threadIdx.x < max_depth) {
        unsigned long long n = 0;
        for (uint32_t e = threadIdx.x; e < max_depth; ++e) n += s_ends[e];
        // (PRT_RAY_STAT_SLOTS copies of the counters, summed by the host when it reads them: a single word would execute only
        // ~87 of these atomics per microsecond, a floor of 23 us for 2,048 blocks, half of a one-sample k_accumulate)
        if (n) atomicAdd(&ray_stats[(blockIdx.x & (PRT_RAY_STAT_SLOTS - 1u)) * PRT_MAX_DEPTH + threadIdx.x], n);
    }
}

// Un-tile `world` gathered rank payloads (each `stride` float4) into the Film layout
// (m_Accum[3*(y*W+x)+c], m_Weights[y*W+x]; src/core/film.h:54-60).
__global__ void __launch_bounds__(256) k_resolve(const float4* __restrict__ gathered, uint32_t world, uint32_t stride,
                                                 uint32_t W, uint32_t H, float* __restrict__ rgb,
                                                 float* __restrict__ weight) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= W * H) return;
    const uint32_t x = idx % W, y = idx / W;
    const uint32_t tiles_x = (W + 7u) >> 3;
    const uint32_t gt = (y >> 3) * tiles_x + (x >> 3);
    const uint32_t rank = gt % world, lt = gt / world;
    const uint32_t pl = lt * 64u + (y & 7u) * 8u + (x & 7u);
    const float4 f = gathered[(size_t)rank * stride + pl];
    rgb[3 * (size_t)idx + 0] = f.x;
    rgb[3 * (size_t)idx + 1] = f.y;
    rgb[3 * (size_t)idx + 2] = f.z;
    weight[idx] = f.w;
}

// Film::UpdateDisplayGPU (updateDisplayKernel, src/core/film.cu:101-121) with the bounds check the
// reference lacks; one thread per channel.
__global__ void __launch_bounds__(256) k_tonemap(const float* __restrict__ rgb, const float* __restrict__ weight,
                                                 uint32_t n_pix, float exposure, float inv_gamma,
                                                 uint8_t* __restrict__ out) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= n_pix * 4u) return;
    const uint32_t p = idx >> 2, c = idx & 3u;
    if (c == 3u) {
        out[idx] = 255;
        return;
    }
    const float w = weight[p];
    float value = 0.0f;
    if (w > 0.0f) {
        const float invW = 1.0f / w;
        value = rgb[3 * (size_t)p + c] * invW;
        value = value * exposure;
        value = value / (1.0f + value);
        value = powf(value, inv_gamma);
    }
    value = __builtin_fmaxf(0.0f, __builtin_fminf(1.0f, value));
    out[idx] = (uint8_t)(value * 255.0f + 0.5f);
}

// ---------------------------------------------------------------------------------------------------------
// Function-level kernels behind prt_camera_rays / prt_closest_hit / prt_scatter (parity tests).
// ---------------------------------------------------------------------------------------------------------
__global__ void k_camera_rays(DevCamera cam, uint32_t n, const float* __restrict__ px, const float* __restrict__ py,
                              float* __restrict__ o_out, float* __restrict__ d_out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    f3 o, d;
    camera_ray(cam, px[i], py[i], o, d);
    o_out[3 * i + 0] = o.x;
    o_out[3 * i + 1] = o.y;
    o_out[3 * i + 2] = o.z;
    d_out[3 * i + 0] = d.x;
    d_out[3 * i + 1] = d.y;
    d_out[3 * i + 2] = d.z;
}

__global__ void k_pack_rays(uint32_t n, const float* __restrict__ o, const float* __restrict__ d,
                            float4* __restrict__ ro, float4* __restrict__ rd, uint32_t* __restrict__ counts) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0) counts[0] = n;  // front count of the slot it was given
    if (i >= n) return;
    ro[i] = make_float4(o[3 * i], o[3 * i + 1], o[3 * i + 2], __uint_as_float(i));
    rd[i] = make_float4(d[3 * i], d[3 * i + 1], d[3 * i + 2], 0.f);
}

__global__ void k_hit_records(DevScene sc, uint32_t n, const float4* __restrict__ ro, const float4* __restrict__ rd,
                              const uint32_t* __restrict__ hit, PrtHit* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    PrtHit h;
    h.prim = -1;
    h.front_face = 0;
    h.material_id = 0xFFFFFFFFu;
    h.d2 = 3.402823466e+38f;
    h.position[0] = h.position[1] = h.position[2] = 0.f;
    h.normal[0] = h.normal[1] = h.normal[2] = 0.f;
    const uint32_t id = hit[i];
    if (id != HIT_MISS && id != HIT_DEAD) {
        const float4 O = ro[i], D = rd[i];
        WorldHit w;
        if (sc.n_insts)
            world_hit_from_id<true>(sc, id, mk3(O.x, O.y, O.z), mk3(D.x, D.y, D.z), w);
        else
            world_hit_from_id<false>(sc, id, mk3(O.x, O.y, O.z), mk3(D.x, D.y, D.z), w);
        if (w.has) {
            h.prim = w.prim;
            h.front_face = w.front ? 1u : 0u;
            h.material_id = w.material;
            h.d2 = w.d2;
            h.position[0] = w.pos.x;
            h.position[1] = w.pos.y;
            h.position[2] = w.pos.z;
            h.normal[0] = w.normal.x;
            h.normal[1] = w.normal.y;
            h.normal[2] = w.normal.z;
        }
    }
    out[i] = h;
}

__global__ void k_scatter_test(DevScene sc, uint32_t n, const float* __restrict__ in_d, const PrtHit* __restrict__ hits,
                               uint32_t* __restrict__ rng_io, uint32_t* __restrict__ scattered,
                               float* __restrict__ atten_o, float* __restrict__ emit_o, float* __restrict__ o_out,
                               float* __restrict__ d_out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const PrtHit h = hits[i];
    uint32_t rng = rng_io[i];
    f3 atten, emitted, so, sd;
    const uint32_t m = h.material_id;
    const bool sc_ = material_scatter(sc.mat_type[m], sc.mat_rgbs[m], mk3(in_d[3 * i], in_d[3 * i + 1], in_d[3 * i + 2]),
                                      mk3(h.position[0], h.position[1], h.position[2]),
                                      mk3(h.normal[0], h.normal[1], h.normal[2]), h.front_face != 0u, rng, atten,
                                      emitted, so, sd);
    rng_io[i] = rng;
    scattered[i] = sc_ ? 1u : 0u;
    atten_o[3 * i + 0] = atten.x;
    atten_o[3 * i + 1] = atten.y;
    atten_o[3 * i + 2] = atten.z;
    emit_o[3 * i + 0] = emitted.x;
    emit_o[3 * i + 1] = emitted.y;
    emit_o[3 * i + 2] = emitted.z;
    o_out[3 * i + 0] = so.x;
    o_out[3 * i + 1] = so.y;
    o_out[3 * i + 2] = so.z;
    d_out[3 * i + 0] = sd.x;
    d_out[3 * i + 1] = sd.y;
    d_out[3 * i + 2] = sd.z;
}

// ---------------------------------------------------------------------------------------------------------
// Host launchers (plain functions; prt_api.cpp has no kernel syntax).
// ---------------------------------------------------------------------------------------------------------
static inline uint32_t blocks_for(uint64_t n) { return (uint32_t)((n + 255u) / 256u); }

void prt_launch_raygen(hipStream_t st, const DevScene& sc, const DevCamera& cam, const PrtTileMap& tm, uint32_t n_paths,
                       uint32_t first_sample, uint32_t seed, const PrtRayBuf& out, float4* rad, uint32_t* counts,
                       uint32_t* work, uint32_t max_depth, const PrtSampling& sp, float4* compact_pix) {
    const uint32_t S = tm.n_pix_local ? n_paths / tm.n_pix_local : 0u;
    const uint32_t group = sp.jitter ? (uint32_t)RAYGEN_GROUP : RAYGEN_GROUP_NOJITTER;
    const dim3 grid((tm.n_pix_local + PRODUCER_BLOCK - 1) / PRODUCER_BLOCK, (S + group - 1) / group);
#define PRT_RAYGEN(J, SA, AB)                                                                                       \
    hipLaunchKernelGGL((k_raygen<J, SA, AB>), grid, dim3(PRODUCER_BLOCK), 0, st, sc, cam, tm, S, first_sample, seed,  \
                       out.o, out.d, out.t, out.hit, out.hd2, rad, counts, work, max_depth, sp, nullptr)
    const bool sa = sp.rr_depth != 0u || sp.clamp > 0.0f;
    if (sc.abvh_nodes) {  // many analytic primitives: the general instances with the BVH scan
        if (sp.jitter) PRT_RAYGEN(true, true, true); else PRT_RAYGEN(false, true, true);
    } else if (sp.jitter) {
        if (sa) PRT_RAYGEN(true, true, false); else PRT_RAYGEN(true, false, false);
    } else if (compact_pix && !sa) {
        hipLaunchKernelGGL((k_raygen<false, false, false, true>), grid, dim3(PRODUCER_BLOCK), 0, st, sc, cam, tm, S, first_sample,
                           seed, out.o, out.d, out.t, out.hit, out.hd2, rad, counts, work, max_depth, sp, compact_pix);
    } else {
        if (sa) PRT_RAYGEN(false, true, false); else PRT_RAYGEN(false, false, false);
    }
#undef PRT_RAYGEN
}

void prt_launch_scan_prims(hipStream_t st, const DevScene& sc, const PrtRayBuf& in, const uint32_t* count_ptr,
                           uint32_t* work, uint32_t max_rays, unsigned long long* stats) {
    hipLaunchKernelGGL(k_scan_prims, dim3(blocks_for(max_rays)), dim3(256), 0, st, sc, in.o, in.d, in.hit, in.hd2,
                       count_ptr, work, stats);
}

// k_fix_cursors: between the fast traversal and its overflow re-run: reset the chunk cursors.
__global__ void k_reset_cursors(uint32_t* work) {
    if (threadIdx.x < 8u) work[32u * threadIdx.x] = 0u;
}

bool prt_traverse_takes_primary(const DevScene& sc, const PrtTravTuning& tune) {
    // the 5-waves instance of the 8-wide kernel, and a tree shallow enough that no ray can reach the overflow list (its
    // re-traversal reads full ray records)
    return sc.nodes8 && !sc.n_insts && (tune.wide == 2u || !sc.nodes4) && (tune.stack_lds == 0u || tune.stack_lds == 6u) &&
           tune.stack_cap == 0u && sc.depth8 <= 9u;
}

// Which traversal kernel instance prt_launch_traverse runs for this scene and these tunables: ONE decision, shared by the
// launch, the occupancy report and prt_kernel_instance (bench.py names the instance in its line and takes the loops'
// static instruction counts of exactly that instance).
//   inst12_4waves  placed mesh copies: two-level walk, 12 stack entries, 4 waves per SIMD
//   wide11_5waves  A/B (stack_lds = 5): 11 entries, 5 waves per SIMD, ray and best hit in registers
//   lean8_5waves   default: 8 entries, 5 waves per SIMD (ray and best hit in LDS); trees of <= 9 levels, and deeper
//                  HOST-built trees, whose rare deeper rays go through the overflow list to the 4-wide tree (C5)
//   deep15_4waves  deeper trees without a 4-wide fallback (device-built): 15 entries, 4 waves per SIMD
enum PrtT8Kind { T8_NONE = 0, T8_INST12_4, T8_WIDE11_5, T8_LEAN8_5, T8_DEEP15_4 };
static PrtT8Kind t8_kind(const DevScene& sc, const PrtTravTuning& tune) {
    if (!((tune.wide == 2u || sc.n_insts || !sc.nodes4) && sc.nodes8)) return T8_NONE;
    if (sc.n_insts) return T8_INST12_4;
    if (tune.stack_lds == 5u) return T8_WIDE11_5;
    const bool lean = tune.stack_lds == 6u || (tune.stack_lds == 0u && (sc.depth8 <= 9u || sc.nodes4 != nullptr));
    return lean ? T8_LEAN8_5 : T8_DEEP15_4;
}
const char* prt_traverse_instance(const DevScene& sc, const PrtTravTuning& tune) {
    switch (t8_kind(sc, tune)) {
        case T8_INST12_4: return "inst12_4waves";
        case T8_WIDE11_5: return "wide11_5waves";
        case T8_LEAN8_5: return "lean8_5waves";
        case T8_DEEP15_4: return "deep15_4waves";
        default: return tune.wide ? "bvh4" : "bvh2";
    }
}

void prt_launch_traverse(hipStream_t st, const DevScene& sc, const PrtRayBuf& in, const uint32_t* count_ptr,
                         uint32_t* work, uint32_t* spill, uint32_t max_rays, uint32_t tree_depth, uint32_t stack4,
                         const PrtTravTuning& tune_in, unsigned long long* stats, const PrtPrimary* primary) {
    // exit_max "auto" (0xFFFFFFFF, the context's default): the two-level instance leaves its node loop at <= 32 walkers
    // (its lanes wait for level switches that the wave does together; C5I +4.4 % against 16, gpurun_out/r3_sweep_C5I_b.log),
    // every other instance at <= 16
    PrtTravTuning tune = tune_in;
    if (tune.exit_max == 0xFFFFFFFFu) tune.exit_max = t8_kind(sc, tune_in) == T8_INST12_4 ? 32u : 16u;
    // tri_min "auto" (0, the context's default): a triangle phase starts after 24 queueing lane-steps, but after 12 on one-level
    // trees far beyond the caches (one node per 128-B line: C5 +1.7 %; 16: +0.8 to +1.9 %), whose walks gain more from an earlier
    // culling bound than a phase's fixed part costs.  On trees the caches hold 12 / 16 cost 0.5-2 % (C2, C3), and the two-level
    // instance at its exit_max of 32 is best at 24 too (16: -1.5 %).  tools/sweep.py, TUNING.md
    if (tune.tri_min == 0u) tune.tri_min = (t8_kind(sc, tune_in) != T8_INST12_4 && sc.node_stride == 8u) ? 12u : 24u;
#ifdef PRT_PROBE_REBOUND
    // diagnostic build (tools/bounce_stats.py --rebound): the instrumented launch is preceded by a plain one that leaves
    // every ray's FINAL hit distance as its initial culling bound, so the instrumented walk's visit counts are those of a
    // traversal that knew the answer from the start: the floor for any visiting order / triangle-test schedule
    if (stats) {
        prt_launch_traverse(st, sc, in, count_ptr, work, spill, max_rays, tree_depth, stack4, tune_in, nullptr, primary);
        hipLaunchKernelGGL(k_reset_cursors, dim3(1), dim3(64), 0, st, work);
    }
#endif
    uint32_t g = tune.grid_blocks;
    const uint32_t need_blocks = blocks_for(max_rays);
    if (g > need_blocks) g = need_blocks;
    if (g == 0) g = 1;
    const dim3 grid(g), block(256);
    uint32_t* ovf = work + 512;  // [0] = count, [1..] = ray slots that overflowed the LDS stack
    const uint32_t* no_list = nullptr;
#define PRT_LAUNCH_T(KERNEL, L, W, MODE, GRID, COUNT, LIST)                                                        \
    do {                                                                                                           \
        if (stats)                                                                                                 \
            hipLaunchKernelGGL((KERNEL<L, W, MODE, true>), GRID, block, 0, st, sc, in.o, in.d, in.hit, in.hd2,     \
                               COUNT, work, spill, LIST, ovf, tune, stats);                                        \
        else                                                                                                       \
            hipLaunchKernelGGL((KERNEL<L, W, MODE, false>), GRID, block, 0, st, sc, in.o, in.d, in.hit, in.hd2,    \
                               COUNT, work, spill, LIST, ovf, tune, stats);                                        \
    } while (0)
    const PrtT8Kind kind = t8_kind(sc, tune);
    if (kind != T8_NONE) {  // (device-built scenes only have the 8-wide tree)
        // default: compressed 8-wide tree; a ray needs at most depth8 - 1 stacked node groups.  15 entries at
        // 4 waves/SIMD or 11 entries at 5 waves/SIMD (tune.stack_lds == 5); deeper rays take the overflow list.
#define PRT_LAUNCH_8(L, W, IN)                                                                                     \
    do {                                                                                                           \
        if (stats)                                                                                                 \
            hipLaunchKernelGGL((k_traverse8_persistent<L, W, true, IN>), grid, block, 0, st, sc, in.o, in.d,       \
                               in.hit, in.hd2, count_ptr, work, ovf, tune, stats, PrtPrimary{}, PrtPathArgs{});    \
        else                                                                                                       \
            hipLaunchKernelGGL((k_traverse8_persistent<L, W, false, IN>), grid, block, 0, st, sc, in.o, in.d,      \
                               in.hit, in.hd2, count_ptr, work, ovf, tune, stats, PrtPrimary{}, PrtPathArgs{});    \
    } while (0)
        if (kind == T8_INST12_4) {  // placed mesh copies: two-level walk; a stack overflow is an error (the host checks the depths).
            // 12 stack entries + the lanes' world rays in LDS = the same 40 KB per block as the one-level instance
            PRT_LAUNCH_8(12, 4, true);
            return;
        }
        // default: the 5-waves-per-SIMD instance (96 VGPRs: ray and best hit live in LDS, 8 stack entries) for trees of
        // at most 9 levels (C3: 9), else the 4-waves instance with 15 entries (C5: 11 levels); stack_lds 4 / 5 / 6
        // force one for A/B runs
        // (deeper host-built trees: rays that need more than 8 entries go through the overflow list to the 4-wide
        // spill-capable instance; C5, 11 levels: deepest stack 8, no such ray)
        const uint32_t stack_l = kind == T8_WIDE11_5 ? 11u : kind == T8_LEAN8_5 ? 8u : 15u;
        if (kind == T8_WIDE11_5) {
            PRT_LAUNCH_8(11, 5, false);
        } else if (kind == T8_LEAN8_5) {
            const dim3 grid5(g == tune.grid_blocks ? g + g / 4u : g);  // 5 instead of 4 resident blocks per CU
            PrtTravTuning t5 = tune;
            if (tune.stack_cap != 0u || sc.depth8 > stack_l + 1u) t5.steal = 0u;  // (a helper cannot hand a ray to the overflow list)
            if (stats && primary)
                hipLaunchKernelGGL((k_traverse8_persistent<8, 5, true, false, true, true>), grid5, block, 0, st, sc, in.o, in.d,
                                   in.hit, in.hd2, count_ptr, work, ovf, t5, stats, *primary, PrtPathArgs{});
            else if (stats)
                hipLaunchKernelGGL((k_traverse8_persistent<8, 5, true, false, true>), grid5, block, 0, st, sc, in.o, in.d,
                                   in.hit, in.hd2, count_ptr, work, ovf, t5, stats, PrtPrimary{}, PrtPathArgs{});
            else if (primary)  // bounce 0 of a batch whose k_raygen stored compact primary rays
                hipLaunchKernelGGL((k_traverse8_persistent<8, 5, false, false, true, true>), grid5, block, 0, st, sc, in.o, in.d,
                                   in.hit, in.hd2, count_ptr, work, ovf, t5, stats, *primary, PrtPathArgs{});
            else
                hipLaunchKernelGGL((k_traverse8_persistent<8, 5, false, false, true>), grid5, block, 0, st, sc, in.o, in.d,
                                   in.hit, in.hd2, count_ptr, work, ovf, t5, stats, PrtPrimary{}, PrtPathArgs{});
        } else {
            PRT_LAUNCH_8(15, 4, false);
        }
#undef PRT_LAUNCH_8
        // overflow list -> the spill-capable 4-wide instance.  Not launched when the tree is too shallow for any ray to
        // overflow (C3: 9 levels = 8 stacked groups at most): two tiny launches and their gaps are ~12 us per bounce
        if (sc.nodes4 && (tune.stack_cap != 0u || sc.depth8 > stack_l + 1u)) {
            hipLaunchKernelGGL(k_reset_cursors, dim3(1), dim3(64), 0, st, work);
            PRT_LAUNCH_T(k_traverse4_persistent, 27, 5, 1, dim3(8), ovf, ovf + 1);
        }
    } else if (tune.wide) {
        if (stack4 <= 22 && tune.stack_lds == 24) PRT_LAUNCH_T(k_traverse4_persistent, 22, 6, 0, grid, count_ptr, no_list);
        else if (stack4 <= 27) PRT_LAUNCH_T(k_traverse4_persistent, 27, 5, 0, grid, count_ptr, no_list);
        else if (tune.stack_lds == 39 && stack4 <= 35) PRT_LAUNCH_T(k_traverse4_persistent, 35, 4, 0, grid, count_ptr, no_list);
        else if (tune.stack_lds == 2) {
            // A/B: LDS-only stack with overflow check, then the spill-capable instance over the (normally empty)
            // overflow list.  Measured equal to the always-spill instance on C3, which is therefore the default.
            PRT_LAUNCH_T(k_traverse4_persistent, 27, 5, 2, grid, count_ptr, no_list);
            hipLaunchKernelGGL(k_reset_cursors, dim3(1), dim3(64), 0, st, work);
            PRT_LAUNCH_T(k_traverse4_persistent, 27, 5, 1, dim3(8), ovf, ovf + 1);
        } else if (tune.stack_lds == 3) {
            // branch-free node step on an LDS-only stack of 25 entries (+3 rows of slack), overflow list -> MODE 1
            PRT_LAUNCH_T(k_traverse4_persistent, 24, 5, 3, grid, count_ptr, no_list);
            hipLaunchKernelGGL(k_reset_cursors, dim3(1), dim3(64), 0, st, work);
            PRT_LAUNCH_T(k_traverse4_persistent, 27, 5, 1, dim3(8), ovf, ovf + 1);
        } else if (tune.stack_lds == 1) {
            PRT_LAUNCH_T(k_traverse4_persistent, 27, 5, 1, grid, count_ptr, no_list);  // A/B: 28 entries in LDS + global spill
        } else {
            // default: branch-free node step, LDS-only stack of 32 entries (the deepest stack any C3 ray reaches is 17;
            // the host's worst-case bound is 36), 4 waves/SIMD so that the 116 VGPRs need no scratch (the 5-wave
            // instances spill 50-70 B/lane and are 9 % slower); a ray that would overflow goes to the overflow list
            // and is re-traversed by the spill-capable instance.
            PRT_LAUNCH_T(k_traverse4_persistent, 32, 4, 3, grid, count_ptr, no_list);
            hipLaunchKernelGGL(k_reset_cursors, dim3(1), dim3(64), 0, st, work);
            PRT_LAUNCH_T(k_traverse4_persistent, 27, 5, 1, dim3(8), ovf, ovf + 1);
        }
    } else {
        const uint32_t pushes = tree_depth ? tree_depth - 1u : 0u;
        if (pushes <= 24 && tune.stack_lds == 24) PRT_LAUNCH_T(k_traverse_persistent, 24, 6, 0, grid, count_ptr, no_list);
        else if (pushes <= 31) PRT_LAUNCH_T(k_traverse_persistent, 31, 5, 0, grid, count_ptr, no_list);
        else PRT_LAUNCH_T(k_traverse_persistent, 31, 5, 1, grid, count_ptr, no_list);
    }
#undef PRT_LAUNCH_T
}

// The PATH instance (PrtPathArgs): whole paths in one launch, for small batches.  One-level scenes with the 8-wide tree,
// a tree its 15-entry stack holds, and no primitive BVH (classify_ray's walk keeps a per-thread LDS stack of its own).
bool prt_path_kernel_applies(const DevScene& sc, const PrtTravTuning& tune) {
    return sc.nodes8 && !sc.n_insts && !sc.abvh_nodes && sc.depth8 <= 16u && tune.wide == 2u && tune.stack_cap == 0u;
}
void prt_launch_path(hipStream_t st, const DevScene& sc, const PrtPathArgs& pa, uint32_t* work, const PrtTravTuning& tune_in) {
    PrtTravTuning tune = tune_in;
    if (tune.exit_max == 0xFFFFFFFFu) tune.exit_max = 16u;
    if (tune.tri_min == 0u) tune.tri_min = 24u;
    uint32_t g = tune.grid_blocks;
    const uint32_t need_blocks = blocks_for(pa.n_paths);
    if (g > need_blocks) g = need_blocks;
    if (g == 0) g = 1;
    hipLaunchKernelGGL((k_traverse8_persistent<15, 4, false, false, false, false, true>), dim3(g), dim3(256), 0, st, sc, (const float4*)nullptr,
                       (const float4*)nullptr, (uint32_t*)nullptr, (const float*)nullptr, (const uint32_t*)nullptr, work, work + 512, tune,
                       (unsigned long long*)nullptr, PrtPrimary{}, pa);
}

// Static occupancy of the default traversal kernel instance for this scene (what the wavefront-occupancy figure of
// the bench line is computed from): resident 256-thread blocks per CU by the runtime's occupancy calculator, and the
// kernel's register / LDS footprint.
int prt_traverse_occupancy(const DevScene& sc, const PrtTravTuning& tune, int* blocks_per_cu, int* vgprs, int* sgprs, int* lds_bytes) {
    const PrtT8Kind kind = t8_kind(sc, tune);
    const void* fn = kind == T8_INST12_4 ? (const void*)k_traverse8_persistent<12, 4, false, true>
                     : kind == T8_WIDE11_5 ? (const void*)k_traverse8_persistent<11, 5, false, false>
                     : kind == T8_LEAN8_5 ? (const void*)k_traverse8_persistent<8, 5, false, false, true>
                     : kind == T8_DEEP15_4 ? (const void*)k_traverse8_persistent<15, 4, false, false>
                                           : (const void*)k_traverse4_persistent<32, 4, 3, false>;
    hipFuncAttributes at;
    hipError_t e = hipFuncGetAttributes(&at, fn);
    if (e != hipSuccess) return (int)e;
    int nb = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, 0);
    if (e != hipSuccess) return (int)e;
    *blocks_per_cu = nb;
    *vgprs = at.numRegs;
    *sgprs = 0;
    *lds_bytes = (int)at.sharedSizeBytes;
    return 0;
}

// Variants 1 (while-while) and 2 (one-loop): the fused one-thread-per-ray kernel, kept for A/B runs.
void prt_launch_intersect(hipStream_t st, const DevScene& sc, const PrtRayBuf& in, const uint32_t* count_ptr, uint32_t max_rays, int stack_depth, int variant,
                          unsigned long long* stats) {
    const dim3 grid(blocks_for(max_rays)), block(256);
#define PRT_LAUNCH_I(S, T, V) \
    hipLaunchKernelGGL((k_intersect<S, T, V>), grid, block, 0, st, sc, in.o, in.d, in.hit, count_ptr, stats)
    const bool deep = stack_depth > 31;
    if (stats) {
        if (variant == 2) { if (deep) PRT_LAUNCH_I(63, true, 1); else PRT_LAUNCH_I(31, true, 1); }
        else              { if (deep) PRT_LAUNCH_I(63, true, 0); else PRT_LAUNCH_I(31, true, 0); }
    } else {
        if (variant == 2) { if (deep) PRT_LAUNCH_I(63, false, 1); else PRT_LAUNCH_I(31, false, 1); }
        else              { if (deep) PRT_LAUNCH_I(63, false, 0); else PRT_LAUNCH_I(31, false, 0); }
    }
#undef PRT_LAUNCH_I
}

void prt_launch_shade(hipStream_t st, const DevScene& sc, const PrtRayBuf& in, const PrtRayBuf& out, float4* rad,
                      uint32_t* counts, uint32_t* work, uint32_t depth, uint32_t max_depth, uint32_t cap,
                      uint32_t fuse_max, const PrtSampling& sp, uint32_t n_rays_known, const PrtPrimary* primary) {
    // n_rays_known: the ray count of this bounce if the host has it already (0xFFFFFFFF: size the grid for `cap`)
    const uint32_t n_for_grid = n_rays_known == 0xFFFFFFFFu ? cap : (n_rays_known ? n_rays_known : 1u);
    const dim3 grid((uint32_t)((n_for_grid + SHADE_BLOCK - 1) / SHADE_BLOCK));
#define PRT_SHADE(F, SA, IN, AB)                                                                                    \
    hipLaunchKernelGGL((k_shade<F, SA, IN, AB>), grid, dim3(SHADE_BLOCK), 0, st, sc, in.o, in.d, in.t, in.hit, out.o, \
                       out.d, out.t, out.hit, out.hd2, rad, counts, work, depth, max_depth, cap, sp, PrtPrimary{})
    const bool sa = sp.rr_depth != 0u || sp.clamp > 0.0f;
    if (primary) {  // (the host only asks for this with the default instance's conditions: no placed copies, no primitive BVH, no sampling options, no fusion)
        hipLaunchKernelGGL((k_shade<0, false, false, false, true>), grid, dim3(SHADE_BLOCK), 0, st, sc, in.o, in.d, in.t, in.hit,
                           out.o, out.d, out.t, out.hit, out.hd2, rad, counts, work, depth, max_depth, cap, sp, *primary);
    } else if (sc.abvh_nodes) {  // many analytic primitives: general instances with the BVH scan
        if (sc.n_insts) PRT_SHADE(0, true, true, true); else PRT_SHADE(0, true, false, true);
    } else if (sc.n_insts) {  // scenes with placed mesh copies: one general instance
        PRT_SHADE(0, true, true, false);
    } else if (fuse_max) {
        if (sa) PRT_SHADE(1, true, false, false); else PRT_SHADE(1, false, false, false);
    } else {
        if (sa) PRT_SHADE(0, true, false, false); else PRT_SHADE(0, false, false, false);
    }
#undef PRT_SHADE
}

void prt_launch_shade_divstats(hipStream_t st, const DevScene& sc, const PrtRayBuf& in, const uint32_t* counts, uint32_t iter,
                               uint32_t cap, unsigned long long* out, const PrtPrimary* primary) {
    hipLaunchKernelGGL(k_shade_divstats, dim3((cap + SHADE_BLOCK - 1u) / SHADE_BLOCK), dim3(SHADE_BLOCK), 0, st, sc, in.hit, counts, iter, cap, out,
                       primary ? *primary : PrtPrimary{}, primary ? 1 : 0);
}

void prt_launch_primary_hit(hipStream_t st, const DevScene& sc, const PrtPrimary& pr, const uint32_t* hit, float4* pix,
                            const uint32_t* counts) {
    hipLaunchKernelGGL(k_primary_hit, dim3(blocks_for(pr.n_pix_local)), dim3(256), 0, st, sc, pr, hit, pix, counts);
}

void prt_launch_accumulate(hipStream_t st, const float4* rad, float4* film_local, const PrtTileMap& tm, uint32_t S,
                           uint32_t max_depth, bool update_film, unsigned long long* ray_stats, const float4* pix_end) {
    const uint32_t nb = blocks_for(tm.n_pix_local ? tm.n_pix_local : 1);
    hipLaunchKernelGGL(k_accumulate, dim3(nb < 8192u ? nb : 8192u), dim3(256), 0, st, rad, film_local, tm, S, max_depth,
                       update_film ? 1 : 0, ray_stats, pix_end);
}

void prt_launch_resolve(hipStream_t st, const float4* gathered, uint32_t world, uint32_t stride, uint32_t W,
                        uint32_t H, float* rgb, float* weight) {
    hipLaunchKernelGGL(k_resolve, dim3(blocks_for((uint64_t)W * H)), dim3(256), 0, st, gathered, world, stride, W, H,
                       rgb, weight);
}

void prt_launch_tonemap(hipStream_t st, const float* rgb, const float* weight, uint32_t n_pix, float exposure,
                        float inv_gamma, uint8_t* out) {
    hipLaunchKernelGGL(k_tonemap, dim3(blocks_for((uint64_t)n_pix * 4u)), dim3(256), 0, st, rgb, weight, n_pix,
                       exposure, inv_gamma, out);
}

void prt_launch_camera_rays(hipStream_t st, const DevCamera& cam, uint32_t n, const float* px, const float* py,
                            float* o, float* d) {
    hipLaunchKernelGGL(k_camera_rays, dim3(blocks_for(n)), dim3(256), 0, st, cam, n, px, py, o, d);
}

void prt_launch_pack_rays(hipStream_t st, uint32_t n, const float* o, const float* d, const PrtRayBuf& out,
                          uint32_t* counts) {
    hipLaunchKernelGGL(k_pack_rays, dim3(blocks_for(n)), dim3(256), 0, st, n, o, d, out.o, out.d, counts);
}

void prt_launch_hit_records(hipStream_t st, const DevScene& sc, uint32_t n, const PrtRayBuf& in, PrtHit* out) {
    hipLaunchKernelGGL(k_hit_records, dim3(blocks_for(n)), dim3(256), 0, st, sc, n, in.o, in.d, in.hit, out);
}

void prt_launch_scatter_test(hipStream_t st, const DevScene& sc, uint32_t n, const float* in_d, const PrtHit* hits,
                             uint32_t* rng_io, uint32_t* scattered, float* atten, float* emitted, float* o_out,
                             float* d_out) {
    hipLaunchKernelGGL(k_scatter_test, dim3(blocks_for(n)), dim3(256), 0, st, sc, n, in_d, hits, rng_io, scattered,
                       atten, emitted, o_out, d_out);
}
