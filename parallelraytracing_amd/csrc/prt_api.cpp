// prt_api.cpp — implementation of the C-ABI in include/prt.h on top of the HIP kernels.
//
// Host-side mirror of what the reference's CudaWavefrontRenderer does around its kernels
// (src/backend/cuda_wavefront/renderer.cu:351-547: Init / ProgressiveRender / SetCamera / Allocate*),
// re-designed: flat SoA scene upload in one piece (no per-object cudaMalloc, soa.cpp:60-61,86-87),
// dense double-buffered ray records, device-side counters (no 1-thread reset kernels,
// renderer.cu:399-414), no host synchronisation inside a sample.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/prt.h"
#include "bvh.h"
#include "bvh_gpu.h"
#include "prt_kernels.h"

namespace {

constexpr float kPadCoeff = 1.0f / 262144.0f;  // 2^-18, see traverse() in prt_kernels.hip
constexpr uint32_t kMaxLeaf = 3;  // the compressed 8-wide node encodes at most 3 triangles per leaf (bvh.h)
constexpr size_t kRayStatWords = (size_t)PRT_RAY_STAT_SLOTS * PRT_MAX_DEPTH;
constexpr uint32_t kTravStatsWords = 16 + PRT_TIMELINE_WORDS * (PRT_MAX_DEPTH + 1);  // counters + one launch timeline per bounce
constexpr uint32_t kMaxStack = 63;  // LDS stack entries per lane: 31 (5 blocks/CU) or 63 (2 blocks/CU)

struct EventPair {
    hipEvent_t a, b;
    int kind;  // 0 raygen, 1 intersect (dominant kernel), 2 shade, 3 accumulate, 4 analytic-primitive scan
};

}  // namespace

struct PrtContext {
    int device = -1;
    bool has_device = false;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;

    // ---- scene ----
    bool has_scene = false;
    std::vector<PrtMaterial> materials;
    std::vector<DevPrim> prims;
    BvhBuild bvh;
    std::vector<float> tri_records;   // 12 floats per triangle, leaf order
    std::vector<float> nrm_records;   // 12 floats per triangle, leaf order
    double gpu_build_ms = 0.0;
    bool scene_device_built = false;  // the scene's 8-wide tree came from the device-side builder (no binary / 4-wide tree)
    std::vector<uint32_t> mesh_sizes;  // per world-space mesh of the scene: n_vertices, n_triangles (prt_refit_meshes checks them)
    double refit_ms = 0.0;             // device time of the last prt_refit_meshes (records + boxes + quantization)
    std::vector<uint32_t> nodes8_all;  // scenes with placed mesh copies: top-level tree + every mesh's tree
    std::vector<DevInstance> dev_insts;
    std::vector<uint32_t> tlas_inst;   // top-level leaf slot -> instance
    BvhBuild abvh;                     // BVH over the analytic primitives' world boxes (scenes with many of them)
    PrtBvhInfo bvh_info{};
    DevScene dsc{};
    void* d_prims = nullptr;
    void* d_mat_rgbs = nullptr;
    void* d_mat_type = nullptr;
    void* d_nodes = nullptr;
    void* d_nodes4 = nullptr;
    void* d_nodes8 = nullptr;
    void* d_insts = nullptr;
    void* d_tlas_inst = nullptr;
    void* d_abvh_nodes = nullptr;
    void* d_abvh_order = nullptr;
    void* d_tris = nullptr;
    void* d_nrms = nullptr;

    // ---- camera ----
    bool has_camera = false;
    DevCamera cam{};

    // ---- film / partition ----
    bool has_film = false;
    PrtTileMap tm{};
    uint32_t valid_local = 0;  // pixels of this rank inside the image
    float4* d_film_local = nullptr;

    // ---- path state ----
    uint32_t S = 1;
    uint64_t cap_paths = 0;
    PrtRayBuf rb[2] = {{nullptr, nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr, nullptr}};
    float4* d_rad = nullptr;
    uint32_t* d_counts = nullptr;             // (PRT_MAX_DEPTH + 2) x PRT_CNT_STRIDE: front/back ray counts per bounce
    unsigned long long* d_ray_stats = nullptr;  // [2][PRT_RAY_STAT_SLOTS][PRT_MAX_DEPTH]: the context's counters, then a scratch set for measurement runs
    unsigned long long* ray_stats_target = nullptr;  // the set k_accumulate adds to (d_ray_stats, or its scratch half during a measurement)
    bool pix_records_blank = false;  // d_pix's primary-hit records all say "no record" (batches of one sample)
    uint32_t* h_flag = nullptr;  // pinned: the traversal kernels' error flags (d_work[256]) as of the last prt_synchronize
    unsigned long long* d_trav_stats = nullptr; // 3

    // ---- scratch for the function-level entry points ----
    void* d_scratch = nullptr;
    size_t scratch_bytes = 0;

    // ---- stats ----
    bool timing = false;
    std::vector<EventPair> events;
    std::vector<EventPair> event_pool;
    PrtStats stats{};
    uint64_t dead_paths = 0;
    int variant = 0;
    int abvh_enabled = 1;  // prt_set_param("prim_bvh", 0): keep the reference's linear scan over the analytic primitives
    int measure_spp = 1;  // samples of the instrumented batch of prt_measure_traversal
    float pad_coeff = kPadCoeff;  // prt_set_param("pad_log2", n): culling pad = 2^-n of the coordinates' magnitude (A/B; before prt_set_scene)
    int node_stride = 0;      // prt_set_param("node_stride", 5 | 8): uint4 per 8-wide node slot, 0 = by tree size (upload_scene); before prt_set_scene
    int compact_primary = 1;  // prt_set_param("compact_primary", 0): k_raygen stores full ray records (A/B)
    int gpu_build = 0;  // prt_set_param("gpu_build", 1): the next prt_set_scene builds the 8-wide tree on the device
    PrtSampling sampling{0u, 0u, 0.0f};
    // grid 256 CUs x 4 blocks, 256-ray chunks, refill at 16 idle lanes, leave the node loop at <= 16 walkers, triangle
    // phase after 24 queueing lane-steps, 8-wide tree (all measured best on C3, tools/sweep.py); XCD affinity off
    PrtTravTuning tune{1024u, 256u, 16u, 0xFFFFFFFFu /* exit_max: auto */, 0u, 2u, 0u /* tri_min: auto */, 0u, 0u, 0u, 1u, 8u, 1u, 0u, nullptr, 0u, 2500000u, 2u /* big */, 8u /* static_small */, 96u /* big_min */, 32u /* big_keep */, 1u};
    unsigned long long* d_shade_div = nullptr;  // diagnostic (prt_measure_shade_divergence): 16 words per bounce, or null
    uint32_t sort_rays = 0;       // measurement aid: 1 / 2 = bounces >= 1 (and jittered bounce 0) walk their rays in sorted order
    uint32_t* d_sort = nullptr;   // keys, keys2, idx, idx2 (n_paths each) + rocPRIM's temporary storage
    size_t sort_entries = 0, sort_temp = 0;
    uint32_t* h_counts = nullptr;  // pinned: the front / back ray counts of each bounce as the host learns them
    hipEvent_t ev_counts[PRT_MAX_DEPTH + 2] = {};
    hipEvent_t ev_prod[PRT_MAX_DEPTH + 2] = {};  // "the producer of bounce d's counts is done" (render stream -> count stream)
    hipStream_t count_stream = nullptr;          // the counts' copies to the host run beside the render stream, not in it
    uint32_t* d_work = nullptr;   // chunk cursor of the persistent traversal kernel
    float4* d_pix = nullptr;      // compact primary rays: one record per local pixel (PrtPrimary)
    uint32_t pix_entries = 0;
    uint32_t* d_spill = nullptr;  // global part of the per-lane traversal stacks
    size_t spill_entries = 0;
};

namespace {

int fail(PrtContext* c, int code, const char* fmt, ...) {
    if (c) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        c->err = buf;
    }
    return code;
}

#define HIPCHECK(c, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return fail((c), PRT_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

int need_device(PrtContext* c) {
    if (!c) return PRT_ERR_INVALID;
    if (!c->has_device)
        return fail(c, PRT_ERR_NO_DEVICE,
                    "no HIP device bound to this context: the MI355X kernels are the only compute path (no CPU fallback)");
    // every device entry point starts here: make the context's GPU the calling thread's current device (a process may
    // hold contexts on several GPUs, each driven from its own host thread: prt_group_*)
    const hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return fail(c, PRT_ERR_HIP, "hipSetDevice(%d) failed: %s", c->device, hipGetErrorString(e));
    return PRT_OK;
}

void free_dev(void*& p) {
    if (p) {
        (void)hipFree(p);
        p = nullptr;
    }
}
template <class T>
void free_dev(T*& p) {
    if (p) {
        (void)hipFree((void*)p);
        p = nullptr;
    }
}

void free_path_state(PrtContext* c) {
    for (int i = 0; i < 2; ++i) {
        free_dev(c->rb[i].o);
        free_dev(c->rb[i].d);
        free_dev(c->rb[i].t);
        free_dev(c->rb[i].hit);
        free_dev(c->rb[i].hd2);
    }
    free_dev(c->d_rad);
    c->cap_paths = 0;
}

int ensure_path_state(PrtContext* c, uint64_t n_paths) {
    if (n_paths <= c->cap_paths) return PRT_OK;
    free_path_state(c);
    const uint64_t n = std::max<uint64_t>(n_paths, 256);
    for (int i = 0; i < 2; ++i) {
        HIPCHECK(c, hipMalloc((void**)&c->rb[i].o, n * sizeof(float4)));
        HIPCHECK(c, hipMalloc((void**)&c->rb[i].d, n * sizeof(float4)));
        HIPCHECK(c, hipMalloc((void**)&c->rb[i].t, n * sizeof(float4)));
        HIPCHECK(c, hipMalloc((void**)&c->rb[i].hit, n * sizeof(uint32_t)));
        HIPCHECK(c, hipMalloc((void**)&c->rb[i].hd2, n * sizeof(float)));
    }
    HIPCHECK(c, hipMalloc((void**)&c->d_rad, n * sizeof(float4)));
    c->cap_paths = n;
    return PRT_OK;
}

int ensure_counters(PrtContext* c) {
    if (c->d_counts) return PRT_OK;
    HIPCHECK(c, hipMalloc((void**)&c->d_counts, (PRT_MAX_DEPTH + 2) * PRT_CNT_STRIDE * sizeof(uint32_t)));
    HIPCHECK(c, hipMalloc((void**)&c->d_ray_stats, kRayStatWords * 2 * sizeof(unsigned long long)));
    c->ray_stats_target = c->d_ray_stats;
    HIPCHECK(c, hipMalloc((void**)&c->d_trav_stats, kTravStatsWords * sizeof(unsigned long long)));
    HIPCHECK(c, hipMemset(c->d_counts, 0, (PRT_MAX_DEPTH + 2) * PRT_CNT_STRIDE * sizeof(uint32_t)));
    HIPCHECK(c, hipMemset(c->d_ray_stats, 0, kRayStatWords * 2 * sizeof(unsigned long long)));
    HIPCHECK(c, hipMemset(c->d_trav_stats, 0, kTravStatsWords * sizeof(unsigned long long)));
    // [0..255] chunk cursors (one 128-B line per XCD), [256] watchdog flag, [512] overflow count, [513..] overflow list
    HIPCHECK(c, hipMalloc((void**)&c->d_work, (513 + (1u << 20)) * sizeof(uint32_t)));
    HIPCHECK(c, hipMemset(c->d_work, 0, (513 + (1u << 20)) * sizeof(uint32_t)));
    return PRT_OK;
}

// global spill area of the traversal stacks: [63 - stack_lds entries][grid threads]
int ensure_spill(PrtContext* c) {
    const size_t need = (size_t)c->tune.grid_blocks * 256u * 64u;
    if (need <= c->spill_entries) return PRT_OK;
    free_dev(c->d_spill);
    c->spill_entries = 0;
    HIPCHECK(c, hipMalloc((void**)&c->d_spill, need * sizeof(uint32_t)));
    c->spill_entries = need;
    return PRT_OK;
}

int ensure_scratch(PrtContext* c, size_t bytes) {
    if (bytes <= c->scratch_bytes) return PRT_OK;
    free_dev(c->d_scratch);
    c->scratch_bytes = 0;
    HIPCHECK(c, hipMalloc(&c->d_scratch, bytes));
    c->scratch_bytes = bytes;
    return PRT_OK;
}

void free_scene(PrtContext* c) {
    free_dev(c->d_prims);
    free_dev(c->d_mat_rgbs);
    free_dev(c->d_mat_type);
    free_dev(c->d_nodes);
    free_dev(c->d_nodes4);
    free_dev(c->d_nodes8);
    free_dev(c->d_insts);
    free_dev(c->d_tlas_inst);
    free_dev(c->d_abvh_nodes);
    free_dev(c->d_abvh_order);
    free_dev(c->d_tris);
    free_dev(c->d_nrms);
    c->has_scene = false;
}

// timing helpers ----------------------------------------------------------------------------------
// the per-depth ray counters of one set: the sum over its slots (prt_kernels.h PRT_RAY_STAT_SLOTS)
int read_ray_stats(PrtContext* c, const unsigned long long* d_set, unsigned long long (&out)[PRT_MAX_DEPTH]) {
    std::vector<unsigned long long> h(kRayStatWords);
    HIPCHECK(c, hipMemcpy(h.data(), d_set, kRayStatWords * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int d = 0; d < PRT_MAX_DEPTH; ++d) {
        out[d] = 0;
        for (size_t sl = 0; sl < PRT_RAY_STAT_SLOTS; ++sl) out[d] += h[sl * PRT_MAX_DEPTH + (size_t)d];
    }
    return PRT_OK;
}

int begin_event(PrtContext* c, int kind, EventPair* ep) {
    if (!c->timing) return PRT_OK;
    if (!c->event_pool.empty()) {
        *ep = c->event_pool.back();
        c->event_pool.pop_back();
    } else {
        HIPCHECK(c, hipEventCreate(&ep->a));
        HIPCHECK(c, hipEventCreate(&ep->b));
    }
    ep->kind = kind;
    HIPCHECK(c, hipEventRecord(ep->a, c->stream));
    return PRT_OK;
}
int end_event(PrtContext* c, EventPair* ep) {
    if (!c->timing) return PRT_OK;
    HIPCHECK(c, hipEventRecord(ep->b, c->stream));
    c->events.push_back(*ep);
    return PRT_OK;
}
int drain_events(PrtContext* c) {
    for (EventPair& ep : c->events) {
        float ms = 0.0f;
        HIPCHECK(c, hipEventSynchronize(ep.b));
        HIPCHECK(c, hipEventElapsedTime(&ms, ep.a, ep.b));
        switch (ep.kind) {
            case 0: c->stats.raygen_ms += ms; break;
            case 1: c->stats.intersect_ms += ms; break;
            case 2: c->stats.shade_ms += ms; break;
            case 4: c->stats.scan_ms += ms; break;
            default: c->stats.accumulate_ms += ms; break;
        }
        c->event_pool.push_back(ep);
    }
    c->events.clear();
    return PRT_OK;
}

void to_dev_mat(const float* m16, float* m12) {
    for (int col = 0; col < 4; ++col)
        for (int r = 0; r < 3; ++r) m12[col * 3 + r] = m16[col * 4 + r];
}

// glm-order helpers for the camera basis (Camera::Camera, src/core/camera.h:10-16)
f3 h_normalize(f3 v) {
    const float d = (v.x * v.x + v.y * v.y) + v.z * v.z;
    const float s = 1.0f / std::sqrt(d);
    return f3{v.x * s, v.y * s, v.z * s};
}
f3 h_cross(f3 a, f3 b) { return f3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }

// One batch of S_cur samples: raygen -> (intersect, shade) x max_depth -> [accumulate]
int run_batch(PrtContext* c, uint32_t S_cur, uint32_t max_depth, uint32_t seed, uint32_t first_sample, bool accumulate,
              unsigned long long* trav_stats) {
    const uint64_t n_paths64 = (uint64_t)S_cur * c->tm.n_pix_local;
    if (n_paths64 == 0) return PRT_OK;
    if (n_paths64 > 0xFFFFFF00ull) return fail(c, PRT_ERR_INVALID, "too many paths in flight");
    const uint32_t n_paths = (uint32_t)n_paths64;
    int rc = ensure_path_state(c, n_paths);
    if (rc) return rc;
    const int stack_depth = c->bvh.max_depth <= 31 ? 31 : 63;
    if ((rc = ensure_spill(c))) return rc;
    // k_shade shades one analytic-only segment in place per call (never stored, never re-read: shade -24 % on C3) when
    // the scene has a BVH and few analytic primitives; with many of them (RANDOM_BALLS presets) compacting between
    // bounces is the better deal
    const uint32_t fuse = (c->dsc.n_nodes && c->dsc.n_prims <= 16u) ? c->tune.fuse : 0u;
    // The ray count of a bounce is only known on the device.  With big batches a k_shade grid sized for the worst case is
    // a million blocks, most of which find nothing to do (~0.5 ms per launch, 4 % of a C3 step).  The host therefore
    // reads the counts of bounce d back WHILE the traversal kernel of bounce d runs (the copy is enqueued right after
    // the producer that wrote them, the host waits for it after enqueueing the traversal): the GPU never waits for the
    // host, k_shade gets an exact grid, and a batch stops at the first bounce without rays.
    const bool exact = c->dsc.n_nodes != 0u && c->tune.exact_grids != 0u && (n_paths > (16384u * 512u) || c->tune.exact_grids == 2u);  // 2: always (tests)
    if (exact && !c->h_counts) {
        HIPCHECK(c, hipHostMalloc((void**)&c->h_counts, (PRT_MAX_DEPTH + 2) * 64 * sizeof(uint32_t), hipHostMallocDefault));
        for (hipEvent_t& e : c->ev_counts) HIPCHECK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t& e : c->ev_prod) HIPCHECK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPCHECK(c, hipStreamCreateWithFlags(&c->count_stream, hipStreamNonBlocking));
    }
    auto read_back = [&](uint32_t d) -> hipError_t {  // counts of bounce d: words 0 (front) and 32 (back) of its stride
        // (on a stream of its own behind an event: in the render stream the 4-us copy kernel and the gaps around it were
        // 10-20 us per bounce, 1 % of a step of a rank of eight)
        hipError_t e = hipEventRecord(c->ev_prod[d], c->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->count_stream, c->ev_prod[d], 0);
        if (e == hipSuccess)
            e = hipMemcpyAsync(c->h_counts + 64 * (size_t)d, c->d_counts + (size_t)d * PRT_CNT_STRIDE, 33 * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, c->count_stream);
        return e != hipSuccess ? e : hipEventRecord(c->ev_counts[d], c->count_stream);
    };
    EventPair ep{};
    // Small batches (the reference's contract: ONE sample per ProgressiveRender call, cpu/renderer.cpp:49) can run as one
    // launch of the PATH instance of the traversal kernel, which carries whole paths (prt_kernels.h PrtPathArgs), instead
    // of raygen + 2 x max_depth launches.  Same arithmetic, same draws, same rad[] / k_accumulate: the frame is
    // bit-identical (tests run both routes).  OFF by default (prt_set_param("path_kernel", 1 | 2)): measured, it ties with
    // the pipeline up to ~250 k paths per call and loses above (profiles/r3_path_kernel.txt, TUNING.md): both are bound
    // by a path's chain of dependent node fetches, and the pipeline shades with full waves.
    const bool path_route = c->tune.path_kernel != 0u && (c->tune.path_kernel == 2u || S_cur == 1u) && n_paths <= c->tune.path_max &&
                            !trav_stats && !c->d_shade_div && c->variant == 0 && fuse == 0u && c->sort_rays == 0u &&
                            prt_path_kernel_applies(c->dsc, c->tune);
    if (path_route) {
        HIPCHECK(c, hipMemsetAsync(c->d_work, 0, 256 * sizeof(uint32_t), c->stream));  // the cursors; the error flags in [256] stay for prt_synchronize
        PrtPathArgs pa{c->cam, c->tm, c->sampling, c->d_rad, first_sample, seed, max_depth, n_paths};
        if ((rc = begin_event(c, 1, &ep))) return rc;
        prt_launch_path(c->stream, c->dsc, pa, c->d_work, c->tune);
        if ((rc = end_event(c, &ep))) return rc;
        ++c->stats.intersect_launches;
        if ((rc = begin_event(c, 3, &ep))) return rc;
        prt_launch_accumulate(c->stream, c->d_rad, c->d_film_local, c->tm, S_cur, max_depth, accumulate, c->ray_stats_target, nullptr);
        if ((rc = end_event(c, &ep))) return rc;
        if (accumulate) c->stats.samples += S_cur;
        HIPCHECK(c, hipGetLastError());
        return PRT_OK;
    }
    // front/back counters of every bounce start at zero (the producers add to them atomically)
    HIPCHECK(c, hipMemsetAsync(c->d_counts, 0, (size_t)(max_depth + 1) * PRT_CNT_STRIDE * sizeof(uint32_t), c->stream));
    // compact primary rays (PrtPrimary): the default pipeline without jitter / roulette / clamp / fusion
    const bool compact = c->compact_primary && c->variant == 0 && c->dsc.n_nodes != 0u && !c->dsc.abvh_nodes &&
                         c->sampling.jitter == 0u && c->sampling.rr_depth == 0u && !(c->sampling.clamp > 0.0f) && fuse == 0u &&
                         prt_traverse_takes_primary(c->dsc, c->tune);
    if (compact && c->pix_entries < c->tm.n_pix_local) {
        free_dev(c->d_pix);
        c->pix_entries = 0;
        HIPCHECK(c, hipMalloc((void**)&c->d_pix, 4 * (size_t)c->tm.n_pix_local * sizeof(float4)));
        c->pix_entries = c->tm.n_pix_local;
        c->pix_records_blank = false;
    }
    const PrtPrimary primary{(const uint32_t*)c->rb[0].t, c->d_pix, {c->cam.pos.x, c->cam.pos.y, c->cam.pos.z}, c->tm.n_pix_local,
                             1.0f / (float)c->tm.n_pix_local, first_sample, seed};
    if ((rc = begin_event(c, 0, &ep))) return rc;
    prt_launch_raygen(c->stream, c->dsc, c->cam, c->tm, n_paths, first_sample, seed, c->rb[0], c->d_rad, c->d_counts,
                      c->d_work, max_depth, c->sampling, compact ? c->d_pix : nullptr);
    if ((rc = end_event(c, &ep))) return rc;
    if (exact) HIPCHECK(c, read_back(0));
    for (uint32_t d = 0; d < max_depth; ++d) {
        const PrtRayBuf& in = c->rb[d & 1];
        const PrtRayBuf& out = c->rb[(d + 1) & 1];
        const uint32_t* front_count = c->d_counts + (size_t)d * PRT_CNT_STRIDE;
        if (c->dsc.n_nodes) {  // only the front part of the buffer can hit a triangle
            if ((rc = begin_event(c, 1, &ep))) return rc;
            if (c->variant == 0 || c->dsc.n_insts || !c->dsc.nodes) {  // placed copies / device-built trees: the 8-wide kernel only
                PrtTravTuning tune = c->tune;
                tune.probe_slot = d;
                tune.perm = nullptr;
                if (c->sort_rays && !(compact && d == 0) && (d > 0 || c->sampling.jitter)) {
                    // measurement aid (tools/sort_ab.py): the host needs the ray count, so this path synchronises; the sort
                    // is timed as its own stage (scan_ms), the traversal that follows as usual
                    uint32_t n_front = 0;
                    HIPCHECK(c, hipMemcpyAsync(&n_front, front_count, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
                    HIPCHECK(c, hipStreamSynchronize(c->stream));
                    if (n_front > 1u) {
                        if (c->sort_entries < n_paths) {
                            free_dev(c->d_sort);
                            c->sort_entries = 0;
                            c->sort_temp = prt_sort_rays_temp_bytes(n_paths);
                            HIPCHECK(c, hipMalloc((void**)&c->d_sort, 4 * (size_t)n_paths * sizeof(uint32_t) + c->sort_temp));
                            c->sort_entries = n_paths;
                        }
                        EventPair es{};
                        if ((rc = begin_event(c, 4, &es))) return rc;
                        uint32_t* q = c->d_sort;
                        if (prt_sort_rays(c->stream, in.o, in.d, n_front, c->dsc.root_min, c->dsc.root_max, c->sort_rays, q, q + n_paths,
                                          q + 2 * (size_t)n_paths, q + 3 * (size_t)n_paths, q + 4 * (size_t)n_paths, c->sort_temp))
                            return fail(c, PRT_ERR_HIP, "ray sort failed");
                        if ((rc = end_event(c, &es))) return rc;
                        if (c->timing) HIPCHECK(c, hipEventRecord(ep.a, c->stream));  // the traversal's own time starts after the sort
                        tune.perm = q + 3 * (size_t)n_paths;
                    }
                }
                prt_launch_traverse(c->stream, c->dsc, in, front_count, c->d_work, c->d_spill, n_paths, c->bvh.max_depth,
                                    c->bvh.max_stack4, tune, trav_stats, (compact && d == 0) ? &primary : nullptr);
            }
            else
                prt_launch_intersect(c->stream, c->dsc, in, front_count, n_paths, stack_depth, c->variant, trav_stats);
            if ((rc = end_event(c, &ep))) return rc;
            ++c->stats.intersect_launches;
            if (compact && d == 0) {
                // (a batch of ONE sample has nothing to share: k_shade rebuilds the hit itself, one launch less)
                if (c->tune.primary_hit && S_cur > 1u) {  // (timed with the shade stage)
                    if ((rc = begin_event(c, 2, &ep))) return rc;
                    prt_launch_primary_hit(c->stream, c->dsc, primary, in.hit, c->d_pix, c->d_counts);
                    if ((rc = end_event(c, &ep))) return rc;
                    c->pix_records_blank = false;
                } else if (!c->pix_records_blank) {  // "no record" for every pixel (hit id 0xFFFFFFFF never equals a hit k_shade looks up); stays so until k_primary_hit runs again
                    HIPCHECK(c, hipMemsetAsync(c->d_pix + 2 * (size_t)c->tm.n_pix_local, 0xFF, 2 * (size_t)c->tm.n_pix_local * sizeof(float4), c->stream));
                    c->pix_records_blank = true;
                }
            }
        }
        uint32_t n_rays_known = 0xFFFFFFFFu;
        if (exact) {
            HIPCHECK(c, hipEventSynchronize(c->ev_counts[d]));
            n_rays_known = c->h_counts[64 * (size_t)d] + c->h_counts[64 * (size_t)d + 32];
            if (n_rays_known == 0u) break;  // every path has ended: the later bounces have nothing to do
        }
        if (c->d_shade_div) prt_launch_shade_divstats(c->stream, c->dsc, in, c->d_counts, d, n_paths, c->d_shade_div, (compact && d == 0) ? &primary : nullptr);
        if ((rc = begin_event(c, 2, &ep))) return rc;
        prt_launch_shade(c->stream, c->dsc, in, out, c->d_rad, c->d_counts, c->d_work, d, max_depth, n_paths, fuse, c->sampling,
                         n_rays_known, (compact && d == 0) ? &primary : nullptr);
        if ((rc = end_event(c, &ep))) return rc;
        if (exact && d + 1 < max_depth) HIPCHECK(c, read_back(d + 1));
    }
    // film += the batch's samples (unless this is a measurement run) and per-depth ray counts from the paths' last
    // segment indices
    if ((rc = begin_event(c, 3, &ep))) return rc;
    prt_launch_accumulate(c->stream, c->d_rad, c->d_film_local, c->tm, S_cur, max_depth, accumulate, c->ray_stats_target,
                          compact ? c->d_pix + c->tm.n_pix_local : nullptr);
    if ((rc = end_event(c, &ep))) return rc;
    if (accumulate) c->stats.samples += S_cur;
    HIPCHECK(c, hipGetLastError());
    return PRT_OK;
}

// Second half of prt_set_scene / prt_clone_scene: the context's host copies -> its device.  `gb` (device-side build):
// the builder's arrays on this device become the scene's arrays; otherwise trees and triangle records are uploaded from
// the host copies (a scene built on ANOTHER device arrives that way too: its 8-wide tree and records were read back).
int upload_scene(PrtContext* c, PrtGpuBvh* gb) {
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    free_scene(c);
    if (gb) {  // owned by the context from here on (free_scene)
        c->d_nodes8 = gb->d_nodes8;
        c->d_tris = gb->d_tris;
        c->d_nrms = gb->d_nrms;
    }
    // Node slots.  An 80-B node at a 16-B aligned address lies across two 128-B lines half of the time, so a visit that
    // misses the caches moves 1.5 lines = 192 B (tools/gather_calib.hip).  Trees far beyond the L2s (C5: 1.6 M nodes =
    // 131 MB, HBM-bound traversal) get one node per 128-B line instead: every miss is one line, at 1.6x the array size;
    // trees the caches hold (C3: 16 MB) stay packed, where the smaller footprint is worth more than the line count.
    const std::vector<uint32_t>& n8 = c->nodes8_all.empty() ? c->bvh.nodes8 : c->nodes8_all;
    const size_t n_nodes8 = gb ? (size_t)gb->n_nodes : n8.size() / 20;
    c->dsc.node_stride = c->node_stride ? (uint32_t)c->node_stride : (n_nodes8 * 80 > ((size_t)64 << 20) ? 8u : 5u);
    if (c->dsc.node_stride == 8u && n_nodes8) {
        void* wide = nullptr;
        HIPCHECK(c, hipMalloc(&wide, 128 * n_nodes8));
        hipError_t e = hipMemset(wide, 0, 128 * n_nodes8);
        if (e == hipSuccess)
            e = gb ? hipMemcpy2D(wide, 128, c->d_nodes8, 80, 80, n_nodes8, hipMemcpyDeviceToDevice)
                   : hipMemcpy2D(wide, 128, n8.data(), 80, 80, n_nodes8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipDeviceSynchronize();  // (device-to-device copies on the null stream return early; renders use c->stream)
        if (e != hipSuccess) {
            (void)hipFree(wide);
            HIPCHECK(c, e);
        }
        free_dev(c->d_nodes8);
        c->d_nodes8 = wide;
    }
    auto upload = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) return e;
        if (bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    std::vector<float> rgbs(4 * c->materials.size());
    std::vector<uint32_t> mtype(c->materials.size());
    for (size_t i = 0; i < c->materials.size(); ++i) {
        rgbs[4 * i + 0] = c->materials[i].rgb[0];
        rgbs[4 * i + 1] = c->materials[i].rgb[1];
        rgbs[4 * i + 2] = c->materials[i].rgb[2];
        rgbs[4 * i + 3] = c->materials[i].scalar;
        mtype[i] = c->materials[i].type;
    }
    HIPCHECK(c, upload(&c->d_prims, c->prims.data(), c->prims.size() * sizeof(DevPrim)));
    HIPCHECK(c, upload(&c->d_mat_rgbs, rgbs.data(), rgbs.size() * 4));
    HIPCHECK(c, upload(&c->d_mat_type, mtype.data(), mtype.size() * 4));
    if (!c->scene_device_built) {  // (device-built scenes have no binary / 4-wide tree: null pointers select the 8-wide kernel)
        HIPCHECK(c, upload(&c->d_nodes, c->bvh.nodes.data(), c->bvh.nodes.size() * 4));
        HIPCHECK(c, upload(&c->d_nodes4, c->bvh.nodes4.data(), c->bvh.nodes4.size() * 4));
    }
    if (!gb && !n8.empty() && c->dsc.node_stride != 8u) HIPCHECK(c, upload(&c->d_nodes8, n8.data(), n8.size() * 4));
    if (!c->abvh.nodes4.empty()) {
        HIPCHECK(c, upload(&c->d_abvh_nodes, c->abvh.nodes4.data(), c->abvh.nodes4.size() * 4));
        HIPCHECK(c, upload(&c->d_abvh_order, c->abvh.order.data(), c->abvh.order.size() * 4));
    }
    if (!c->dev_insts.empty()) {
        HIPCHECK(c, upload(&c->d_insts, c->dev_insts.data(), c->dev_insts.size() * sizeof(DevInstance)));
        HIPCHECK(c, upload(&c->d_tlas_inst, c->tlas_inst.data(), c->tlas_inst.size() * 4));
    }
    if (!gb) {
        HIPCHECK(c, upload(&c->d_tris, c->tri_records.data(), c->tri_records.size() * 4));
        HIPCHECK(c, upload(&c->d_nrms, c->nrm_records.data(), c->nrm_records.size() * 4));
    }
    DevScene& d = c->dsc;
    d.prims = (const DevPrim*)c->d_prims;
    d.mat_rgbs = (const float4*)c->d_mat_rgbs;
    d.mat_type = (const uint32_t*)c->d_mat_type;
    d.nodes = (const float4*)c->d_nodes;
    d.nodes4 = (const float4*)c->d_nodes4;
    d.abvh_nodes = (const float4*)c->d_abvh_nodes;
    d.abvh_order = (const uint32_t*)c->d_abvh_order;
    d.depth8 = c->bvh_info.depth8;
    d.insts = (const DevInstance*)c->d_insts;
    d.tlas_inst = (const uint32_t*)c->d_tlas_inst;
    d.nodes8 = (const uint4*)c->d_nodes8;  // null when the tree has no compressed 8-wide form: the 4-wide kernel runs
    d.tris = (const float4*)c->d_tris;
    d.tri_normals = (const float4*)c->d_nrms;
    const int rc_cnt = ensure_counters(c);
    if (rc_cnt) return rc_cnt;
    c->has_scene = true;
    return PRT_OK;
}

// Device-side build (csrc/bvh_gpu.hip) of the 8-wide tree over n triangles given as 9 floats each (+ normals, + a material
// per triangle): nodes and the triangle / normal records in the tree's slot order come back to the host copies the
// rest of prt_set_scene works with; with `keep` the device arrays stay allocated and are handed to the caller.
constexpr int kDeviceBuildGaveUp = -1000;  // internal: not a PRT_ERR_* code
int device_build(PrtContext* c, const float* verts, const float* norms, const uint32_t* tri_mat, uint32_t n, uint32_t n_prims,
                 std::vector<uint32_t>& nodes8, uint32_t& depth, float* tri_rec, float* nrm_rec, PrtGpuBvh* keep, float leaf_cost = 0.0f) {
    float cmin[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmax[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t t = 0; t < (size_t)n; ++t)
        for (int a = 0; a < 3; ++a) {
            const float* v = &verts[9 * t];
            const float lo = std::min(v[a], std::min(v[3 + a], v[6 + a])), hi = std::max(v[a], std::max(v[3 + a], v[6 + a]));
            const float cc = 0.5f * lo + 0.5f * hi;
            cmin[a] = std::min(cmin[a], cc);
            cmax[a] = std::max(cmax[a], cc);
        }
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    void *dv = nullptr, *dn = nullptr, *dm = nullptr;
    auto drop = [&]() {
        (void)hipFree(dv);
        (void)hipFree(dn);
        (void)hipFree(dm);
    };
    hipError_t e = hipMalloc(&dv, 36 * (size_t)n);
    if (e == hipSuccess) e = hipMalloc(&dn, 36 * (size_t)n);
    if (e == hipSuccess) e = hipMalloc(&dm, 4 * (size_t)n);
    if (e == hipSuccess) e = hipMemcpy(dv, verts, 36 * (size_t)n, hipMemcpyHostToDevice);
    // (fills on the stream the builder's kernels run on: c->stream is non-blocking, nothing orders it after the null stream)
    if (e == hipSuccess) e = norms ? hipMemcpy(dn, norms, 36 * (size_t)n, hipMemcpyHostToDevice) : hipMemsetAsync(dn, 0, 36 * (size_t)n, c->stream);
    if (e == hipSuccess) e = tri_mat ? hipMemcpy(dm, tri_mat, 4 * (size_t)n, hipMemcpyHostToDevice) : hipMemsetAsync(dm, 0, 4 * (size_t)n, c->stream);
    if (e != hipSuccess) {
        drop();
        return fail(c, PRT_ERR_HIP, "device-side BVH build: %s", hipGetErrorString(e));
    }
    PrtGpuBvh gb{};
    const auto t0 = std::chrono::steady_clock::now();
    // gpu_build 1: the quality builder (PLOC + optimal collapse); 2: the Morton octree (fastest build, slower to traverse)
    const int brc = c->gpu_build == 2
                        ? prt_gpu_bvh8_build(c->stream, (const float*)dv, (const float*)dn, (const uint32_t*)dm, n, n_prims, cmin, cmax, &gb)
                        : prt_gpu_bvh8_build_ploc(c->stream, (const float*)dv, (const float*)dn, (const uint32_t*)dm, n, n_prims, cmin, cmax, &gb, leaf_cost);
    c->gpu_build_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    drop();
    if (brc < 0) {  // the builder itself gave up (too many passes / levels for this input): the caller may take the host builder
        (void)fail(c, PRT_ERR_INVALID, "device-side BVH build gave up (%d)", brc);
        return kDeviceBuildGaveUp;
    }
    if (brc) return fail(c, PRT_ERR_HIP, "device-side BVH build failed (%d)", brc);
    nodes8.resize(20 * (size_t)gb.n_nodes);
    depth = gb.depth;
    e = hipMemcpy(nodes8.data(), gb.d_nodes8, nodes8.size() * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(tri_rec, gb.d_tris, 48 * (size_t)n, hipMemcpyDeviceToHost);
    if (e == hipSuccess && nrm_rec) e = hipMemcpy(nrm_rec, gb.d_nrms, 48 * (size_t)n, hipMemcpyDeviceToHost);
    if (keep && e == hipSuccess) {
        *keep = gb;
    } else {
        (void)hipFree(gb.d_nodes8);
        (void)hipFree(gb.d_tris);
        (void)hipFree(gb.d_nrms);
    }
    if (e != hipSuccess) return fail(c, PRT_ERR_HIP, "device-side BVH build: %s", hipGetErrorString(e));
    return PRT_OK;
}

int check_ready(PrtContext* c) {
    int rc = need_device(c);
    if (rc) return rc;
    if (!c->has_scene) return fail(c, PRT_ERR_INVALID, "prt_set_scene has not been called");
    if (!c->has_camera) return fail(c, PRT_ERR_INVALID, "prt_set_camera has not been called");
    if (!c->has_film) return fail(c, PRT_ERR_INVALID, "prt_set_film has not been called");
    return PRT_OK;
}

}  // namespace

extern "C" {

int prt_version(void) { return PRT_VERSION; }

int prt_create(int device_id, PrtContext** out) {
    if (!out) return PRT_ERR_INVALID;
    PrtContext* c = new PrtContext();
    *out = c;
    c->device = device_id;
    if (device_id < 0) return PRT_OK;  // host-only context
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(c, PRT_ERR_NO_DEVICE, "no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device_id >= n) return fail(c, PRT_ERR_NO_DEVICE, "device %d out of range (%d devices)", device_id, n);
    HIPCHECK(c, hipSetDevice(device_id));
    HIPCHECK(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    c->has_device = true;
    return PRT_OK;
}

void prt_destroy(PrtContext* c) {
    if (!c) return;
    if (c->has_device) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        free_scene(c);
        free_path_state(c);
        free_dev(c->d_film_local);
        free_dev(c->d_counts);
        free_dev(c->d_ray_stats);
        if (c->h_flag) (void)hipHostFree(c->h_flag);
        free_dev(c->d_trav_stats);
        if (c->h_counts) {
            (void)hipHostFree(c->h_counts);
            for (hipEvent_t& e : c->ev_counts) (void)hipEventDestroy(e);
            for (hipEvent_t& e : c->ev_prod) (void)hipEventDestroy(e);
            if (c->count_stream) (void)hipStreamDestroy(c->count_stream);
        }
        free_dev(c->d_work);
        free_dev(c->d_pix);
        free_dev(c->d_spill);
        free_dev(c->d_sort);
        free_dev(c->d_scratch);
        for (EventPair& ep : c->events) {
            (void)hipEventDestroy(ep.a);
            (void)hipEventDestroy(ep.b);
        }
        for (EventPair& ep : c->event_pool) {
            (void)hipEventDestroy(ep.a);
            (void)hipEventDestroy(ep.b);
        }
        if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    }
    delete c;
}

const char* prt_last_error(const PrtContext* c) { return c ? c->err.c_str() : "null context"; }

int prt_set_stream(PrtContext* c, void* hip_stream) {
    int rc = need_device(c);
    if (rc) return rc;
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return PRT_OK;
}

int prt_get_stream(PrtContext* c, void** hip_stream) {
    int rc = need_device(c);
    if (rc) return rc;
    if (!hip_stream) return PRT_ERR_INVALID;
    *hip_stream = (void*)c->stream;
    return PRT_OK;
}

int prt_get_device(const PrtContext* c) { return c ? c->device : -1; }

int prt_set_scene(PrtContext* c, const PrtSceneDesc* s) {
    if (!c || !s) return PRT_ERR_INVALID;
    if ((s->n_materials && !s->materials) || (s->n_primitives && !s->primitives) || (s->n_meshes && !s->meshes) ||
        (s->n_instanced_meshes && !s->instanced_meshes) || (s->n_instances && !s->instances))
        return fail(c, PRT_ERR_INVALID, "null array in scene description");
    // From here on the context's host copies and c->dsc are rewritten in place: a scene the context held before is gone
    // whatever happens, so every failure below leaves the context WITHOUT a scene (the next render fails with
    // PRT_ERR_INVALID instead of walking half-built arrays); has_scene is set again only after the last upload.
    c->has_scene = false;
    // ---- validate + flatten (host) ----
    c->materials.assign(s->materials, s->materials + s->n_materials);
    c->prims.clear();
    for (uint32_t i = 0; i < s->n_primitives; ++i) {
        const PrtPrimitive& p = s->primitives[i];
        if (p.shape_type != PRT_SHAPE_CIRCLE && p.shape_type != PRT_SHAPE_QUAD)
            return fail(c, PRT_ERR_INVALID, "primitive %u: analytic shapes are CIRCLE or QUAD (triangles come as meshes)", i);
        if (p.material_id >= s->n_materials) return fail(c, PRT_ERR_INVALID, "primitive %u: material out of range", i);
        DevPrim d;
        d.shape_type = p.shape_type;
        d.p0 = p.shape_param[0];
        d.p1 = p.shape_param[1];
        d.material = p.material_id;
        to_dev_mat(p.mat, d.mat);
        to_dev_mat(p.inv, d.inv);
        c->prims.push_back(d);
    }
    uint64_t n_tris = 0;
    for (uint32_t m = 0; m < s->n_meshes; ++m) {
        const PrtMesh& me = s->meshes[m];
        if (me.n_triangles && (!me.positions || !me.normals || !me.indices))
            return fail(c, PRT_ERR_INVALID, "mesh %u: positions, normals and indices are required", m);
        if (me.material_id >= s->n_materials) return fail(c, PRT_ERR_INVALID, "mesh %u: material out of range", m);
        n_tris += me.n_triangles;
    }
    if (n_tris >= (1ull << 26)) return fail(c, PRT_ERR_INVALID, "too many triangles (limit 2^26 - 1)");
    std::vector<float> verts(9 * (size_t)n_tris);
    std::vector<float> norms(9 * (size_t)n_tris);
    std::vector<uint32_t> tri_mat((size_t)n_tris);
    float extent = 0.0f;
    {
        size_t t = 0;
        for (uint32_t m = 0; m < s->n_meshes; ++m) {
            const PrtMesh& me = s->meshes[m];
            for (uint32_t k = 0; k < me.n_triangles; ++k, ++t) {
                for (int v = 0; v < 3; ++v) {
                    const uint32_t vi = me.indices[3 * (size_t)k + v];
                    if (vi >= me.n_vertices) return fail(c, PRT_ERR_INVALID, "mesh %u: vertex index out of range", m);
                    for (int a = 0; a < 3; ++a) {
                        const float pv = me.positions[3 * (size_t)vi + a];
                        if (!std::isfinite(pv)) return fail(c, PRT_ERR_INVALID, "mesh %u: non-finite vertex", m);
                        verts[9 * t + 3 * v + a] = pv;
                        norms[9 * t + 3 * v + a] = me.normals[3 * (size_t)vi + a];
                        extent = std::max(extent, std::fabs(pv));
                    }
                }
                tri_mat[t] = me.material_id;
            }
        }
    }
    c->mesh_sizes.clear();
    for (uint32_t m = 0; m < s->n_meshes; ++m) {
        c->mesh_sizes.push_back(s->meshes[m].n_vertices);
        c->mesh_sizes.push_back(s->meshes[m].n_triangles);
    }
    const uint32_t n_prims = (uint32_t)c->prims.size();
    // device-side build (prt_set_param("gpu_build", 1)): Morton-ordered 8-wide tree straight on the GPU (bvh_gpu.hip);
    // only for world-space meshes on a context with a device; anything else takes the host builder below
    const bool gpu_any = c->gpu_build && c->has_device;  // placed copies and the top-level tree take the device builder too
    bool gpu_build = gpu_any && n_tris > 0;
    PrtGpuBvh gb{};
    c->gpu_build_ms = 0.0;
    const auto t_build0 = std::chrono::steady_clock::now();
    if (gpu_build) {
        // host copies for the read-back entry points (prt_bvh_read / prt_bvh_read8) and for scenes whose node array is put
        // together on the host (placed copies); the binary and 4-wide trees of the A/B kernels are not built in this mode
        c->bvh = BvhBuild();
        c->bvh.max_leaf = 3;
        c->tri_records.assign(12 * (size_t)n_tris, 0.0f);
        c->nrm_records.assign(12 * (size_t)n_tris, 0.0f);
        const int brc = device_build(c, verts.data(), norms.data(), tri_mat.data(), (uint32_t)n_tris, n_prims, c->bvh.nodes8, c->bvh.depth8,
                                     c->tri_records.data(), c->nrm_records.data(), s->n_instances == 0 ? &gb : nullptr);
        if (brc && brc != kDeviceBuildGaveUp) return brc;
        if (!brc && c->bvh.depth8 > 15u) {
            (void)hipFree(gb.d_nodes8);
            (void)hipFree(gb.d_tris);
            (void)hipFree(gb.d_nrms);
        }
        // A valid mesh is never refused because the DEVICE builder could not cope with it (a tree deeper than the kernels'
        // stacks, or more clustering passes than its guard allows: degenerate inputs such as thousands of coincident
        // triangles): the host builder, with its forced median splits, takes over.
        if (brc == kDeviceBuildGaveUp || c->bvh.depth8 > 15u) {
            gb = PrtGpuBvh{};
            gpu_build = false;
            c->bvh = BvhBuild();
        }
    }
    if (!gpu_build && !bvh_build(verts.data(), (uint32_t)n_tris, kMaxLeaf, 0, kMaxStack, &c->bvh)) {
        return fail(c, PRT_ERR_INVALID, "BVH deeper than the traversal stack (%u > %u)", c->bvh.max_depth, kMaxStack);
    }
    const double build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build0).count();
    if (!gpu_build) {
        c->tri_records.assign(12 * (size_t)n_tris, 0.0f);
        c->nrm_records.assign(12 * (size_t)n_tris, 0.0f);
    }
    for (size_t slot = 0; !gpu_build && slot < (size_t)n_tris; ++slot) {
        const uint32_t t = c->bvh.order[slot];
        float* r = &c->tri_records[12 * slot];
        float* q = &c->nrm_records[12 * slot];
        for (int v = 0; v < 3; ++v)
            for (int a = 0; a < 3; ++a) {
                r[4 * v + a] = verts[9 * (size_t)t + 3 * v + a];
                q[4 * v + a] = norms[9 * (size_t)t + 3 * v + a];
            }
        const uint32_t prim = n_prims + t;  // global primitive index: analytic first, then triangles in input order
        memcpy(&r[3], &prim, 4);
        memcpy(&r[7], &tri_mat[t], 4);
    }
    PrtBvhInfo& bi = c->bvh_info;
    bi.n_nodes = (uint32_t)(c->bvh.nodes.size() / 16);
    bi.n_triangles = (uint32_t)n_tris;
    bi.max_depth = c->bvh.max_depth;
    bi.max_leaf_size = c->bvh.max_leaf;
    bi.sah_cost = c->bvh.sah_cost;
    bi.pad_abs = c->pad_coeff;
    bi.node_bytes = (uint64_t)c->bvh.nodes4.size() * 4;
    bi.n_nodes4 = (uint32_t)(c->bvh.nodes4.size() / 32);
    bi.max_stack4 = c->bvh.max_stack4;
    bi.n_nodes8 = (uint32_t)(c->bvh.nodes8.size() / 20);
    bi.depth8 = c->bvh.depth8;
    bi.build_ms = (float)(gpu_build ? c->gpu_build_ms : build_ms);
    bi.built_on_device = (gpu_build || (gpu_any && s->n_instances)) ? 1u : 0u;
    bi.refit_ms = 0.0f;
    bi.refits = 0u;
    bi.tri_bytes = (uint64_t)c->tri_records.size() * 4;

    DevScene& d = c->dsc;
    memset(&d, 0, sizeof(d));
    d.n_prims = n_prims;
    d.n_nodes = gpu_build ? (uint32_t)(c->bvh.nodes8.size() / 20) : bi.n_nodes;  // "the scene has a BVH" for the producers' classification
    d.n_tris = (uint32_t)n_tris;
    d.pad = c->pad_coeff;
    d.extent = extent;
    memcpy(d.sky, s->sky, sizeof(d.sky));
    for (int k = 0; k < 3; ++k) {
        d.root_min[k] = FLT_MAX;
        d.root_max[k] = -FLT_MAX;
    }
    for (size_t t = 0; t < (size_t)n_tris * 9; ++t) {
        d.root_min[t % 3] = std::min(d.root_min[t % 3], verts[t]);
        d.root_max[t % 3] = std::max(d.root_max[t % 3], verts[t]);
    }
    // ---- BVH over the analytic primitives (only when there are many: the reference scans all of them for every ray,
    // primitive.cpp:26; its default scene RANDOM_BALLS_LARGE has 809).  World boxes are only valid bounds of the
    // reference's hits when the primitive's transform is rotation + uniform scale + translation; one primitive that is
    // not keeps the linear scan for the whole scene. ----
    c->abvh = BvhBuild();
    float extent_prims = 0.0f;
    double quad_pad[3] = {0.0, 0.0, 0.0};
    if (c->abvh_enabled && n_prims > 16u) {
        std::vector<float> pv(9 * (size_t)n_prims);
        bool ok = true;
        for (uint32_t i = 0; i < n_prims && ok; ++i) {
            const PrtPrimitive& p = s->primitives[i];
            const float* M = p.mat;
            double g[3][3];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b)
                    g[a][b] = (double)M[4 * a] * M[4 * b] + (double)M[4 * a + 1] * M[4 * b + 1] + (double)M[4 * a + 2] * M[4 * b + 2];
            const double s2 = g[0][0];
            ok = s2 > 1e-20 && std::isfinite(s2) && M[3] == 0.0f && M[7] == 0.0f && M[11] == 0.0f && M[15] == 1.0f;
            for (int a = 0; a < 3 && ok; ++a)
                for (int b = 0; b < 3; ++b)
                    if (std::fabs(g[a][b] - (a == b ? s2 : 0.0)) > 1e-4 * s2) ok = false;
            if (!ok) break;
            float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            if (p.shape_type == PRT_SHAPE_CIRCLE) {  // sphere of radius r around the local origin
                const double R = std::fabs((double)p.shape_param[0]) * std::sqrt(s2);
                for (int a = 0; a < 3; ++a) {
                    mn[a] = (float)((double)M[12 + a] - R);
                    mx[a] = (float)((double)M[12 + a] + R);
                }
                // phantom hits of the fp32 discriminant: up to K * dist^2 / R outside the sphere, dist <= |o|_1 + |c|_1
                // (K = 1e-6: measured worst 2.0e-7 over 3.6e7 grazing rays at 3..1000 units, analytic bound 4.8e-7)
                if (R > 0.0) {
                    const double q = 1e-6 / R, c1 = std::fabs((double)M[12]) + std::fabs((double)M[13]) + std::fabs((double)M[14]);
                    quad_pad[0] = std::max(quad_pad[0], q);
                    quad_pad[1] = std::max(quad_pad[1], 2.0 * q * c1);
                    quad_pad[2] = std::max(quad_pad[2], q * c1 * c1);
                }
            } else {  // quad in the local plane y = 0
                for (int corner = 0; corner < 4; ++corner) {
                    const float lx = ((corner & 1) ? 0.5f : -0.5f) * p.shape_param[0], lz = ((corner & 2) ? 0.5f : -0.5f) * p.shape_param[1];
                    for (int a = 0; a < 3; ++a) {
                        const float wv = (M[a] * lx + M[4 + a] * 0.0f) + (M[8 + a] * lz + M[12 + a]);
                        mn[a] = std::min(mn[a], wv);
                        mx[a] = std::max(mx[a], wv);
                    }
                }
            }
            float mag = 0.0f;
            for (int a = 0; a < 3; ++a) mag = std::max(mag, std::max(std::fabs(mn[a]), std::fabs(mx[a])));
            const float slack = 1e-5f * (mag + (float)std::sqrt(s2) * (std::fabs(p.shape_param[0]) + std::fabs(p.shape_param[1]))) + 1e-30f;
            for (int a = 0; a < 3; ++a) {
                mn[a] -= slack;
                mx[a] += slack;
                if (!std::isfinite(mn[a]) || !std::isfinite(mx[a])) ok = false;
                extent_prims = std::max(extent_prims, std::max(std::fabs(mn[a]), std::fabs(mx[a])));
            }
            const float tri[9] = {mn[0], mn[1], mn[2], mx[0], mx[1], mx[2], mn[0], mx[1], mn[2]};  // spans the box
            memcpy(&pv[9 * (size_t)i], tri, sizeof(tri));
        }
        if (ok && (!bvh_build(pv.data(), n_prims, kMaxLeaf, 1, kMaxStack, &c->abvh) || c->abvh.nodes4.empty()))  // (a walk that would need more than ABVH_STACK entries falls back to the scan)
            ok = false;
        if (!ok) c->abvh = BvhBuild();
    }
    if (!c->abvh.nodes4.empty()) {
        extent = std::max(extent, extent_prims);  // the culling pad of the primitive walk scales with the scene
        d.extent = extent;
        for (int k = 0; k < 3; ++k) d.abvh_q[k] = (float)(quad_pad[k] * 1.0000002);  // (rounded up)
    }

    // ---- placed mesh copies (PrtInstance): one tree per instanced mesh in its own space + a top-level tree over the
    // copies' world boxes; the world-space meshes above become one identity instance ----
    c->nodes8_all.clear();
    c->dev_insts.clear();
    c->tlas_inst.clear();
    if (s->n_instances) {
        const uint32_t n_world = (uint32_t)n_tris;
        if (n_world && c->bvh.nodes8.empty()) return fail(c, PRT_ERR_INVALID, "instances need the 8-wide tree (leaves <= 3 triangles)");
        struct Blas {
            BvhBuild bvh;
            uint32_t slot_base = 0, node_base = 0, n_tris = 0;
            float mn[3], mx[3], extent = 0.0f;
        };
        std::vector<Blas> blas(s->n_instanced_meshes);
        uint64_t slots = n_world;
        for (uint32_t m = 0; m < s->n_instanced_meshes; ++m) {
            const PrtMesh& me = s->instanced_meshes[m];
            if (!me.n_triangles || !me.positions || !me.normals || !me.indices)
                return fail(c, PRT_ERR_INVALID, "instanced mesh %u: positions, normals and indices are required", m);
            Blas& B = blas[m];
            B.n_tris = me.n_triangles;
            std::vector<float> v(9 * (size_t)me.n_triangles), nn(9 * (size_t)me.n_triangles);
            for (int a = 0; a < 3; ++a) {
                B.mn[a] = FLT_MAX;
                B.mx[a] = -FLT_MAX;
            }
            for (uint32_t k = 0; k < me.n_triangles; ++k)
                for (int vv = 0; vv < 3; ++vv) {
                    const uint32_t vi = me.indices[3 * (size_t)k + vv];
                    if (vi >= me.n_vertices) return fail(c, PRT_ERR_INVALID, "instanced mesh %u: vertex index out of range", m);
                    for (int a = 0; a < 3; ++a) {
                        const float pv = me.positions[3 * (size_t)vi + a];
                        if (!std::isfinite(pv)) return fail(c, PRT_ERR_INVALID, "instanced mesh %u: non-finite vertex", m);
                        v[9 * (size_t)k + 3 * vv + a] = pv;
                        nn[9 * (size_t)k + 3 * vv + a] = me.normals[3 * (size_t)vi + a];
                        B.mn[a] = std::min(B.mn[a], pv);
                        B.mx[a] = std::max(B.mx[a], pv);
                        B.extent = std::max(B.extent, std::fabs(pv));
                    }
                }
            B.slot_base = (uint32_t)slots;
            slots += me.n_triangles;
            if (slots >= (1ull << 26)) return fail(c, PRT_ERR_INVALID, "too many triangles (limit 2^26 - 1)");
            // triangle / normal records in this mesh's leaf order: {P0, face index}, {P1, -}, {P2, -}
            c->tri_records.resize(12 * (size_t)slots, 0.0f);
            c->nrm_records.resize(12 * (size_t)slots, 0.0f);
            if (gpu_any) {  // the mesh's tree in its own space on the device; the records come back in its slot order
                const int brc = device_build(c, v.data(), nn.data(), nullptr, me.n_triangles, 0u, B.bvh.nodes8, B.bvh.depth8,
                                             &c->tri_records[12 * (size_t)B.slot_base], &c->nrm_records[12 * (size_t)B.slot_base], nullptr);
                if (brc && brc != kDeviceBuildGaveUp) return brc;
                if (!brc) continue;
                B.bvh = BvhBuild();  // the device builder gave up on this mesh: the host builder takes it
            }
            if (!bvh_build(v.data(), me.n_triangles, kMaxLeaf, 0, kMaxStack, &B.bvh) || B.bvh.nodes8.empty())
                return fail(c, PRT_ERR_INVALID, "instanced mesh %u: BVH construction failed", m);
            for (uint32_t sl = 0; sl < me.n_triangles; ++sl) {
                const uint32_t t = B.bvh.order[sl];
                float* r = &c->tri_records[12 * ((size_t)B.slot_base + sl)];
                float* q = &c->nrm_records[12 * ((size_t)B.slot_base + sl)];
                for (int vv = 0; vv < 3; ++vv)
                    for (int a = 0; a < 3; ++a) {
                        r[4 * vv + a] = v[9 * (size_t)t + 3 * vv + a];
                        q[4 * vv + a] = nn[9 * (size_t)t + 3 * vv + a];
                    }
                memcpy(&r[3], &t, 4);
            }
        }
        // the instance table: [identity instance of the world-space meshes] + the placed copies
        auto identity12 = [](float* m12) {
            for (int k = 0; k < 12; ++k) m12[k] = 0.0f;
            m12[0] = m12[4] = m12[8] = 1.0f;
        };
        std::vector<std::array<float, 6>> boxes;
        if (n_world) {
            DevInstance I{};
            identity12(I.mat);
            identity12(I.inv);
            I.root = 0;  // fixed up below
            I.slot_base = 0;
            I.prim_base = 0;  // the world triangles' records carry their full primitive index
            I.virt_base = 0;
            I.material = 0xFFFFFFFFu;  // per triangle record
            I.n_tris = n_world;
            I.inv_scale = 1.0f;
            I.extent = extent;
            c->dev_insts.push_back(I);
            boxes.push_back({d.root_min[0], d.root_min[1], d.root_min[2], d.root_max[0], d.root_max[1], d.root_max[2]});
        }
        uint32_t virt = n_world, prim = n_prims + n_world;
        for (uint32_t i = 0; i < s->n_instances; ++i) {
            const PrtInstance& pi = s->instances[i];
            if (pi.mesh >= s->n_instanced_meshes) return fail(c, PRT_ERR_INVALID, "instance %u: mesh out of range", i);
            if (pi.material_id >= s->n_materials) return fail(c, PRT_ERR_INVALID, "instance %u: material out of range", i);
            // rotation + uniform scale + translation only: transpose(M3) * M3 = s^2 * I, and inv * mat = I
            const float* M = pi.mat;
            double g[3][3];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b)
                    g[a][b] = (double)M[4 * a] * M[4 * b] + (double)M[4 * a + 1] * M[4 * b + 1] + (double)M[4 * a + 2] * M[4 * b + 2];
            const double s2 = g[0][0];
            bool ok = s2 > 1e-20 && std::isfinite(s2);
            for (int a = 0; a < 3 && ok; ++a)
                for (int b = 0; b < 3; ++b)
                    if (std::fabs(g[a][b] - (a == b ? s2 : 0.0)) > 1e-4 * s2) ok = false;
            for (int r = 0; r < 4 && ok; ++r)
                for (int cc = 0; cc < 4; ++cc) {
                    double acc = 0.0;
                    for (int kk = 0; kk < 4; ++kk) acc += (double)pi.inv[4 * kk + r] * (double)pi.mat[4 * cc + kk];
                    if (std::fabs(acc - (r == cc ? 1.0 : 0.0)) > 1e-3) ok = false;
                }
            if (!ok || M[3] != 0.0f || M[7] != 0.0f || M[11] != 0.0f || M[15] != 1.0f)
                return fail(c, PRT_ERR_INVALID,
                            "instance %u: the transform must be rotation + uniform scale + translation with inv = inverse(mat) "
                            "(the reference's local ray, primitive.cpp:29-30, is only a ray transform for those)", i);
            const Blas& B = blas[pi.mesh];
            DevInstance I{};
            to_dev_mat(pi.mat, I.mat);
            to_dev_mat(pi.inv, I.inv);
            I.slot_base = B.slot_base;
            I.prim_base = prim;
            I.virt_base = virt;
            I.material = pi.material_id;
            I.n_tris = B.n_tris;
            I.inv_scale = (float)(1.0 / std::sqrt(s2));
            I.extent = B.extent;
            I.root = pi.mesh;  // mesh index for now; node base below
            c->dev_insts.push_back(I);
            virt += B.n_tris;
            prim += B.n_tris;
            // world box: the 8 corners of the mesh box through Mat, widened by a relative slack for the fp32 rounding
            // of Mat * p anywhere inside the box
            std::array<float, 6> bx{FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
            float mag = 0.0f;
            for (int corner = 0; corner < 8; ++corner) {
                const float p3[3] = {(corner & 1) ? B.mx[0] : B.mn[0], (corner & 2) ? B.mx[1] : B.mn[1], (corner & 4) ? B.mx[2] : B.mn[2]};
                for (int a = 0; a < 3; ++a) {
                    const float wv = (M[a] * p3[0] + M[4 + a] * p3[1]) + (M[8 + a] * p3[2] + M[12 + a]);
                    bx[a] = std::min(bx[a], wv);
                    bx[3 + a] = std::max(bx[3 + a], wv);
                    mag = std::max(mag, std::fabs(wv));
                }
            }
            for (int a = 0; a < 3; ++a) {
                bx[a] -= 1e-5f * (mag + 1e-30f);
                bx[3 + a] += 1e-5f * (mag + 1e-30f);
            }
            boxes.push_back(bx);
        }
        if ((uint64_t)virt + n_prims >= 0xFFFFFFF0ull) return fail(c, PRT_ERR_INVALID, "too many placed triangles");
        // top-level tree: the same builder over one degenerate "triangle" per instance that spans its world box
        const uint32_t n_inst_total = (uint32_t)c->dev_insts.size();
        std::vector<float> pv(9 * (size_t)n_inst_total);
        for (uint32_t i = 0; i < n_inst_total; ++i) {
            const std::array<float, 6>& bx = boxes[i];
            const float tri[9] = {bx[0], bx[1], bx[2], bx[3], bx[4], bx[5], bx[0], bx[4], bx[2]};
            memcpy(&pv[9 * (size_t)i], tri, sizeof(tri));
        }
        BvhBuild top;
        if (gpu_any) {  // the same device builder over the copies' boxes; a record's primitive index is the instance it stands for
            std::vector<float> rec(12 * (size_t)n_inst_total);
            // (an instance in a hit leaf is ENTERED, a level switch of ~150 instructions, without a box test of its own:
            // a leaf cost this high makes the optimisation put every instance into a leaf of its own wherever the boxes
            // differ; copies whose boxes coincide may still share a leaf, which costs a redundant entry, never a result)
            const int brc = device_build(c, pv.data(), nullptr, nullptr, n_inst_total, 0u, top.nodes8, top.depth8, rec.data(), nullptr, nullptr, 64.0f);
            if (brc == kDeviceBuildGaveUp) return fail(c, PRT_ERR_INVALID, "top-level tree: the device builder gave up; use gpu_build = 0 for this scene");
            if (brc) return brc;
            top.order.resize(n_inst_total);
            for (uint32_t sl = 0; sl < n_inst_total; ++sl) memcpy(&top.order[sl], &rec[12 * (size_t)sl + 3], 4);
        } else if (!bvh_build(pv.data(), n_inst_total, kMaxLeaf, 1, kMaxStack, &top) || top.nodes8.empty()) {
            return fail(c, PRT_ERR_INVALID, "top-level BVH construction failed");
        }
        c->tlas_inst = top.order;
        // one node array: [top level][world meshes' tree][instanced meshes' trees]; child_base / tri_base made absolute
        c->nodes8_all = top.nodes8;
        uint32_t max_blas_depth = 0;
        auto append = [&](const std::vector<uint32_t>& n8, uint32_t slot_base) -> uint32_t {
            const uint32_t node_base = (uint32_t)(c->nodes8_all.size() / 20);
            const size_t at = c->nodes8_all.size();
            c->nodes8_all.insert(c->nodes8_all.end(), n8.begin(), n8.end());
            for (size_t k = at; k < c->nodes8_all.size(); k += 20) {
                c->nodes8_all[k + 4] += node_base;
                c->nodes8_all[k + 5] += slot_base;
            }
            return node_base;
        };
        uint32_t world_root = 0;
        if (n_world) {
            world_root = append(c->bvh.nodes8, 0);
            max_blas_depth = c->bvh.depth8;
        }
        for (Blas& B : blas) {
            B.node_base = append(B.bvh.nodes8, B.slot_base);
            max_blas_depth = std::max(max_blas_depth, B.bvh.depth8);
        }
        for (size_t i = 0; i < c->dev_insts.size(); ++i) {
            DevInstance& I = c->dev_insts[i];
            I.root = (n_world && i == 0) ? world_root : blas[I.root].node_base;
        }
        if (top.depth8 + max_blas_depth > 12u)
            return fail(c, PRT_ERR_INVALID, "two-level BVH too deep for the traversal stack (%u + %u > 12)", top.depth8, max_blas_depth);
        // scene-wide quantities the producers use
        for (int a = 0; a < 3; ++a) {
            d.root_min[a] = FLT_MAX;
            d.root_max[a] = -FLT_MAX;
        }
        for (const std::array<float, 6>& bx : boxes)
            for (int a = 0; a < 3; ++a) {
                d.root_min[a] = std::min(d.root_min[a], bx[a]);
                d.root_max[a] = std::max(d.root_max[a], bx[3 + a]);
                extent = std::max(extent, std::max(std::fabs(bx[a]), std::fabs(bx[3 + a])));
            }
        d.extent = extent;
        d.n_insts = n_inst_total;
        d.n_nodes = std::max(d.n_nodes, 1u);  // "the scene has a BVH"
        d.n_tris = (uint32_t)slots;
        bi.n_nodes8 = (uint32_t)(c->nodes8_all.size() / 20);
        bi.depth8 = top.depth8 + max_blas_depth;
        bi.n_triangles = (uint32_t)slots;
        bi.tri_bytes = (uint64_t)c->tri_records.size() * 4;
        // world meshes + every placed mesh + the top level: device time, or the host builders' wall time
        bi.build_ms = gpu_any ? (float)c->gpu_build_ms
                              : (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build0).count();
    }
    c->scene_device_built = gpu_build || (gpu_any && s->n_instances != 0u);
    if (!c->has_device) {  // host-only context: BVH built, nothing to upload
        c->has_scene = true;
        return PRT_OK;
    }
    return upload_scene(c, (gpu_build && s->n_instances == 0) ? &gb : nullptr);
}

// Replicates the scene of `src` (host copies of the flattened primitives, trees and triangle records) onto the device
// of `dst` without building anything again: the multi-GPU host path builds the BVH once and clones it N - 1 times.
int prt_clone_scene(PrtContext* dst, const PrtContext* src) {
    if (!dst || !src) return PRT_ERR_INVALID;
    if (!src->has_scene) return fail(dst, PRT_ERR_INVALID, "prt_clone_scene: the source context has no scene");
    if (dst == src) return PRT_OK;
    dst->has_scene = false;
    dst->materials = src->materials;
    dst->prims = src->prims;
    dst->bvh = src->bvh;
    dst->tri_records = src->tri_records;
    dst->nrm_records = src->nrm_records;
    dst->gpu_build_ms = src->gpu_build_ms;
    dst->nodes8_all = src->nodes8_all;
    dst->dev_insts = src->dev_insts;
    dst->tlas_inst = src->tlas_inst;
    dst->abvh = src->abvh;
    dst->bvh_info = src->bvh_info;
    dst->dsc = src->dsc;  // scalar fields; every device pointer is replaced by upload_scene
    dst->scene_device_built = src->scene_device_built;
    dst->mesh_sizes = src->mesh_sizes;
    if (!dst->has_device) {
        dst->has_scene = true;
        return PRT_OK;
    }
    return upload_scene(dst, nullptr);
}

// Deforming geometry: the world-space meshes of the current scene with NEW vertex positions / normals (same vertex and
// triangle counts, same index buffers as at prt_set_scene).  The 8-wide tree keeps its topology and is refitted on the
// device (csrc/bvh_gpu.hip prt_gpu_bvh8_refit): records rewritten, boxes recomputed bottom-up, nodes re-quantized.
int prt_refit_meshes(PrtContext* c, const PrtMesh* meshes, uint32_t n_meshes) {
    int rc = need_device(c);
    if (rc) return rc;
    if (!c->has_scene) return fail(c, PRT_ERR_INVALID, "prt_set_scene has not been called");
    if (!meshes && n_meshes) return fail(c, PRT_ERR_INVALID, "null mesh array");
    if (c->dsc.n_insts) return fail(c, PRT_ERR_INVALID, "prt_refit_meshes: scenes with placed copies are rebuilt, not refitted");
    if (!c->dsc.nodes8 || c->bvh.nodes8.empty()) return fail(c, PRT_ERR_INVALID, "prt_refit_meshes needs the compressed 8-wide tree");
    if (2 * (size_t)n_meshes != c->mesh_sizes.size()) return fail(c, PRT_ERR_INVALID, "prt_refit_meshes: the scene has %zu meshes", c->mesh_sizes.size() / 2);
    uint64_t n_tris = 0;
    for (uint32_t m = 0; m < n_meshes; ++m) {
        const PrtMesh& me = meshes[m];
        if (me.n_vertices != c->mesh_sizes[2 * m] || me.n_triangles != c->mesh_sizes[2 * m + 1])
            return fail(c, PRT_ERR_INVALID, "prt_refit_meshes: mesh %u has another topology than at prt_set_scene", m);
        if (me.n_triangles && (!me.positions || !me.normals || !me.indices)) return fail(c, PRT_ERR_INVALID, "mesh %u: positions, normals and indices are required", m);
        n_tris += me.n_triangles;
    }
    if (n_tris != c->dsc.n_tris || n_tris == 0) return fail(c, PRT_ERR_INVALID, "prt_refit_meshes: triangle count mismatch");
    std::vector<float> verts(9 * (size_t)n_tris), norms(9 * (size_t)n_tris);
    float extent = 0.0f;
    {
        size_t t = 0;
        for (uint32_t m = 0; m < n_meshes; ++m) {
            const PrtMesh& me = meshes[m];
            for (uint32_t k = 0; k < me.n_triangles; ++k, ++t)
                for (int v = 0; v < 3; ++v) {
                    const uint32_t vi = me.indices[3 * (size_t)k + v];
                    if (vi >= me.n_vertices) return fail(c, PRT_ERR_INVALID, "mesh %u: vertex index out of range", m);
                    for (int a = 0; a < 3; ++a) {
                        const float pv = me.positions[3 * (size_t)vi + a];
                        if (!std::isfinite(pv)) return fail(c, PRT_ERR_INVALID, "mesh %u: non-finite vertex", m);
                        verts[9 * t + 3 * v + a] = pv;
                        norms[9 * t + 3 * v + a] = me.normals[3 * (size_t)vi + a];
                        extent = std::max(extent, std::fabs(pv));
                    }
                }
        }
    }
    // the nodes of every tree level, from the host copy of the tree (breadth-first search from the root: the builders emit
    // their nodes in different orders, none of which the refit relies on)
    const std::vector<uint32_t>& n8 = c->bvh.nodes8;
    const uint32_t n_nodes = (uint32_t)(n8.size() / 20);
    std::vector<uint32_t> level_nodes{0u}, level_start{0u, 1u};
    level_nodes.reserve(n_nodes);
    for (;;) {
        const uint32_t b = level_start[level_start.size() - 2], e = level_start.back();
        for (uint32_t li = b; li < e; ++li) {
            const uint32_t nd = level_nodes[li];
            const uint32_t imask = n8[20 * (size_t)nd + 3] >> 24, child_base = n8[20 * (size_t)nd + 4];
            const uint32_t kids = (uint32_t)__builtin_popcount(imask);
            if ((uint64_t)child_base + kids > n_nodes || level_nodes.size() + kids > n_nodes)
                return fail(c, PRT_ERR_INVALID, "prt_refit_meshes: malformed tree (node %u)", nd);
            for (uint32_t k = 0; k < kids; ++k) level_nodes.push_back(child_base + k);
        }
        if (level_nodes.size() == e) break;
        level_start.push_back((uint32_t)level_nodes.size());
    }
    if (level_nodes.size() != n_nodes) return fail(c, PRT_ERR_INVALID, "prt_refit_meshes: %zu of %u nodes reachable from the root", level_nodes.size(), n_nodes);
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    void *dv = nullptr, *dn = nullptr;
    hipError_t e = hipMalloc(&dv, 36 * (size_t)n_tris);
    if (e == hipSuccess) e = hipMalloc(&dn, 36 * (size_t)n_tris);
    if (e == hipSuccess) e = hipMemcpy(dv, verts.data(), 36 * (size_t)n_tris, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dn, norms.data(), 36 * (size_t)n_tris, hipMemcpyHostToDevice);
    float root_box[6] = {0, 0, 0, 0, 0, 0};
    int brc = 0;
    if (e == hipSuccess) {
        const auto t0 = std::chrono::steady_clock::now();
        brc = prt_gpu_bvh8_refit(c->stream, (uint32_t*)c->d_nodes8, c->dsc.node_stride * 4u, n_nodes, level_nodes.data(), level_start.data(),
                                 (uint32_t)level_start.size() - 1u, (const float*)dv, (const float*)dn, (uint32_t)n_tris, c->dsc.n_prims,
                                 (float4*)c->d_tris, (float4*)c->d_nrms, root_box);
        c->refit_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    (void)hipFree(dv);
    (void)hipFree(dn);
    if (e != hipSuccess || brc) {
        c->has_scene = false;  // the device arrays may be half rewritten
        return fail(c, PRT_ERR_HIP, "prt_refit_meshes: %s (%d)", e != hipSuccess ? hipGetErrorString(e) : "refit failed", brc);
    }
    // host copies (prt_bvh_read8 / prt_bvh_read, prt_clone_scene) follow the device
    e = hipMemcpy2D(c->bvh.nodes8.data(), 80, c->d_nodes8, (size_t)c->dsc.node_stride * 16, 80, n_nodes, hipMemcpyDeviceToHost);
    if (e == hipSuccess && c->tri_records.size() == 12 * (size_t)n_tris) e = hipMemcpy(c->tri_records.data(), c->d_tris, 48 * (size_t)n_tris, hipMemcpyDeviceToHost);
    if (e == hipSuccess && c->nrm_records.size() == 12 * (size_t)n_tris) e = hipMemcpy(c->nrm_records.data(), c->d_nrms, 48 * (size_t)n_tris, hipMemcpyDeviceToHost);
    HIPCHECK(c, e);
    // the binary and 4-wide trees (A/B kernels, overflow fallback of deep host-built trees) still describe the OLD
    // geometry: they go, and the instance selection falls to the 8-wide kernels that need neither (prt_launch_traverse)
    free_dev(c->d_nodes);
    free_dev(c->d_nodes4);
    c->dsc.nodes = nullptr;
    c->dsc.nodes4 = nullptr;
    c->bvh.nodes.clear();
    c->bvh.nodes4.clear();
    c->scene_device_built = true;
    c->variant = 0;
    for (int a = 0; a < 3; ++a) {
        c->dsc.root_min[a] = root_box[a];
        c->dsc.root_max[a] = root_box[3 + a];
    }
    float ext_all = extent;
    if (!c->abvh.nodes4.empty()) ext_all = std::max(ext_all, c->dsc.extent);  // (the primitive walk's pad scales with the larger of the two)
    c->dsc.extent = ext_all;
    c->bvh_info.refit_ms = (float)c->refit_ms;
    ++c->bvh_info.refits;
    c->bvh_info.n_nodes = 0;      // (the binary and 4-wide trees are gone)
    c->bvh_info.n_nodes4 = 0;
    c->bvh_info.node_bytes = 0;
    c->bvh_info.max_stack4 = 0;
    return PRT_OK;
}

int prt_set_camera(PrtContext* c, const PrtCameraDesc* cam) {
    if (!c || !cam) return PRT_ERR_INVALID;
    if (!(cam->width > 0.0f) || !(cam->height > 0.0f)) return fail(c, PRT_ERR_INVALID, "camera width/height must be > 0");
    DevCamera& k = c->cam;
    k.pos = f3{cam->position[0], cam->position[1], cam->position[2]};
    k.front = h_normalize(f3{cam->front[0], cam->front[1], cam->front[2]});
    k.right = h_normalize(h_cross(k.front, f3{0.0f, 1.0f, 0.0f}));
    k.up = h_normalize(h_cross(k.right, k.front));
    k.W = cam->width;
    k.H = cam->height;
    k.tan_fov_y = tanf(0.5f);  // src/core/camera.h:111
    c->has_camera = true;
    return PRT_OK;
}

int prt_set_film(PrtContext* c, uint32_t width, uint32_t height, uint32_t rank, uint32_t world) {
    if (!c) return PRT_ERR_INVALID;
    if (width == 0 || height == 0 || world == 0 || rank >= world)
        return fail(c, PRT_ERR_INVALID, "bad film size or partition (%ux%u, rank %u of %u)", width, height, rank, world);
    if ((uint64_t)width * height > 0x7FFFFFFFull) return fail(c, PRT_ERR_INVALID, "film too large");
    PrtTileMap& tm = c->tm;
    tm.W = width;
    tm.H = height;
    tm.tiles_x = (width + 7) / 8;
    tm.tiles_y = (height + 7) / 8;
    tm.rank = rank;
    tm.world = world;
    const uint32_t tiles = tm.tiles_x * tm.tiles_y;
    tm.n_tiles_local = tiles > rank ? (tiles - rank + world - 1) / world : 0;
    tm.n_pix_local = tm.n_tiles_local * 64;
    tm.stride = ((tiles + world - 1) / world) * 64;
    uint32_t valid = 0;
    for (uint32_t lt = 0; lt < tm.n_tiles_local; ++lt) {
        const uint32_t gt = lt * world + rank;
        const uint32_t tx = gt % tm.tiles_x, ty = gt / tm.tiles_x;
        const uint32_t w = std::min(8u, width - tx * 8), h = std::min(8u, height - ty * 8);
        valid += w * h;
    }
    c->valid_local = valid;
    c->has_film = true;
    c->pix_records_blank = false;  // (the records' place in d_pix depends on the pixel count)
    if (!c->has_device) return PRT_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    free_dev(c->d_film_local);
    HIPCHECK(c, hipMalloc((void**)&c->d_film_local, std::max<size_t>((size_t)tm.stride, 64) * sizeof(float4)));
    int rc = ensure_counters(c);
    if (rc) return rc;
    return prt_film_clear(c);
}

int prt_film_clear(PrtContext* c) {
    int rc = need_device(c);
    if (rc) return rc;
    if (!c->has_film) return fail(c, PRT_ERR_INVALID, "prt_set_film has not been called");
    HIPCHECK(c, hipMemsetAsync(c->d_film_local, 0, std::max<size_t>((size_t)c->tm.stride, 64) * sizeof(float4), c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return PRT_OK;
}

int prt_set_sampling(PrtContext* c, const PrtSampling* sp) {
    if (!c) return PRT_ERR_INVALID;
    if (sp && (sp->jitter > 1u || sp->rr_depth > PRT_MAX_DEPTH || !(sp->clamp >= 0.0f)))
        return fail(c, PRT_ERR_INVALID, "bad sampling options");
    c->sampling = sp ? *sp : PrtSampling{0u, 0u, 0.0f};
    return PRT_OK;
}

int prt_set_samples_in_flight(PrtContext* c, uint32_t n) {
    if (!c || n == 0 || n > 1024) return fail(c, PRT_ERR_INVALID, "samples in flight must be 1..1024");
    c->S = n;
    return PRT_OK;
}

int prt_render_async(PrtContext* c, uint32_t spp, uint32_t max_depth, uint32_t seed, uint32_t first_sample) {
    int rc = check_ready(c);
    if (rc) return rc;
    if (max_depth == 0 || max_depth > PRT_MAX_DEPTH) return fail(c, PRT_ERR_INVALID, "max_depth must be 1..%d", PRT_MAX_DEPTH);
    HIPCHECK(c, hipSetDevice(c->device));
    uint32_t done = 0;
    while (done < spp) {
        const uint32_t S_cur = std::min(c->S, spp - done);
        rc = run_batch(c, S_cur, max_depth, seed, first_sample + done, true, nullptr);
        if (rc) return rc;
        done += S_cur;
    }
    return PRT_OK;
}

int prt_synchronize(PrtContext* c) {
    int rc = need_device(c);
    if (rc) return rc;
    // (the flag's copy is enqueued behind the kernels and ONE wait covers both: a blocking copy after the wait was a second
    // host round trip, ~30 us of a 0.9 ms one-sample call)
    if (c->d_work && !c->h_flag) {
        HIPCHECK(c, hipHostMalloc((void**)&c->h_flag, 64, hipHostMallocDefault));
        *c->h_flag = 0u;
    }
    if (c->d_work) HIPCHECK(c, hipMemcpyAsync(c->h_flag, c->d_work + 256, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    if (c->d_work) {  // watchdog flag of the persistent traversal kernel
        const uint32_t w = *c->h_flag;
        if (w) {
            HIPCHECK(c, hipMemset(c->d_work, 0, 2052));
            return fail(c, PRT_ERR_HIP, w & 2u ? "traversal stack-overflow list full" : "traversal watchdog tripped: a wave exceeded its iteration cap");
        }
    }
    return PRT_OK;
}

int prt_render(PrtContext* c, uint32_t spp, uint32_t max_depth, uint32_t seed, uint32_t first_sample) {
    int rc = prt_render_async(c, spp, max_depth, seed, first_sample);
    if (rc) return rc;
    return prt_synchronize(c);
}

int prt_film_local(PrtContext* c, void** d_ptr, uint64_t* n_floats) {
    int rc = need_device(c);
    if (rc) return rc;
    if (!c->has_film) return fail(c, PRT_ERR_INVALID, "prt_set_film has not been called");
    if (d_ptr) *d_ptr = c->d_film_local;
    if (n_floats) *n_floats = (uint64_t)c->tm.stride * 4;
    return PRT_OK;
}

int prt_film_resolve_on(PrtContext* c, void* hip_stream, const void* d_gathered, uint32_t world, void* d_rgb, void* d_weight) {
    int rc = need_device(c);
    if (rc) return rc;
    if (!c->has_film || !d_gathered || !d_rgb || !d_weight || world != c->tm.world)
        return fail(c, PRT_ERR_INVALID, "bad arguments to prt_film_resolve");
    HIPCHECK(c, hipSetDevice(c->device));
    prt_launch_resolve(hip_stream ? (hipStream_t)hip_stream : c->stream, (const float4*)d_gathered, world, c->tm.stride, c->tm.W,
                       c->tm.H, (float*)d_rgb, (float*)d_weight);
    HIPCHECK(c, hipGetLastError());
    return PRT_OK;
}

int prt_film_resolve(PrtContext* c, const void* d_gathered, uint32_t world, void* d_rgb, void* d_weight) {
    return prt_film_resolve_on(c, nullptr, d_gathered, world, d_rgb, d_weight);
}

int prt_film_tonemap(PrtContext* c, const void* d_rgb, const void* d_weight, float exposure, float gamma, void* d_rgba8) {
    int rc = need_device(c);
    if (rc) return rc;
    if (!c->has_film || !d_rgb || !d_weight || !d_rgba8) return fail(c, PRT_ERR_INVALID, "bad arguments to prt_film_tonemap");
    prt_launch_tonemap(c->stream, (const float*)d_rgb, (const float*)d_weight, c->tm.W * c->tm.H, exposure, 1.0f / gamma,
                       (uint8_t*)d_rgba8);
    HIPCHECK(c, hipGetLastError());
    return PRT_OK;
}

// Builds [world][stride] with only this rank's payload filled, resolves it, and leaves rgb / weight in scratch.
static int resolve_own(PrtContext* c, float** d_rgb, float** d_w, uint8_t** d_rgba) {
    const PrtTileMap& tm = c->tm;
    const size_t npix = (size_t)tm.W * tm.H;
    const size_t gathered = (size_t)tm.world * tm.stride * sizeof(float4);
    const size_t off_rgb = (gathered + 255) & ~(size_t)255;
    const size_t off_w = off_rgb + ((npix * 12 + 255) & ~(size_t)255);
    const size_t off_b = off_w + ((npix * 4 + 255) & ~(size_t)255);
    int rc = ensure_scratch(c, off_b + npix * 4);
    if (rc) return rc;
    char* base = (char*)c->d_scratch;
    HIPCHECK(c, hipMemsetAsync(base, 0, gathered, c->stream));
    HIPCHECK(c, hipMemcpyAsync(base + (size_t)tm.rank * tm.stride * sizeof(float4), c->d_film_local,
                               (size_t)tm.stride * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    *d_rgb = (float*)(base + off_rgb);
    *d_w = (float*)(base + off_w);
    *d_rgba = (uint8_t*)(base + off_b);
    prt_launch_resolve(c->stream, (const float4*)base, tm.world, tm.stride, tm.W, tm.H, *d_rgb, *d_w);
    HIPCHECK(c, hipGetLastError());
    return PRT_OK;
}

int prt_film_read(PrtContext* c, float* rgb_sum, float* weight) {
    int rc = need_device(c);
    if (rc) return rc;
    if (!c->has_film) return fail(c, PRT_ERR_INVALID, "prt_set_film has not been called");
    float *d_rgb, *d_w;
    uint8_t* d_b;
    if ((rc = resolve_own(c, &d_rgb, &d_w, &d_b))) return rc;
    const size_t npix = (size_t)c->tm.W * c->tm.H;
    if (rgb_sum) HIPCHECK(c, hipMemcpyAsync(rgb_sum, d_rgb, npix * 12, hipMemcpyDeviceToHost, c->stream));
    if (weight) HIPCHECK(c, hipMemcpyAsync(weight, d_w, npix * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return PRT_OK;
}

int prt_film_display(PrtContext* c, float exposure, float gamma, uint8_t* rgba8) {
    int rc = need_device(c);
    if (rc) return rc;
    if (!c->has_film || !rgba8) return fail(c, PRT_ERR_INVALID, "bad arguments to prt_film_display");
    float *d_rgb, *d_w;
    uint8_t* d_b;
    if ((rc = resolve_own(c, &d_rgb, &d_w, &d_b))) return rc;
    const uint32_t npix = c->tm.W * c->tm.H;
    prt_launch_tonemap(c->stream, d_rgb, d_w, npix, exposure, 1.0f / gamma, d_b);
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, hipMemcpyAsync(rgba8, d_b, (size_t)npix * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return PRT_OK;
}

// ---- function-level entry points ------------------------------------------------------------------
int prt_camera_rays(PrtContext* c, uint32_t n, const float* px, const float* py, float* origins, float* dirs) {
    int rc = need_device(c);
    if (rc) return rc;
    if (!c->has_camera) return fail(c, PRT_ERR_INVALID, "prt_set_camera has not been called");
    if (n == 0) return PRT_OK;
    if (!px || !py || !origins || !dirs) return fail(c, PRT_ERR_INVALID, "null array");
    const size_t b1 = (size_t)n * 4, b3 = (size_t)n * 12;
    if ((rc = ensure_scratch(c, 2 * b1 + 2 * b3))) return rc;
    char* base = (char*)c->d_scratch;
    float* d_px = (float*)base;
    float* d_py = (float*)(base + b1);
    float* d_o = (float*)(base + 2 * b1);
    float* d_d = (float*)(base + 2 * b1 + b3);
    HIPCHECK(c, hipMemcpyAsync(d_px, px, b1, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(d_py, py, b1, hipMemcpyHostToDevice, c->stream));
    prt_launch_camera_rays(c->stream, c->cam, n, d_px, d_py, d_o, d_d);
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, hipMemcpyAsync(origins, d_o, b3, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(dirs, d_d, b3, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return PRT_OK;
}

int prt_closest_hit(PrtContext* c, uint32_t n, const float* origins, const float* dirs, PrtHit* hits) {
    int rc = need_device(c);
    if (rc) return rc;
    if (!c->has_scene) return fail(c, PRT_ERR_INVALID, "prt_set_scene has not been called");
    if (n == 0) return PRT_OK;
    if (!origins || !dirs || !hits) return fail(c, PRT_ERR_INVALID, "null array");
    if ((rc = ensure_path_state(c, n))) return rc;
    if ((rc = ensure_counters(c))) return rc;
    const size_t b3 = (size_t)n * 12;
    const size_t bh = (size_t)n * sizeof(PrtHit);
    if ((rc = ensure_scratch(c, 2 * b3 + bh + 64))) return rc;
    char* base = (char*)c->d_scratch;
    float* d_o = (float*)base;
    float* d_d = (float*)(base + b3);
    PrtHit* d_h = (PrtHit*)(base + ((2 * b3 + 15) & ~(size_t)15));
    HIPCHECK(c, hipMemcpyAsync(d_o, origins, b3, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(d_d, dirs, b3, hipMemcpyHostToDevice, c->stream));
    uint32_t* cnt = c->d_counts + (size_t)(PRT_MAX_DEPTH + 1) * PRT_CNT_STRIDE;  // a counter slot the render loop never uses
    prt_launch_pack_rays(c->stream, n, d_o, d_d, c->rb[0], cnt);
    const int stack_depth = c->bvh.max_depth <= 31 ? 31 : 63;
    if ((rc = ensure_spill(c))) return rc;
    prt_launch_scan_prims(c->stream, c->dsc, c->rb[0], cnt, c->d_work, n, nullptr);
    if (c->dsc.n_nodes) {
        if (c->variant == 0 || c->dsc.n_insts || !c->dsc.nodes)
            prt_launch_traverse(c->stream, c->dsc, c->rb[0], cnt, c->d_work, c->d_spill, n, c->bvh.max_depth,
                                c->bvh.max_stack4, c->tune, nullptr);
        else
            prt_launch_intersect(c->stream, c->dsc, c->rb[0], cnt, n, stack_depth, c->variant, nullptr);
    }
    prt_launch_hit_records(c->stream, c->dsc, n, c->rb[0], d_h);
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, hipMemcpyAsync(hits, d_h, bh, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return PRT_OK;
}

int prt_scatter(PrtContext* c, uint32_t n, const float* in_dirs, const PrtHit* hits, uint32_t* rng_state,
                uint32_t* scattered, float* attenuation, float* emitted, float* out_origins, float* out_dirs) {
    int rc = need_device(c);
    if (rc) return rc;
    if (!c->has_scene) return fail(c, PRT_ERR_INVALID, "prt_set_scene has not been called");
    if (n == 0) return PRT_OK;
    if (!in_dirs || !hits || !rng_state || !scattered || !attenuation || !emitted || !out_origins || !out_dirs)
        return fail(c, PRT_ERR_INVALID, "null array");
    for (uint32_t i = 0; i < n; ++i)
        if (hits[i].material_id >= c->materials.size()) return fail(c, PRT_ERR_INVALID, "hit %u: material out of range", i);
    const size_t b3 = (size_t)n * 12, b1 = (size_t)n * 4, bh = (size_t)n * sizeof(PrtHit);
    if ((rc = ensure_scratch(c, 5 * b3 + 2 * b1 + bh + 64))) return rc;
    char* base = (char*)c->d_scratch;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = base + off;
        off += (bytes + 15) & ~(size_t)15;
        return p;
    };
    PrtHit* d_h = (PrtHit*)take(bh);
    float* d_in = (float*)take(b3);
    uint32_t* d_rng = (uint32_t*)take(b1);
    uint32_t* d_sc = (uint32_t*)take(b1);
    float* d_at = (float*)take(b3);
    float* d_em = (float*)take(b3);
    float* d_oo = (float*)take(b3);
    float* d_od = (float*)take(b3);
    HIPCHECK(c, hipMemcpyAsync(d_h, hits, bh, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(d_in, in_dirs, b3, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(d_rng, rng_state, b1, hipMemcpyHostToDevice, c->stream));
    prt_launch_scatter_test(c->stream, c->dsc, n, d_in, d_h, d_rng, d_sc, d_at, d_em, d_oo, d_od);
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, hipMemcpyAsync(rng_state, d_rng, b1, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(scattered, d_sc, b1, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(attenuation, d_at, b3, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(emitted, d_em, b3, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(out_origins, d_oo, b3, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(out_dirs, d_od, b3, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return PRT_OK;
}

// ---- measurement -----------------------------------------------------------------------------------
int prt_enable_timing(PrtContext* c, int on) {
    if (!c) return PRT_ERR_INVALID;
    c->timing = on != 0;
    return PRT_OK;
}

int prt_get_stats(PrtContext* c, PrtStats* out) {
    int rc = need_device(c);
    if (rc) return rc;
    if (!out) return PRT_ERR_INVALID;
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    if ((rc = drain_events(c))) return rc;
    PrtStats s = c->stats;
    if (c->d_ray_stats) {
        unsigned long long h[PRT_MAX_DEPTH];
        if ((rc = read_ray_stats(c, c->d_ray_stats, h))) return rc;
        s.rays_total = 0;
        for (int d = 0; d < PRT_MAX_DEPTH; ++d) {
            s.rays_per_depth[d] = h[d];
            s.rays_total += h[d];
        }
    }
    *out = s;
    return PRT_OK;
}

int prt_reset_stats(PrtContext* c) {
    int rc = need_device(c);
    if (rc) return rc;
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    if ((rc = drain_events(c))) return rc;
    memset(&c->stats, 0, sizeof(c->stats));
    c->dead_paths = 0;
    if (c->d_ray_stats) HIPCHECK(c, hipMemset(c->d_ray_stats, 0, kRayStatWords * sizeof(unsigned long long)));
    return PRT_OK;
}

int prt_measure_traversal(PrtContext* c, uint32_t max_depth, uint32_t seed, uint32_t sample, PrtStats* out) {
    int rc = check_ready(c);
    if (rc) return rc;
    if (!out || max_depth == 0 || max_depth > PRT_MAX_DEPTH) return fail(c, PRT_ERR_INVALID, "bad arguments");
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    {
        std::vector<unsigned long long> init(kTravStatsWords, 0ull);
        for (uint32_t d = 0; d <= PRT_MAX_DEPTH; ++d) init[16 + PRT_TIMELINE_WORDS * d] = init[16 + PRT_TIMELINE_WORDS * d + 2] = ~0ull;  // minima
        HIPCHECK(c, hipMemcpy(c->d_trav_stats, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    }
    const bool timing = c->timing;
    const uint64_t launches = c->stats.intersect_launches;
    c->timing = false;
    // the per-depth ray counters are cumulative: this run counts into the scratch set
    unsigned long long before[PRT_MAX_DEPTH] = {}, after[PRT_MAX_DEPTH];
    HIPCHECK(c, hipMemset(c->d_ray_stats + kRayStatWords, 0, kRayStatWords * sizeof(unsigned long long)));
    c->ray_stats_target = c->d_ray_stats + kRayStatWords;
    // measure_spp (prt_set_param) samples in one batch: the counters scale, the per-phase cycle split becomes that of a
    // loaded kernel
    rc = run_batch(c, (uint32_t)std::max(1, c->measure_spp), max_depth, seed, sample, false, c->d_trav_stats);
    c->ray_stats_target = c->d_ray_stats;
    c->timing = timing;
    c->stats.intersect_launches = launches;
    if (rc) return rc;
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    if ((rc = read_ray_stats(c, c->d_ray_stats + kRayStatWords, after))) return rc;
    unsigned long long t[kTravStatsWords];
    std::vector<uint32_t> cnt((size_t)(PRT_MAX_DEPTH + 2) * PRT_CNT_STRIDE);
    HIPCHECK(c, hipMemcpy(t, c->d_trav_stats, sizeof(t), hipMemcpyDeviceToHost));
    if (getenv("PRT_TAIL_PROBE")) {  // diagnostic: where a launch of the 8-wide kernel spends its wall time (100 MHz ticks -> us)
        for (uint32_t d = 0; d < max_depth; ++d) {
            const unsigned long long* tl = t + 16 + PRT_TIMELINE_WORDS * d;
            if (tl[6] == 0ull) continue;
            fprintf(stderr,
                    "[prt tail probe] bounce %u: launch %.1f us, first wave out of rays at %.1f us, last at %.1f us, drain after "
                    "the first %.1f us; per wave: mean life %.1f us, mean drain %.1f us (%llu waves); longest ray %llu node steps\n",
                    d, (tl[1] - tl[0]) * 0.01, (tl[2] - tl[0]) * 0.01, (tl[3] - tl[0]) * 0.01, (tl[1] - tl[2]) * 0.01,
                    tl[5] * 0.01 / tl[6], tl[4] * 0.01 / tl[6], tl[6], tl[7]);
        }
    }
    HIPCHECK(c, hipMemcpy(cnt.data(), c->d_counts, cnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    memset(out, 0, sizeof(*out));
    uint64_t front = 0;
    for (uint32_t d = 0; d < max_depth; ++d) {
        front += cnt[(size_t)d * PRT_CNT_STRIDE];  // rays handed to the traversal kernel in bounce iteration d
        out->rays_per_depth[d] = after[d] - before[d];
        out->rays_total += out->rays_per_depth[d];
    }
    out->rays_traversed = front;
    out->samples = (uint64_t)std::max(1, c->measure_spp);
    out->bvh_node_visits = t[0];
    out->bvh_tri_tests = t[1];
    out->prim_tests = (uint64_t)c->prims.size() * out->rays_total;  // every ray scans every analytic primitive
    out->node_lane_slots = t[3];
    out->tri_lane_slots = t[4];
    out->max_stack_used = t[5];
    out->wave_cycles_refill = t[6];
    out->wave_cycles_node = t[7];
    out->wave_cycles_tri = t[8];
    return PRT_OK;
}

// Diagnostic: one batch of measure_spp samples (film untouched) with k_shade_divstats in front of every k_shade.
int prt_measure_shade_divergence(PrtContext* c, uint32_t max_depth, uint32_t seed, uint32_t sample, uint64_t* out) {
    int rc = check_ready(c);
    if (rc) return rc;
    if (!out || max_depth == 0 || max_depth > PRT_MAX_DEPTH) return PRT_ERR_INVALID;
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    const size_t bytes = 16 * (size_t)PRT_MAX_DEPTH * sizeof(unsigned long long);
    unsigned long long* buf = nullptr;
    HIPCHECK(c, hipMalloc((void**)&buf, bytes));
    hipError_t e = hipMemsetAsync(buf, 0, bytes, c->stream);
    if (e == hipSuccess) {
        c->d_shade_div = buf;
        c->ray_stats_target = c->d_ray_stats + kRayStatWords;  // (the run's rays are not the context's)
        const bool timing = c->timing;
        const uint64_t launches = c->stats.intersect_launches;
        c->timing = false;
        rc = run_batch(c, (uint32_t)std::max(1, c->measure_spp), max_depth, seed, sample, false, nullptr);
        c->timing = timing;
        c->stats.intersect_launches = launches;
        c->d_shade_div = nullptr;
        c->ray_stats_target = c->d_ray_stats;
        if (!rc) e = hipStreamSynchronize(c->stream);
        if (!rc && e == hipSuccess) e = hipMemcpy(out, buf, 16 * (size_t)max_depth * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    }
    (void)hipFree(buf);
    if (rc) return rc;
    HIPCHECK(c, e);
    return PRT_OK;
}

int prt_kernel_occupancy(PrtContext* c, PrtOccupancy* out) {
    int rc = need_device(c);
    if (rc) return rc;
    if (!out) return PRT_ERR_INVALID;
    if (!c->has_scene) return fail(c, PRT_ERR_INVALID, "prt_set_scene has not been called");
    int nb = 0, vg = 0, sg = 0, lds = 0;
    if (prt_traverse_occupancy(c->dsc, c->tune, &nb, &vg, &sg, &lds)) return fail(c, PRT_ERR_HIP, "occupancy query failed");
    hipDeviceProp_t prop;
    HIPCHECK(c, hipGetDeviceProperties(&prop, c->device));
    out->blocks_per_cu = (uint32_t)nb;
    out->waves_per_cu = (uint32_t)nb * 4u;  // 256-thread blocks = 4 wave64
    out->max_waves_per_cu = (uint32_t)(prop.maxThreadsPerMultiProcessor / 64);
    out->vgprs = (uint32_t)vg;
    out->lds_bytes_per_block = (uint32_t)lds;
    out->compute_units = (uint32_t)prop.multiProcessorCount;
    out->resident_grid_blocks = !strcmp(prt_traverse_instance(c->dsc, c->tune), "lean8_5waves")
                                    ? c->tune.grid_blocks + c->tune.grid_blocks / 4u : c->tune.grid_blocks;
    return PRT_OK;
}

int prt_kernel_instance(PrtContext* c, char* name, uint32_t capacity) {
    if (!c || !name || capacity == 0u) return PRT_ERR_INVALID;
    if (!c->has_scene) return fail(c, PRT_ERR_INVALID, "prt_set_scene has not been called");
    snprintf(name, capacity, "%s", prt_traverse_instance(c->dsc, c->tune));
    return PRT_OK;
}

int prt_bvh_info(PrtContext* c, PrtBvhInfo* out) {
    if (!c || !out) return PRT_ERR_INVALID;
    if (!c->has_scene) return fail(c, PRT_ERR_INVALID, "prt_set_scene has not been called");
    *out = c->bvh_info;
    return PRT_OK;
}

int prt_bvh_read4(PrtContext* c, float* nodes4) {
    if (!c) return PRT_ERR_INVALID;
    if (!c->has_scene) return fail(c, PRT_ERR_INVALID, "prt_set_scene has not been called");
    if (nodes4) memcpy(nodes4, c->bvh.nodes4.data(), c->bvh.nodes4.size() * 4);
    return PRT_OK;
}

int prt_bvh_read8(PrtContext* c, uint32_t* nodes8) {
    if (!c) return PRT_ERR_INVALID;
    if (!c->has_scene) return fail(c, PRT_ERR_INVALID, "prt_set_scene has not been called");
    const std::vector<uint32_t>& n8 = c->nodes8_all.empty() ? c->bvh.nodes8 : c->nodes8_all;
    if (nodes8) memcpy(nodes8, n8.data(), n8.size() * 4);
    return PRT_OK;
}

int prt_bvh_read(PrtContext* c, float* nodes, float* tris) {
    if (!c) return PRT_ERR_INVALID;
    if (!c->has_scene) return fail(c, PRT_ERR_INVALID, "prt_set_scene has not been called");
    if (nodes) memcpy(nodes, c->bvh.nodes.data(), c->bvh.nodes.size() * 4);
    if (tris) memcpy(tris, c->tri_records.data(), c->tri_records.size() * 4);
    return PRT_OK;
}

int prt_set_param(PrtContext* c, const char* name, int value) {
    if (!c || !name) return PRT_ERR_INVALID;
    const std::string n = name;
    if (n == "variant") c->variant = value;
    else if (n == "grid_blocks" && value > 0 && value <= 8192) c->tune.grid_blocks = (uint32_t)value;
    else if (n == "chunk" && value >= 64 && value % 64 == 0) c->tune.chunk = (uint32_t)value;
    else if (n == "xcd_affinity" && (value == 0 || value == 1)) c->tune.xcd_affinity = (uint32_t)value;
    else if (n == "wide" && (value == 0 || value == 1 || value == 2)) c->tune.wide = (uint32_t)value;
    else if (n == "stack_lds" && (value == 0 || value == 1 || value == 2 || value == 3 || value == 4 || value == 5 || value == 6 || value == 24 || value == 39)) c->tune.stack_lds = (uint32_t)value;
    else if (n == "exact_grids" && (value == 0 || value == 1 || value == 2)) c->tune.exact_grids = (uint32_t)value;
    else if (n == "steal" && value >= 0 && value <= 64) c->tune.steal = (uint32_t)value;
    else if (n == "primary_hit" && (value == 0 || value == 1)) c->tune.primary_hit = (uint32_t)value;
    else if (n == "path_kernel" && value >= 0 && value <= 2) c->tune.path_kernel = (uint32_t)value;
    else if (n == "path_max" && value >= 1) c->tune.path_max = (uint32_t)value;
    else if (n == "sort_rays" && value >= 0 && value <= 2) c->sort_rays = (uint32_t)value;
    else if (n == "compact_primary" && (value == 0 || value == 1)) c->compact_primary = value;
    else if (n == "node_stride" && (value == 0 || value == 5 || value == 8)) c->node_stride = value;
    else if (n == "pad_log2" && value >= 8 && value <= 22) c->pad_coeff = std::ldexp(1.0f, -value);
    else if (n == "tail" && value >= 0 && value <= 64) c->tune.tail = (uint32_t)value;
    else if (n == "big" && value >= 1 && value <= 16) c->tune.big = (uint32_t)value;
    else if (n == "static_small" && value >= 0 && value <= 4096) c->tune.static_small = (uint32_t)value;
    else if (n == "big_min" && value >= 1 && value <= 100000) c->tune.big_min = (uint32_t)value;
    else if (n == "big_keep" && value >= 0 && value <= 1024) c->tune.big_keep = (uint32_t)value;
    else if (n == "stack_cap" && value >= 0 && value <= 64) c->tune.stack_cap = (uint32_t)value;
    else if (n == "prim_bvh" && (value == 0 || value == 1)) c->abvh_enabled = value;
    else if (n == "measure_spp" && value >= 1 && value <= 1024) c->measure_spp = value;
    else if (n == "gpu_build" && (value == 0 || value == 1 || value == 2)) c->gpu_build = value;
    else if (n == "fuse" && (value == 0 || value == 1)) c->tune.fuse = (uint32_t)value;
    else if (n == "tri_min" && value >= 0 && value <= 1024) c->tune.tri_min = (uint32_t)value;  // 0 = auto
    else if (n == "refill_min" && value >= 1 && value <= 64) c->tune.refill_min = (uint32_t)value;
    else if (n == "exit_max" && value >= 0 && value < 64) c->tune.exit_max = (uint32_t)value;
    else return fail(c, PRT_ERR_INVALID, "unknown parameter or bad value: %s = %d", name, value);
    return PRT_OK;
}

int prt_set_variant(PrtContext* c, int variant) {
    if (!c) return PRT_ERR_INVALID;
    c->variant = variant;
    return PRT_OK;
}

}  // extern "C"
