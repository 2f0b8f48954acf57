// prt_group.cpp — several GPUs of one node behind one renderer: the C/C++ multi-GPU host path (include/prt.h, prt_group_*).
//
// The reference is single-GPU (cudaSetDevice(0), src/backend/optix/renderer.cpp:217); BASELINE's north star asks for the
// image tiled across the GPUs of a node with a final RCCL gather of per-tile radiance, driven from C/C++ host code.  This
// file is a pure CLIENT of the single-GPU C-ABI (prt_create / prt_set_film(rank, world) / prt_render_async / prt_film_local /
// prt_film_resolve_on ...): one context per GPU, one host thread per context, no kernel of its own.  RCCL is loaded on
// demand (dlopen): libprt.so has no link-time dependency on it, and a Python process that already holds torch's copy of
// librccl binds to that one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/prt.h"

namespace {

// The slice of the RCCL API the gather needs (rccl.h: ncclResult_t is an enum with ncclSuccess = 0, ncclFloat = 7).
typedef void* ncclComm_t;
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load() {
        // a copy some other module of the process already loaded (torch bundles one) first, then the system's
        for (const char* name : {"librccl.so", "librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (lib) break;
        }
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) return false;
#define PRT_SYM(F) F = (decltype(F))dlsym(lib, "nccl" #F)
        PRT_SYM(CommInitAll); PRT_SYM(CommDestroy); PRT_SYM(GroupStart); PRT_SYM(GroupEnd); PRT_SYM(Send); PRT_SYM(Recv);
        PRT_SYM(AllGather); PRT_SYM(GetErrorString);
#undef PRT_SYM
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && AllGather && GetErrorString;
    }
};
constexpr int kNcclFloat = 7;

}  // namespace

struct PrtGroup {
    std::vector<int> devices;
    std::vector<PrtContext*> ctx;
    std::string err;
    std::string transport = "none";
    Rccl rccl;
    std::vector<ncclComm_t> comms;
    // rank 0's device: [n][stride] gathered payloads, un-tiled film, display
    uint32_t W = 0, H = 0;
    uint64_t payload_floats = 0;
    float* d_gathered = nullptr;
    float* d_rgb = nullptr;
    float* d_weight = nullptr;
    uint8_t* d_rgba = nullptr;
    std::vector<hipEvent_t> ev;  // per rank: payload copied / rendering done
    bool film_current = false;   // d_rgb / d_weight hold the films' current contents
};

namespace {

int gfail(PrtGroup* g, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (g) g->err = buf;
    return code;
}

#define GHIP(g, call)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return gfail((g), PRT_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// f(rank) for every rank; the first non-zero status (with that context's message) wins.  threads = true: one host thread
// per rank (renders, uploads: each call drives its GPU for a while); false: in turn on the caller's thread (setters that
// only store a value: a thread each would cost more than the call).  No exception leaves this function: a thread that
// cannot be started (std::system_error) turns into PRT_ERR_HIP after the threads that did start have been joined.
int for_each_rank(PrtGroup* g, const std::function<int(uint32_t)>& f, bool threads = true) {
    const uint32_t n = (uint32_t)g->ctx.size();
    std::vector<int> rc(n, 0);
    if (n == 1 || !threads) {
        for (uint32_t r = 0; r < n; ++r) rc[r] = f(r);
    } else {
        std::vector<std::thread> th;
        bool spawn_failed = false;
        try {
            th.reserve(n);
            for (uint32_t r = 0; r < n; ++r) th.emplace_back([&, r] { rc[r] = f(r); });
        } catch (...) {
            spawn_failed = true;
        }
        for (std::thread& t : th) t.join();
        if (spawn_failed) return gfail(g, PRT_ERR_HIP, "could not start a host thread per rank (%zu of %u started)", th.size(), n);
    }
    for (uint32_t r = 0; r < n; ++r)
        if (rc[r]) return gfail(g, rc[r], "rank %u (device %d): %s", r, g->devices[r], prt_last_error(g->ctx[r]));
    return PRT_OK;
}

void free_film_buffers(PrtGroup* g) {
    if (g->devices.empty()) return;
    (void)hipSetDevice(g->devices[0]);
    for (void* p : {(void*)g->d_gathered, (void*)g->d_rgb, (void*)g->d_weight, (void*)g->d_rgba})
        if (p) (void)hipFree(p);
    g->d_gathered = g->d_rgb = g->d_weight = nullptr;
    g->d_rgba = nullptr;
}

// All ranks' payloads -> rank 0's device, un-tiled into d_rgb / d_weight.  Every rank's stream is idle on entry.
int gather_and_resolve(PrtGroup* g) {
    const uint32_t n = (uint32_t)g->ctx.size();
    if (!g->d_gathered) return gfail(g, PRT_ERR_INVALID, "prt_group_set_film has not been called");
    std::vector<void*> local(n), stream(n);
    uint64_t nf = 0;
    for (uint32_t r = 0; r < n; ++r) {
        int rc = prt_film_local(g->ctx[r], &local[r], &nf);
        if (!rc) rc = prt_get_stream(g->ctx[r], &stream[r]);
        if (rc) return gfail(g, rc, "rank %u: %s", r, prt_last_error(g->ctx[r]));
        if (nf != g->payload_floats) return gfail(g, PRT_ERR_INVALID, "rank %u: payload size mismatch", r);
    }
    const size_t bytes = (size_t)nf * sizeof(float);
    hipStream_t s0 = (hipStream_t)stream[0];
    if (g->transport == "rccl" && n == 1) {  // 1-rank all-gather: the RCCL path's hardware smoke test
        GHIP(g, hipSetDevice(g->devices[0]));
        const int rc = g->rccl.AllGather(local[0], g->d_gathered, (size_t)nf, kNcclFloat, g->comms[0], s0);
        if (rc) return gfail(g, PRT_ERR_HIP, "ncclAllGather: %s", g->rccl.GetErrorString(rc));
    } else if (g->transport == "rccl") {
        // rank 0 receives n - 1 payloads, every other rank sends one: a single group call (single-process RCCL)
        GHIP(g, hipSetDevice(g->devices[0]));
        GHIP(g, hipMemcpyAsync(g->d_gathered, local[0], bytes, hipMemcpyDeviceToDevice, s0));
        int rc = g->rccl.GroupStart();
        for (uint32_t r = 1; r < n && !rc; ++r) {
            rc = g->rccl.Recv(g->d_gathered + (size_t)r * nf, (size_t)nf, kNcclFloat, (int)r, g->comms[0], s0);
            if (!rc) rc = g->rccl.Send(local[r], (size_t)nf, kNcclFloat, 0, g->comms[r], (hipStream_t)stream[r]);
        }
        const int rc_end = g->rccl.GroupEnd();
        if (rc || rc_end) return gfail(g, PRT_ERR_HIP, "RCCL gather: %s", g->rccl.GetErrorString(rc ? rc : rc_end));
    } else {
        // peer copies: each rank's stream pushes its payload into rank 0's buffer; rank 0's stream waits for all of them
        for (uint32_t r = 0; r < n; ++r) {
            GHIP(g, hipSetDevice(g->devices[r]));
            float* dst = g->d_gathered + (size_t)r * nf;
            if (g->devices[r] == g->devices[0])
                GHIP(g, hipMemcpyAsync(dst, local[r], bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream[r]));
            else
                GHIP(g, hipMemcpyPeerAsync(dst, g->devices[0], local[r], g->devices[r], bytes, (hipStream_t)stream[r]));
            GHIP(g, hipEventRecord(g->ev[r], (hipStream_t)stream[r]));
        }
        GHIP(g, hipSetDevice(g->devices[0]));
        for (uint32_t r = 1; r < n; ++r) GHIP(g, hipStreamWaitEvent(s0, g->ev[r], 0));
    }
    int rc = prt_film_resolve_on(g->ctx[0], nullptr, g->d_gathered, n, g->d_rgb, g->d_weight);
    if (rc) return gfail(g, rc, "rank 0: %s", prt_last_error(g->ctx[0]));
    for (uint32_t r = 0; r < n; ++r)  // the senders' streams too: the payload may be rendered into again after this call
        if ((rc = prt_synchronize(g->ctx[r]))) return gfail(g, rc, "rank %u: %s", r, prt_last_error(g->ctx[r]));
    g->film_current = true;
    return PRT_OK;
}

}  // namespace

extern "C" {

int prt_group_create(const int* device_ids, uint32_t n, PrtGroup** out) {
    if (!out) return PRT_ERR_INVALID;
    PrtGroup* g = new PrtGroup();
    *out = g;
    if (!device_ids || n == 0 || n > 64) return gfail(g, PRT_ERR_INVALID, "prt_group_create: 1..64 device ids");
    g->devices.assign(device_ids, device_ids + n);
    for (uint32_t r = 0; r < n; ++r) {
        PrtContext* c = nullptr;
        const int rc = prt_create(device_ids[r], &c);
        g->ctx.push_back(c);
        if (rc) return gfail(g, rc, "rank %u (device %d): %s", r, device_ids[r], prt_last_error(c));
        if (device_ids[r] < 0) return gfail(g, PRT_ERR_NO_DEVICE, "rank %u: a group needs HIP devices (no CPU fallback)", r);
    }
    g->ev.resize(n);
    for (uint32_t r = 0; r < n; ++r) {
        GHIP(g, hipSetDevice(device_ids[r]));
        GHIP(g, hipEventCreateWithFlags(&g->ev[r], hipEventDisableTiming));
    }
    // transport: RCCL over xGMI when every rank has its own device, peer copies otherwise
    bool distinct = true;
    for (uint32_t a = 0; a < n; ++a)
        for (uint32_t b = a + 1; b < n; ++b) distinct = distinct && device_ids[a] != device_ids[b];
    const char* want = getenv("PRT_GROUP_TRANSPORT");
    const bool force_rccl = want && !strcmp(want, "rccl"), force_peer = want && !strcmp(want, "peer");
    if (force_rccl && !distinct) return gfail(g, PRT_ERR_INVALID, "PRT_GROUP_TRANSPORT=rccl needs distinct devices (two RCCL ranks cannot share one)");
    if (!force_peer && distinct && (n > 1 || force_rccl)) {
        if (g->rccl.load()) {
            g->comms.assign(n, nullptr);
            const int rc = g->rccl.CommInitAll(g->comms.data(), (int)n, device_ids);
            if (rc == 0) {
                g->transport = "rccl";
            } else {
                g->comms.clear();
                if (force_rccl) return gfail(g, PRT_ERR_HIP, "ncclCommInitAll: %s", g->rccl.GetErrorString(rc));
            }
        } else if (force_rccl) {
            return gfail(g, PRT_ERR_HIP, "librccl.so could not be loaded: %s", dlerror());
        }
    }
    if (g->transport == "none" && n > 1) {
        g->transport = "peer";
        for (uint32_t r = 1; r < n; ++r)  // direct xGMI copies where the devices allow it (an error here only means staged copies)
            if (device_ids[r] != device_ids[0]) {
                (void)hipSetDevice(device_ids[r]);
                (void)hipDeviceEnablePeerAccess(device_ids[0], 0);
                (void)hipGetLastError();
            }
    }
    return PRT_OK;
}

void prt_group_destroy(PrtGroup* g) {
    if (!g) return;
    for (ncclComm_t c : g->comms)
        if (c) (void)g->rccl.CommDestroy(c);
    free_film_buffers(g);
    for (size_t r = 0; r < g->ev.size(); ++r)
        if (g->ev[r]) {
            (void)hipSetDevice(g->devices[r]);
            (void)hipEventDestroy(g->ev[r]);
        }
    for (PrtContext* c : g->ctx) prt_destroy(c);
    delete g;
}

const char* prt_group_last_error(const PrtGroup* g) { return g ? g->err.c_str() : "null group"; }
uint32_t prt_group_size(const PrtGroup* g) { return g ? (uint32_t)g->ctx.size() : 0u; }
const char* prt_group_transport(const PrtGroup* g) { return g ? g->transport.c_str() : "none"; }
PrtContext* prt_group_context(PrtGroup* g, uint32_t rank) { return (g && rank < g->ctx.size()) ? g->ctx[rank] : nullptr; }

int prt_group_set_scene(PrtGroup* g, const PrtSceneDesc* scene) {
    if (!g || g->ctx.empty()) return PRT_ERR_INVALID;
    int rc = prt_set_scene(g->ctx[0], scene);  // flatten + BVH build once ...
    if (rc) return gfail(g, rc, "rank 0: %s", prt_last_error(g->ctx[0]));
    return for_each_rank(g, [&](uint32_t r) { return r == 0 ? PRT_OK : prt_clone_scene(g->ctx[r], g->ctx[0]); });  // ... uploads in parallel
}

int prt_group_refit_meshes(PrtGroup* g, const PrtMesh* meshes, uint32_t n_meshes) {
    if (!g || g->ctx.empty()) return PRT_ERR_INVALID;
    g->film_current = false;
    return for_each_rank(g, [&](uint32_t r) { return prt_refit_meshes(g->ctx[r], meshes, n_meshes); });
}

int prt_group_set_camera(PrtGroup* g, const PrtCameraDesc* cam) {
    if (!g) return PRT_ERR_INVALID;
    return for_each_rank(g, [&](uint32_t r) { return prt_set_camera(g->ctx[r], cam); }, false);
}

int prt_group_set_film(PrtGroup* g, uint32_t width, uint32_t height) {
    if (!g || g->ctx.empty()) return PRT_ERR_INVALID;
    const uint32_t n = (uint32_t)g->ctx.size();
    int rc = for_each_rank(g, [&](uint32_t r) { return prt_set_film(g->ctx[r], width, height, r, n); });
    if (rc) return rc;
    void* p = nullptr;
    uint64_t nf = 0;
    if ((rc = prt_film_local(g->ctx[0], &p, &nf))) return gfail(g, rc, "rank 0: %s", prt_last_error(g->ctx[0]));
    free_film_buffers(g);
    g->W = width;
    g->H = height;
    g->payload_floats = nf;
    GHIP(g, hipSetDevice(g->devices[0]));
    const size_t npix = (size_t)width * height;
    GHIP(g, hipMalloc((void**)&g->d_gathered, (size_t)n * nf * sizeof(float)));
    GHIP(g, hipMalloc((void**)&g->d_rgb, npix * 3 * sizeof(float)));
    GHIP(g, hipMalloc((void**)&g->d_weight, npix * sizeof(float)));
    GHIP(g, hipMalloc((void**)&g->d_rgba, npix * 4));
    g->film_current = false;
    return PRT_OK;
}

int prt_group_film_clear(PrtGroup* g) {
    if (!g) return PRT_ERR_INVALID;
    g->film_current = false;
    return for_each_rank(g, [&](uint32_t r) { return prt_film_clear(g->ctx[r]); }, false);
}

int prt_group_set_sampling(PrtGroup* g, const PrtSampling* s) {
    if (!g) return PRT_ERR_INVALID;
    return for_each_rank(g, [&](uint32_t r) { return prt_set_sampling(g->ctx[r], s); }, false);
}

int prt_group_set_samples_in_flight(PrtGroup* g, uint32_t n_samples) {
    if (!g) return PRT_ERR_INVALID;
    return for_each_rank(g, [&](uint32_t r) { return prt_set_samples_in_flight(g->ctx[r], n_samples); }, false);
}

int prt_group_set_param(PrtGroup* g, const char* name, int value) {
    if (!g) return PRT_ERR_INVALID;
    return for_each_rank(g, [&](uint32_t r) { return prt_set_param(g->ctx[r], name, value); }, false);
}

int prt_group_render(PrtGroup* g, uint32_t spp, uint32_t max_depth, uint32_t seed, uint32_t first_sample) {
    if (!g) return PRT_ERR_INVALID;
    g->film_current = false;
    int rc = for_each_rank(g, [&](uint32_t r) {  // one host thread per GPU: enqueue all batches, then wait (watchdog checked)
        const int e = prt_render_async(g->ctx[r], spp, max_depth, seed, first_sample);
        return e ? e : prt_synchronize(g->ctx[r]);
    });
    if (rc) return rc;
    return gather_and_resolve(g);
}

int prt_group_film_read(PrtGroup* g, float* rgb_sum, float* weight) {
    if (!g || g->ctx.empty()) return PRT_ERR_INVALID;
    if (!g->film_current) {
        const int rc = gather_and_resolve(g);
        if (rc) return rc;
    }
    GHIP(g, hipSetDevice(g->devices[0]));
    const size_t npix = (size_t)g->W * g->H;
    if (rgb_sum) GHIP(g, hipMemcpy(rgb_sum, g->d_rgb, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (weight) GHIP(g, hipMemcpy(weight, g->d_weight, npix * sizeof(float), hipMemcpyDeviceToHost));
    return PRT_OK;
}

int prt_group_film_display(PrtGroup* g, float exposure, float gamma, uint8_t* rgba8) {
    if (!g || g->ctx.empty() || !rgba8) return PRT_ERR_INVALID;
    if (!g->film_current) {
        const int rc = gather_and_resolve(g);
        if (rc) return rc;
    }
    int rc = prt_film_tonemap(g->ctx[0], g->d_rgb, g->d_weight, exposure, gamma, g->d_rgba);
    if (!rc) rc = prt_synchronize(g->ctx[0]);
    if (rc) return gfail(g, rc, "rank 0: %s", prt_last_error(g->ctx[0]));
    GHIP(g, hipSetDevice(g->devices[0]));
    GHIP(g, hipMemcpy(rgba8, g->d_rgba, (size_t)g->W * g->H * 4, hipMemcpyDeviceToHost));
    return PRT_OK;
}

int prt_group_get_stats(PrtGroup* g, PrtStats* out) {
    if (!g || !out) return PRT_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    for (size_t r = 0; r < g->ctx.size(); ++r) {
        PrtStats s;
        const int rc = prt_get_stats(g->ctx[r], &s);
        if (rc) return gfail(g, rc, "rank %zu: %s", r, prt_last_error(g->ctx[r]));
        out->rays_total += s.rays_total;
        for (int d = 0; d < PRT_MAX_DEPTH; ++d) out->rays_per_depth[d] += s.rays_per_depth[d];
        out->samples = std::max(out->samples, s.samples);
        out->intersect_launches = std::max(out->intersect_launches, s.intersect_launches);
        out->intersect_ms = std::max(out->intersect_ms, s.intersect_ms);
        out->shade_ms = std::max(out->shade_ms, s.shade_ms);
        out->raygen_ms = std::max(out->raygen_ms, s.raygen_ms);
        out->accumulate_ms = std::max(out->accumulate_ms, s.accumulate_ms);
    }
    return PRT_OK;
}

}  // extern "C"
