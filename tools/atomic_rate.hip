// atomic_rate.hip — what the producers' slot reservation costs: one returning atomicAdd per counter per block.
//
// k_shade / k_raygen reserve output slots with ONE global atomic per side per block (block_alloc2).  A counter word is
// served by one L2 channel, which executes returning atomics to the same address one after the other; this measures
// how many per microsecond, in the producers' own pattern: B blocks of T threads, thread 0 adds to C counters (each
// on its own 256-B line), barrier, every thread stores 48 B at its slot.  MODE 0: no atomic (slots from blockIdx),
// MODE 1: the atomics as the kernels do them, MODE 2: the same number of atomics spread over 8 words per counter
// (blockIdx & 7), i.e. what per-XCD sub-counters would cost.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/atomic_rate.hip -o gpurun_out/atomic_rate && gpurun_out/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int T, int MODE>
__global__ void __launch_bounds__(T) k_reserve(uint32_t* counters, float4* out, uint32_t n_counters, uint32_t cap, uint32_t work) {
    __shared__ uint32_t s_base[2];
    float acc = (float)threadIdx.x;  // `work` dependent FMAs per thread before the reservation, as a producer computes before it reserves
    for (uint32_t i = 0; i < work; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(acc));
    if (threadIdx.x == 0) {
        for (uint32_t c = 0; c < n_counters; ++c) {
            if (MODE == 0) s_base[c] = blockIdx.x * (T / n_counters);
            if (MODE == 1) s_base[c] = atomicAdd(&counters[64 * c], T / n_counters);
            if (MODE == 2) s_base[c] = atomicAdd(&counters[64 * c + 1024 * (1 + (blockIdx.x & 7u))], T / n_counters);
        }
    }
    __syncthreads();
    const uint32_t c = threadIdx.x % n_counters;
    uint32_t slot = s_base[c] + threadIdx.x / n_counters;
    if (MODE == 2) slot = (slot + (blockIdx.x & 7u) * (cap / 8u)) % cap;
    slot %= cap;
    const float v = (float)slot + (acc == 1.5f ? 1.0f : 0.0f);
    out[3 * (size_t)slot + 0] = make_float4(v, v, v, v);
    out[3 * (size_t)slot + 1] = make_float4(v, v, v, v);
    out[3 * (size_t)slot + 2] = make_float4(v, v, v, v);
}

template <int T, int MODE>
static float run(uint32_t blocks, uint32_t* counters, float4* out, uint32_t n_counters, uint32_t cap, uint32_t work = 0) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(counters, 0, 1 << 20));
        CHECK(hipEventRecord(e0));
        k_reserve<T, MODE><<<blocks, T>>>(counters, out, n_counters, cap, work);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const uint32_t rays = 80u << 20;  // ~ the rays the first k_shade of a C3 batch of 64 samples stores
    uint32_t* counters; float4* out;
    CHECK(hipMalloc(&counters, 1 << 20));
    CHECK(hipMalloc(&out, 48 * (size_t)rays));
    printf("slot reservation on MI355X (tools/atomic_rate.hip): %u M slots of 48 B, one returning atomicAdd per counter per block\n", rays >> 20);
    printf("block | counters | blocks | no atomic ms | atomics ms | same count over 8 words ms | atomics per us per counter (extra time)\n");
#define ROW(T, NC)                                                                                      \
    {                                                                                                   \
        const uint32_t blocks = rays / T;                                                               \
        const float a = run<T, 0>(blocks, counters, out, NC, rays), b = run<T, 1>(blocks, counters, out, NC, rays), \
                    c = run<T, 2>(blocks, counters, out, NC, rays);                                    \
        printf("%4d | %d | %u | %.3f | %.3f | %.3f | %.1f\n", T, NC, blocks, a, b, c, blocks / ((b - a) * 1e3 + 1e-9)); \
    }
    ROW(512, 2) ROW(512, 1) ROW(1024, 2) ROW(1024, 1) ROW(256, 2)
    printf("with `work` dependent FMAs per thread before the reservation (512-thread blocks, 2 counters):\n");
    printf("work | no atomic ms | atomics ms | over 8 words ms\n");
    for (uint32_t work : {0u, 250u, 500u, 1000u, 2000u, 4000u}) {
        const uint32_t blocks = rays / 512;
        printf("%u | %.3f | %.3f | %.3f\n", work, run<512, 0>(blocks, counters, out, 2, rays, work), run<512, 1>(blocks, counters, out, 2, rays, work),
               run<512, 2>(blocks, counters, out, 2, rays, work));
    }
    return 0;
}
