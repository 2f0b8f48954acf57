// gather_rate.hip — what a dependent chain of 80-B record gathers costs on an MI355X CU, per step, at 1..8 waves per SIMD.
//
// One traversal step of k_traverse8_persistent is "address from the last node -> five 16-B loads of the next node ->
// ~260 VALU instructions -> address".  This strips the step to that skeleton: every lane walks its own pseudo-random
// chain through a table of 80-B records (next index = hash of the words just loaded), with LOADS 16-B loads per step
// and FILL dependent VALU instructions behind them, at a chosen table size (L1 / L2 / Infinity Cache / HBM resident)
// and a chosen number of waves per SIMD (LDS allocation limits the blocks per CU).  Output: wave cycles per step, and
// the lane-loads per clock per CU that corresponds to.  If the step time stops falling with more waves while VALU is
// idle (FILL 0), the vector-memory path of the CU (address coalescer / L1 tag rate) is what bounds a step.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/gather_rate.hip -o gpurun_out/gather_rate && gpurun_out/gather_rate
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                    \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

template <int LOADS, int FILL, int CHAINS>
__global__ void __launch_bounds__(256) k_chain(const uint4* __restrict__ table, uint32_t n_records, uint32_t iters,
                                                unsigned long long* cycles, uint32_t* sink) {
    extern __shared__ uint32_t s_pad[];
    if (iters == 0xFFFFFFFFu) s_pad[threadIdx.x] = 1;
    uint32_t idx[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        idx[c] = ((blockIdx.x * 256u + threadIdx.x) * 2654435761u + 40503u * c) % n_records;
    float acc = 0.0f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
    for (uint32_t it = 0; it < iters; ++it) {
        uint4 w[CHAINS][LOADS];
#pragma unroll
        for (int c = 0; c < CHAINS; ++c)
#pragma unroll
            for (int l = 0; l < LOADS; ++l) w[c][l] = table[(size_t)idx[c] * 5u + l];
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            uint32_t h = 0;
#pragma unroll
            for (int l = 0; l < LOADS; ++l) h += w[c][l].x ^ w[c][l].y ^ w[c][l].z ^ w[c][l].w;
            float f = __uint_as_float((h & 0x007FFFFFu) | 0x3F800000u);
#pragma unroll
            for (int k = 0; k < FILL; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f) : "v"(acc));
            acc += f;
            h += __float_as_uint(f) & 1u;
            idx[c] = (h * 2654435761u + it) % n_records;
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : : "memory");
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * 256u + threadIdx.x) >> 6] = t1 - t0;
    if (acc == 123.456f) sink[0] = idx[0];
}

struct Result {
    double cycles_per_step;
    double ms;
};

template <int LOADS, int FILL, int CHAINS>
static Result run(const uint4* table, uint32_t n_records, int waves_per_simd, uint32_t iters, unsigned long long* d_cycles, uint32_t* d_sink) {
    const int n_cu = 256;
    const int blocks = n_cu * waves_per_simd;  // 256 threads = one wave on each SIMD of a CU
    const size_t lds = (size_t)(160 * 1024 / waves_per_simd) & ~(size_t)1023;
    CHECK(hipFuncSetAttribute((const void*)k_chain<LOADS, FILL, CHAINS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    k_chain<LOADS, FILL, CHAINS><<<blocks, 256, lds>>>(table, n_records, iters / 8, d_cycles, d_sink);  // warm
    CHECK(hipEventRecord(e0));
    k_chain<LOADS, FILL, CHAINS><<<blocks, 256, lds>>>(table, n_records, iters, d_cycles, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), d_cycles, h.size() * 8, hipMemcpyDeviceToHost));
    double sum = 0;
    for (auto v : h) sum += (double)v;
    // s_memtime counts the 100 MHz-derived shader clock reference on this part? no: it returns the GPU clock counter;
    // the ratio to the event time is printed so the reader can check (cycles per step x steps / time = clock in MHz)
    return {sum / h.size() / iters, ms};
}

int main() {
    const size_t sizes[] = {16u << 10, 2u << 20, 16u << 20, 64u << 20, 1u << 30};
    const char* names[] = {"16 KiB (L1)", "2 MiB (L2)", "16 MiB (C3's tree)", "64 MiB (MALL)", "1 GiB (HBM)"};
    const size_t max_bytes = sizes[4];
    uint4* table;
    CHECK(hipMalloc(&table, max_bytes));
    {
        std::vector<uint32_t> h(max_bytes / 4);
        uint32_t s = 12345u;
        for (auto& v : h) {
            s = s * 1664525u + 1013904223u;
            v = s;
        }
        CHECK(hipMemcpy(table, h.data(), max_bytes, hipMemcpyHostToDevice));
    }
    unsigned long long* d_cycles;
    uint32_t* d_sink;
    CHECK(hipMalloc(&d_cycles, 256 * 8 * 4 * 8));
    CHECK(hipMalloc(&d_sink, 4));
    const uint32_t iters = 2000;
    printf("dependent-chain gathers of 80-B records on MI355X (tools/gather_rate.hip), %u steps per lane\n", iters);
    printf("table | loads x 16 B | VALU fill | chains/lane | waves/SIMD | s_memtime ticks per step | us per step (events) | lane-loads per us per CU\n");
    for (int si = 0; si < 5; ++si) {
        const uint32_t n = (uint32_t)(sizes[si] / 80);
#define ROW(L, F, C, W)                                                                                                  \
    {                                                                                                                    \
        Result r = run<L, F, C>(table, n, W, iters, d_cycles, d_sink);                                                    \
        const double us = r.ms * 1e3 / iters;                                                                            \
        printf("%-18s | %d | %3d | %d | %d | %8.1f | %7.4f | %9.1f\n", names[si], L, F, C, W, r.cycles_per_step, us,      \
               (double)W * 4 * 64 * L * C / us);                                                                          \
        fflush(stdout);                                                                                                  \
    }
        for (int w : {1, 2, 3, 4, 5, 6, 8}) ROW(5, 0, 1, w)
        for (int w : {1, 3, 5, 8}) ROW(1, 0, 1, w)
        for (int w : {3, 5, 8}) ROW(5, 256, 1, w)
        for (int w : {3, 5}) ROW(5, 0, 2, w)
        for (int w : {3, 5}) ROW(5, 256, 2, w)
        for (int w : {5}) ROW(1, 256, 1, w)
    }
    return 0;
}
