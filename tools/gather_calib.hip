// gather_calib.hip — what does rocprofv3's FETCH_SIZE report for the access shapes of the traversal kernel?
//
// MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports exactly HALF the bytes of a wide coalesced streaming read (128-B
// requests tallied at 64 B), "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern".  The traversal kernel's pattern is a per-lane gather of one 80-B node (5 x global_load_dwordx4 from a 16-B
// aligned record) or one 48-B triangle record (3 x dwordx4) at unrelated addresses.  This program runs those shapes over
// a table far larger than the 256 MB Infinity Cache (so every record comes from HBM) with a KNOWN number of records, next
// to a plain streaming read of the same table; tools/gather_calib.sh divides the counter by the record counts.
//   64-B lines touched per record: 80-B records at 16-B granularity ALWAYS straddle two 64-B lines (128 B);
//   48-B records touch 1.5 lines on average (96 B); a lone 16-B load one line (64 B).
// build: hipcc --offload-arch=gfx950 -O3 tools/gather_calib.hip -o gpurun_out/gather_calib
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}

// every lane reads WORDS consecutive uint4 (= one record of 16 * WORDS bytes) of a pseudo-random record
template <int WORDS>
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ table, uint64_t n_records, uint32_t* __restrict__ sink,
                                                uint32_t salt) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint64_t rec = ((uint64_t)pcg(i ^ salt) * n_records) >> 32;  // uniform in [0, n_records)
    const uint4* p = table + rec * WORDS;
    uint32_t acc = 0;
#pragma unroll
    for (int w = 0; w < WORDS; ++w) {
        const uint4 v = p[w];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = i;  // never true for the zero-filled table: no store traffic
}

__global__ void __launch_bounds__(256) k_stream(const uint4* __restrict__ table, uint64_t n_words, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * 256ull) {
        const uint4 v = table[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = 1;
}

int main() {
    const uint64_t bytes = 4ull << 30;  // 4 GiB table: 16x the Infinity Cache
    uint4* table = nullptr;
    uint32_t* sink = nullptr;
    CHECK(hipMalloc((void**)&table, bytes));
    CHECK(hipMalloc((void**)&sink, 64));
    CHECK(hipMemset(table, 0, bytes));
    CHECK(hipMemset(sink, 0, 64));
    CHECK(hipDeviceSynchronize());
    const uint32_t lanes = 1u << 26;  // 67,108,864 records per gather kernel
    const uint32_t blocks = lanes / 256u;
    // salts differ so that no kernel re-reads the records of the one before it
    hipLaunchKernelGGL((k_gather<5>), dim3(blocks), dim3(256), 0, 0, table, bytes / 80, sink, 0x1111u);
    hipLaunchKernelGGL((k_gather<3>), dim3(blocks), dim3(256), 0, 0, table, bytes / 48, sink, 0x2222u);
    hipLaunchKernelGGL((k_gather<1>), dim3(blocks), dim3(256), 0, 0, table, bytes / 16, sink, 0x3333u);
    hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, 0, table, bytes / 16, sink);
    CHECK(hipDeviceSynchronize());
    printf("records_per_gather %u table_bytes %llu\n", lanes, (unsigned long long)bytes);
    CHECK(hipFree(table));
    CHECK(hipFree(sink));
    return 0;
}
