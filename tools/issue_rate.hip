// issue_rate.hip — what one SIMD of an MI355X CU sustains, per instruction kind, at 1..8 waves per SIMD.
//
// The traversal kernel is bound by instruction issue, not by HBM; to price it (bench.py's roofline) the cost of an
// instruction on a SIMD has to be known for the kinds that kernel is made of, not only for v_fma_f32.  Each test is a
// loop over 64 independent instructions of one kind (8 registers x 8), every wave stamps s_memtime around its loop;
// cycles per instruction per SIMD = wave cycles / instructions / waves on the SIMD.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/issue_rate.hip -o gpurun_out/issue_rate && gpurun_out/issue_rate
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                        \
    do {                                                                \
        hipError_t e_ = (x);                                            \
        if (e_ != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));     \
            exit(1);                                                    \
        }                                                               \
    } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define ROUND8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

enum { OP_FMA, OP_FMA_MIX, OP_PERM, OP_CVT_UBYTE, OP_CNDMASK, OP_MAX3, OP_LSHL, OP_BFE, OP_MUL_LO, OP_SALU, OP_FMA_SALU, OP_CMP, OP_MIN, OP_AND_OR, OP_SAVEEXEC, OP_READLANE, OP_CNDMASK_SGPR, OP_CNDMASK_IND, OP_BFI, OP_SUB, OP_ASHR, OP_OR3, OP_DS_WRITE, OP_DS_READ, N_OPS };
static const char* kNames[N_OPS] = {"v_fma_f32", "v_fma_mix_f32 (f16 src0)", "v_perm_b32", "v_cvt_f32_ubyte0", "v_cndmask_b32",
                                    "v_max3_f32", "v_lshlrev_b32", "v_bfe_u32", "v_mul_lo_u32", "s_and_b32 (SALU)",
                                    "v_fma_f32 + s_and_b32 interleaved (per pair)", "v_cmp_le_f32 (-> vcc)", "v_min_f32", "v_and_or_b32",
                                    "s_and_saveexec_b64 + s_mov exec (per pair)", "v_readlane_b32 (-> sgpr)",
                                    "v_cndmask_b32_e64 (sgpr-pair condition)", "v_cndmask_b32 vcc, dst != src", "v_bfi_b32", "v_sub_f32",
                                    "v_ashrrev_i32", "v_or3_b32", "ds_write_b64 (own slot)", "ds_read_b64 (own slot)"};

template <int OP>
__global__ void __launch_bounds__(256) k_issue(uint32_t iters, unsigned long long* cycles, float* sink, float a, float b) {
    float r[8];
    uint32_t s[8];
    for (int i = 0; i < 8; ++i) {
        r[i] = a + (float)(threadIdx.x + i);
        s[i] = (uint32_t)i + 12345u * blockIdx.x;
    }
    unsigned long long t0, t1;
    __shared__ unsigned long long s_lds[8 * 256];
    const uint32_t lds_addr = (uint32_t)(threadIdx.x * 8u);
    unsigned long long pair = threadIdx.x;
    const unsigned long long m64 = 0x5555AAAA3333CCCCull + blockIdx.x;
    if (iters == 0xFFFFFFFFu) s_lds[threadIdx.x] = 1;  // keeps the array
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r[0]), "v"(b) : "vcc");  // some lanes set: v_cndmask's condition
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
    for (uint32_t it = 0; it < iters; ++it) {
        if (OP == OP_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            ROUND8(X)
#undef X
        } else if (OP == OP_FMA_MIX) {
#define X(i) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(r[i]) : "v"(a), "v"(b));
            ROUND8(X)
#undef X
        } else if (OP == OP_PERM) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(0x00050004u));
            ROUND8(X)
#undef X
        } else if (OP == OP_CVT_UBYTE) {
#define X(i) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(r[i]));
            ROUND8(X)
#undef X
        } else if (OP == OP_CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a));
            ROUND8(X)
#undef X
        } else if (OP == OP_MAX3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            ROUND8(X)
#undef X
        } else if (OP == OP_LSHL) {
#define X(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r[i]));
            ROUND8(X)
#undef X
        } else if (OP == OP_BFE) {
#define X(i) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(r[i]));
            ROUND8(X)
#undef X
        } else if (OP == OP_MUL_LO) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            ROUND8(X)
#undef X
        } else if (OP == OP_SALU) {
#define X(i) asm volatile("s_and_b32 %0, %0, 0x7fffffff" : "+s"(s[i]));
            ROUND8(X)
#undef X
        } else if (OP == OP_FMA_SALU) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\ts_and_b32 %1, %1, 0x7fffffff" : "+v"(r[i]), "+s"(s[i]) : "v"(a), "v"(b));
            ROUND8(X)
#undef X
        } else if (OP == OP_MIN) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            ROUND8(X)
#undef X
        } else if (OP == OP_AND_OR) {
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            ROUND8(X)
#undef X
        } else if (OP == OP_SAVEEXEC) {
#define X(i) asm volatile("s_and_saveexec_b64 s[20:21], vcc\n\ts_mov_b64 exec, s[20:21]" : : : "s20", "s21", "scc");
            ROUND8(X)
#undef X
        } else if (OP == OP_READLANE) {
#define X(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s[i]) : "v"(r[i]));
            ROUND8(X)
#undef X
        } else if (OP == OP_CNDMASK_SGPR) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "s"(m64));
            ROUND8(X)
#undef X
        } else if (OP == OP_CNDMASK_IND) {
#define X(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r[i]) : "v"(a), "v"(b));
            ROUND8(X)
#undef X
        } else if (OP == OP_BFI) {
#define X(i) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            ROUND8(X)
#undef X
        } else if (OP == OP_SUB) {
#define X(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            ROUND8(X)
#undef X
        } else if (OP == OP_ASHR) {
#define X(i) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(r[i]));
            ROUND8(X)
#undef X
        } else if (OP == OP_OR3) {
#define X(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            ROUND8(X)
#undef X
        } else if (OP == OP_DS_WRITE) {
#define X(i) asm volatile("ds_write_b64 %0, %1 offset:" #i "*2048" : : "v"(lds_addr), "v"(pair) : "memory");
            ROUND8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == OP_DS_READ) {
#define X(i) asm volatile("ds_read_b64 %0, %1 offset:" #i "*2048" : "=v"(pair) : "v"(lds_addr) : "memory");
            ROUND8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == OP_CMP) {
#define X(i) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(r[i]), "v"(a) : "vcc");
            ROUND8(X)
#undef X
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : : "memory");
    float acc = 0.f;
    for (int i = 0; i < 8; ++i) acc += r[i] + (float)s[i];
    acc += (float)pair + (float)s_lds[threadIdx.x & 7];
    if (acc == 1.2345f) sink[0] = acc;
    if ((threadIdx.x & 63u) == 0u) cycles[blockIdx.x * 4u + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
static void run(unsigned long long* d_cyc, float* d_sink, int n_cu) {
    const uint32_t iters = 2000;
    printf("%-48s", kNames[OP]);
    for (int w : {1, 2, 4, 5, 8}) {
        // w blocks of 256 threads per CU = w waves on every SIMD (blocks are dealt over the CUs evenly when they all fit)
        const int blocks = n_cu * w;
        hipLaunchKernelGGL((k_issue<OP>), dim3(blocks), dim3(256), 0, 0, iters, d_cyc, d_sink, 1.0001f, 0.5f);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(blocks * 4);
        CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (unsigned long long c : h) sum += (double)c;
        const double per_wave = sum / h.size() / ((double)iters * 64.0);  // wave cycles per instruction
        printf("  w=%d: %5.2f/wave %5.2f/SIMD", w, per_wave, per_wave / w);
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    unsigned long long* d_cyc;
    float* d_sink;
    CHECK(hipMalloc((void**)&d_cyc, (size_t)n_cu * 8 * 4 * 8));
    CHECK(hipMalloc((void**)&d_sink, 64));
    printf("issue cost in shader cycles per wave64 instruction (s_memtime), %d CUs; /wave = as one wave sees it, /SIMD = that / waves per SIMD\n", n_cu);
    run<OP_FMA>(d_cyc, d_sink, n_cu);
    run<OP_FMA_MIX>(d_cyc, d_sink, n_cu);
    run<OP_PERM>(d_cyc, d_sink, n_cu);
    run<OP_CVT_UBYTE>(d_cyc, d_sink, n_cu);
    run<OP_CNDMASK>(d_cyc, d_sink, n_cu);
    run<OP_MAX3>(d_cyc, d_sink, n_cu);
    run<OP_LSHL>(d_cyc, d_sink, n_cu);
    run<OP_BFE>(d_cyc, d_sink, n_cu);
    run<OP_MUL_LO>(d_cyc, d_sink, n_cu);
    run<OP_CMP>(d_cyc, d_sink, n_cu);
    run<OP_SALU>(d_cyc, d_sink, n_cu);
    run<OP_FMA_SALU>(d_cyc, d_sink, n_cu);
    run<OP_MIN>(d_cyc, d_sink, n_cu);
    run<OP_AND_OR>(d_cyc, d_sink, n_cu);
    run<OP_SAVEEXEC>(d_cyc, d_sink, n_cu);
    run<OP_READLANE>(d_cyc, d_sink, n_cu);
    run<OP_CNDMASK_SGPR>(d_cyc, d_sink, n_cu);
    run<OP_CNDMASK_IND>(d_cyc, d_sink, n_cu);
    run<OP_BFI>(d_cyc, d_sink, n_cu);
    run<OP_SUB>(d_cyc, d_sink, n_cu);
    run<OP_ASHR>(d_cyc, d_sink, n_cu);
    run<OP_OR3>(d_cyc, d_sink, n_cu);
    run<OP_DS_WRITE>(d_cyc, d_sink, n_cu);
    run<OP_DS_READ>(d_cyc, d_sink, n_cu);
    return 0;
}
