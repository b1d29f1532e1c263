// micro-benchmark: what does one 64-lane gather instruction cost on an MI355X CU, as a function of how the lanes' addresses
// fall into cache lines?  (The fused renderer issues 6.3 M buffer_load_dwordx2 gathers per 4096-ray launch; its gather side
// alone takes 0.70 ms = ~68 clocks per instruction and CU, independent of the number of resident waves.)
// build: hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o /tmp/gather_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

enum { P_RANDOM, P_PAIR_ADJ, P_QUAD_ADJ, P_ROW16_ADJ, P_SAME, P_FAR_SAME, P_PAIR_SAME, P_HALF_OOB, P_X4_ALIGNED, P_X4_UNALIGNED, P_NOLOAD, P_COUNT };
static const char *NAMES[P_COUNT] = {
    "dwordx2, 64 random entries",
    "dwordx2, lanes (2k,2k+1) adjacent entries (one aligned 16 B pair)",
    "dwordx2, 4 adjacent lanes = 4 consecutive entries (32 B)",
    "dwordx2, 16 adjacent lanes = 16 consecutive entries (one 128 B line)",
    "dwordx2, all 64 lanes the same entry",
    "dwordx2, lanes l and l^32 the same entry (32 distinct)",
    "dwordx2, lanes (2k,2k+1) the same entry (32 distinct)",
    "dwordx2, odd lanes out of range (32 random entries)",
    "dwordx4, 64 random aligned pairs",
    "dwordx4, 64 random unaligned pairs (8 B aligned)",
    "no load (address arithmetic only)",
};

template <int P>
__global__ __launch_bounds__(256) void gather(const float *tab, uint32_t entries_mask, uint32_t table_bytes, uint32_t iters, uint32_t *sink)
{
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tab), 0, table_bytes, 0x00020000);
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t grp, sub = 0, mul = 1;
    switch (P) {
        case P_PAIR_ADJ: grp = t >> 1; sub = t & 1u; mul = 2; break;
        case P_QUAD_ADJ: grp = t >> 2; sub = t & 3u; mul = 4; break;
        case P_ROW16_ADJ: grp = t >> 4; sub = t & 15u; mul = 16; break;
        case P_SAME: grp = t >> 6; break;
        case P_FAR_SAME: grp = t & ~32u; break;
        case P_PAIR_SAME: grp = t >> 1; break;
        case P_X4_ALIGNED: grp = t; mul = 2; break;
        default: grp = t; break;
    }
    uint32_t base = hash32(grp * 2654435761u + 12345u);
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            base += 0x9e3779b1u;                                            // odd stride: a different line every time
            uint32_t e = ((base >> 7) * mul + sub) & entries_mask;
            uint32_t off = e * 8u;
            if (P == P_HALF_OOB && (t & 1u)) off = 0xfffffff8u;
            if (P == P_X4_UNALIGNED) off = (e | 1u) * 8u;
            if (P == P_NOLOAD) { acc ^= off; continue; }
            if (P == P_X4_ALIGNED || P == P_X4_UNALIGNED) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            } else {
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
                acc ^= v.x ^ v.y;
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = t;
}

template <int P>
static int run(const float *tab, uint32_t entries, uint32_t *sink, const char *label)
{
    const uint32_t blocks = 2048, iters = 128;                                // 8192 waves = 32 per CU, 1024 gather instructions each
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
        gather<P><<<blocks, 256>>>(tab, entries - 1, entries * 8u, iters, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    const double instr = (double)blocks * 4 * iters * 8;                      // wave-level load instructions
    const double per_cu_ns = best * 1e6 / (instr / 256.0);
    printf("  %-72s %8.3f ms  %6.1f ns per instruction and CU (%5.1f clk @2.4 GHz)  %7.1f G lanes/s\n", NAMES[P], best, per_cu_ns, per_cu_ns * 2.4,
           instr * (P == P_HALF_OOB ? 32 : 64) / best * 1e-6);
    (void)label;
    return 0;
}

int main()
{
    uint32_t *sink; CK(hipMalloc(&sink, 64));
    for (uint32_t entries : { 1u << 23, 1u << 22, 1u << 18, 1u << 11 }) {     // 64 MB (MALL), 32 MB, 2 MB (one XCD's L2), 16 KB (L1)
        float *tab; CK(hipMalloc(&tab, (size_t)entries * 8)); CK(hipMemset(tab, 1, (size_t)entries * 8));
        printf("table: %u entries of 8 B = %.2f MB\n", entries, entries * 8.0 / 1048576.0);
        if (run<P_RANDOM>(tab, entries, sink, "")) return 1;
        if (run<P_PAIR_ADJ>(tab, entries, sink, "")) return 1;
        if (run<P_QUAD_ADJ>(tab, entries, sink, "")) return 1;
        if (run<P_ROW16_ADJ>(tab, entries, sink, "")) return 1;
        if (run<P_SAME>(tab, entries, sink, "")) return 1;
        if (run<P_FAR_SAME>(tab, entries, sink, "")) return 1;
        if (run<P_PAIR_SAME>(tab, entries, sink, "")) return 1;
        if (run<P_HALF_OOB>(tab, entries, sink, "")) return 1;
        if (run<P_X4_ALIGNED>(tab, entries, sink, "")) return 1;
        if (run<P_X4_UNALIGNED>(tab, entries, sink, "")) return 1;
        if (run<P_NOLOAD>(tab, entries, sink, "")) return 1;
        CK(hipFree(tab));
    }
    return 0;
}
