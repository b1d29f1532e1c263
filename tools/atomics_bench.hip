// micro-benchmark: random fp32 atomic adds into a table, agent scope vs workgroup scope with XCD-private copies.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomics_bench.hip -o /tmp/atomics_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf; }   // HW_REG_XCC_ID

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>   // 0: agent scope, one table; 1: workgroup scope, table copy = XCC id; 2: agent scope into copy = XCC id
__global__ __launch_bounds__(256) void k(float *tab, uint32_t entries, uint32_t per_thread, uint32_t *xcc_hist)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t xid = xcc_id();
    if (threadIdx.x == 0 && xcc_hist) atomicAdd(&xcc_hist[(blockIdx.x & 7) * 16 + xid], 1u);
    float *base = tab + (MODE == 0 ? 0 : (size_t)xid * entries * 2);
    for (uint32_t i = 0; i < per_thread; ++i) {
        const uint32_t idx = hash32(t * per_thread + i) % entries;
        float *p = base + 2 * (size_t)idx;
        if (MODE == 1) {
            __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p + 1, 2.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(p + 1, 2.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void reduce8(const float *tab, float *out, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f;
    for (int c = 0; c < 8; ++c) s += tab[c * n + i];
    out[i] = s;
}

int main()
{
    const uint32_t per_thread = 56;
    const uint32_t threads = 524288;
    for (uint32_t entries : { 4913u, 12167u, 79507u, 1u << 19 }) {
        float *tab, *out; uint32_t *hist;
        CK(hipMalloc(&tab, (size_t)entries * 2 * 4 * 8)); CK(hipMalloc(&out, (size_t)entries * 2 * 4)); CK(hipMalloc(&hist, 128 * 4));
        for (int mode = 0; mode < 3; ++mode) {
            CK(hipMemset(tab, 0, (size_t)entries * 2 * 4 * 8)); CK(hipMemset(hist, 0, 128 * 4));
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(a));
                if (mode == 0) k<0><<<threads / 256, 256>>>(tab, entries, per_thread, rep == 0 ? hist : nullptr);
                if (mode == 1) k<1><<<threads / 256, 256>>>(tab, entries, per_thread, rep == 0 ? hist : nullptr);
                if (mode == 2) k<2><<<threads / 256, 256>>>(tab, entries, per_thread, rep == 0 ? hist : nullptr);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
            }
            // correctness: total of channel 0 over all copies must be 4 * threads * per_thread
            reduce8<<<(entries * 2 + 255) / 256, 256>>>(tab, out, (size_t)entries * 2);
            std::vector<float> h((size_t)entries * 2);
            CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
            double s0 = 0, s1 = 0; for (size_t i = 0; i < entries; ++i) { s0 += h[2 * i]; s1 += h[2 * i + 1]; }
            const double expect = 4.0 * threads * per_thread;
            printf("entries %8u mode %d: %.3f ms  %.1f G atomics/s   sum0 %.0f (expect %.0f) sum1 %.0f (expect %.0f)\n", entries, mode, best,
                   2.0 * threads * per_thread / best / 1e6, s0, expect, s1, 2 * expect);
            if (mode == 1) {
                std::vector<uint32_t> hh(128); CK(hipMemcpy(hh.data(), hist, 512, hipMemcpyDeviceToHost));
                printf("  blockIdx%%8 -> XCC histogram:"); for (int r = 0; r < 8; ++r) { printf(" ["); for (int c = 0; c < 8; ++c) printf("%u ", hh[r * 16 + c]); printf("]"); } printf("\n");
            }
        }
        CK(hipFree(tab)); CK(hipFree(out)); CK(hipFree(hist));
    }
    return 0;
}
