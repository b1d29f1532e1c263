// micro-benchmark: LDS atomic throughput per CU -- ds_add_f32 vs ds_add_u32 vs ds_add_rtn_u32 vs plain read-modify-write vs ds_add_u64, random addresses
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomics_bench.hip -o tools/_bin/lds_atomics_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, uint32_t iters, uint32_t nslots)
{
    extern __shared__ float acc[];
    for (uint32_t e = threadIdx.x; e < nslots; e += blockDim.x) acc[e] = 0.0f;
    __syncthreads();
    uint32_t *iacc = reinterpret_cast<uint32_t *>(acc);
    uint32_t sink = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        const uint32_t idx = hash32((blockIdx.x * 1024u + threadIdx.x) * iters + i) % nslots;
        if (MODE == 0) atomicAdd(&acc[idx], 1.0f);
        if (MODE == 1) atomicAdd(&iacc[idx], 1u);
        if (MODE == 2) sink += atomicAdd(&iacc[idx], 1u);
        if (MODE == 3) acc[idx] += 1.0f;                       // racy read-modify-write (wrong sums): the non-atomic cost
        if (MODE == 4) atomicAdd(reinterpret_cast<unsigned long long *>(acc) + (idx >> 1), 0x100000001ull);        // ds_add_u64, half as many slots
    }
    __syncthreads();
    float s = 0.0f;
    for (uint32_t e = threadIdx.x; e < nslots; e += blockDim.x) s += MODE == 1 || MODE == 2 ? (float)iacc[e] : acc[e];
    out[blockIdx.x * 1024 + threadIdx.x] = s + (float)(sink & 1u) * 0.0f;
}

int main()
{
    const uint32_t iters = 2048, blocks = 512, nslots = 16384;
    float *out; CK(hipMalloc(&out, blocks * 1024 * 4));
    const char *names[5] = { "ds_add_f32 (atomicAdd float)", "ds_add_u32", "ds_add_rtn_u32", "plain RMW (racy)", "ds_add_u64" };
    for (int mode = 0; mode < 5; ++mode) {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(a));
            if (mode == 0) k<0><<<blocks, 1024, nslots * 4>>>(out, iters, nslots);
            if (mode == 1) k<1><<<blocks, 1024, nslots * 4>>>(out, iters, nslots);
            if (mode == 2) k<2><<<blocks, 1024, nslots * 4>>>(out, iters, nslots);
            if (mode == 3) k<3><<<blocks, 1024, nslots * 4>>>(out, iters, nslots);
            if (mode == 4) k<4><<<blocks, 1024, nslots * 4>>>(out, iters, nslots);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        const double ops = (double)blocks * 1024 * iters;
        printf("%-34s %.3f ms  %.1f G lane-ops/s  = %.2f lane-ops / clock / CU (2.4 GHz, 256 CUs)\n", names[mode], best, ops / best / 1e6, ops / (best * 1e-3) / 2.4e9 / 256);
    }
    return 0;
}
