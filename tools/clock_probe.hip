// micro-benchmark: which shader clock does an MI355X sustain under the instruction mixes of the fused renderer?
// s_memtime counts shader-clock cycles, s_memrealtime a constant 100 MHz: their ratio inside a kernel is the clock it ran at.
// build: hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MIX>   // 0: VALU fma only, 1: fp32 MFMA only, 2: MFMA + VALU interleaved 1 : 4
__global__ __launch_bounds__(512) void spin(uint32_t iters, unsigned long long *out, float *sink)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
    f32x4 acc0 = { 0, 0, 0, 0 }, acc1 = { 0, 0, 0, 0 };
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MIX != 0) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
            }
            if (MIX != 1) {
#pragma unroll
                for (int v = 0; v < 4; ++v) { c = __builtin_fmaf(c, b, a); d = __builtin_fmaf(d, b, c); }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0; out[2 * w + 1] = r1 - r0;
    }
    if (acc0[0] + acc1[1] + c + d == 123.456f) sink[0] = c;
}

template <int MIX>
static int run(const char *name, uint32_t blocks, uint32_t iters, unsigned long long *out_d, float *sink)
{
    const uint32_t waves = blocks * 8;
    unsigned long long *h = (unsigned long long *)malloc(waves * 16);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        spin<MIX><<<blocks, 512>>>(iters, out_d, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
    }
    CK(hipMemcpy(h, out_d, waves * 16, hipMemcpyDeviceToHost));
    double st = 0, sr = 0;
    for (uint32_t w = 0; w < waves; ++w) { st += (double)h[2 * w]; sr += (double)h[2 * w + 1]; }
    const double ghz = st / sr * 0.1;
    const double mfma = MIX == 0 ? 0.0 : (double)waves * iters * 16, valu = MIX == 1 ? 0.0 : (double)waves * iters * 64;
    printf("  %-44s %5u workgroups  %8.3f ms  shader clock %.3f GHz", name, blocks, ms, ghz);
    if (mfma > 0) printf("  %6.1f TFLOP/s fp32 MFMA (%.1f clk per MFMA and SIMD)", mfma * 2048 / ms * 1e-9, ms * 1e-3 * ghz * 1e9 / (mfma / (blocks < 256 ? blocks * 4.0 : 1024.0)));
    if (valu > 0) printf("  %.2f clk per VALU instruction and SIMD", ms * 1e-3 * ghz * 1e9 / (valu / (blocks < 256 ? blocks * 4.0 : 1024.0)));
    printf("\n");
    free(h);
    return 0;
}

int main()
{
    unsigned long long *out; float *sink;
    CK(hipMalloc(&out, 4096 * 8 * 16)); CK(hipMalloc(&sink, 64));
    for (uint32_t blocks : { 1u, 256u, 512u }) {
        if (run<0>("VALU fma chain (2 waves per SIMD)", blocks, 40000, out, sink)) return 1;
        if (run<1>("fp32 MFMA 16x16x4 (2 waves per SIMD)", blocks, 40000, out, sink)) return 1;
        if (run<2>("MFMA + VALU, 1 : 4 instructions", blocks, 40000, out, sink)) return 1;
    }
    return 0;
}
