// avatarcraft_amd/csrc/hashgrid.hip -- multiresolution hash-grid encoder for gfx950 (stand-alone operator).
//
// Replaces the reference's `_hash_encoder` extension (encoder/hashencoder/src/hashencoder.cu):
//   kernel_grid (:73-220)           -> hash_fwd_kernel
//   kernel_grid_backward (:223-308) -> hash_bwd_kernel
//   kernel_input_backward (:311-337)-> hash_input_bwd_kernel
// The fused renderer (render_fused.hip) has its own in-register copy of the gather; this file is
// the drop-in operator behind `_backend.hash_encode_forward/backward`.
//
// MI355X notes: HBM/L2-bound gather.  One lane = one (point, level); the level is blockIdx.y so
// that a workgroup only touches one level's table (coarse levels stay L2/MALL resident).  The
// C features of a corner are one 4/8/16-byte load; level constants (scale, resolution, offset,
// hashed?) come in as launch constants from the host table, no exp2f on the device (bit-exact
// corner indices, SURVEY section 7 hard part ii).
#include "ac_common.hpp"
#include "ac_devmath.hpp"
#include <hip/hip_fp16.h>

using namespace acdev;

namespace {

template <uint32_t D>
__device__ __forceinline__ uint32_t grid_index(const uint32_t (&pg)[D], uint32_t stride1, uint32_t size,
                                               uint32_t hashed, uint32_t mask)
{
    uint32_t index;
    if (hashed) {
        constexpr uint32_t primes[3] = { 1u, 2654435761u, 805459861u };
        index = 0;
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) index ^= pg[d] * primes[d];
    } else {
        uint32_t stride = 1; index = 0;
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) { index += pg[d] * stride; stride *= stride1; }
    }
    if (mask) index &= mask;
    else if (index >= size) index %= size;
    return index;
}

template <uint32_t C> struct Feat;
template <> struct Feat<1> { float v[1]; __device__ void load(const float *p) { v[0] = p[0]; } };
template <> struct Feat<2> { float v[2]; __device__ void load(const float *p) { float2 t = *reinterpret_cast<const float2 *>(p); v[0] = t.x; v[1] = t.y; } };
template <> struct Feat<4> { float v[4]; __device__ void load(const float *p) { float4 t = *reinterpret_cast<const float4 *>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; } };
template <> struct Feat<8> { float v[8]; __device__ void load(const float *p) {
    float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; } };

template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void hash_fwd_kernel(const float *__restrict__ inputs, const float *__restrict__ grid,
                                                       float *__restrict__ outputs, uint32_t B, ac::LevelTable lt,
                                                       int calc_grad, float *__restrict__ dy_dx)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y, L = lt.L;
    const float scale = lt.scale[level];
    const uint32_t stride1 = lt.stride1[level], size = lt.size[level], hashed = lt.hashed[level], mask = lt.pow2mask[level];
    const float *g = grid + (size_t)lt.offset[level] * C;
    float x[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) { x[d] = inputs[(size_t)b * D + d]; oob |= (x[d] < 0.0f) | (x[d] > 1.0f); }
    float *out = outputs + ((size_t)level * B + b) * C;
    float *dd = dy_dx + (size_t)b * D * L * C + (size_t)level * D * C;
    if (oob) {
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) out[c] = 0.0f;
        if (calc_grad) for (uint32_t i = 0; i < D * C; ++i) dd[i] = 0.0f;
        return;
    }
    float pos[D]; uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        float p = fma_(x[d], scale, 0.5f);
        float fl = __builtin_floorf(p);
        pg[d] = (uint32_t)fl;
        pos[d] = p - (float)pg[d];
    }
    float acc[C];
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) acc[c] = 0.0f;
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); ++idx) {
        float w = 1.0f; uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) {
            if ((idx & (1u << d)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
            else { w *= pos[d]; pl[d] = pg[d] + 1u; }
        }
        const uint32_t index = grid_index<D>(pl, stride1, size, hashed, mask);
        Feat<C> f; f.load(g + (size_t)index * C);
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) acc[c] = fma_(w, f.v[c], acc[c]);
    }
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) out[c] = acc[c];
    if (calc_grad) {
#pragma unroll
        for (uint32_t gd = 0; gd < D; ++gd) {
            float rg[C];
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) rg[c] = 0.0f;
#pragma unroll
            for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
                float w = scale; uint32_t pl[D];
#pragma unroll
                for (uint32_t nd = 0; nd < D - 1; ++nd) {
                    const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                    if ((idx & (1u << nd)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
                    else { w *= pos[d]; pl[d] = pg[d] + 1u; }
                }
                pl[gd] = pg[gd];
                Feat<C> fl; fl.load(g + (size_t)grid_index<D>(pl, stride1, size, hashed, mask) * C);
                pl[gd] = pg[gd] + 1u;
                Feat<C> fr; fr.load(g + (size_t)grid_index<D>(pl, stride1, size, hashed, mask) * C);
#pragma unroll
                for (uint32_t c = 0; c < C; ++c) rg[c] = fma_(w, fr.v[c] - fl.v[c], rg[c]);
            }
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) dd[gd * C + c] = rg[c];
        }
    }
}

template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void hash_bwd_kernel(const float *__restrict__ grad, const float *__restrict__ inputs,
                                                       float *__restrict__ grad_grid, uint32_t B, ac::LevelTable lt)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    const float scale = lt.scale[level];
    const uint32_t stride1 = lt.stride1[level], size = lt.size[level], hashed = lt.hashed[level], mask = lt.pow2mask[level];
    float *gg = grad_grid + (size_t)lt.offset[level] * C;
    float pos[D]; uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        const float x = inputs[(size_t)b * D + d];
        if (x < 0.0f || x > 1.0f) return;           // grad is zero-initialised by the caller
        float p = fma_(x, scale, 0.5f);
        pg[d] = (uint32_t)__builtin_floorf(p);
        pos[d] = p - (float)pg[d];
    }
    float gcur[C];
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) gcur[c] = grad[((size_t)level * B + b) * C + c];
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); ++idx) {
        float w = 1.0f; uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) {
            if ((idx & (1u << d)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
            else { w *= pos[d]; pl[d] = pg[d] + 1u; }
        }
        const uint32_t index = grid_index<D>(pl, stride1, size, hashed, mask);
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) unsafeAtomicAdd(gg + (size_t)index * C + c, w * gcur[c]);
    }
}

// grad_inputs[b,d] = sum_l sum_c grad[l,b,c] * dy_dx[b,l,d,c]   (hashencoder.cu:311-337)
__global__ __launch_bounds__(256) void hash_input_bwd_kernel(const float *__restrict__ grad, const float *__restrict__ dy_dx,
                                                             float *__restrict__ grad_inputs, uint32_t B, uint32_t D,
                                                             uint32_t C, uint32_t L)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float *dd = dy_dx + (size_t)b * L * D * C;
    float r = 0.0f;
    for (uint32_t l = 0; l < L; ++l)
        for (uint32_t ch = 0; ch < C; ++ch)
            r = fma_(grad[((size_t)l * B + b) * C + ch], dd[l * D * C + d * C + ch], r);
    grad_inputs[t] = r;
}

template <uint32_t D>
__global__ __launch_bounds__(256) void hash_corner_kernel(const float *__restrict__ inputs, uint32_t *__restrict__ out,
                                                          uint32_t B, ac::LevelTable lt)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    uint32_t pg[D]; bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        const float x = inputs[(size_t)b * D + d];
        oob |= (x < 0.0f) | (x > 1.0f);
        pg[d] = (uint32_t)__builtin_floorf(fma_(x, lt.scale[level], 0.5f));
    }
    uint32_t *o = out + ((size_t)level * B + b) * (1u << D);
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); ++idx) {
        uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) pl[d] = pg[d] + ((idx >> d) & 1u);
        o[idx] = oob ? 0xffffffffu
                     : grid_index<D>(pl, lt.stride1[level], lt.size[level], lt.hashed[level], lt.pow2mask[level]);
    }
}

// ---- half / double tensors (hashencoder.cu:352,391 AT_DISPATCH_FLOATING_TYPES_AND_HALF) ----------------------------------------------------------------
// Storage type T, accumulator type A.  What does not depend on T, as in the reference: the range test on the stored value, cell position and
// interpolation weights in fp32 from (float)inputs (hashencoder.cu:125-133,142-154).  half: features widened on load, the eight corners accumulated in
// fp32 with fma and rounded ONCE on store (the reference rounds every partial sum to half through c10::Half's operators -- lower accuracy, never
// exercised and not reproducible without its CUDA build: DESIGN.md section 3); table gradient as packed half2 atomics for even C like
// hashencoder.cu:293-299 (scalar half atomics for C = 1).  double: accumulated in double; fp64 atomics.
template <class T> struct HgTy;
template <> struct HgTy<__half> {
    using A = float;
    static __device__ __forceinline__ float up(__half v) { return __half2float(v); }
    // the fp32 value is pinned in a register first: left alone, the compiler fuses the last fma and the conversion into v_fma_mixlo_f16 (ONE rounding of
    // the exact fma to half), which differs from "fp32 result, then rounded to half" whenever the fp32 result sits on a half tie -- common with half operands
    static __device__ __forceinline__ __half down(float v) { asm volatile("" : "+v"(v)); return __float2half(v); }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return fma_(a, b, c); }
};
template <> struct HgTy<double> {
    using A = double;
    static __device__ __forceinline__ double up(double v) { return v; }
    static __device__ __forceinline__ double down(double v) { return v; }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
};

template <class T, uint32_t D>
__device__ __forceinline__ bool locate_typed(const T *__restrict__ in, float scale, float (&pos)[D], uint32_t (&pg)[D])
{
    using Y = HgTy<T>;
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        const typename Y::A xv = Y::up(in[d]);
        oob |= (xv < 0) | (xv > 1);
        const float p = fma_((float)xv, scale, 0.5f);
        pg[d] = (uint32_t)__builtin_floorf(p);
        pos[d] = p - (float)pg[d];
    }
    return oob;
}

template <class T, uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void hash_fwd_typed_kernel(const T *__restrict__ inputs, const T *__restrict__ grid, T *__restrict__ outputs,
                                                             uint32_t B, ac::LevelTable lt, int calc_grad, T *__restrict__ dy_dx)
{
    using Y = HgTy<T>; using A = typename Y::A;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y, L = lt.L;
    const float scale = lt.scale[level];
    const uint32_t stride1 = lt.stride1[level], size = lt.size[level], hashed = lt.hashed[level], mask = lt.pow2mask[level];
    const T *g = grid + (size_t)lt.offset[level] * C;
    T *out = outputs + ((size_t)level * B + b) * C;
    T *dd = dy_dx + (size_t)b * D * L * C + (size_t)level * D * C;
    float pos[D]; uint32_t pg[D];
    if (locate_typed<T, D>(inputs + (size_t)b * D, scale, pos, pg)) {
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) out[c] = Y::down((A)0);
        if (calc_grad) for (uint32_t i = 0; i < D * C; ++i) dd[i] = Y::down((A)0);
        return;
    }
    A acc[C];
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) acc[c] = (A)0;
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); ++idx) {
        float w = 1.0f; uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) {
            if ((idx & (1u << d)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
            else { w *= pos[d]; pl[d] = pg[d] + 1u; }
        }
        const T *f = g + (size_t)grid_index<D>(pl, stride1, size, hashed, mask) * C;
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) acc[c] = Y::fma((A)w, Y::up(f[c]), acc[c]);
    }
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) out[c] = Y::down(acc[c]);
    if (calc_grad) {
#pragma unroll
        for (uint32_t gd = 0; gd < D; ++gd) {
            A rg[C];
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) rg[c] = (A)0;
#pragma unroll
            for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
                float w = scale; uint32_t pl[D];
#pragma unroll
                for (uint32_t nd = 0; nd < D - 1; ++nd) {
                    const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                    if ((idx & (1u << nd)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
                    else { w *= pos[d]; pl[d] = pg[d] + 1u; }
                }
                pl[gd] = pg[gd];
                const T *fl = g + (size_t)grid_index<D>(pl, stride1, size, hashed, mask) * C;
                pl[gd] = pg[gd] + 1u;
                const T *fr = g + (size_t)grid_index<D>(pl, stride1, size, hashed, mask) * C;
#pragma unroll
                for (uint32_t c = 0; c < C; ++c) rg[c] = Y::fma((A)w, Y::up(fr[c]) - Y::up(fl[c]), rg[c]);
            }
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) dd[gd * C + c] = Y::down(rg[c]);
        }
    }
}

template <uint32_t C> __device__ __forceinline__ void scatter_typed(double *t, float w, const double (&gc)[C])
{
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) unsafeAtomicAdd(t + c, (double)w * gc[c]);
}
template <uint32_t C> __device__ __forceinline__ void scatter_typed(__half *t, float w, const float (&gc)[C])
{
    if constexpr (C % 2 == 0) {
#pragma unroll
        for (uint32_t c = 0; c < C; c += 2) unsafeAtomicAdd(reinterpret_cast<__half2 *>(t + c), __halves2half2(HgTy<__half>::down(w * gc[c]), HgTy<__half>::down(w * gc[c + 1])));
    } else {
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) unsafeAtomicAdd(t + c, HgTy<__half>::down(w * gc[c]));
    }
}

template <class T, uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void hash_bwd_typed_kernel(const T *__restrict__ grad, const T *__restrict__ inputs, T *__restrict__ grad_grid,
                                                             uint32_t B, ac::LevelTable lt)
{
    using Y = HgTy<T>; using A = typename Y::A;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    const uint32_t stride1 = lt.stride1[level], size = lt.size[level], hashed = lt.hashed[level], mask = lt.pow2mask[level];
    T *gg = grad_grid + (size_t)lt.offset[level] * C;
    float pos[D]; uint32_t pg[D];
    if (locate_typed<T, D>(inputs + (size_t)b * D, lt.scale[level], pos, pg)) return;           // grad_embeddings is zero-initialised by the caller
    A gcur[C];
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) gcur[c] = Y::up(grad[((size_t)level * B + b) * C + c]);
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); ++idx) {
        float w = 1.0f; uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) {
            if ((idx & (1u << d)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
            else { w *= pos[d]; pl[d] = pg[d] + 1u; }
        }
        scatter_typed<C>(gg + (size_t)grid_index<D>(pl, stride1, size, hashed, mask) * C, w, gcur);
    }
}

template <class T>
__global__ __launch_bounds__(256) void hash_input_bwd_typed_kernel(const T *__restrict__ grad, const T *__restrict__ dy_dx, T *__restrict__ grad_inputs,
                                                                   uint32_t B, uint32_t D, uint32_t C, uint32_t L)
{
    using Y = HgTy<T>; using A = typename Y::A;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const T *dd = dy_dx + (size_t)b * L * D * C;
    A r = (A)0;
    for (uint32_t l = 0; l < L; ++l)
        for (uint32_t ch = 0; ch < C; ++ch)
            r = Y::fma(Y::up(grad[((size_t)l * B + b) * C + ch]), Y::up(dd[l * D * C + d * C + ch]), r);
    grad_inputs[t] = Y::down(r);
}

template <class T, uint32_t D>
int launch_fwd_typed(uint32_t C, dim3 grid, hipStream_t st, const T *in, const T *emb, T *out, uint32_t B, const ac::LevelTable &lt, int cg, T *dy_dx)
{
    switch (C) {
    case 1: hipLaunchKernelGGL((hash_fwd_typed_kernel<T, D, 1>), grid, dim3(256), 0, st, in, emb, out, B, lt, cg, dy_dx); break;
    case 2: hipLaunchKernelGGL((hash_fwd_typed_kernel<T, D, 2>), grid, dim3(256), 0, st, in, emb, out, B, lt, cg, dy_dx); break;
    case 4: hipLaunchKernelGGL((hash_fwd_typed_kernel<T, D, 4>), grid, dim3(256), 0, st, in, emb, out, B, lt, cg, dy_dx); break;
    case 8: hipLaunchKernelGGL((hash_fwd_typed_kernel<T, D, 8>), grid, dim3(256), 0, st, in, emb, out, B, lt, cg, dy_dx); break;
    default: return AC_ERR_BAD_ARG;
    }
    return AC_OK;
}
template <class T, uint32_t D>
int launch_bwd_typed(uint32_t C, dim3 grid, hipStream_t st, const T *grad, const T *in, T *gg, uint32_t B, const ac::LevelTable &lt)
{
    switch (C) {
    case 1: hipLaunchKernelGGL((hash_bwd_typed_kernel<T, D, 1>), grid, dim3(256), 0, st, grad, in, gg, B, lt); break;
    case 2: hipLaunchKernelGGL((hash_bwd_typed_kernel<T, D, 2>), grid, dim3(256), 0, st, grad, in, gg, B, lt); break;
    case 4: hipLaunchKernelGGL((hash_bwd_typed_kernel<T, D, 4>), grid, dim3(256), 0, st, grad, in, gg, B, lt); break;
    case 8: hipLaunchKernelGGL((hash_bwd_typed_kernel<T, D, 8>), grid, dim3(256), 0, st, grad, in, gg, B, lt); break;
    default: return AC_ERR_BAD_ARG;
    }
    return AC_OK;
}

template <class T>
int encode_forward_typed(const T *inputs, const T *embeddings, const int32_t *offsets_host, T *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                         float S, uint32_t H, int calc_grad_inputs, T *dy_dx, hipStream_t st)
{
    ac::LevelTable lt; ac::make_level_table(lt, L, D, S, H, offsets_host);
    const dim3 grid((B + 255) / 256, L);
    return (D == 2) ? launch_fwd_typed<T, 2>(C, grid, st, inputs, embeddings, outputs, B, lt, calc_grad_inputs, dy_dx)
                    : launch_fwd_typed<T, 3>(C, grid, st, inputs, embeddings, outputs, B, lt, calc_grad_inputs, dy_dx);
}
template <class T>
int encode_backward_typed(const T *grad, const T *inputs, const int32_t *offsets_host, T *grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                          uint32_t L, float S, uint32_t H, int calc_grad_inputs, const T *dy_dx, T *grad_inputs, hipStream_t st)
{
    ac::LevelTable lt; ac::make_level_table(lt, L, D, S, H, offsets_host);
    const dim3 grid((B + 255) / 256, L);
    const int rc = (D == 2) ? launch_bwd_typed<T, 2>(C, grid, st, grad, inputs, grad_embeddings, B, lt)
                            : launch_bwd_typed<T, 3>(C, grid, st, grad, inputs, grad_embeddings, B, lt);
    if (rc) return rc;
    if (calc_grad_inputs)
        hipLaunchKernelGGL(hash_input_bwd_typed_kernel<T>, dim3((B * D + 255) / 256), dim3(256), 0, st, grad, dy_dx, grad_inputs, B, D, C, L);
    return AC_OK;
}

template <uint32_t D>
int launch_fwd(uint32_t C, dim3 grid, hipStream_t st, const float *in, const float *emb, float *out, uint32_t B,
               const ac::LevelTable &lt, int cg, float *dy_dx)
{
    switch (C) {
    case 1: hipLaunchKernelGGL((hash_fwd_kernel<D, 1>), grid, dim3(256), 0, st, in, emb, out, B, lt, cg, dy_dx); break;
    case 2: hipLaunchKernelGGL((hash_fwd_kernel<D, 2>), grid, dim3(256), 0, st, in, emb, out, B, lt, cg, dy_dx); break;
    case 4: hipLaunchKernelGGL((hash_fwd_kernel<D, 4>), grid, dim3(256), 0, st, in, emb, out, B, lt, cg, dy_dx); break;
    case 8: hipLaunchKernelGGL((hash_fwd_kernel<D, 8>), grid, dim3(256), 0, st, in, emb, out, B, lt, cg, dy_dx); break;
    default: return AC_ERR_BAD_ARG;
    }
    return AC_OK;
}
template <uint32_t D>
int launch_bwd(uint32_t C, dim3 grid, hipStream_t st, const float *grad, const float *in, float *gg, uint32_t B,
               const ac::LevelTable &lt)
{
    switch (C) {
    case 1: hipLaunchKernelGGL((hash_bwd_kernel<D, 1>), grid, dim3(256), 0, st, grad, in, gg, B, lt); break;
    case 2: hipLaunchKernelGGL((hash_bwd_kernel<D, 2>), grid, dim3(256), 0, st, grad, in, gg, B, lt); break;
    case 4: hipLaunchKernelGGL((hash_bwd_kernel<D, 4>), grid, dim3(256), 0, st, grad, in, gg, B, lt); break;
    case 8: hipLaunchKernelGGL((hash_bwd_kernel<D, 8>), grid, dim3(256), 0, st, grad, in, gg, B, lt); break;
    default: return AC_ERR_BAD_ARG;
    }
    return AC_OK;
}

int check_cfg(uint32_t D, uint32_t C, uint32_t L, const int32_t *offsets_host)
{
    if (!(D == 2 || D == 3) || !(C == 1 || C == 2 || C == 4 || C == 8)) {
        ac::set_error("GridEncoding: C must be 1, 2, 4, or 8 (and D 2 or 3); got D=%u C=%u", D, C);
        return AC_ERR_BAD_ARG;
    }
    if (L == 0 || L > AC_MAX_LEVELS) { ac::set_error("GridEncoding: L=%u out of range (1..%d)", L, AC_MAX_LEVELS); return AC_ERR_BAD_ARG; }
    if (!offsets_host) { ac::set_error("GridEncoding: offsets_host is NULL"); return AC_ERR_BAD_ARG; }
    return AC_OK;
}

}  // namespace

AC_API int ac_hash_encode_forward(const float *inputs, const float *embeddings, const int32_t *offsets,
                                  const int32_t *offsets_host, float *outputs, uint32_t B, uint32_t D, uint32_t C,
                                  uint32_t L, float S, uint32_t H, int calc_grad_inputs, float *dy_dx, ac_stream_t stream)
{
    (void)offsets;
    if (int rc = check_cfg(D, C, L, offsets_host)) return rc;
    if (B == 0) return AC_OK;
    if (!inputs || !embeddings || !outputs || (calc_grad_inputs && !dy_dx)) { ac::set_error("hash_encode_forward: NULL buffer"); return AC_ERR_BAD_ARG; }
    ac::LevelTable lt; ac::make_level_table(lt, L, D, S, H, offsets_host);
    dim3 grid((B + 255) / 256, L);
    hipStream_t st = (hipStream_t)stream;
    int rc = (D == 2) ? launch_fwd<2>(C, grid, st, inputs, embeddings, outputs, B, lt, calc_grad_inputs, dy_dx)
                      : launch_fwd<3>(C, grid, st, inputs, embeddings, outputs, B, lt, calc_grad_inputs, dy_dx);
    if (rc) return rc;
    return ac::check_launch("hash_encode_forward");
}

AC_API int ac_hash_encode_backward(const float *grad, const float *inputs, const float *embeddings, const int32_t *offsets,
                                   const int32_t *offsets_host, float *grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                   uint32_t L, float S, uint32_t H, int calc_grad_inputs, const float *dy_dx,
                                   float *grad_inputs, ac_stream_t stream)
{
    (void)offsets; (void)embeddings;
    if (int rc = check_cfg(D, C, L, offsets_host)) return rc;
    if (B == 0) return AC_OK;
    if (!grad || !inputs || !grad_embeddings || (calc_grad_inputs && (!dy_dx || !grad_inputs))) { ac::set_error("hash_encode_backward: NULL buffer"); return AC_ERR_BAD_ARG; }
    ac::LevelTable lt; ac::make_level_table(lt, L, D, S, H, offsets_host);
    dim3 grid((B + 255) / 256, L);
    hipStream_t st = (hipStream_t)stream;
    int rc = (D == 2) ? launch_bwd<2>(C, grid, st, grad, inputs, grad_embeddings, B, lt)
                      : launch_bwd<3>(C, grid, st, grad, inputs, grad_embeddings, B, lt);
    if (rc) return rc;
    if (calc_grad_inputs)
        hipLaunchKernelGGL(hash_input_bwd_kernel, dim3((B * D + 255) / 256), dim3(256), 0, st, grad, dy_dx, grad_inputs, B, D, C, L);
    return ac::check_launch("hash_encode_backward");
}

AC_API int ac_hash_corner_indices(const float *inputs, const int32_t *offsets_host, uint32_t *corner_idx, uint32_t B,
                                  uint32_t D, uint32_t L, float S, uint32_t H, ac_stream_t stream)
{
    if (int rc = check_cfg(D, 1, L, offsets_host)) return rc;
    if (B == 0) return AC_OK;
    ac::LevelTable lt; ac::make_level_table(lt, L, D, S, H, offsets_host);
    dim3 grid((B + 255) / 256, L);
    hipStream_t st = (hipStream_t)stream;
    if (D == 2) hipLaunchKernelGGL((hash_corner_kernel<2>), grid, dim3(256), 0, st, inputs, corner_idx, B, lt);
    else hipLaunchKernelGGL((hash_corner_kernel<3>), grid, dim3(256), 0, st, inputs, corner_idx, B, lt);
    return ac::check_launch("hash_corner_indices");
}

AC_API int ac_hash_encode_forward_typed(int dtype, const void *inputs, const void *embeddings, const int32_t *offsets, const int32_t *offsets_host,
                                        void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs,
                                        void *dy_dx, ac_stream_t stream)
{
    if (dtype == AC_DTYPE_F32)
        return ac_hash_encode_forward((const float *)inputs, (const float *)embeddings, offsets, offsets_host, (float *)outputs, B, D, C, L, S, H,
                                      calc_grad_inputs, (float *)dy_dx, stream);
    if (dtype != AC_DTYPE_F16 && dtype != AC_DTYPE_F64) { ac::set_error("hash_encode_forward: inputs must be a floating tensor (dtype code %d)", dtype); return AC_ERR_BAD_ARG; }
    if (int rc = check_cfg(D, C, L, offsets_host)) return rc;
    if (B == 0) return AC_OK;
    if (!inputs || !embeddings || !outputs || (calc_grad_inputs && !dy_dx)) { ac::set_error("hash_encode_forward: NULL buffer"); return AC_ERR_BAD_ARG; }
    const int rc = dtype == AC_DTYPE_F16
        ? encode_forward_typed<__half>((const __half *)inputs, (const __half *)embeddings, offsets_host, (__half *)outputs, B, D, C, L, S, H, calc_grad_inputs, (__half *)dy_dx, (hipStream_t)stream)
        : encode_forward_typed<double>((const double *)inputs, (const double *)embeddings, offsets_host, (double *)outputs, B, D, C, L, S, H, calc_grad_inputs, (double *)dy_dx, (hipStream_t)stream);
    if (rc) return rc;
    return ac::check_launch("hash_encode_forward");
}

AC_API int ac_hash_encode_backward_typed(int dtype, const void *grad, const void *inputs, const void *embeddings, const int32_t *offsets,
                                         const int32_t *offsets_host, void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                         uint32_t H, int calc_grad_inputs, const void *dy_dx, void *grad_inputs, ac_stream_t stream)
{
    if (dtype == AC_DTYPE_F32)
        return ac_hash_encode_backward((const float *)grad, (const float *)inputs, (const float *)embeddings, offsets, offsets_host, (float *)grad_embeddings,
                                       B, D, C, L, S, H, calc_grad_inputs, (const float *)dy_dx, (float *)grad_inputs, stream);
    (void)embeddings;
    if (dtype != AC_DTYPE_F16 && dtype != AC_DTYPE_F64) { ac::set_error("hash_encode_backward: grad must be a floating tensor (dtype code %d)", dtype); return AC_ERR_BAD_ARG; }
    if (int rc = check_cfg(D, C, L, offsets_host)) return rc;
    if (B == 0) return AC_OK;
    if (!grad || !inputs || !grad_embeddings || (calc_grad_inputs && (!dy_dx || !grad_inputs))) { ac::set_error("hash_encode_backward: NULL buffer"); return AC_ERR_BAD_ARG; }
    const int rc = dtype == AC_DTYPE_F16
        ? encode_backward_typed<__half>((const __half *)grad, (const __half *)inputs, offsets_host, (__half *)grad_embeddings, B, D, C, L, S, H, calc_grad_inputs, (const __half *)dy_dx, (__half *)grad_inputs, (hipStream_t)stream)
        : encode_backward_typed<double>((const double *)grad, (const double *)inputs, offsets_host, (double *)grad_embeddings, B, D, C, L, S, H, calc_grad_inputs, (const double *)dy_dx, (double *)grad_inputs, (hipStream_t)stream);
    if (rc) return rc;
    return ac::check_launch("hash_encode_backward");
}
