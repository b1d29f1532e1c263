// avatarcraft_amd/csrc/shencoder.hip -- real spherical-harmonics direction encoder (degree 1..8) for gfx950.
//
// Replaces the reference's `_sh_encoder` extension (encoder/shencoder/src/shencoder.cu:28-384):
//   kernel_sh -> sh_fwd_kernel (values + analytic Jacobian), kernel_sh_backward -> sh_bwd_kernel.
// The 64 basis polynomials are evaluated from the generated monomial table (ac_sh_table.hpp,
// tools/gen_sh_tables.py) on the RAW input, exactly as the reference does (no normalisation).
// Pure ALU + streaming stores; one lane per point, outputs written as contiguous rows.
#include "ac_common.hpp"
#include "ac_devmath.hpp"
#include "ac_sh_table.hpp"
#include <hip/hip_fp16.h>

using namespace acdev;

namespace {


__device__ __forceinline__ float sh_eval(const unsigned short *off, const float *coef, const unsigned char (*ex)[3], int idx,
                                         const float (&px)[8], const float (&py)[8], const float (&pz)[8])
{
    float acc = 0.0f;
    for (int m = off[idx]; m < off[idx + 1]; ++m) {
        const float mono = (px[ex[m][0]] * py[ex[m][1]]) * pz[ex[m][2]];
        acc = fma_(coef[m], mono, acc);
    }
    return acc;
}

__global__ __launch_bounds__(256) void sh_fwd_kernel(const float *__restrict__ inputs, float *__restrict__ outputs, uint32_t B,
                                                     uint32_t C, int calc_grad, float *__restrict__ dy_dx)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t C2 = C * C;
    float p[3][8];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float v = inputs[(size_t)b * 3 + a];
        p[a][0] = 1.0f;
#pragma unroll
        for (int k = 1; k < 8; ++k) p[a][k] = p[a][k - 1] * v;
    }
    for (uint32_t i = 0; i < C2; ++i)
        outputs[(size_t)b * C2 + i] = sh_eval(AC_SH_OFF0, AC_SH_COEF0, AC_SH_EXP0, i, p[0], p[1], p[2]);
    if (calc_grad) {
        float *dx = dy_dx + (size_t)b * 3 * C2, *dy = dx + C2, *dz = dy + C2;
        for (uint32_t i = 0; i < C2; ++i) {
            dx[i] = sh_eval(AC_SH_OFF1, AC_SH_COEF1, AC_SH_EXP1, i, p[0], p[1], p[2]);
            dy[i] = sh_eval(AC_SH_OFF2, AC_SH_COEF2, AC_SH_EXP2, i, p[0], p[1], p[2]);
            dz[i] = sh_eval(AC_SH_OFF3, AC_SH_COEF3, AC_SH_EXP3, i, p[0], p[1], p[2]);
        }
    }
}

// grad_inputs[b,d] += sum_ch grad[b,ch] * dy_dx[b,d,ch]   (shencoder.cu:360-384)
__global__ __launch_bounds__(256) void sh_bwd_kernel(const float *__restrict__ grad, uint32_t B, uint32_t C,
                                                     const float *__restrict__ dy_dx, float *__restrict__ grad_inputs)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = t / 3;
    if (b >= B) return;
    const uint32_t d = t - b * 3, C2 = C * C;
    float acc = grad_inputs[t];
    for (uint32_t ch = 0; ch < C2; ++ch)
        acc = fma_(grad[(size_t)b * C2 + ch], dy_dx[(size_t)b * 3 * C2 + d * C2 + ch], acc);
    grad_inputs[t] = acc;
}

// ---- half / double instantiations (shencoder.cu:337,380 dispatch AT_DISPATCH_FLOATING_TYPES_AND_HALF over the dtype of `inputs`) -----------------------
// Storage type T, arithmetic type A: half tensors are widened on load, evaluated with the fp32 routine's arithmetic and rounded ONCE on store (the
// reference evaluates every product in half through c10::Half's operators: lower accuracy, never exercised -- SURVEY 0.5 --, and not reproducible here
// without its CUDA build; stated in DESIGN.md section 3); double tensors are evaluated in double with the double coefficient tables.
template <class T> struct ShTy;
template <> struct ShTy<__half> {
    using A = float;
    static __device__ __forceinline__ float up(__half v) { return __half2float(v); }
    // the fp32 value is pinned in a register first: left alone, the compiler fuses the last fma and the conversion into v_fma_mixlo_f16 (ONE rounding of
    // the exact fma to half), which differs from "fp32 result, then rounded to half" whenever the fp32 result sits on a half tie -- common with half operands
    static __device__ __forceinline__ __half down(float v) { asm volatile("" : "+v"(v)); return __float2half(v); }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return fma_(a, b, c); }
    static __device__ __forceinline__ const float *coef(int kind) { return kind == 0 ? AC_SH_COEF0 : kind == 1 ? AC_SH_COEF1 : kind == 2 ? AC_SH_COEF2 : AC_SH_COEF3; }
};
template <> struct ShTy<double> {
    using A = double;
    static __device__ __forceinline__ double up(double v) { return v; }
    static __device__ __forceinline__ double down(double v) { return v; }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static __device__ __forceinline__ const double *coef(int kind) { return kind == 0 ? AC_SH_COEF0D : kind == 1 ? AC_SH_COEF1D : kind == 2 ? AC_SH_COEF2D : AC_SH_COEF3D; }
};

template <class T>
__global__ __launch_bounds__(256) void sh_fwd_typed_kernel(const T *__restrict__ inputs, T *__restrict__ outputs, uint32_t B, uint32_t C, int calc_grad,
                                                           T *__restrict__ dy_dx)
{
    using Y = ShTy<T>; using A = typename Y::A;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t C2 = C * C;
    A p[3][8];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const A v = Y::up(inputs[(size_t)b * 3 + a]);
        p[a][0] = (A)1;
#pragma unroll
        for (int k = 1; k < 8; ++k) p[a][k] = p[a][k - 1] * v;
    }
    const unsigned short *offs[4] = { AC_SH_OFF0, AC_SH_OFF1, AC_SH_OFF2, AC_SH_OFF3 };
    const unsigned char (*exps[4])[3] = { AC_SH_EXP0, AC_SH_EXP1, AC_SH_EXP2, AC_SH_EXP3 };
    for (int kind = 0; kind < (calc_grad ? 4 : 1); ++kind) {
        T *dst = kind == 0 ? outputs + (size_t)b * C2 : dy_dx + (size_t)b * 3 * C2 + (size_t)(kind - 1) * C2;
        const auto *coef = Y::coef(kind);
        for (uint32_t i = 0; i < C2; ++i) {
            A acc = (A)0;
            for (int m = offs[kind][i]; m < offs[kind][i + 1]; ++m) {
                const A mono = (p[0][exps[kind][m][0]] * p[1][exps[kind][m][1]]) * p[2][exps[kind][m][2]];
                acc = Y::fma((A)coef[m], mono, acc);
            }
            dst[i] = Y::down(acc);
        }
    }
}

template <class T>
__global__ __launch_bounds__(256) void sh_bwd_typed_kernel(const T *__restrict__ grad, uint32_t B, uint32_t C, const T *__restrict__ dy_dx,
                                                           T *__restrict__ grad_inputs)
{
    using Y = ShTy<T>; using A = typename Y::A;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = t / 3;
    if (b >= B) return;
    const uint32_t d = t - b * 3, C2 = C * C;
    A acc = Y::up(grad_inputs[t]);
    for (uint32_t ch = 0; ch < C2; ++ch)
        acc = Y::fma(Y::up(grad[(size_t)b * C2 + ch]), Y::up(dy_dx[(size_t)b * 3 * C2 + d * C2 + ch]), acc);
    grad_inputs[t] = Y::down(acc);
}

}  // namespace

AC_API int ac_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D, uint32_t C, int calc_grad_inputs,
                                float *dy_dx, ac_stream_t stream)
{
    if (D != 3) { ac::set_error("SH encoder only support input dim == 3 (got %u)", D); return AC_ERR_BAD_ARG; }
    if (C < 1 || C > 8) { ac::set_error("SH encoder only supports degree in [1, 8] (got %u)", C); return AC_ERR_BAD_ARG; }
    if (B == 0) return AC_OK;
    if (!inputs || !outputs || (calc_grad_inputs && !dy_dx)) { ac::set_error("sh_encode_forward: NULL buffer"); return AC_ERR_BAD_ARG; }
    hipLaunchKernelGGL(sh_fwd_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, inputs, outputs, B, C,
                       calc_grad_inputs, dy_dx);
    return ac::check_launch("sh_encode_forward");
}

AC_API int ac_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D, uint32_t C, const float *dy_dx,
                                 float *grad_inputs, ac_stream_t stream)
{
    (void)inputs;
    if (D != 3 || C < 1 || C > 8) { ac::set_error("SH encoder: unsupported input_dim=%u degree=%u", D, C); return AC_ERR_BAD_ARG; }
    if (B == 0) return AC_OK;
    if (!grad || !dy_dx || !grad_inputs) { ac::set_error("sh_encode_backward: NULL buffer"); return AC_ERR_BAD_ARG; }
    hipLaunchKernelGGL(sh_bwd_kernel, dim3((B * 3 + 255) / 256), dim3(256), 0, (hipStream_t)stream, grad, B, C, dy_dx, grad_inputs);
    return ac::check_launch("sh_encode_backward");
}

AC_API int ac_sh_encode_forward_typed(int dtype, const void *inputs, void *outputs, uint32_t B, uint32_t D, uint32_t C, int calc_grad_inputs,
                                      void *dy_dx, ac_stream_t stream)
{
    if (dtype == AC_DTYPE_F32) return ac_sh_encode_forward((const float *)inputs, (float *)outputs, B, D, C, calc_grad_inputs, (float *)dy_dx, stream);
    if (dtype != AC_DTYPE_F16 && dtype != AC_DTYPE_F64) { ac::set_error("sh_encode_forward: inputs must be a floating tensor (dtype code %d)", dtype); return AC_ERR_BAD_ARG; }
    if (D != 3) { ac::set_error("SH encoder only support input dim == 3 (got %u)", D); return AC_ERR_BAD_ARG; }
    if (C < 1 || C > 8) { ac::set_error("SH encoder only supports degree in [1, 8] (got %u)", C); return AC_ERR_BAD_ARG; }
    if (B == 0) return AC_OK;
    if (!inputs || !outputs || (calc_grad_inputs && !dy_dx)) { ac::set_error("sh_encode_forward: NULL buffer"); return AC_ERR_BAD_ARG; }
    const dim3 grid((B + 255) / 256);
    if (dtype == AC_DTYPE_F16)
        hipLaunchKernelGGL(sh_fwd_typed_kernel<__half>, grid, dim3(256), 0, (hipStream_t)stream, (const __half *)inputs, (__half *)outputs, B, C, calc_grad_inputs, (__half *)dy_dx);
    else
        hipLaunchKernelGGL(sh_fwd_typed_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, (const double *)inputs, (double *)outputs, B, C, calc_grad_inputs, (double *)dy_dx);
    return ac::check_launch("sh_encode_forward");
}

AC_API int ac_sh_encode_backward_typed(int dtype, const void *grad, const void *inputs, uint32_t B, uint32_t D, uint32_t C, const void *dy_dx,
                                       void *grad_inputs, ac_stream_t stream)
{
    if (dtype == AC_DTYPE_F32) return ac_sh_encode_backward((const float *)grad, (const float *)inputs, B, D, C, (const float *)dy_dx, (float *)grad_inputs, stream);
    if (dtype != AC_DTYPE_F16 && dtype != AC_DTYPE_F64) { ac::set_error("sh_encode_backward: grad must be a floating tensor (dtype code %d)", dtype); return AC_ERR_BAD_ARG; }
    if (D != 3 || C < 1 || C > 8) { ac::set_error("SH encoder: unsupported input_dim=%u degree=%u", D, C); return AC_ERR_BAD_ARG; }
    if (B == 0) return AC_OK;
    if (!grad || !dy_dx || !grad_inputs) { ac::set_error("sh_encode_backward: NULL buffer"); return AC_ERR_BAD_ARG; }
    const dim3 grid((B * 3 + 255) / 256);
    if (dtype == AC_DTYPE_F16)
        hipLaunchKernelGGL(sh_bwd_typed_kernel<__half>, grid, dim3(256), 0, (hipStream_t)stream, (const __half *)grad, B, C, (const __half *)dy_dx, (__half *)grad_inputs);
    else
        hipLaunchKernelGGL(sh_bwd_typed_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, (const double *)grad, B, C, (const double *)dy_dx, (double *)grad_inputs);
    return ac::check_launch("sh_encode_backward");
}
