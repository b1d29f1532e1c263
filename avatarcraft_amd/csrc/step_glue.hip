// avatarcraft_amd/csrc/step_glue.hip -- the small per-step pieces around the training render, so that a stylisation step (stylize.py:95-199)
// launches no torch kernel between the guidance gradient and the optimizer:
//   ac_weight_norm_forward : W = v * (g / ||v||_row) of every layer in one launch (torch.nn.utils.weight_norm, dim 0; models/instant_nsr.py:557-590)
//   ac_param_grads         : its backward for every layer + the bias gradients + d loss / d variance from the per-ray d loss / d inv_s,
//                            ACCUMULATED into the parameters' .grad buffers, one launch
//   ac_sds_upstream        : d / d weights_sum of smooth_l1(clamp(ws, 0, 1), clamp(ws_gt, 0, 1)) * scale (stylize.py:183-193) and the loss value
//   ac_adam_step           : optimizer.step() of torch.optim.Adam (stylize.py:199, :355-363) for every parameter tensor in ONE launch, optionally
//                            clearing the gradients it has just consumed (the next step's optimizer.zero_grad(), stylize.py:143)
// All reductions are wave-local trees in a fixed order: results do not depend on the launch.
#include "ac_common.hpp"

namespace {

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct WnArgs { ac_wn_layer l[AC_WN_MAX_LAYERS]; uint32_t row0[AC_WN_MAX_LAYERS + 1]; uint32_t n; };

__global__ __launch_bounds__(64) void weight_norm_fwd_kernel(const WnArgs a)
{
    // static indexing of the by-value argument (a dynamic index would move the whole struct to scratch memory)
    ac_wn_layer L = a.l[0];
    uint32_t r = blockIdx.x;
#pragma unroll
    for (int i = 1; i < AC_WN_MAX_LAYERS; ++i)
        if ((uint32_t)i < a.n && blockIdx.x >= a.row0[i]) { L = a.l[i]; r = blockIdx.x - a.row0[i]; }
    const int lane = threadIdx.x;
    const float *v = L.v + (size_t)r * L.cols;
    float s = 0.0f;
    for (uint32_t c = lane; c < L.cols; c += 64) s += v[c] * v[c];
    const float norm = __builtin_sqrtf(wave_sum(s));
    const float k = L.g[r] / norm;
    float *w = L.w + (size_t)r * L.w_stride;
    for (uint32_t c = lane; c < L.cols; c += 64) w[c] = v[c] * k;
}

struct PgArgs { ac_pg_entry e[AC_PG_MAX_ENTRIES]; uint32_t blk0[AC_PG_MAX_ENTRIES + 1]; uint32_t n; };

constexpr int PG_BLOCK = 256;
__device__ __forceinline__ float block_sum(float v, float *red)       // PG_BLOCK threads; every thread gets the total (fixed order)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ __launch_bounds__(PG_BLOCK) void param_grads_kernel(const PgArgs a)
{
    __shared__ float red[4];
    // static indexing of the by-value argument (a dynamic index would move the whole struct to scratch memory)
    ac_pg_entry E = a.e[0];
    uint32_t r = blockIdx.x;
#pragma unroll
    for (int i = 1; i < AC_PG_MAX_ENTRIES; ++i)
        if ((uint32_t)i < a.n && blockIdx.x >= a.blk0[i]) { E = a.e[i]; r = blockIdx.x - a.blk0[i]; }
    const uint32_t t = threadIdx.x;
    if (E.kind == AC_PG_WEIGHT_NORM) {
        // w = v g / n  =>  dg = (dw . v) / n,  dv = (g / n) (dw - v (dw . v) / n^2)
        const float *v = E.v + (size_t)r * E.cols, *gw = E.src + (size_t)r * E.src_stride;
        float s = 0.0f, d = 0.0f;
        for (uint32_t c = t; c < E.cols; c += PG_BLOCK) { s += v[c] * v[c]; d += gw[c] * v[c]; }
        s = block_sum(s, red); d = block_sum(d, red);
        const float norm = __builtin_sqrtf(s), g = E.g[r];
        if (t == 0) E.dst2[r] += d / norm;
        const float k = g / norm, m = d / s;
        float *gv = E.dst + (size_t)r * E.cols;
        for (uint32_t c = t; c < E.cols; c += PG_BLOCK) gv[c] += k * (gw[c] - v[c] * m);
    } else if (E.kind == AC_PG_ADD) {
        for (uint32_t i = t; i < E.rows; i += PG_BLOCK) E.dst[i] += E.src[(size_t)i * E.src_stride];
    } else {
        // AC_PG_VARIANCE: inv_s = clip(exp(10 variance), 1e-6, 1e6) (models/instant_nsr.py:35-45, 666-667): d / d variance = 10 inv_s sum_rays(d / d inv_s)
        // inside the clip range, 0 outside (torch.clip's backward passes the gradient on [min, max])
        float s = 0.0f;
        for (uint32_t i = t; i < E.rows; i += PG_BLOCK) s += E.src[i];
        s = block_sum(s, red);
        const float inv_s = E.g[0];
        if (t == 0 && inv_s > 1e-6f && inv_s < 1e6f) E.dst[0] += 10.0f * inv_s * s;
    }
}

__global__ __launch_bounds__(1024) void sds_upstream_kernel(const float *__restrict__ ws, const float *__restrict__ ws_gt, uint32_t N, float scale,
                                                            float *__restrict__ g_ws, float *__restrict__ loss)
{
    __shared__ float red[16];
    float acc = 0.0f;
    for (uint32_t i = threadIdx.x; i < N; i += 1024) {
        const float a = ws[i], b = ws_gt[i];
        const float ca = a < 0.0f ? 0.0f : (a > 1.0f ? 1.0f : a), cb = b < 0.0f ? 0.0f : (b > 1.0f ? 1.0f : b);
        const float d = ca - cb, ad = __builtin_fabsf(d);
        acc += ad < 1.0f ? 0.5f * d * d : ad - 0.5f;                        // smooth_l1, beta = 1
        const float gd = ad < 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f);
        if (g_ws) g_ws[i] = (a >= 0.0f && a <= 1.0f) ? gd * scale : 0.0f;  // clamp passes the gradient on [0, 1]
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && loss) {
        float t = 0.0f;
        for (int w = 0; w < 16; ++w) t += red[w];
        loss[0] = t * scale;
    }
}

// Adam (torch.optim.Adam without amsgrad / weight decay / maximize; the arithmetic of torch's update, torch/optim/adam.py _single_tensor_adam):
//   m = m + (1 - beta1) (g - m);  v = beta2 v + (1 - beta2) g g;  p = p - step_size * m / (sqrt(v) / sqrt(bias_correction2) + eps)
// with step_size = lr / bias_correction1, sqrt(bias_correction2), 1 - beta1 and 1 - beta2 formed on the host in double like torch does (1 - 0.999f is
// 1.3e-5 off 0.001).  HBM-bound streaming: 4 reads + 3 writes (+ 1 with
// zero_grad) of 4 bytes per element; 16-byte accesses, 4 x 16 bytes in flight per lane.
struct AdamArgs { ac_adam_entry e[AC_ADAM_MAX_TENSORS]; uint32_t blk0[AC_ADAM_MAX_TENSORS + 1]; uint32_t n; float step_size, beta1, omb1, beta2, omb2, eps, bc2_sqrt; int zero_grad; };
constexpr int ADAM_BLOCK = 256, ADAM_PER_BLOCK = ADAM_BLOCK * 16;     // elements per workgroup: four float4 per lane

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const AdamArgs &a)
{
    m = m + a.omb1 * (g - m);
    v = a.beta2 * v + a.omb2 * g * g;
    const float denom = __builtin_sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p - a.step_size * (m / denom);
}

__global__ __launch_bounds__(ADAM_BLOCK) void adam_step_kernel(const AdamArgs a)
{
    ac_adam_entry E = a.e[0];                          // static indexing of the by-value argument (see weight_norm_fwd_kernel)
    uint32_t blk = blockIdx.x;
#pragma unroll
    for (int i = 1; i < AC_ADAM_MAX_TENSORS; ++i)
        if ((uint32_t)i < a.n && blockIdx.x >= a.blk0[i]) { E = a.e[i]; blk = blockIdx.x - a.blk0[i]; }
    const size_t base = (size_t)blk * ADAM_PER_BLOCK;
    const bool vec = ((reinterpret_cast<uintptr_t>(E.param) | reinterpret_cast<uintptr_t>(E.grad) | reinterpret_cast<uintptr_t>(E.exp_avg) |
                       reinterpret_cast<uintptr_t>(E.exp_avg_sq)) & 15u) == 0 && base + ADAM_PER_BLOCK <= E.n;
    if (vec) {
        float4 p[4], g[4], m[4], v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = base / 4 + (size_t)u * ADAM_BLOCK + threadIdx.x;
            p[u] = reinterpret_cast<const float4 *>(E.param)[i]; g[u] = reinterpret_cast<const float4 *>(E.grad)[i];
            m[u] = reinterpret_cast<const float4 *>(E.exp_avg)[i]; v[u] = reinterpret_cast<const float4 *>(E.exp_avg_sq)[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            adam_one(p[u].x, g[u].x, m[u].x, v[u].x, a); adam_one(p[u].y, g[u].y, m[u].y, v[u].y, a);
            adam_one(p[u].z, g[u].z, m[u].z, v[u].z, a); adam_one(p[u].w, g[u].w, m[u].w, v[u].w, a);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = base / 4 + (size_t)u * ADAM_BLOCK + threadIdx.x;
            reinterpret_cast<float4 *>(E.param)[i] = p[u]; reinterpret_cast<float4 *>(E.exp_avg)[i] = m[u]; reinterpret_cast<float4 *>(E.exp_avg_sq)[i] = v[u];
            if (a.zero_grad) reinterpret_cast<float4 *>(E.grad)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        return;
    }
    for (size_t i = base + threadIdx.x; i < base + ADAM_PER_BLOCK && i < E.n; i += ADAM_BLOCK) {       // ragged tail / unaligned tensor
        float p = E.param[i], m = E.exp_avg[i], v = E.exp_avg_sq[i];
        adam_one(p, E.grad[i], m, v, a);
        E.param[i] = p; E.exp_avg[i] = m; E.exp_avg_sq[i] = v;
        if (a.zero_grad) E.grad[i] = 0.0f;
    }
}

// inv_s = forward_variance() of the model: ones([1, 1]) * exp(variance * 10) clipped to [1e-6, 1e6] (models/instant_nsr.py:35-45, 666-667) -- five torch
// launches of ~5 us each per parameter version, one here.  expf is the device library's exp torch's kernel calls: the same bits; NaN passes like torch.clip's.
__global__ __launch_bounds__(64) void variance_forward_kernel(const float *__restrict__ variance, float *__restrict__ inv_s)
{
    if (threadIdx.x != 0) return;
    const float e = expf(variance[0] * 10.0f);
    inv_s[0] = e != e ? e : __builtin_fminf(__builtin_fmaxf(e, 1e-6f), 1e6f);
}

}  // namespace

AC_API int ac_adam_step(const ac_adam_entry *tensors, uint32_t n, float step_size, float beta1, float one_minus_beta1, float beta2, float one_minus_beta2,
                        float eps, float bias_correction2_sqrt, int zero_grad, ac_stream_t stream)
{
    if (n == 0) return AC_OK;
    if (!tensors || n > AC_ADAM_MAX_TENSORS) { ac::set_error("adam_step: NULL tensors or more than %d", AC_ADAM_MAX_TENSORS); return AC_ERR_BAD_ARG; }
    if (!(bias_correction2_sqrt > 0.0f)) { ac::set_error("adam_step: sqrt(bias_correction2) must be positive"); return AC_ERR_BAD_ARG; }
    AdamArgs a;
    a.n = n; a.blk0[0] = 0; a.step_size = step_size; a.beta1 = beta1; a.omb1 = one_minus_beta1; a.beta2 = beta2; a.omb2 = one_minus_beta2; a.eps = eps; a.bc2_sqrt = bias_correction2_sqrt; a.zero_grad = zero_grad;
    for (uint32_t i = 0; i < n; ++i) {
        const ac_adam_entry &e = tensors[i];
        if (!e.param || !e.grad || !e.exp_avg || !e.exp_avg_sq || e.n == 0) { ac::set_error("adam_step: tensor %u: NULL buffer or empty", i); return AC_ERR_BAD_ARG; }
        const uint64_t blocks = (e.n + ADAM_PER_BLOCK - 1) / ADAM_PER_BLOCK;
        if (a.blk0[i] + blocks > 0x7fffffffull) { ac::set_error("adam_step: too many elements"); return AC_ERR_BAD_ARG; }
        a.e[i] = e; a.blk0[i + 1] = a.blk0[i] + (uint32_t)blocks;
    }
    hipLaunchKernelGGL(adam_step_kernel, dim3(a.blk0[n]), dim3(ADAM_BLOCK), 0, (hipStream_t)stream, a);
    return ac::check_launch("adam_step");
}

AC_API int ac_variance_forward(const float *variance, float *inv_s, ac_stream_t stream)
{
    if (!variance || !inv_s) { ac::set_error("variance_forward: NULL buffer"); return AC_ERR_BAD_ARG; }
    hipLaunchKernelGGL(variance_forward_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, variance, inv_s);
    return ac::check_launch("variance_forward");
}

AC_API int ac_weight_norm_forward(const ac_wn_layer *layers, uint32_t n, ac_stream_t stream)
{
    if (n == 0) return AC_OK;
    if (!layers || n > AC_WN_MAX_LAYERS) { ac::set_error("weight_norm_forward: NULL layers or more than %d", AC_WN_MAX_LAYERS); return AC_ERR_BAD_ARG; }
    WnArgs a;
    a.n = n; a.row0[0] = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (!layers[i].v || !layers[i].g || !layers[i].w || layers[i].rows == 0 || layers[i].cols == 0 || layers[i].w_stride < layers[i].cols) {
            ac::set_error("weight_norm_forward: layer %u: NULL buffer, empty shape or w_stride < cols", i); return AC_ERR_BAD_ARG;
        }
        a.l[i] = layers[i]; a.row0[i + 1] = a.row0[i] + layers[i].rows;
    }
    hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3(a.row0[n]), dim3(64), 0, (hipStream_t)stream, a);
    return ac::check_launch("weight_norm_forward");
}

AC_API int ac_param_grads(const ac_pg_entry *entries, uint32_t n, ac_stream_t stream)
{
    if (n == 0) return AC_OK;
    if (!entries || n > AC_PG_MAX_ENTRIES) { ac::set_error("param_grads: NULL entries or more than %d", AC_PG_MAX_ENTRIES); return AC_ERR_BAD_ARG; }
    PgArgs a;
    a.n = n; a.blk0[0] = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const ac_pg_entry &e = entries[i];
        const bool wn = e.kind == AC_PG_WEIGHT_NORM;
        if (!e.src || !e.dst || e.rows == 0 || (wn && (!e.v || !e.g || !e.dst2 || e.cols == 0 || e.src_stride < e.cols)) || (e.kind == AC_PG_VARIANCE && !e.g) ||
            (e.kind != AC_PG_WEIGHT_NORM && e.kind != AC_PG_ADD && e.kind != AC_PG_VARIANCE)) {
            ac::set_error("param_grads: entry %u: NULL buffer, empty shape or unknown kind", i); return AC_ERR_BAD_ARG;
        }
        a.e[i] = e; a.blk0[i + 1] = a.blk0[i] + (wn ? e.rows : 1u);
    }
    hipLaunchKernelGGL(param_grads_kernel, dim3(a.blk0[n]), dim3(PG_BLOCK), 0, (hipStream_t)stream, a);
    return ac::check_launch("param_grads");
}

AC_API int ac_sds_upstream(const float *weights_sum, const float *weights_sum_gt, uint32_t n_rays, float scale, float *g_weights_sum, float *loss,
                           ac_stream_t stream)
{
    if (n_rays == 0) return AC_OK;
    if (!weights_sum || !weights_sum_gt || (!g_weights_sum && !loss)) { ac::set_error("sds_upstream: NULL buffer"); return AC_ERR_BAD_ARG; }
    hipLaunchKernelGGL(sds_upstream_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, weights_sum, weights_sum_gt, n_rays, scale, g_weights_sum, loss);
    return ac::check_launch("sds_upstream");
}
