// avatarcraft_amd/csrc/step_glue.hip -- the small per-step pieces around the training render, so that a stylisation step (stylize.py:95-199)
// launches no torch kernel between the guidance gradient and the optimizer:
//   ac_weight_norm_forward : W = v * (g / ||v||_row) of every layer in one launch (torch.nn.utils.weight_norm, dim 0; models/instant_nsr.py:557-590)
//   ac_param_grads         : its backward for every layer + the bias gradients + d loss / d variance from the per-ray d loss / d inv_s,
//                            ACCUMULATED into the parameters' .grad buffers, one launch
//   ac_sds_upstream        : d / d weights_sum of smooth_l1(clamp(ws, 0, 1), clamp(ws_gt, 0, 1)) * scale (stylize.py:183-193) and the loss value
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

}  // namespace

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
