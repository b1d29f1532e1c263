// avatarcraft_amd/csrc/raymarching.hip -- occupancy-grid ray marcher + packed-sample compositor for gfx950.
//
// Replaces the reference's `_raymarching` extension (raymarching/src/raymarching.cu):
//   kernel_march_rays_train (:56-222)                -> march_count_kernel + scan + march_write_kernel
//   kernel_composite_rays_train_forward (:232-301)   -> composite_train_fwd_kernel
//   kernel_composite_rays_train_backward (:315-391)  -> composite_train_bwd_kernel
//   kernel_march_rays (:497-599)                     -> march_rays_kernel
//   kernel_composite_rays (:611-707)                 -> composite_rays_kernel
//   kernel_compact_rays (:730-747)                   -> flags + scan + compact_scatter_kernel
//
// Differences by design (MI355X-first, and required for reproducibility):
//   * the reference reserves output slots with atomicAdd(counter, num_steps) from inside the ray loop,
//     which makes the packed layout run-dependent.  Here pass 1 only counts, a single-workgroup
//     exclusive scan (wave64 DPP scan + LDS carry) turns counts into offsets IN RAY ORDER, pass 2
//     writes.  rays[N,3] = (id, offset, n_steps) is therefore bit-reproducible and equal to the
//     serial order (the order of the SURVEY A.4 known answers).  Same for compact_rays.
//   * the walk's grid look-ups are issued RM_BATCH at a time (rm_device.hpp: rm_march_batch, round 5) -- the reference's decisions replayed over
//     positions that are known in advance, the same samples bit for bit, a fraction of the dependent round trips to L2.
//   * arithmetic follows the reference without fma contraction (the A.4 KATs were produced that way;
//     build flag -ffp-contract=off), the `0.5 *` voxel conversion is evaluated in double like the
//     reference's double literal.
#include "ac_common.hpp"
#include "ac_devmath.hpp"
#include "rm_device.hpp"

using namespace acdev;

namespace {

// pass 1: number of occupied steps per ray -- and WHICH positions of the ray's recurrence they are (rec: RM_REC_WORDS words per ray, rm_device.hpp)
__global__ __launch_bounds__(256) void march_count_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                          const float *__restrict__ grid, float mean_density, float bound,
                                                          uint32_t N, uint32_t H, uint32_t perturb, int32_t *__restrict__ counts,
                                                          uint32_t *__restrict__ rec, int32_t *__restrict__ ovf, uint32_t *__restrict__ wmask)
{
    __shared__ float edge[RM_EDGE_MAX];                           // the voxel faces (rm_skip_target_tab): H + 1 values; a larger grid computes them per position
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    {
        RayCtx c0{}; c0.H = H; c0.bound = bound;
        for (uint32_t m = threadIdx.x; m <= H && m < (uint32_t)RM_EDGE_MAX; m += blockDim.x) edge[m] = rm_edge(c0, m);
        __syncthreads();
    }
    if (n >= N) return;
    const float *etab = H < (uint32_t)RM_EDGE_MAX ? edge : nullptr;
    RayCtx c; rm_setup(c, rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, grid, mean_density, bound, H);
    float near, far; rm_near_far(c, near, far);
    float t = ray_t0(c, near, n, perturb), skip_tt = RM_NO_SKIP;
    uint32_t room = RM_MAX_STEPS, kpos = 0;                       // the reference's walk (`while (t < far && num_steps < MAX)`), look-ups RM_BATCH at a time: rm_march_batch
    RayRecorder rr; rr.begin(rec + (size_t)n * RM_REC_WORDS);
    while (room > 0 && rm_march_batch<RM_BATCH, true>(c, t, skip_tt, far, room, kpos, [&](float, float, float, float, float, uint32_t k) { rr.add(k); }, etab)) {}
    rr.end();
    counts[n] = (int32_t)(RM_MAX_STEPS - room);
    ovf[n] = rr.ovf ? 1 : 0; wmask[n] = rr.wmask;
}

// single-workgroup exclusive scan of vals[0..n) (in place) starting at base[0]; then base[0] += total,
// base[1] += add1 (if base1_add >= 0).  1024 threads, sequential chunk per thread + LDS tree.
__global__ __launch_bounds__(1024) void exclusive_scan_kernel(int32_t *__restrict__ vals, uint32_t n, int32_t *__restrict__ base,
                                                              int32_t base1_add)
{
    __shared__ int32_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t b0 = t * per, b1 = (b0 + per < n) ? b0 + per : n;
    int32_t s = 0;
    for (uint32_t i = b0; i < b1; ++i) s += vals[i];
    part[t] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {          // Hillis-Steele inclusive scan of the partials
        int32_t v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const int32_t start = base[0];
    int32_t run = start + ((t == 0) ? 0 : part[t - 1]);
    for (uint32_t i = b0; i < b1; ++i) { const int32_t v = vals[i]; vals[i] = run; run += v; }
    __syncthreads();
    if (t == 0) { base[0] = start + part[1023]; if (base1_add >= 0) base[1] += base1_add; }
}

// pass 2: write samples at the scanned offsets
__global__ __launch_bounds__(256) void march_write_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                          const float *__restrict__ grid, float mean_density, float bound,
                                                          uint32_t N, uint32_t H, uint32_t M, uint32_t perturb,
                                                          const int32_t *__restrict__ counts, const int32_t *__restrict__ offsets,
                                                          const int32_t *__restrict__ ray_base, float *__restrict__ xyzs, float *__restrict__ dirs,
                                                          float *__restrict__ deltas, int32_t *__restrict__ rays, const uint32_t *__restrict__ rec,
                                                          const int32_t *__restrict__ ovf, const uint32_t *__restrict__ wmask)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint32_t num_steps = (uint32_t)counts[n], point_index = (uint32_t)offsets[n], ray_index = (uint32_t)ray_base[0] + n;
    rays[ray_index * 3] = (int32_t)n; rays[ray_index * 3 + 1] = (int32_t)point_index; rays[ray_index * 3 + 2] = (int32_t)num_steps;
    if (num_steps == 0) return;
    if (point_index + num_steps >= M) return;
    RayCtx c; rm_setup(c, rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, grid, mean_density, bound, H);
    float near, far; rm_near_far(c, near, far);
    float *px = xyzs + (size_t)point_index * 3, *pd = dirs + (size_t)point_index * 3, *pt = deltas + point_index;
    const float t0 = ray_t0(c, near, n, perturb);
    if (!ovf[n]) {                                                // the counting pass recorded the samples' positions: replay the recurrence, no second walk
        rm_replay(c, t0, rec + (size_t)n * RM_REC_WORDS, wmask[n], num_steps, [&](float x, float y, float z, float dt) {
            px[0] = x; px[1] = y; px[2] = z; pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
            pt[0] = dt; px += 3; pd += 3; pt++;
        });
        return;
    }
    float t = t0, skip_tt = RM_NO_SKIP;
    uint32_t room = num_steps, kpos = 0;
    auto emit = [&](float x, float y, float z, float dt, float, uint32_t) {
        px[0] = x; px[1] = y; px[2] = z; pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
        pt[0] = dt; px += 3; pd += 3; pt++;
    };
    while (room > 0 && rm_march_batch<RM_BATCH>(c, t, skip_tt, far, room, kpos, emit)) {}
}

__global__ __launch_bounds__(256) void composite_train_fwd_kernel(const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                                  const int32_t *__restrict__ rays, uint32_t M, uint32_t N,
                                                                  float *__restrict__ weights_sum, float *__restrict__ image)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps >= M) {
        weights_sum[index] = 0; image[index * 3] = 0; image[index * 3 + 1] = 0; image[index * 3 + 2] = 0;
        return;
    }
    const float *s = sigmas + offset, *c = rgbs + (size_t)offset * 3;
    float T = 1.0f, r = 0, g = 0, b = 0;
    for (uint32_t step = 0; step < num_steps; ++step) {
        if (T < 1e-4f) break;
        const float alpha = s[step], w = alpha * T;
        r += w * c[3 * step]; g += w * c[3 * step + 1]; b += w * c[3 * step + 2];
        T *= 1.0f - alpha;
    }
    weights_sum[index] = 1.0f - T;
    image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
}

__global__ __launch_bounds__(256) void composite_train_bwd_kernel(const float *__restrict__ grad_weights_sum, const float *__restrict__ grad,
                                                                  const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                                  const float *__restrict__ deltas, const int32_t *__restrict__ rays,
                                                                  const float *__restrict__ weights_sum, const float *__restrict__ image,
                                                                  uint32_t M, uint32_t N, float *__restrict__ grad_sigmas,
                                                                  float *__restrict__ grad_rgbs)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps >= M) return;
    const float gws = grad_weights_sum[index];
    const float g0 = grad[(size_t)index * 3], g1 = grad[(size_t)index * 3 + 1], g2 = grad[(size_t)index * 3 + 2];
    const float rf = image[index * 3], gf = image[index * 3 + 1], bf = image[index * 3 + 2], Tf = 1 - weights_sum[index];
    const float *s = sigmas + offset, *c = rgbs + (size_t)offset * 3, *dl = deltas + offset;
    float *gs = grad_sigmas + offset, *gc = grad_rgbs + (size_t)offset * 3;
    float T = 1.0f, r = 0, g = 0, b = 0;
    for (uint32_t step = 0; step < num_steps; ++step) {
        const float alpha = s[step], w = alpha * T;
        r += w * c[3 * step]; g += w * c[3 * step + 1]; b += w * c[3 * step + 2];
        T *= 1.0f - alpha;
        gc[3 * step] = g0 * w; gc[3 * step + 1] = g1 * w; gc[3 * step + 2] = g2 * w;
        const float a0 = g0 * (T * c[3 * step] - (rf - r));
        const float a1 = g1 * (T * c[3 * step + 1] - (gf - g));
        const float a2 = g2 * (T * c[3 * step + 2] - (bf - b));
        gs[step] = dl[step] * (((a0 + a1) + a2) + gws * Tf);
    }
}

__global__ __launch_bounds__(256) void march_rays_kernel(uint32_t n_alive, uint32_t n_step, const int32_t *__restrict__ rays_alive,
                                                         const float *__restrict__ rays_t, const float *__restrict__ rays_o,
                                                         const float *__restrict__ rays_d, float bound, uint32_t H,
                                                         const float *__restrict__ grid, float mean_density,
                                                         const float *__restrict__ fars, float *__restrict__ xyzs,
                                                         float *__restrict__ dirs, float *__restrict__ deltas, uint32_t perturb)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    float t = rays_t[n];
    RayCtx c; rm_setup(c, rays_o + 3 * (size_t)index, rays_d + 3 * (size_t)index, grid, mean_density, bound, H);
    const float far = fars[index];
    float *px = xyzs + (size_t)n * n_step * 3, *pd = dirs + (size_t)n * n_step * 3, *pt = deltas + (size_t)n * n_step * 2;
    if (perturb) t += c.dt_min * pcg_first_float((uint64_t)n, (uint64_t)perturb);
    float last_t = t, skip_tt = RM_NO_SKIP;
    uint32_t room = n_step, kpos = 0;
    auto emit = [&](float x, float y, float z, float dt, float t_after, uint32_t) {
        px[0] = x; px[1] = y; px[2] = z; pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
        pt[0] = dt; pt[1] = t_after - last_t; last_t = t_after;
        px += 3; pd += 3; pt += 2;
    };
    while (room > 0 && rm_march_batch<RM_BATCH>(c, t, skip_tt, far, room, kpos, emit)) {}
}

__global__ __launch_bounds__(256) void composite_rays_kernel(uint32_t n_alive, uint32_t n_step, const int32_t *__restrict__ rays_alive,
                                                             float *__restrict__ rays_t, const float *__restrict__ sigmas,
                                                             const float *__restrict__ rgbs, const float *__restrict__ normals,
                                                             const float *__restrict__ deltas, float *__restrict__ weights_sum,
                                                             float *__restrict__ depth, float *__restrict__ image,
                                                             float *__restrict__ normal_map)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    float t = rays_t[n];
    const float *s = sigmas + (size_t)n * n_step, *c = rgbs + (size_t)n * n_step * 3;
    const float *dl = deltas + (size_t)n * n_step * 2, *nr = normals + (size_t)n * n_step * 3;
    float ws = weights_sum[index], d = depth[index];
    float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
    float nx = normal_map[index * 3], ny = normal_map[index * 3 + 1], nz = normal_map[index * 3 + 2];
    uint32_t step = 0;
    while (step < n_step) {
        if (dl[0] == 0) break;
        const float alpha = s[0], T = 1 - ws, w = alpha * T;
        ws += w;
        t += dl[1];
        d += w * t;
        r += w * c[0]; g += w * c[1]; b += w * c[2];
        nx += w * nr[0]; ny += w * nr[1]; nz += w * nr[2];
        if ((double)T < 1e-2) break;
        s++; c += 3; dl += 2; nr += 3; step++;
    }
    rays_t[n] = (step < n_step) ? -1.0f : t;
    weights_sum[index] = ws; depth[index] = d;
    image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    normal_map[index * 3] = nx; normal_map[index * 3 + 1] = ny; normal_map[index * 3 + 2] = nz;
}

__global__ __launch_bounds__(256) void alive_flags_kernel(uint32_t n_alive, const float *__restrict__ rays_t_old, int32_t *__restrict__ flags)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < n_alive) flags[n] = rays_t_old[n] >= 0 ? 1 : 0;
}
__global__ __launch_bounds__(256) void compact_scatter_kernel(uint32_t n_alive, const int32_t *__restrict__ rays_alive_old,
                                                              const float *__restrict__ rays_t_old, const int32_t *__restrict__ pos,
                                                              int32_t *__restrict__ rays_alive, float *__restrict__ rays_t)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const float t = rays_t_old[n];
    if (t >= 0) { const int32_t p = pos[n]; rays_alive[p] = rays_alive_old[n]; rays_t[p] = t; }
}

}  // namespace

AC_API int ac_march_rays_train(const float *rays_o, const float *rays_d, const float *grid, float mean_density, int iter_density,
                               float bound, uint32_t N, uint32_t H, uint32_t M, float *xyzs, float *dirs, float *deltas,
                               int32_t *rays, int32_t *counter, uint32_t perturb, int32_t *scratch, ac_stream_t stream)
{
    (void)iter_density;
    if (N == 0) return AC_OK;
    if (!rays_o || !rays_d || !grid || !xyzs || !dirs || !deltas || !rays || !counter || !scratch || H < 2) {
        ac::set_error("march_rays_train: NULL buffer or H < 2"); return AC_ERR_BAD_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    int32_t *counts = scratch;                            // [N] occupied steps per ray
    int32_t *offsets = scratch + N;                       // [N] exclusive scan of counts (+ counter[0] on entry)
    int32_t *ray_base = scratch + 2 * (size_t)N;          // [1] counter[1] on entry (first free ray slot) (+ 1 pad)
    int32_t *ovf = scratch + 2 * (size_t)N + 2;           // [N] 1: a sample beyond the record (the writer walks that ray again)
    uint32_t *wmask = reinterpret_cast<uint32_t *>(scratch + 3 * (size_t)N + 2);    // [N] which words of a ray's record are in use
    uint32_t *rec = reinterpret_cast<uint32_t *>(scratch + 4 * (size_t)N + 2);      // [N][RM_REC_WORDS] the samples' positions (rm_device.hpp)
    hipLaunchKernelGGL(march_count_kernel, dim3((N + 255) / 256), dim3(256), 0, st, rays_o, rays_d, grid, mean_density, bound, N, H,
                       perturb, counts, rec, ovf, wmask);
    hipMemcpyAsync(offsets, counts, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToDevice, st);
    hipMemcpyAsync(ray_base, counter + 1, sizeof(int32_t), hipMemcpyDeviceToDevice, st);
    hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, st, offsets, N, counter, (int32_t)N);
    hipLaunchKernelGGL(march_write_kernel, dim3((N + 255) / 256), dim3(256), 0, st, rays_o, rays_d, grid, mean_density, bound, N, H, M,
                       perturb, counts, offsets, ray_base, xyzs, dirs, deltas, rays, rec, ovf, wmask);
    return ac::check_launch("march_rays_train");
}

AC_API size_t ac_march_rays_train_scratch(uint32_t N) { return (4 + (size_t)RM_REC_WORDS) * (size_t)N + 2; }

AC_API int ac_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays, float bound,
                                           uint32_t M, uint32_t N, float *weights_sum, float *image, ac_stream_t stream)
{
    (void)bound; (void)deltas;
    if (N == 0) return AC_OK;
    if (!sigmas || !rgbs || !rays || !weights_sum || !image) { ac::set_error("composite_rays_train_forward: NULL buffer"); return AC_ERR_BAD_ARG; }
    hipLaunchKernelGGL(composite_train_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, sigmas, rgbs, rays, M, N,
                       weights_sum, image);
    return ac::check_launch("composite_rays_train_forward");
}

AC_API int ac_composite_rays_train_backward(const float *grad_weights_sum, const float *grad, const float *sigmas, const float *rgbs,
                                            const float *deltas, const int32_t *rays, const float *weights_sum, const float *image,
                                            float bound, uint32_t M, uint32_t N, float *grad_sigmas, float *grad_rgbs, ac_stream_t stream)
{
    (void)bound;
    if (N == 0) return AC_OK;
    if (!grad_weights_sum || !grad || !sigmas || !rgbs || !deltas || !rays || !weights_sum || !image || !grad_sigmas || !grad_rgbs) {
        ac::set_error("composite_rays_train_backward: NULL buffer"); return AC_ERR_BAD_ARG;
    }
    hipLaunchKernelGGL(composite_train_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, grad_weights_sum, grad, sigmas,
                       rgbs, deltas, rays, weights_sum, image, M, N, grad_sigmas, grad_rgbs);
    return ac::check_launch("composite_rays_train_backward");
}

AC_API int ac_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t, const float *rays_o,
                         const float *rays_d, float bound, uint32_t H, const float *grid, float mean_density, const float *near,
                         const float *far, float *xyzs, float *dirs, float *deltas, uint32_t perturb, ac_stream_t stream)
{
    (void)near;
    if (n_alive == 0) return AC_OK;
    if (!rays_alive || !rays_t || !rays_o || !rays_d || !grid || !far || !xyzs || !dirs || !deltas || H < 2) {
        ac::set_error("march_rays: NULL buffer or H < 2"); return AC_ERR_BAD_ARG;
    }
    hipLaunchKernelGGL(march_rays_kernel, dim3((n_alive + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_alive, n_step, rays_alive, rays_t,
                       rays_o, rays_d, bound, H, grid, mean_density, far, xyzs, dirs, deltas, perturb);
    return ac::check_launch("march_rays");
}

AC_API int ac_composite_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, float *rays_t, const float *sigmas,
                             const float *rgbs, const float *normals, const float *deltas, float *weights_sum, float *depth, float *image,
                             float *normal_map, ac_stream_t stream)
{
    if (n_alive == 0) return AC_OK;
    if (!rays_alive || !rays_t || !sigmas || !rgbs || !normals || !deltas || !weights_sum || !depth || !image || !normal_map) {
        ac::set_error("composite_rays: NULL buffer"); return AC_ERR_BAD_ARG;
    }
    hipLaunchKernelGGL(composite_rays_kernel, dim3((n_alive + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_alive, n_step, rays_alive,
                       rays_t, sigmas, rgbs, normals, deltas, weights_sum, depth, image, normal_map);
    return ac::check_launch("composite_rays");
}

AC_API int ac_compact_rays(uint32_t n_alive, int32_t *rays_alive, const int32_t *rays_alive_old, float *rays_t, const float *rays_t_old,
                           int32_t *alive_counter, int32_t *scratch, ac_stream_t stream)
{
    if (n_alive == 0) return AC_OK;
    if (!rays_alive || !rays_alive_old || !rays_t || !rays_t_old || !alive_counter || !scratch) {
        ac::set_error("compact_rays: NULL buffer"); return AC_ERR_BAD_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(alive_flags_kernel, dim3((n_alive + 255) / 256), dim3(256), 0, st, n_alive, rays_t_old, scratch);
    hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, st, scratch, n_alive, alive_counter, -1);
    hipLaunchKernelGGL(compact_scatter_kernel, dim3((n_alive + 255) / 256), dim3(256), 0, st, n_alive, rays_alive_old, rays_t_old, scratch,
                       rays_alive, rays_t);
    return ac::check_launch("compact_rays");
}
