/*
 * oracle/ac_oracle_ops.c -- TEST INFRASTRUCTURE (see ac_oracle.c header).
 * CPU restatement of the reference's SH encoder and `raymarching` operators.
 *
 *   SH encoder      : encoder/shencoder/src/shencoder.cu:28-384
 *   raymarching ops : raymarching/src/raymarching.cu:56-222,232-301,315-391,497-599,611-707,730-747
 *
 * The raymarching restatement is serial in ray order, which is also the canonical slot
 * order for "packed-ray (id, offset, n_steps) bit-exact" (the reference
 * reserves slots with atomicAdd, i.e. in a run-dependent order; the serial order is the one
 * the Appendix A.4 known-answer values were taken in).
 *
 * fp convention for the marcher: the A.4 KATs were produced without fma contraction, so the
 * marcher uses separately rounded mul/add (ORC_RM_FMA=0).  The HIP kernel follows the same.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "ac_math.h"
#include "ac_sh_table.h"

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* SH                                                                   */
/* ------------------------------------------------------------------ */
static float sh_eval(const unsigned short *off, const float *coef, const unsigned char (*ex)[3],
                     int idx, const float px[8], const float py[8], const float pz[8])
{
    float acc = 0.0f;
    for (int m = off[idx]; m < off[idx + 1]; m++) {
        float mono = (px[ex[m][0]] * py[ex[m][1]]) * pz[ex[m][2]];
        acc = fmaf(coef[m], mono, acc);
    }
    return acc;
}

#include "ac_oracle.h"
void orc_sh16(const float d[3], float sh[16])
{
    float p[3][8];
    for (int a = 0; a < 3; a++) { p[a][0] = 1.0f; for (int k = 1; k < 8; k++) p[a][k] = p[a][k - 1] * d[a]; }
    for (int i = 0; i < 16; i++) sh[i] = sh_eval(AC_SH_OFF0, AC_SH_COEF0, AC_SH_EXP0, i, p[0], p[1], p[2]);
}

/* _backend.sh_encode_forward(inputs, outputs, B, D, C=degree, calc_grad_inputs, dy_dx):
 * outputs [B, C*C]; dy_dx [B, 3, C*C] (shencoder.cu:128-130). */
ORC_API int orc_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D, uint32_t C,
                                  int calc_grad_inputs, float *dy_dx)
{
    if (D != 3 || C < 1 || C > 8) return 1;
    const uint32_t C2 = C * C;
    #pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)B; b++) {
        float p[3][8];
        for (int a = 0; a < 3; a++) {
            p[a][0] = 1.0f;
            for (int k = 1; k < 8; k++) p[a][k] = p[a][k - 1] * inputs[b * 3 + a];
        }
        for (uint32_t i = 0; i < C2; i++)
            outputs[b * C2 + i] = sh_eval(AC_SH_OFF0, AC_SH_COEF0, AC_SH_EXP0, i, p[0], p[1], p[2]);
        if (calc_grad_inputs) {
            float *dx = dy_dx + (size_t)b * 3 * C2, *dy = dx + C2, *dz = dy + C2;
            for (uint32_t i = 0; i < C2; i++) {
                dx[i] = sh_eval(AC_SH_OFF1, AC_SH_COEF1, AC_SH_EXP1, i, p[0], p[1], p[2]);
                dy[i] = sh_eval(AC_SH_OFF2, AC_SH_COEF2, AC_SH_EXP2, i, p[0], p[1], p[2]);
                dz[i] = sh_eval(AC_SH_OFF3, AC_SH_COEF3, AC_SH_EXP3, i, p[0], p[1], p[2]);
            }
        }
    }
    return 0;
}

/* _backend.sh_encode_backward (shencoder.cu:360-384): grad_inputs[b,d] += sum_ch grad*dy_dx */
ORC_API int orc_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D, uint32_t C,
                                   const float *dy_dx, float *grad_inputs)
{
    (void)inputs;
    if (D != 3 || C < 1 || C > 8) return 1;
    const uint32_t C2 = C * C;
    for (uint32_t t = 0; t < B * D; t++) {
        uint32_t b = t / D, d = t - b * D;
        float acc = grad_inputs[t];
        for (uint32_t ch = 0; ch < C2; ch++)
            acc = fmaf(grad[b * C2 + ch], dy_dx[(size_t)b * D * C2 + d * C2 + ch], acc);
        grad_inputs[t] = acc;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* raymarching                                                          */
/* ------------------------------------------------------------------ */
#ifndef ORC_RM_FMA
#define ORC_RM_FMA 0
#endif
static inline float rm_madd(float a, float b, float c)   /* c + a*b as the marcher rounds it */
{
#if ORC_RM_FMA
    return fmaf(a, b, c);
#else
    return c + a * b;
#endif
}
static inline float rm_clamp(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }
static inline float rm_sign(float x) { return copysignf(1.0f, x); }

#define RM_MAX_STEPS 1024
#define RM_SQRT3 1.73205080757f
#define RM_MIN_NEAR 0.05f

typedef struct { float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz; float bound, rbound; uint32_t H;
                 float dt_min, dt_max, dt_gamma, thresh; const float *grid; } rm_ctx;

static void rm_setup(rm_ctx *c, const float *o, const float *d, const float *grid, float mean_density,
                     float bound, uint32_t H)
{
    c->ox = o[0]; c->oy = o[1]; c->oz = o[2]; c->dx = d[0]; c->dy = d[1]; c->dz = d[2];
    c->rdx = 1 / c->dx; c->rdy = 1 / c->dy; c->rdz = 1 / c->dz;
    c->bound = bound; c->rbound = 1 / bound; c->H = H; c->grid = grid;
    c->dt_min = (2 * RM_SQRT3 / RM_MAX_STEPS) * bound;      /* raymarching.cu:24,100 */
    c->dt_max = 2 * bound / (float)(H - 1);                 /* :101 */
    c->dt_gamma = bound > 1 ? (1.f / 256.f) : 0.0f;         /* :102 */
    c->thresh = fminf(10.0f, mean_density);                 /* :75 */
}
/* voxel lookup :118-128; the 0.5 literal is a double in the reference */
static inline float rm_density(const rm_ctx *c, float t, float *x, float *y, float *z, int *nx, int *ny, int *nz)
{
    *x = rm_clamp(rm_madd(t, c->dx, c->ox), -c->bound, c->bound);
    *y = rm_clamp(rm_madd(t, c->dy, c->oy), -c->bound, c->bound);
    *z = rm_clamp(rm_madd(t, c->dz, c->oz), -c->bound, c->bound);
    const float hm1 = (float)(c->H - 1);
    *nx = (int)rm_clamp((float)(0.5 * (double)(*x * c->rbound + 1) * (double)c->H), 0.0f, hm1);
    *ny = (int)rm_clamp((float)(0.5 * (double)(*y * c->rbound + 1) * (double)c->H), 0.0f, hm1);
    *nz = (int)rm_clamp((float)(0.5 * (double)(*z * c->rbound + 1) * (double)c->H), 0.0f, hm1);
    uint32_t index = (uint32_t)*nx * c->H * c->H + (uint32_t)*ny * c->H + (uint32_t)*nz;
    return c->grid[index];
}
/* skip to the next voxel :139-148 */
static inline float rm_skip(const rm_ctx *c, float t, float x, float y, float z, int nx, int ny, int nz)
{
    const float hm1 = (float)(c->H - 1);
    float tx = (((nx + 0.5f + 0.5f * rm_sign(c->dx)) / hm1 * 2 - 1) * c->bound - x) * c->rdx;
    float ty = (((ny + 0.5f + 0.5f * rm_sign(c->dy)) / hm1 * 2 - 1) * c->bound - y) * c->rdy;
    float tz = (((nz + 0.5f + 0.5f * rm_sign(c->dz)) / hm1 * 2 - 1) * c->bound - z) * c->rdz;
    float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do { t += rm_clamp(t * c->dt_gamma, c->dt_min, c->dt_max); } while (t < tt);
    return t;
}
static void rm_near_far(const rm_ctx *c, float *near, float *far)
{
    float nx = (-c->bound - c->ox) * c->rdx, fx = (c->bound - c->ox) * c->rdx;
    if (nx > fx) { float s = nx; nx = fx; fx = s; }
    float ny = (-c->bound - c->oy) * c->rdy, fy = (c->bound - c->oy) * c->rdy;
    if (ny > fy) { float s = ny; ny = fy; fy = s; }
    float nz = (-c->bound - c->oz) * c->rdz, fz = (c->bound - c->oz) * c->rdz;
    if (nz > fz) { float s = nz; nz = fz; fz = s; }
    *near = fmaxf(fmaxf(nx, fmaxf(ny, nz)), RM_MIN_NEAR);
    *far = fminf(fx, fminf(fy, fz));
}

static float pcg_first_float(uint64_t initstate, uint64_t initseq)
{
    uint64_t st[2];
    extern void orc_pcg32_seed(uint64_t *, uint64_t, uint64_t);
    extern float orc_pcg32_next_float(uint64_t *);
    orc_pcg32_seed(st, initstate, initseq);
    return orc_pcg32_next_float(st);
}

/* _backend.march_rays_train (raymarching.cu:56-222): deterministic (ray-order) slot reservation. */
ORC_API int orc_march_rays_train(const float *rays_o, const float *rays_d, const float *grid,
                                 float mean_density, int iter_density, float bound, uint32_t N, uint32_t H,
                                 uint32_t M, float *xyzs, float *dirs, float *deltas, int32_t *rays,
                                 int32_t *counter, uint32_t perturb)
{
    (void)iter_density;
    for (uint32_t n = 0; n < N; n++) {
        rm_ctx c; rm_setup(&c, rays_o + 3 * n, rays_d + 3 * n, grid, mean_density, bound, H);
        float near, far; rm_near_far(&c, &near, &far);
        float t0 = near;
        if (perturb) t0 += c.dt_min * pcg_first_float((uint64_t)n, 1);
        float t = t0; uint32_t num_steps = 0;
        float x, y, z; int nx, ny, nz;
        while (t < far && num_steps < RM_MAX_STEPS) {
            float den = rm_density(&c, t, &x, &y, &z, &nx, &ny, &nz);
            if (den > c.thresh) { num_steps++; t += rm_clamp(t * c.dt_gamma, c.dt_min, c.dt_max); }
            else t = rm_skip(&c, t, x, y, z, nx, ny, nz);
        }
        uint32_t point_index = (uint32_t)counter[0]; counter[0] += (int32_t)num_steps;
        uint32_t ray_index = (uint32_t)counter[1]; counter[1] += 1;
        rays[ray_index * 3] = (int32_t)n; rays[ray_index * 3 + 1] = (int32_t)point_index;
        rays[ray_index * 3 + 2] = (int32_t)num_steps;
        if (num_steps == 0) continue;
        if (point_index + num_steps >= M) continue;
        float *px = xyzs + (size_t)point_index * 3, *pd = dirs + (size_t)point_index * 3, *pt = deltas + point_index;
        t = t0; uint32_t step = 0;
        while (t < far && step < num_steps) {
            float den = rm_density(&c, t, &x, &y, &z, &nx, &ny, &nz);
            if (den > c.thresh) {
                px[0] = x; px[1] = y; px[2] = z; pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
                float dt = rm_clamp(t * c.dt_gamma, c.dt_min, c.dt_max);
                t += dt; pt[0] = dt; px += 3; pd += 3; pt++; step++;
            } else t = rm_skip(&c, t, x, y, z, nx, ny, nz);
        }
    }
    return 0;
}

/* _backend.composite_rays_train_forward (raymarching.cu:232-301) */
ORC_API int orc_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                             const int32_t *rays, float bound, uint32_t M, uint32_t N,
                                             float *weights_sum, float *image)
{
    (void)bound; (void)deltas;
    for (uint32_t n = 0; n < N; n++) {
        uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps >= M) {
            weights_sum[index] = 0; image[index * 3] = image[index * 3 + 1] = image[index * 3 + 2] = 0;
            continue;
        }
        const float *s = sigmas + offset, *c = rgbs + (size_t)offset * 3;
        float T = 1.0f, r = 0, g = 0, b = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            if (T < 1e-4f) break;
            float alpha = s[step], w = alpha * T;
            r = rm_madd(w, c[3 * step], r); g = rm_madd(w, c[3 * step + 1], g); b = rm_madd(w, c[3 * step + 2], b);
            T *= 1.0f - alpha;
        }
        weights_sum[index] = 1.0f - T;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
    return 0;
}

/* _backend.composite_rays_train_backward (raymarching.cu:315-391), incl. the sigma-form
 * derivative scaled by deltas and the missing early stop. */
ORC_API int orc_composite_rays_train_backward(const float *grad_weights_sum, const float *grad,
                                              const float *sigmas, const float *rgbs, const float *deltas,
                                              const int32_t *rays, const float *weights_sum, const float *image,
                                              float bound, uint32_t M, uint32_t N, float *grad_sigmas, float *grad_rgbs)
{
    (void)bound;
    for (uint32_t n = 0; n < N; n++) {
        uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps >= M) continue;
        const float gws = grad_weights_sum[index];
        const float *gr = grad + (size_t)index * 3;
        const float rf = image[index * 3], gf = image[index * 3 + 1], bf = image[index * 3 + 2];
        const float Tf = 1 - weights_sum[index];
        const float *s = sigmas + offset, *c = rgbs + (size_t)offset * 3, *dl = deltas + offset;
        float *gs = grad_sigmas + offset, *gc = grad_rgbs + (size_t)offset * 3;
        float T = 1.0f, r = 0, g = 0, b = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            float alpha = s[step], w = alpha * T;
            r = rm_madd(w, c[3 * step], r); g = rm_madd(w, c[3 * step + 1], g); b = rm_madd(w, c[3 * step + 2], b);
            T *= 1.0f - alpha;
            gc[3 * step] = gr[0] * w; gc[3 * step + 1] = gr[1] * w; gc[3 * step + 2] = gr[2] * w;
            float a0 = gr[0] * (rm_madd(T, c[3 * step], -(rf - r)));
            float a1 = gr[1] * (rm_madd(T, c[3 * step + 1], -(gf - g)));
            float a2 = gr[2] * (rm_madd(T, c[3 * step + 2], -(bf - b)));
            gs[step] = dl[step] * (((a0 + a1) + a2) + gws * Tf);
        }
    }
    return 0;
}

/* _backend.march_rays (raymarching.cu:497-599) */
ORC_API int orc_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                           const float *rays_o, const float *rays_d, float bound, uint32_t H, const float *grid,
                           float mean_density, const float *nears, const float *fars, float *xyzs, float *dirs,
                           float *deltas, uint32_t perturb)
{
    for (uint32_t n = 0; n < n_alive; n++) {
        const int index = rays_alive[n];
        float t = rays_t[n];
        rm_ctx c; rm_setup(&c, rays_o + 3 * (size_t)index, rays_d + 3 * (size_t)index, grid, mean_density, bound, H);
        const float far = fars[index];
        float *px = xyzs + (size_t)n * n_step * 3, *pd = dirs + (size_t)n * n_step * 3, *pt = deltas + (size_t)n * n_step * 2;
        if (perturb) t += c.dt_min * pcg_first_float((uint64_t)n, (uint64_t)perturb);
        float last_t = t; uint32_t step = 0;
        float x, y, z; int nx, ny, nz;
        (void)nears;
        while (t < far && step < n_step) {
            float den = rm_density(&c, t, &x, &y, &z, &nx, &ny, &nz);
            if (den > c.thresh) {
                px[0] = x; px[1] = y; px[2] = z; pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
                float dt = rm_clamp(t * c.dt_gamma, c.dt_min, c.dt_max);
                t += dt; pt[0] = dt; pt[1] = t - last_t; last_t = t;
                px += 3; pd += 3; pt += 2; step++;
            } else t = rm_skip(&c, t, x, y, z, nx, ny, nz);
        }
    }
    return 0;
}

/* _backend.composite_rays (raymarching.cu:611-707) */
ORC_API int orc_composite_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, float *rays_t,
                               const float *sigmas, const float *rgbs, const float *normals, const float *deltas,
                               float *weights_sum, float *depth, float *image, float *normal_map)
{
    for (uint32_t n = 0; n < n_alive; n++) {
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
            d = rm_madd(w, t, d);
            r = rm_madd(w, c[0], r); g = rm_madd(w, c[1], g); b = rm_madd(w, c[2], b);
            nx = rm_madd(w, nr[0], nx); ny = rm_madd(w, nr[1], ny); nz = rm_madd(w, nr[2], nz);
            if ((double)T < 1e-2) break;   /* double literal in the reference (raymarching.cu:680) */
            s++; c += 3; dl += 2; nr += 3; step++;
        }
        rays_t[n] = (step < n_step) ? -1.0f : t;
        weights_sum[index] = ws; depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
        normal_map[index * 3] = nx; normal_map[index * 3 + 1] = ny; normal_map[index * 3 + 2] = nz;
    }
    return 0;
}

/* _backend.compact_rays (raymarching.cu:730-747), order-preserving (serial) compaction */
ORC_API int orc_compact_rays(uint32_t n_alive, int32_t *rays_alive, const int32_t *rays_alive_old,
                             float *rays_t, const float *rays_t_old, int32_t *alive_counter)
{
    for (uint32_t n = 0; n < n_alive; n++) {
        if (rays_t_old[n] >= 0) {
            int idx = alive_counter[0]++;
            rays_alive[idx] = rays_alive_old[n];
            rays_t[idx] = rays_t_old[n];
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* SMPL-guided warp (utils/ray_utils.py:62-90, 277-308)                 */
/* ------------------------------------------------------------------ */
/* geometry_guided_near_far_torch (ray_utils.py:277-294): per ray, over V vertex-spheres of radius r */
ORC_API int orc_mesh_near_far(const float *rays_o, const float *rays_d, const float *verts, uint32_t N, uint32_t V,
                              float geo_threshold, float *near, float *far)
{
    const float r2 = (float)((double)geo_threshold * (double)geo_threshold);   /* geo_threshold**2 is a python float */
    #pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const float *o = rays_o + 3 * n, *d = rays_d + 3 * n;
        float nr = INFINITY, fr = -INFINITY;
        for (uint32_t v = 0; v < V; v++) {
            const float x = verts[3 * v] - o[0], y = verts[3 * v + 1] - o[1], z = verts[3 * v + 2] - o[2];
            const float z0 = (x * d[0] + y * d[1]) + z * d[2];
            const float nrm = sqrtf((x * x + y * y) + z * z);
            const float dz = sqrtf(r2 - (nrm * nrm - z0 * z0));
            const float a = z0 - dz, b = z0 + dz;
            if (a == a && a < nr) nr = a;          /* NaN -> +inf / -inf */
            if (b == b && b > fr) fr = b;
        }
        near[n] = nr; far[n] = fr;
    }
    return 0;
}

/* closest point on triangle (a,b,c) to p, all double: Ericson, Real-Time Collision Detection 5.1.5 */
static void closest_pt_tri(const double p[3], const double a[3], const double b[3], const double c[3], double out[3])
{
    double ab[3], ac[3], ap[3], bp[3], cp[3];
    for (int i = 0; i < 3; i++) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; }
    #define DOT(u, v) ((u)[0] * (v)[0] + (u)[1] * (v)[1] + (u)[2] * (v)[2])
    const double d1 = DOT(ab, ap), d2 = DOT(ac, ap);
    if (d1 <= 0.0 && d2 <= 0.0) { for (int i = 0; i < 3; i++) out[i] = a[i]; return; }
    for (int i = 0; i < 3; i++) bp[i] = p[i] - b[i];
    const double d3 = DOT(ab, bp), d4 = DOT(ac, bp);
    if (d3 >= 0.0 && d4 <= d3) { for (int i = 0; i < 3; i++) out[i] = b[i]; return; }
    const double vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) { const double v = d1 / (d1 - d3); for (int i = 0; i < 3; i++) out[i] = a[i] + v * ab[i]; return; }
    for (int i = 0; i < 3; i++) cp[i] = p[i] - c[i];
    const double d5 = DOT(ab, cp), d6 = DOT(ac, cp);
    if (d6 >= 0.0 && d5 <= d6) { for (int i = 0; i < 3; i++) out[i] = c[i]; return; }
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) { const double w = d2 / (d2 - d6); for (int i = 0; i < 3; i++) out[i] = a[i] + w * ac[i]; return; }
    const double va = d3 * d6 - d5 * d4;
    if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
        const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        for (int i = 0; i < 3; i++) out[i] = b[i] + w * (c[i] - b[i]);
        return;
    }
    const double denom = 1.0 / (va + vb + vc), v = vb * denom, w = vc * denom;
    for (int i = 0; i < 3; i++) out[i] = a[i] + ab[i] * v + ac[i] * w;
}

/* 4x4 inverse, Gauss-Jordan with partial pivoting (np.linalg.inv restated) */
static int inv4(const double m[16], double out[16])
{
    double a[4][8];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { a[i][j] = m[4 * i + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int col = 0; col < 4; col++) {
        int piv = col; double best = fabs(a[col][col]);
        for (int r = col + 1; r < 4; r++) if (fabs(a[r][col]) > best) { best = fabs(a[r][col]); piv = r; }
        if (best == 0.0) return 1;
        if (piv != col) for (int j = 0; j < 8; j++) { double t = a[col][j]; a[col][j] = a[piv][j]; a[piv][j] = t; }
        const double ip = 1.0 / a[col][col];
        for (int j = 0; j < 8; j++) a[col][j] *= ip;
        for (int r = 0; r < 4; r++) if (r != col) { const double f = a[r][col]; if (f != 0.0) for (int j = 0; j < 8; j++) a[r][j] -= f * a[col][j]; }
    }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) out[4 * i + j] = a[i][4 + j];
    return 0;
}

/* warp_samples_to_canonical (ray_utils.py:62-90): closest point on the mesh (brute force over all faces, double;
 * the reference uses libigl's AABB tree -- same mathematical result), mask = dist^2 < threshold,
 * barycentric blend of the three per-vertex 4x4 (double), inverse, apply.  Ties between faces: lowest face id.
 * Outputs: can_pts [P,3] double, closest [P,3] double, dist2 [P] double, face_id [P], mask [P] uint8. */
ORC_API int orc_warp_samples(const float *pts, const float *verts, const int32_t *faces, const double *T, uint32_t P, uint32_t F,
                             double threshold, double *can_pts, double *closest, double *dist2, int32_t *face_id, uint8_t *mask)
{
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)P; i++) {
        const double p[3] = { pts[3 * i], pts[3 * i + 1], pts[3 * i + 2] };
        double best = INFINITY, bc[3] = {0, 0, 0}; int bf = 0;
        for (uint32_t f = 0; f < F; f++) {
            double a[3], b[3], c[3], q[3];
            for (int k = 0; k < 3; k++) { a[k] = verts[3 * faces[3 * f] + k]; b[k] = verts[3 * faces[3 * f + 1] + k]; c[k] = verts[3 * faces[3 * f + 2] + k]; }
            closest_pt_tri(p, a, b, c, q);
            const double dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2], d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < best) { best = d2; bf = (int)f; bc[0] = q[0]; bc[1] = q[1]; bc[2] = q[2]; }
        }
        /* igl.barycentric_coordinates_tri(closest, a, b, c) */
        double a[3], b[3], c[3], v0[3], v1[3], v2[3];
        const int32_t *fv = faces + 3 * bf;
        for (int k = 0; k < 3; k++) { a[k] = verts[3 * fv[0] + k]; b[k] = verts[3 * fv[1] + k]; c[k] = verts[3 * fv[2] + k];
                                      v0[k] = b[k] - a[k]; v1[k] = c[k] - a[k]; v2[k] = bc[k] - a[k]; }
        const double d00 = DOT(v0, v0), d01 = DOT(v0, v1), d11 = DOT(v1, v1), d20 = DOT(v2, v0), d21 = DOT(v2, v1);
        const double den = d00 * d11 - d01 * d01;
        const double bv = (d11 * d20 - d01 * d21) / den, bw = (d00 * d21 - d01 * d20) / den, bu = 1.0 - bv - bw;
        double M[16], Mi[16];
        for (int e = 0; e < 16; e++) M[e] = T[16 * (size_t)fv[0] + e] * bu + T[16 * (size_t)fv[1] + e] * bv + T[16 * (size_t)fv[2] + e] * bw;
        inv4(M, Mi);
        for (int r = 0; r < 3; r++) can_pts[3 * i + r] = Mi[4 * r] * p[0] + Mi[4 * r + 1] * p[1] + Mi[4 * r + 2] * p[2] + Mi[4 * r + 3];
        if (closest) for (int k = 0; k < 3; k++) closest[3 * i + k] = bc[k];
        if (dist2) dist2[i] = best;
        if (face_id) face_id[i] = bf;
        mask[i] = best < threshold ? 1 : 0;
    }
    return 0;
    #undef DOT
}
