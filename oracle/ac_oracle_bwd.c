/*
 * oracle/ac_oracle_bwd.c -- TEST INFRASTRUCTURE (see ac_oracle.c: only tests/, smoke() and bench.py's cpu_baseline leg load the oracle).
 *
 * Backward of the render core of NeRFRenderer.run under autograd -- what `loss.backward()` computes for
 *     loss = <g_image, image> + <g_wsum, weights_sum> + <g_depth, depth> + <g_nmap, normal_map> + g_eik * gradient_error
 * (reference models/instant_nsr.py:190-299 differentiated by torch; the callers are stylize.py:163-193 and reconstruct.py:101-112) --
 * restated as one analytic reverse pass in DOUBLE precision, including the scatter into the hash table.  It is an independent witness for the
 * HIP backward (sdf_train.hip / hash_stencil.hip), which runs in fp32 with recomputed activations and a binned scatter: the two share no
 * code, no summation order and no number format.  Pinned against the reference's own autograd by tests/golden/train_grad.npz.
 *
 * What is differentiated, and what is a constant (exactly as in the reference):
 *   - the sample positions z_vals come from the no-grad sampling stage (:176-184): constants; so are deltas, mid points, the clamped
 *     sample points (:190-207, fp32 arithmetic, taken over bit for bit), the hash cell indices and interpolation weights (functions of
 *     the points only, fp32 like hashencoder.cu:122-134), the `relax` indicator |x| < 1.2 (:266-268, detached) and the ray geometry;
 *   - differentiable: the hash table entries, the EFFECTIVE (weight-normed) matrices / biases of sdf_net and color_net, and
 *     inv_s = forward_variance() (:219, :665-667).  weight_norm and exp(10 variance) stay with the caller (the tests chain them in float64).
 *
 * Per sample (reference lines):
 *   sdf_out = forward_sdf(x) (:210-212, :627-642): enc = HashEncoder(x) (hashencoder.cu:94-175), h = softplus100(W1 [x, enc] + b1), W2 h + b2
 *   gradient_k = 0.5 (sdf(clamp(x + eps e_k)) - sdf(clamp(x - eps e_k))) / eps (:214, :687-704); normal = gradient / (1e-5 + |gradient|) (:215)
 *   color = sigmoid(Wc3 relu(Wc2 relu(Wc1 [x, normal, feat]))) (:217, :644-663)
 *   true_cos = d . normal; iter_cos = -(softplus100(-true_cos / 2 + 1/2) (1 - car) + softplus100(-true_cos) car) (:222-233)
 *   alpha = clip((sigmoid((sdf - iter_cos delta / 2) inv_s) - sigmoid((sdf + iter_cos delta / 2) inv_s) + 1e-5) / (sigmoid(prev) + 1e-5), 0, 1) (:236-243)
 * Per ray: weights = alpha cumprod([1, 1 - alpha + 1e-7])[:-1] (:250); weights_sum, image (+ (1 - weights_sum) bg, :294), normal_map,
 *   depth = sum weights clamp((z - near) / (far - near), 0, 1) (:252-263); batch: gradient_error = sum relax (|gradient| - 1)^2 / (sum relax + 1e-5) (:266-272).
 * softplus100 is torch.nn.Softplus(beta=100, threshold=20): derivative sigmoid(100 x), 1 beyond the threshold.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "ac_oracle.h"

#define NW1 (64 * 35)
#define NW2 (16 * 64)
#define NC1 (64 * 21)
#define NC2 (64 * 64)
#define NC3 (3 * 64)
#define NPAR (NW1 + 64 + NW2 + 16 + NC1 + NC2 + NC3)
#define OFF_W1 0
#define OFF_B1 (OFF_W1 + NW1)
#define OFF_W2 (OFF_B1 + 64)
#define OFF_B2 (OFF_W2 + NW2)
#define OFF_C1 (OFF_B2 + 16)
#define OFF_C2 (OFF_C1 + NC1)
#define OFF_C3 (OFF_C2 + NC2)
#define BWD_MAXT 128

typedef struct {
    double *g_table;     /* [n_entries * 2], accumulated into (caller zero-fills) */
    double *g_params;    /* [NPAR]: W1 [64,35], b1 [64], W2 [16,64], b2 [16], Wc1 [64,21], Wc2 [64,64], Wc3 [3,64]; overwritten.  A field with view
                          * directions (orc_field.Wsh): [NPAR + 1024], the last 1024 = d Wsh [64,16] */
    double *g_inv_s;     /* [1]; overwritten */
    double *fwd;         /* optional [N, 8]: image (3), weights_sum, depth, normal_map (3) of the fp64 forward (cross-check with orc_render_rays) */
    double *gradient_error; /* optional [1]: the fp64 forward's gradient_error */
} orc_core_grads;

static inline double sigm(double x) { return 1.0 / (1.0 + exp(-x)); }
static inline double sp100(double x) { double t = 100.0 * x; return t > 20.0 ? x : log1p(exp(t)) / 100.0; }     /* torch Softplus(beta=100, threshold=20) */
static inline double dsp100(double x) { double t = 100.0 * x; return t > 20.0 ? 1.0 : sigm(t); }
static inline float clampf_(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* the 8 corners of one (point, level): table entry index (x2 = first channel) and interpolation weight -- index / position arithmetic in fp32
 * exactly as hashencoder.cu:122-166 (they are constants of the differentiation), the weight product widened to double */
typedef struct { uint32_t idx[16][8]; double w[16][8]; int oob; } enc_geo;

static void encode_geo(const orc_field *f, const float x[3], float bound, enc_geo *g)
{
    const float two_b = (float)(2.0 * (double)bound);
    float u[3];
    g->oob = 0;
    for (int d = 0; d < 3; d++) { u[d] = (x[d] + bound) / two_b; if (u[d] < 0.0f || u[d] > 1.0f) g->oob = 1; }
    if (g->oob) return;
    for (int l = 0; l < 16; l++) {
        const uint32_t hs = (uint32_t)(f->offsets[l + 1] - f->offsets[l]);
        float pos[3]; uint32_t pg[3];
        for (int d = 0; d < 3; d++) {
            pos[d] = fmaf(u[d], f->scale[l], 0.5f);
            pg[d] = (uint32_t)floorf(pos[d]);
            pos[d] -= (float)pg[d];
        }
        for (uint32_t c = 0; c < 8; c++) {
            double w = 1.0; uint32_t pl[3];
            for (int d = 0; d < 3; d++) {
                if ((c & (1u << d)) == 0) { w *= 1.0 - (double)pos[d]; pl[d] = pg[d]; }
                else { w *= (double)pos[d]; pl[d] = pg[d] + 1; }
            }
            g->idx[l][c] = (uint32_t)f->offsets[l] * 2u + orc_grid_index(3, 2, 0, hs, f->res[l], pl);
            g->w[l][c] = w;
        }
    }
}

typedef struct { double enc[32], pre[64], hid[64], out[16]; } sdf_fwd;

static void sdf_forward(const orc_field *f, const float x[3], const enc_geo *g, int n_out, sdf_fwd *s)
{
    for (int l = 0; l < 16; l++) {
        double a0 = 0.0, a1 = 0.0;
        if (!g->oob)
            for (int c = 0; c < 8; c++) { a0 += g->w[l][c] * (double)f->table[g->idx[l][c]]; a1 += g->w[l][c] * (double)f->table[g->idx[l][c] + 1]; }
        s->enc[2 * l] = a0; s->enc[2 * l + 1] = a1;
    }
    for (int u = 0; u < 64; u++) {
        const float *w = f->W1 + u * 35;
        double acc = (double)f->b1[u] + (double)w[0] * x[0] + (double)w[1] * x[1] + (double)w[2] * x[2];
        for (int k = 0; k < 32; k++) acc += (double)w[3 + k] * s->enc[k];
        s->pre[u] = acc; s->hid[u] = sp100(acc);
    }
    for (int o = 0; o < n_out; o++) {
        const float *w = f->W2 + o * 64;
        double acc = (double)f->b2[o];
        for (int u = 0; u < 64; u++) acc += (double)w[u] * s->hid[u];
        s->out[o] = acc;
    }
}

/* reverse of sdf_forward for the upstream gradient g_out[0..n_out): parameter gradients into gp, table gradient into gt (atomic) */
static void sdf_backward(const orc_field *f, const float x[3], const enc_geo *g, const sdf_fwd *s, const double *g_out, int n_out, double *gp, double *gt)
{
    double g_hid[64], g_enc[32];
    int any = 0;
    for (int o = 0; o < n_out; o++) any |= g_out[o] != 0.0;
    if (!any) return;
    memset(g_hid, 0, sizeof g_hid);
    for (int o = 0; o < n_out; o++) {
        if (g_out[o] == 0.0) continue;
        const float *w = f->W2 + o * 64;
        gp[OFF_B2 + o] += g_out[o];
        for (int u = 0; u < 64; u++) { gp[OFF_W2 + o * 64 + u] += g_out[o] * s->hid[u]; g_hid[u] += g_out[o] * (double)w[u]; }
    }
    memset(g_enc, 0, sizeof g_enc);
    for (int u = 0; u < 64; u++) {
        const double gpre = g_hid[u] * dsp100(s->pre[u]);
        const float *w = f->W1 + u * 35;
        gp[OFF_B1 + u] += gpre;
        for (int k = 0; k < 3; k++) gp[OFF_W1 + u * 35 + k] += gpre * (double)x[k];
        for (int k = 0; k < 32; k++) { gp[OFF_W1 + u * 35 + 3 + k] += gpre * s->enc[k]; g_enc[k] += gpre * (double)w[3 + k]; }
    }
    if (g->oob) return;
    for (int l = 0; l < 16; l++)
        for (int c = 0; c < 8; c++) {
            const double v0 = g->w[l][c] * g_enc[2 * l], v1 = g->w[l][c] * g_enc[2 * l + 1];
            #pragma omp atomic
            gt[g->idx[l][c]] += v0;
            #pragma omp atomic
            gt[g->idx[l][c] + 1] += v1;
        }
}

typedef struct { double in[21], p1[64], h1[64], p2[64], h2[64], o[3], rgb[3]; } col_fwd;

/* sh: the 16 spherical harmonics of the ray direction in double (from the fp32 values of orc_sh16: constants of the differentiation), or NULL */
static void color_forward(const orc_field *f, const float x[3], const double n[3], const double *sdf_out, const double *sh, col_fwd *c)
{
    for (int k = 0; k < 3; k++) { c->in[k] = x[k]; c->in[3 + k] = n[k]; }
    for (int k = 0; k < 15; k++) c->in[6 + k] = sdf_out[1 + k];
    for (int u = 0; u < 64; u++) {
        double acc = 0.0;
        if (sh) for (int j = 0; j < 16; j++) acc += (double)f->Wsh[u * 16 + j] * sh[j];
        for (int k = 0; k < 21; k++) acc += (double)f->Wc1[u * 21 + k] * c->in[k];
        c->p1[u] = acc; c->h1[u] = acc > 0.0 ? acc : 0.0;
    }
    for (int u = 0; u < 64; u++) {
        double acc = 0.0;
        for (int k = 0; k < 64; k++) acc += (double)f->Wc2[u * 64 + k] * c->h1[k];
        c->p2[u] = acc; c->h2[u] = acc > 0.0 ? acc : 0.0;
    }
    for (int o = 0; o < 3; o++) {
        double acc = 0.0;
        for (int k = 0; k < 64; k++) acc += (double)f->Wc3[o * 64 + k] * c->h2[k];
        c->o[o] = acc; c->rgb[o] = sigm(acc);
    }
}

/* g_rgb -> parameter gradients, g_n (normal, accumulated), g_feat (sdf_out[1..15], accumulated into g_sdf_out[1..]) */
static void color_backward(const orc_field *f, const col_fwd *c, const double g_rgb[3], double *gp, double g_n[3], double *g_sdf_out,
                           const double *sh, double *g_wsh)
{
    double g_h2[64], g_h1[64], g_in[21];
    memset(g_h2, 0, sizeof g_h2); memset(g_h1, 0, sizeof g_h1); memset(g_in, 0, sizeof g_in);
    for (int o = 0; o < 3; o++) {
        const double go = g_rgb[o] * c->rgb[o] * (1.0 - c->rgb[o]);
        for (int k = 0; k < 64; k++) { gp[OFF_C3 + o * 64 + k] += go * c->h2[k]; g_h2[k] += go * (double)f->Wc3[o * 64 + k]; }
    }
    for (int u = 0; u < 64; u++) {
        if (!(c->p2[u] > 0.0)) continue;
        for (int k = 0; k < 64; k++) { gp[OFF_C2 + u * 64 + k] += g_h2[u] * c->h1[k]; g_h1[k] += g_h2[u] * (double)f->Wc2[u * 64 + k]; }
    }
    for (int u = 0; u < 64; u++) {
        if (!(c->p1[u] > 0.0)) continue;
        for (int k = 0; k < 21; k++) { gp[OFF_C1 + u * 21 + k] += g_h1[u] * c->in[k]; g_in[k] += g_h1[u] * (double)f->Wc1[u * 21 + k]; }
        if (sh && g_wsh) for (int j = 0; j < 16; j++) g_wsh[u * 16 + j] += g_h1[u] * sh[j];
    }
    for (int k = 0; k < 3; k++) g_n[k] += g_in[3 + k];
    for (int k = 0; k < 15; k++) g_sdf_out[1 + k] += g_in[6 + k];
}

static void ray_near_far(const float *o, const float *d, float bound, float *near_, float *far_)
{
    float near = -INFINITY, far = INFINITY;          /* near_far_from_bound, cube (instant_nsr.py:58-77) */
    for (int k = 0; k < 3; k++) {
        float dd = d[k] + 1e-15f;
        float tmin = (-bound - o[k]) / dd, tmax = (bound - o[k]) / dd;
        float lo = tmin < tmax ? tmin : tmax, hi = tmin > tmax ? tmin : tmax;
        if (k == 0 || lo > near) near = lo;
        if (k == 0 || hi < far) far = hi;
    }
    if (near < 0.05f) near = 0.05f;
    *near_ = near; *far_ = far;
}

/* the clamped sample points of one ray and their spacing, fp32 like the forward (:190-207) */
typedef struct { const float *ext_pts; const uint8_t *mask; const float *near_m, *far_m; } posed_args;     /* all optional, see orc_render_core_backward_posed */

static void ray_points(const orc_render_opts *op, const float *o, const float *d, const float *z, float *pts, float *delta, float *zn, float *near_, float *far_,
                       const posed_args *pa, int r)
{
    const int T = op->num_steps + op->upsample_steps;
    float near, far;
    ray_near_far(o, d, op->bound, &near, &far);
    if (pa && pa->near_m) {                          /* :148-153 the mesh-guided range where the ray passes the body */
        if (!isinf(pa->near_m[r])) near = pa->near_m[r];
        if (!isinf(pa->far_m[r])) far = pa->far_m[r];
    }
    const float span = far - near, sample_dist = span / (float)op->num_steps;
    for (int i = 0; i < T; i++) {
        delta[i] = (i < T - 1) ? z[i + 1] - z[i] : sample_dist;
        const float zmid = (i < T - 1) ? z[i] + 0.5f * delta[i] : z[i];
        for (int k = 0; k < 3; k++) pts[3 * i + k] = clampf_(o[k] + d[k] * zmid, -op->bound, op->bound);
        if (pa && pa->ext_pts)                       /* :198-207 posed space: the SMPL inverse warp of the mid points (numpy in the reference: constants) */
            for (int k = 0; k < 3; k++) pts[3 * i + k] = clampf_(pa->ext_pts[((size_t)r * T + i) * 3 + k], -op->bound, op->bound);
        zn[i] = clampf_((z[i] - near) / span, 0.0f, 1.0f);
    }
    *near_ = near; *far_ = far;
}

/* Posed space (run(render_can=False), instant_nsr.py:166-172,198-207,246-249): ext_pts [N,T,3] = the warped mid points the field is evaluated at
 * (the view directions stay the rays'), mask [N,T] multiplies alpha, near_m / far_m [N] replace the cube's range where finite.  Any of them NULL =
 * the canonical-space render. */
ORC_API int orc_render_core_backward_posed(const orc_field *f, const orc_render_opts *op, const float *rays_o, const float *rays_d, const float *bg,
                                           const float *z_vals, const float *ext_pts, const uint8_t *mask, const float *near_m, const float *far_m,
                                           const float *g_image, const float *g_wsum, const float *g_depth, const float *g_nmap,
                                           double g_eik, const orc_core_grads *out)
{
    const posed_args pa_ = { ext_pts, mask, near_m, far_m }, *pa = &pa_;
    const int N = op->n_rays, T = op->num_steps + op->upsample_steps;
    if (T > BWD_MAXT || T <= 0 || !(op->fd_eps > 0.0f)) return 1;
    const float bound = op->bound, eps = op->fd_eps;
    const double inv_s = (double)op->inv_s, car = (double)op->cos_anneal_ratio;
    /* the eikonal denominator needs every sample of the batch first (it depends on the points only) */
    double e_den = 0.0;
    #pragma omp parallel for schedule(static) reduction(+ : e_den)
    for (int r = 0; r < N; r++) {
        float pts[BWD_MAXT * 3], delta[BWD_MAXT], zn[BWD_MAXT], near, far;
        ray_points(op, rays_o + 3 * r, rays_d + 3 * r, z_vals + (size_t)r * T, pts, delta, zn, &near, &far, pa, r);
        for (int i = 0; i < T; i++) {
            const float *p = pts + 3 * i;
            const float pn = sqrtf((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]);
            if (pn < 1.2f) e_den += 1.0;
        }
    }
    e_den += 1e-5;
    const int NPV = NPAR + (f->Wsh ? 64 * 16 : 0);
    memset(out->g_params, 0, NPV * sizeof(double));
    double g_inv_s = 0.0, e_num = 0.0;
    #pragma omp parallel
    {
        double *gp = (double *)calloc(NPAR + 64 * 16, sizeof(double));
        double gs_local = 0.0, en_local = 0.0;
        #pragma omp for schedule(dynamic, 1)
        for (int r = 0; r < N; r++) {
            const float *o = rays_o + 3 * r, *d = rays_d + 3 * r, *z = z_vals + (size_t)r * T;
            float pts[BWD_MAXT * 3], delta[BWD_MAXT], zn[BWD_MAXT], near, far;
            ray_points(op, o, d, z, pts, delta, zn, &near, &far, pa, r);
            double shd[16]; const double *sh = NULL;             /* use_viewdirs: sh(d) of this ray (fp32 values, constants of the differentiation) */
            if (f->Wsh) { float shf[16]; orc_sh16(d, shf); for (int j = 0; j < 16; j++) shd[j] = shf[j]; sh = shd; }
            /* ---- forward of every sample (kept: the reverse pass needs the ray's transmittance first) ---- */
            static __thread enc_geo *geo = NULL;        /* [T][7] */
            static __thread sdf_fwd *sf = NULL;         /* [T][7] */
            static __thread col_fwd *cf = NULL;         /* [T] */
            if (!geo) { geo = malloc(sizeof(enc_geo) * BWD_MAXT * 7); sf = malloc(sizeof(sdf_fwd) * BWD_MAXT * 7); cf = malloc(sizeof(col_fwd) * BWD_MAXT); }
            double grad[BWD_MAXT][3], gn[BWD_MAXT], nrm[BWD_MAXT][3], alpha[BWD_MAXT], raw[BWD_MAXT], pc[BWD_MAXT], nc[BWD_MAXT], tc[BWD_MAXT],
                   half[BWD_MAXT], Tr[BWD_MAXT], w[BWD_MAXT];
            float q7[BWD_MAXT][7][3];
            int relax[BWD_MAXT];
            for (int i = 0; i < T; i++) {
                const float *p = pts + 3 * i;
                for (int e = 0; e < 7; e++) {
                    float *q = q7[i][e];
                    q[0] = p[0]; q[1] = p[1]; q[2] = p[2];
                    if (e) { const int k = (e - 1) >> 1; q[k] = clampf_(p[k] + ((e - 1) & 1 ? -eps : eps), -bound, bound); }
                    encode_geo(f, q, bound, &geo[i * 7 + e]);
                    sdf_forward(f, q, &geo[i * 7 + e], e ? 1 : 16, &sf[i * 7 + e]);
                }
                for (int k = 0; k < 3; k++) grad[i][k] = 0.5 * (sf[i * 7 + 1 + 2 * k].out[0] - sf[i * 7 + 2 + 2 * k].out[0]) / (double)eps;
                gn[i] = sqrt(grad[i][0] * grad[i][0] + grad[i][1] * grad[i][1] + grad[i][2] * grad[i][2]);
                for (int k = 0; k < 3; k++) nrm[i][k] = grad[i][k] / (1e-5 + gn[i]);
                color_forward(f, p, nrm[i], sf[i * 7].out, sh, &cf[i]);
                tc[i] = (double)d[0] * nrm[i][0] + (double)d[1] * nrm[i][1] + (double)d[2] * nrm[i][2];
                const double ic = -(sp100(-tc[i] * 0.5 + 0.5) * (1.0 - car) + sp100(-tc[i]) * car);
                half[i] = ic * (double)delta[i] * 0.5;
                const double sdf = sf[i * 7].out[0];
                pc[i] = sigm((sdf - half[i]) * inv_s); nc[i] = sigm((sdf + half[i]) * inv_s);
                raw[i] = (pc[i] - nc[i] + 1e-5) / (pc[i] + 1e-5);
                alpha[i] = raw[i] < 0.0 ? 0.0 : (raw[i] > 1.0 ? 1.0 : raw[i]);
                if (pa->mask && !pa->mask[(size_t)r * T + i]) alpha[i] = 0.0;                  /* :246-249 alpha * alpha_mask */
                const float pn = sqrtf((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]);
                relax[i] = pn < 1.2f;
                if (relax[i]) en_local += (gn[i] - 1.0) * (gn[i] - 1.0);
            }
            double tr = 1.0, wsum = 0.0, img[3] = { 0, 0, 0 }, nm[3] = { 0, 0, 0 }, dep = 0.0;
            for (int i = 0; i < T; i++) {
                Tr[i] = tr; w[i] = alpha[i] * tr; tr *= 1.0 - alpha[i] + 1e-7;
                wsum += w[i]; dep += w[i] * (double)zn[i];
                for (int k = 0; k < 3; k++) { img[k] += w[i] * cf[i].rgb[k]; nm[k] += w[i] * nrm[i][k]; }
            }
            const double b3[3] = { bg ? bg[3 * r] : 1.0, bg ? bg[3 * r + 1] : 1.0, bg ? bg[3 * r + 2] : 1.0 };
            if (out->fwd) {
                double *fo = out->fwd + (size_t)r * 8;
                for (int k = 0; k < 3; k++) { fo[k] = img[k] + (1.0 - wsum) * b3[k]; fo[5 + k] = nm[k]; }
                fo[3] = wsum; fo[4] = dep;
            }
            /* ---- reverse pass ---- */
            const double gi[3] = { g_image ? g_image[3 * r] : 0.0, g_image ? g_image[3 * r + 1] : 0.0, g_image ? g_image[3 * r + 2] : 0.0 };
            const double gw = g_wsum ? g_wsum[r] : 0.0, gd = g_depth ? g_depth[r] : 0.0;
            const double gm[3] = { g_nmap ? g_nmap[3 * r] : 0.0, g_nmap ? g_nmap[3 * r + 1] : 0.0, g_nmap ? g_nmap[3 * r + 2] : 0.0 };
            const double gi_bg = gi[0] * b3[0] + gi[1] * b3[1] + gi[2] * b3[2];
            double suffix = 0.0;                             /* sum_{j > i} dL/dw_j w_j */
            for (int i = T - 1; i >= 0; i--) {
                const double dw = gi[0] * cf[i].rgb[0] + gi[1] * cf[i].rgb[1] + gi[2] * cf[i].rgb[2] - gi_bg + gw + gd * (double)zn[i]
                                  + gm[0] * nrm[i][0] + gm[1] * nrm[i][1] + gm[2] * nrm[i][2];
                double g_alpha = dw * Tr[i] - suffix / (1.0 - alpha[i] + 1e-7);
                if (pa->mask && !pa->mask[(size_t)r * T + i]) g_alpha = 0.0;                   /* d (alpha * 0) / d alpha */
                suffix += dw * w[i];
                double g_rgb[3], g_n[3], g_out16[16];
                memset(g_out16, 0, sizeof g_out16);
                for (int k = 0; k < 3; k++) { g_rgb[k] = gi[k] * w[i]; g_n[k] = gm[k] * w[i]; }
                /* alpha -> sdf, half, inv_s */
                double g_half = 0.0;
                if (raw[i] >= 0.0 && raw[i] <= 1.0 && g_alpha != 0.0) {
                    const double den = pc[i] + 1e-5;
                    const double g_pc = g_alpha * (nc[i] / (den * den)), g_nc = -g_alpha / den;
                    const double sdf = sf[i * 7].out[0];
                    const double dpc = pc[i] * (1.0 - pc[i]), dnc = nc[i] * (1.0 - nc[i]);
                    g_out16[0] += (g_pc * dpc + g_nc * dnc) * inv_s;
                    g_half += (-g_pc * dpc + g_nc * dnc) * inv_s;
                    gs_local += g_pc * dpc * (sdf - half[i]) + g_nc * dnc * (sdf + half[i]);
                }
                /* half -> iter_cos -> true_cos -> normal */
                const double g_ic = g_half * (double)delta[i] * 0.5;
                const double g_tc = g_ic * (0.5 * (1.0 - car) * dsp100(-tc[i] * 0.5 + 0.5) + car * dsp100(-tc[i]));
                for (int k = 0; k < 3; k++) g_n[k] += g_tc * (double)d[k];
                color_backward(f, &cf[i], g_rgb, gp, g_n, g_out16, sh, gp + NPAR);
                /* normal = g / (1e-5 + |g|), eikonal term */
                double g_g[3] = { 0, 0, 0 };
                if (gn[i] > 0.0) {
                    const double den = 1e-5 + gn[i];
                    const double dot = g_n[0] * grad[i][0] + g_n[1] * grad[i][1] + g_n[2] * grad[i][2];
                    for (int k = 0; k < 3; k++) g_g[k] = g_n[k] / den - dot / (den * den) * (grad[i][k] / gn[i]);
                    if (relax[i] && g_eik != 0.0)
                        for (int k = 0; k < 3; k++) g_g[k] += g_eik * 2.0 * (gn[i] - 1.0) / e_den * (grad[i][k] / gn[i]);
                }
                sdf_backward(f, q7[i][0], &geo[i * 7], &sf[i * 7], g_out16, 16, gp, out->g_table);
                for (int k = 0; k < 3; k++) {
                    const double gs = 0.5 * g_g[k] / (double)eps, gneg = -gs;
                    sdf_backward(f, q7[i][1 + 2 * k], &geo[i * 7 + 1 + 2 * k], &sf[i * 7 + 1 + 2 * k], &gs, 1, gp, out->g_table);
                    sdf_backward(f, q7[i][2 + 2 * k], &geo[i * 7 + 2 + 2 * k], &sf[i * 7 + 2 + 2 * k], &gneg, 1, gp, out->g_table);
                }
            }
        }
        #pragma omp critical
        {
            for (int k = 0; k < NPV; k++) out->g_params[k] += gp[k];
            g_inv_s += gs_local; e_num += en_local;
        }
        free(gp);
    }
    out->g_inv_s[0] = g_inv_s;
    if (out->gradient_error) out->gradient_error[0] = e_num / e_den;
    return 0;
}

ORC_API int orc_render_core_backward(const orc_field *f, const orc_render_opts *op, const float *rays_o, const float *rays_d, const float *bg,
                                     const float *z_vals, const float *g_image, const float *g_wsum, const float *g_depth, const float *g_nmap,
                                     double g_eik, const orc_core_grads *out)
{
    return orc_render_core_backward_posed(f, op, rays_o, rays_d, bg, z_vals, NULL, NULL, NULL, NULL, g_image, g_wsum, g_depth, g_nmap, g_eik, out);
}
