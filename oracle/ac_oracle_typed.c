/* oracle/ac_oracle_typed.c -- TEST INFRASTRUCTURE: the double-precision instantiation of the two encoder extensions.
 *
 * The reference dispatches its kernels over the dtype of their tensors (AT_DISPATCH_FLOATING_TYPES_AND_HALF: encoder/hashencoder/src/
 * hashencoder.cu:352,391; encoder/shencoder/src/shencoder.cu:337,380).  This file restates scalar_t = double:
 *   kernel_grid (hashencoder.cu:73-220): the range test on the double input; `float pos[D]` from (float)inputs[d] and the fp32 interpolation
 *     weights exactly as in the float instantiation (:125-154); results[C] accumulated in double (`results[ch] += w * grid[index + ch]`,
 *     float x double -> double; the compiler contracts it to one fma, stated here as fma());
 *   kernel_grid_backward (:223-308): grad_grid[index + c] += w * grad_cur[c] in double;  kernel_input_backward (:311-337) in double;
 *   kernel_sh / kernel_sh_backward (shencoder.cu:28-384): the basis evaluated in double (coefficients = the double literals of the reference,
 *     here the generated double tables of ac_sh_table.h, evaluation order of ac_oracle_ops.c's fp32 routine).
 * scalar_t = at::Half is restated in oracle.py (widen, the fp32 routines, round once): see the note there.
 * PARITY UNPINNED for both non-float dtypes: the reference never exercises them and its CUDA sources cannot be built
 * here, so no output of these instantiations exists to compare with; the fp32 routines they derive from are pinned (ac_oracle.c header). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "ac_oracle.h"
#include "ac_sh_table.h"

#ifndef ORC_API
#define ORC_API __attribute__((visibility("default")))
#endif
#define TYPED_MAX_LEVELS 32

/* cell position and fractional part of one coordinate, fp32 whatever the dtype (hashencoder.cu:125-133) */
static int locate_f64(const double *x, uint32_t D, float scale, float *pos, uint32_t *pg)
{
    int oob = 0;
    for (uint32_t d = 0; d < D; d++) if (x[d] < 0 || x[d] > 1) oob = 1;
    if (oob) return 1;
    for (uint32_t d = 0; d < D; d++) {
        float p = fmaf((float)x[d], scale, 0.5f);
        pg[d] = (uint32_t)floorf(p);
        pos[d] = p - (float)pg[d];
    }
    return 0;
}

ORC_API int orc_hash_encode_forward_f64(const double *inputs, const double *grid, const int32_t *offsets, double *outputs, uint32_t B, uint32_t D,
                                        uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, double *dy_dx)
{
    if (!(D == 2 || D == 3) || !(C == 1 || C == 2 || C == 4 || C == 8) || L > TYPED_MAX_LEVELS) return 1;
    float scale[TYPED_MAX_LEVELS]; uint32_t res[TYPED_MAX_LEVELS];
    orc_hash_level_table(L, S, H, scale, res);
    for (uint32_t l = 0; l < L; l++) {
        const double *g = grid + (size_t)(uint32_t)offsets[l] * C;
        const uint32_t hs = (uint32_t)(offsets[l + 1] - offsets[l]);
        #pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < (int64_t)B; b++) {
            double *out = outputs + ((size_t)l * B + b) * C;
            double *dd = calc_grad_inputs ? dy_dx + (size_t)b * D * L * C + (size_t)l * D * C : NULL;
            float pos[3]; uint32_t pg[3];
            if (locate_f64(inputs + b * D, D, scale[l], pos, pg)) {
                for (uint32_t c = 0; c < C; c++) out[c] = 0;
                if (dd) for (uint32_t i = 0; i < D * C; i++) dd[i] = 0;
                continue;
            }
            double acc[8] = {0};
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1; uint32_t pl[3];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                    else { w *= pos[d]; pl[d] = pg[d] + 1; }
                }
                const uint32_t index = orc_grid_index(D, C, 0, hs, res[l], pl);
                for (uint32_t c = 0; c < C; c++) acc[c] = fma((double)w, g[index + c], acc[c]);
            }
            for (uint32_t c = 0; c < C; c++) out[c] = acc[c];
            if (dd) {
                for (uint32_t gd = 0; gd < D; gd++) {
                    double rg[8] = {0};
                    for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                        float w = scale[l]; uint32_t pl[3];
                        for (uint32_t nd = 0; nd < D - 1; nd++) {
                            const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                            if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                            else { w *= pos[d]; pl[d] = pg[d] + 1; }
                        }
                        pl[gd] = pg[gd];
                        const uint32_t il = orc_grid_index(D, C, 0, hs, res[l], pl);
                        pl[gd] = pg[gd] + 1;
                        const uint32_t ir = orc_grid_index(D, C, 0, hs, res[l], pl);
                        for (uint32_t c = 0; c < C; c++) rg[c] = fma((double)w, g[ir + c] - g[il + c], rg[c]);
                    }
                    for (uint32_t c = 0; c < C; c++) dd[gd * C + c] = rg[c];
                }
            }
        }
    }
    return 0;
}

/* serial in b per level: the canonical accumulation order (the GPU adds with fp64 atomics, order-free: compared with a tolerance) */
ORC_API int orc_hash_encode_backward_f64(const double *grad, const double *inputs, const int32_t *offsets, double *grad_grid, uint32_t B, uint32_t D,
                                         uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, const double *dy_dx, double *grad_inputs)
{
    if (!(D == 2 || D == 3) || !(C == 1 || C == 2 || C == 4 || C == 8) || L > TYPED_MAX_LEVELS) return 1;
    float scale[TYPED_MAX_LEVELS]; uint32_t res[TYPED_MAX_LEVELS];
    orc_hash_level_table(L, S, H, scale, res);
    #pragma omp parallel for schedule(dynamic, 1)
    for (int64_t l = 0; l < (int64_t)L; l++) {
        double *gg = grad_grid + (size_t)(uint32_t)offsets[l] * C;
        const uint32_t hs = (uint32_t)(offsets[l + 1] - offsets[l]);
        for (uint32_t b = 0; b < B; b++) {
            float pos[3]; uint32_t pg[3];
            if (locate_f64(inputs + (size_t)b * D, D, scale[l], pos, pg)) continue;
            const double *g = grad + ((size_t)l * B + b) * C;
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1; uint32_t pl[3];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                    else { w *= pos[d]; pl[d] = pg[d] + 1; }
                }
                const uint32_t index = orc_grid_index(D, C, 0, hs, res[l], pl);
                for (uint32_t c = 0; c < C; c++) gg[index + c] += (double)w * g[c];
            }
        }
    }
    if (calc_grad_inputs) {
        for (uint32_t t = 0; t < B * D; t++) {
            const uint32_t b = t / D, d = t - b * D;
            const double *dd = dy_dx + (size_t)b * L * D * C;
            double r = 0;
            for (uint32_t l = 0; l < L; l++)
                for (uint32_t ch = 0; ch < C; ch++)
                    r = fma(grad[((size_t)l * B + b) * C + ch], dd[l * D * C + d * C + ch], r);
            grad_inputs[t] = r;
        }
    }
    return 0;
}

static double sh_eval_f64(const unsigned short *off, const double *coef, const unsigned char (*ex)[3], int idx, const double px[8], const double py[8],
                          const double pz[8])
{
    double acc = 0.0;
    for (int m = off[idx]; m < off[idx + 1]; m++) {
        const double mono = (px[ex[m][0]] * py[ex[m][1]]) * pz[ex[m][2]];
        acc = fma(coef[m], mono, acc);
    }
    return acc;
}

ORC_API int orc_sh_encode_forward_f64(const double *inputs, double *outputs, uint32_t B, uint32_t D, uint32_t C, int calc_grad_inputs, double *dy_dx)
{
    if (D != 3 || C < 1 || C > 8) return 1;
    const uint32_t C2 = C * C;
    #pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)B; b++) {
        double p[3][8];
        for (int a = 0; a < 3; a++) {
            p[a][0] = 1.0;
            for (int k = 1; k < 8; k++) p[a][k] = p[a][k - 1] * inputs[b * 3 + a];
        }
        for (uint32_t i = 0; i < C2; i++) outputs[b * C2 + i] = sh_eval_f64(AC_SH_OFF0, AC_SH_COEF0D, AC_SH_EXP0, i, p[0], p[1], p[2]);
        if (calc_grad_inputs) {
            double *dx = dy_dx + (size_t)b * 3 * C2, *dy = dx + C2, *dz = dy + C2;
            for (uint32_t i = 0; i < C2; i++) {
                dx[i] = sh_eval_f64(AC_SH_OFF1, AC_SH_COEF1D, AC_SH_EXP1, i, p[0], p[1], p[2]);
                dy[i] = sh_eval_f64(AC_SH_OFF2, AC_SH_COEF2D, AC_SH_EXP2, i, p[0], p[1], p[2]);
                dz[i] = sh_eval_f64(AC_SH_OFF3, AC_SH_COEF3D, AC_SH_EXP3, i, p[0], p[1], p[2]);
            }
        }
    }
    return 0;
}

ORC_API int orc_sh_encode_backward_f64(const double *grad, uint32_t B, uint32_t D, uint32_t C, const double *dy_dx, double *grad_inputs)
{
    if (D != 3 || C < 1 || C > 8) return 1;
    const uint32_t C2 = C * C;
    for (uint32_t t = 0; t < B * 3; t++) {
        const uint32_t b = t / 3, d = t - b * 3;
        double acc = grad_inputs[t];
        for (uint32_t ch = 0; ch < C2; ch++) acc = fma(grad[(size_t)b * C2 + ch], dy_dx[(size_t)b * 3 * C2 + d * C2 + ch], acc);
        grad_inputs[t] = acc;
    }
    return 0;
}
