/*
 * oracle/ac_oracle.h -- TEST INFRASTRUCTURE.  Types shared by the translation units of the CPU oracle (ac_oracle.c: forward of the hot
 * path; ac_oracle_bwd.c: fp64 backward of the render core).  See ac_oracle.c for what the oracle is and who may load it.
 */
#ifndef AC_ORACLE_H
#define AC_ORACLE_H
#include <stdint.h>

#define ORC_API __attribute__((visibility("default")))

/* effective (weight-normed) parameters of the default NeRFNetwork (models/instant_nsr.py:478-591) */
typedef struct {
    const float *table;        /* embeddings [n_entries, 2] */
    const int32_t *offsets;    /* [17] */
    float scale[16];           /* per-level table (orc_hash_level_table) */
    uint32_t res[16];
    const float *W1, *b1;      /* effective (weight-normed) sdf_net.0: [64,35], [64] */
    const float *W2, *b2;      /* sdf_net.1: [16,64], [16] */
    const float *Wc1;          /* color_net.0: [64,21] */
    const float *Wc2;          /* color_net.1: [64,64] */
    const float *Wc3;          /* color_net.2: [3,64] */
    const float *Wsh;          /* NeRFNetwork(use_viewdirs=True) only, else NULL: the 16 columns of color_net.0 that multiply sh(d) -- the degree-4 spherical
                                * harmonics of the raw ray direction (models/instant_nsr.py:565-569, 644-653: h = cat[x, sh(d), n, geo_feat]) -- [64,16];
                                * Wc1 then holds the other 21 columns (x, n, geo_feat) */
} orc_field;

/* degree-4 spherical harmonics of a raw direction: the first 16 values of orc_sh_encode_forward (encoder/shencoder/src/shencoder.cu:28-...), same operations */
void orc_sh16(const float d[3], float sh[16]);

/* NeRFRenderer.run options (models/instant_nsr.py:133-299) */
typedef struct {
    int32_t n_rays;
    int32_t num_steps;        /* coarse samples T0: multiple of 16, 16..64  */
    int32_t upsample_steps;   /* multiple of 16, T0+up <= 128 */
    float bound;
    float inv_s;              /* forward_variance(): exp(10*variance).clip(1e-6,1e6) */
    float cos_anneal_ratio;
    float fd_eps;             /* 0.005*(1-normal_epsilon_ratio) */
    int32_t perturb;          /* training && perturb_overwrite: use noise[n, T0] */
} orc_render_opts;

ORC_API uint32_t orc_grid_index(uint32_t D, uint32_t C, uint32_t ch, uint32_t hashmap_size, uint32_t resolution, const uint32_t *pos_grid);
ORC_API void orc_hash_level_table(uint32_t L, float S, uint32_t H, float *scale, uint32_t *res);

#endif
