/*
 * include/avatarcraft_hip.h -- C ABI of libavatarcraft_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary of the AvatarCraft hot path: every entry point replaces one
 * function that the reference binds through pybind11 from its JIT-built CUDA extensions, with
 * the same argument meaning and order, plus (a) an explicit stream and (b) an int status
 * instead of a C++ exception (0 = ok; non-zero = bad arguments / launch failure, message in
 * ac_last_error(); the Python wrappers raise RuntimeError exactly where the reference's
 * TORCH_CHECK / std::runtime_error would).
 *
 * Conventions (same as the reference):
 *   - the CALLER allocates every buffer; the library never allocates user-visible memory and
 *     keeps no pointer after the call returns (small internal scratch is per-call);
 *   - all pointers are DEVICE pointers unless the parameter is documented "host";
 *   - all float data is fp32, all index data int32/uint32;
 *   - kernels are enqueued on `stream` (a hipStream_t; NULL = the null stream) and the call
 *     returns without synchronising.
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   encoder/hashencoder/src/hashencoder.h:13-14   hash_encode_forward / hash_encode_backward
 *   encoder/shencoder/src/shencoder.h:11,14       sh_encode_forward / sh_encode_backward
 *   raymarching/src/raymarching.h:8-17            march_rays_train, composite_rays_train_forward/
 *                                                 _backward, march_rays, composite_rays, compact_rays
 *   models/instant_nsr.py:133-299,358-408         NeRFRenderer.run / render (the fused path:
 *                                                 ac_render_rays; the reference has no native
 *                                                 entry for it -- `run_cuda` is missing)
 */
#ifndef AVATARCRAFT_HIP_H
#define AVATARCRAFT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *ac_stream_t; /* hipStream_t */

#define AC_OK 0
#define AC_ERR_BAD_ARG 1
#define AC_ERR_LAUNCH 2
#define AC_MAX_LEVELS 32

/* library identification / diagnostics */
int ac_version(void);                /* ABI version, currently 10 (round 6, second half: ac_sdf_stencil_backward_inputs, ac_hash_stencil_input_backward -- the position gradient
                                      * of the stencil query, for the curvature term; 9, round 6: ac_warp_mesh.seed_faces / seed_stride -- the struct grew --, ac_set_occupancy_barrier_ms, ac_debug_hold_cus; the phased occupancy launches are cooperative launches sized by the runtime's
                                      * occupancy figure; 7, round 5, second half: ac_render_rays_occupancy_phased, ac_render_rays_occupancy_train, ac_march_rays_train_scratch -- the
                                      * marcher's scratch grew --; 6, round 5: ac_render_rays_occupancy's max_steps, ac_field_sdf_grid, ac_marching_cubes*,
                                      * ac_density_grid_update, the SH colour input of ac_field; round 4 = 5: ac_render_opts.opacity_only -- the struct grew from 64 to 72 bytes --,
                                      * ac_field_samples, ac_render_rays_occupancy, the measurement / liveness accessors, the *_typed encoder entries) */
const char *ac_last_error(void);     /* message of the last failing call on this thread */
/* test utility: `blocks` workgroups of 1024 threads + lds_bytes of LDS each spin for `millis` ms of wall clock on `stream` -- a FOREIGN workload that holds
 * compute units (no memory traffic), for the liveness tests of the kernels that wait for each other (tests/test_gpu_run_cuda.py, tests/test_gpu_render.py) */
int ac_debug_hold_cus(uint32_t blocks, uint32_t lds_bytes, uint32_t millis, ac_stream_t stream);
/* test utility: the division-free unit coordinate (q = a * RN(1 / d); r = fma(-q, d, a); u = fma(r, RN(1 / d), q)) against the IEEE quotient a / d
 * on the device, for EVERY fp32 dividend whose bit pattern lies in [lo_bits, hi_bits]: *mismatches (device, zeroed by the caller) += the number that differ in a
 * bit (two NaNs count as equal).  tests/test_gpu_ops.py sweeps the renderer's whole dividend domain for the divisors the library accepts. */
int ac_debug_unit_div_check(uint32_t lo_bits, uint32_t hi_bits, float d, unsigned long long *mismatches, ac_stream_t stream);

/* ---- hash-grid encoder -------------------------------------------------------------------
 * replaces hash_encode_forward (encoder/hashencoder/src/hashencoder.cu:413-436)
 *   inputs  [B,D] in [0,1]; embeddings [sum_l T_l, C]; offsets [L+1] int32 (device);
 *   outputs [L,B,C] (level major); dy_dx [B, L*D*C] (only written if calc_grad_inputs).
 *   D in {2,3}, C in {1,2,4,8}, L <= AC_MAX_LEVELS, else AC_ERR_BAD_ARG
 *   ("GridEncoding: C must be 1, 2, 4, or 8." in the reference).
 *   S = log2(per_level_scale) as float, H = base resolution.
 *   offsets_host: the same L+1 offsets in HOST memory (the reference reads them on the device
 *   only; the MI355X kernels take the level table as launch constants). */
int ac_hash_encode_forward(const float *inputs, const float *embeddings, const int32_t *offsets,
                           const int32_t *offsets_host, float *outputs, uint32_t B, uint32_t D, uint32_t C,
                           uint32_t L, float S, uint32_t H, int calc_grad_inputs, float *dy_dx,
                           ac_stream_t stream);

/* replaces hash_encode_backward (hashencoder.cu:438-468): grad [L,B,C]; grad_embeddings is
 * accumulated into (caller zero-fills, hashgrid.py:61); grad_inputs [B,D] written iff
 * calc_grad_inputs. */
int ac_hash_encode_backward(const float *grad, const float *inputs, const float *embeddings,
                            const int32_t *offsets, const int32_t *offsets_host, float *grad_embeddings,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                            int calc_grad_inputs, const float *dy_dx, float *grad_inputs, ac_stream_t stream);

/* debugging / parity: table entry index (before *C) of every trilinear corner, [L,B,2^D] uint32,
 * 0xffffffff for out-of-range inputs */
int ac_hash_corner_indices(const float *inputs, const int32_t *offsets_host, uint32_t *corner_idx, uint32_t B,
                           uint32_t D, uint32_t L, float S, uint32_t H, ac_stream_t stream);

/* host helper: the per-level (scale, resolution) table every kernel uses
 * (hashencoder.cu:122-123), computed once on the host */
void ac_hash_level_table(uint32_t L, float S, uint32_t H, float *scale_host, uint32_t *res_host);

/* ---- spherical-harmonics encoder -----------------------------------------------------------
 * replaces sh_encode_forward / sh_encode_backward (encoder/shencoder/src/shencoder.cu:403-441)
 *   inputs [B,3]; outputs [B, C*C] (C = degree, 1..8); dy_dx [B, 3*C*C]. */
int ac_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D, uint32_t C,
                         int calc_grad_inputs, float *dy_dx, ac_stream_t stream);
int ac_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D, uint32_t C,
                          const float *dy_dx, float *grad_inputs, ac_stream_t stream);

/* ---- half / double tensors of the two encoders ------------------------------------------------
 * The reference dispatches both extensions over the dtype of their tensors (AT_DISPATCH_FLOATING_TYPES_AND_HALF: hashencoder.cu:352,391,
 * shencoder.cu:337,380; `CHECK_IS_FLOATING` accepts Float, Half, Double).  Same arguments as the fp32 entry points above with `dtype` in front and
 * untyped pointers: AC_DTYPE_F32 forwards to them; AC_DTYPE_F16 / AC_DTYPE_F64 are every tensor of the call (inputs, embeddings, outputs, dy_dx, grad,
 * grad_embeddings, grad_inputs) in that type.  Arithmetic: cell positions and interpolation weights in fp32 whatever the dtype (`float pos[D]`,
 * hashencoder.cu:125-133); half tensors are widened on load, accumulated in fp32 and rounded once on store (the table gradient: packed half2 atomics,
 * hashencoder.cu:293-299); double tensors accumulate in double.  Any other code: AC_ERR_BAD_ARG ("... must be a floating tensor"). */
#define AC_DTYPE_F32 0
#define AC_DTYPE_F16 1
#define AC_DTYPE_F64 2
int ac_hash_encode_forward_typed(int dtype, const void *inputs, const void *embeddings, const int32_t *offsets, const int32_t *offsets_host,
                                 void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs,
                                 void *dy_dx, ac_stream_t stream);
int ac_hash_encode_backward_typed(int dtype, const void *grad, const void *inputs, const void *embeddings, const int32_t *offsets,
                                  const int32_t *offsets_host, void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                  uint32_t H, int calc_grad_inputs, const void *dy_dx, void *grad_inputs, ac_stream_t stream);
int ac_sh_encode_forward_typed(int dtype, const void *inputs, void *outputs, uint32_t B, uint32_t D, uint32_t C, int calc_grad_inputs,
                               void *dy_dx, ac_stream_t stream);
int ac_sh_encode_backward_typed(int dtype, const void *grad, const void *inputs, uint32_t B, uint32_t D, uint32_t C, const void *dy_dx,
                                void *grad_inputs, ac_stream_t stream);

/* ---- raymarching operators (raymarching/src/raymarching.cu) ---------------------------------
 * Same arguments as the reference wrappers.  Slot reservation is deterministic: packed samples
 * are laid out in ray order (an exclusive prefix sum replaces the reference's atomicAdd), so
 * rays[N,3] = (ray id, offset, n_steps) is reproducible.  counter[2] (device) is incremented by
 * (total steps, N) as in the reference.  scratch: device int32 buffer of ac_march_rays_train_scratch(N)
 * elements (ABI 7: 36 N + 2 -- counts, offsets and, new, the positions of every ray's samples as a bit mask over its
 * step recurrence, which the counting pass records and the writing pass replays instead of walking the grid a second
 * time; before: 2N+2). */
size_t ac_march_rays_train_scratch(uint32_t N);
int ac_march_rays_train(const float *rays_o, const float *rays_d, const float *grid, float mean_density,
                        int iter_density, float bound, uint32_t N, uint32_t H, uint32_t M, float *xyzs,
                        float *dirs, float *deltas, int32_t *rays, int32_t *counter, uint32_t perturb,
                        int32_t *scratch, ac_stream_t stream);
int ac_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                    const int32_t *rays, float bound, uint32_t M, uint32_t N,
                                    float *weights_sum, float *image, ac_stream_t stream);
int ac_composite_rays_train_backward(const float *grad_weights_sum, const float *grad, const float *sigmas,
                                     const float *rgbs, const float *deltas, const int32_t *rays,
                                     const float *weights_sum, const float *image, float bound, uint32_t M,
                                     uint32_t N, float *grad_sigmas, float *grad_rgbs, ac_stream_t stream);
int ac_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                  const float *rays_o, const float *rays_d, float bound, uint32_t H, const float *grid,
                  float mean_density, const float *near, const float *far, float *xyzs, float *dirs,
                  float *deltas, uint32_t perturb, ac_stream_t stream);
int ac_composite_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, float *rays_t,
                      const float *sigmas, const float *rgbs, const float *normals, const float *deltas,
                      float *weights_sum, float *depth, float *image, float *normal_map, ac_stream_t stream);
/* order-preserving compaction; scratch: >= n_alive+2 int32 */
int ac_compact_rays(uint32_t n_alive, int32_t *rays_alive, const int32_t *rays_alive_old, float *rays_t,
                    const float *rays_t_old, int32_t *alive_counter, int32_t *scratch, ac_stream_t stream);

/* ---- fused Instant-NSR renderer --------------------------------------------------------------
 * One launch = NeRFRenderer.run for a batch of rays (models/instant_nsr.py:133-299,
 * render_can=True): cube near/far, uniform(+jitter) z, 4x NeuS up-sampling (up_sample,
 * sample_pdf, cat_z_vals), hash-grid SDF MLP, finite-difference normals, colour MLP, NeuS
 * alpha, transmittance scan and compositing, eikonal partials.  Default model only:
 * L=16, C=2, D=3 hash grid; SDF MLP 35-64-16; colour MLP 21-64-64-3. */
typedef struct ac_field {
    const float *table;       /* embeddings [offsets[16], 2]                           (device) */
    int32_t offsets[17];      /* level offsets, entries                                (host values) */
    float S;                  /* log2(per_level_scale)  (hashgrid.py:27)                         */
    uint32_t H;               /* base resolution                                                  */
    const float *W1, *b1;     /* effective sdf_net.0 weight [64,35] row-major, bias [64] (device) */
    const float *W2, *b2;     /* effective sdf_net.1 weight [16,64], bias [16]                    */
    const float *Wc1;         /* effective color_net.0 weight [64,21]                             */
    const float *Wc2;         /* effective color_net.1 weight [64,64]                             */
    const float *Wc3;         /* effective color_net.2 weight [3,64]                              */
    const void *prepared;     /* optional: AC_FIELD_PREPARED_BYTES of device memory filled by ac_field_prepare() for THESE
                                 parameters (the weights in the order the renderer keeps them in LDS: its workgroups then
                                 copy 52 KB linearly instead of re-deriving the layout from the row-major matrices, 512 times
                                 per launch).  NULL = derive it in every workgroup.  Must be re-prepared when a parameter changes. */
    const float *Wc1_sh;      /* optional (ABI 6): NeRFNetwork(use_viewdirs=True) -- the colour network's first layer reads
                                 h = cat[x, sh(d), n, geo_feat] (models/instant_nsr.py:565-569, 644-653; sh = degree-4 real spherical harmonics of the RAW
                                 ray direction, encoder/shencoder: 16 values).  Wc1_sh = the 16 columns of the effective color_net.0 weight that multiply
                                 sh(d), [64,16] row-major (columns 3..18 of the [64,37] matrix); Wc1 keeps the other 21 (x, n, geo_feat).  The view
                                 direction is constant along a ray: the renderer folds Wc1_sh sh(d) into a per-ray bias of layer 1 (fp32 fma chain over
                                 the 16 terms in order, then the 21 inputs as without view directions) -- no per-sample cost.  NULL = no view directions. */
} ac_field;
#define AC_FIELD_PREPARED_BYTES 98304
/* fills `prepared` (device, AC_FIELD_PREPARED_BYTES) from the other members of `field`; enqueue on `stream` before the launches that use it */
int ac_field_prepare(const ac_field *field, void *prepared, ac_stream_t stream);

typedef struct ac_render_opts {
    int32_t n_rays;
    int32_t num_steps;        /* coarse samples per ray: 16,32,48 or 64                           */
    int32_t upsample_steps;   /* multiple of 16, num_steps + upsample_steps <= 128                */
    float bound;
    float inv_s;              /* forward_variance() = exp(10*variance).clip(1e-6,1e6)             */
    float cos_anneal_ratio;
    float fd_eps;             /* 0.005 * (1 - normal_epsilon_ratio)                               */
    int32_t perturb;          /* 1: z += (noise-0.5)*sample_dist (training && perturb_overwrite)  */
    const float *inv_s_dev;   /* optional: inv_s as ONE float in device memory (then `inv_s` above is ignored): the trainable
                                 variance stays on the device, no host read-back per render                                */
    const float *near_m, *far_m; /* optional [N]: per-ray sampling range that replaces the cube's where finite (+-inf = keep the cube's):
                                 the mesh-guided range of run(render_can=True, verts=..., use_mesh_guide=True), instant_nsr.py:147-153,
                                 as produced by ac_mesh_near_far.  NULL = cube only.  (ac_render_rays_warped computes its own.)       */
    int32_t precision;        /* 0 = exact: every product an fp32 fma in a stated order (bit-identical to the CPU oracle);
                                 1 = fast: layer 1 of the six finite-difference evaluations of a sample as
                                     l1(x +- eps e_k) = l1(x) [exact fp32] + W1 (h(x +- eps e_k) - h(x)) + W1[:,k] (+-eps)
                                 with the middle product on the bf16 matrix pipe, both factors split into hi + lo bf16 (3 products, fp32
                                 accumulate): the correction term is ~1e-2 of l1, its 2^-16 relative error is below fp32 round-off of l1
                                 itself (normals within 6e-5 of the exact mode), and the colour network (21-64-64-3) in split bf16 as well
                                 (hi + lo, 3 products per layer; colours move by ~1e-6).  Sample positions (everything that feeds
                                 searchsorted / the sort) and the centre SDF evaluation are unaffected: z_vals, indices and sdf stay
                                 bit-identical.  The product's default is 0. */
    int32_t skip_masked;      /* posed-space rendering (ac_render_rays_warped) only: 1 = tiles of 16 samples that the warp masks out entirely
                                 (alpha * 0, instant_nsr.py:246-249) are not evaluated.  image, weights_sum, depth, normal_map and the per-sample
                                 weights / alpha are unchanged bit for bit (their contribution is exactly zero; the transmittance factor
                                 1 + 1e-7 of a masked sample is kept); the per-sample sdf / color / gradient of skipped samples are 0 and
                                 gradient_error covers the evaluated samples only.  0 = evaluate everything like the reference (default). */
    int32_t opacity_only;     /* 1 = the caller wants weights_sum / depth / normal_map / gradient_error only (the frozen avatar of the opacity loss,
                                 stylize.py:176-190, is rendered for its weight_sum alone): the colour network is not evaluated and `image` is the
                                 background blend of a black body.  Every other output is unchanged bit for bit (alpha does not depend on the
                                 colour).  0 = default. */
} ac_render_opts;

typedef struct ac_render_out {
    float *image;             /* [N,3]  rgb incl. background blend                                */
    float *weights_sum;       /* [N]                                                              */
    float *depth;             /* [N]                                                              */
    float *normal_map;        /* [N,3]                                                            */
    float *eik;               /* [N,2]  per-ray (sum relax*(|g|-1)^2, sum relax)                  */
    /* optional (NULL = not written); T = num_steps + upsample_steps */
    float *z_vals;            /* [N,T]                                                            */
    float *weights;           /* [N,T]                                                            */
    float *alpha;             /* [N,T]                                                            */
    float *color;             /* [N,T,3]                                                          */
    float *sdf;               /* [N,T]                                                            */
    float *gradient;          /* [N,T,3]                                                          */
    int32_t *ss_inds;         /* [N, upsample_steps/16, 16]  searchsorted indices of sample_pdf   */
    int32_t *sort_index;      /* [N, upsample_steps/16, 128] sort permutation of cat_z_vals, -1 pad */
    float *sdf_out16;         /* [N,T,16] forward_sdf at the mid points: sdf + the 15 geometry features   */
    float *pts;               /* [N,T,3]  the mid points themselves (clamped to the bound)                */
    float *feat7;             /* [N*T/16,14,64,4] the hash features of the 7 points of every sample's finite-difference stencil, in the kernel's own
                               * lane order: tile of 16 consecutive samples, then float k = 8 e + 2 j + c (evaluation e, level 4 j + g, channel c) of
                               * lane n + 16 g (sample n of the tile, level group g) at [k / 4][lane][k % 4] -- opaque to callers, produced by the
                               * forward and consumed by ac_render_core_backward, which would otherwise gather the table again (0.9 KB per
                               * sample; training renders only)                                                                                */
    float *eik_reduced;       /* optional [2] ([2][2] for ac_render_rays_pair: one pair per copy): (gradient_error, its denominator) =
                               * (sum relax * err / (sum relax + 1e-5), sum relax + 1e-5) over the batch (instant_nsr.py:270-272), reduced by the
                               * render launch itself in ac_eikonal_reduce2's fixed order (bit-identical to calling it on `eik` afterwards);
                               * not written for an empty batch (ABI version 4)                                                               */
} ac_render_out;

/* rays_o, rays_d [N,3]; bg [N,3] or NULL (= white, bg_color None -> 1); noise [N,num_steps] U[0,1)
 * (read iff opts->perturb); lin_z [num_steps] = torch.linspace(0,1,num_steps) and lin_u [16] =
 * torch.linspace(0.5/16, 1-0.5/16, 16) as DEVICE arrays made by the host (instant_nsr.py:155,:34). */
int ac_render_rays(const ac_field *field, const ac_render_opts *opts, const float *rays_o, const float *rays_d,
                   const float *bg, const float *noise, const float *lin_z, const float *lin_u,
                   const ac_render_out *out, ac_stream_t stream);

/* The same N = opts->n_rays rays rendered TWICE in one launch, with two draws of the jitter noise and two backgrounds: the two renders of
 * net_style in one stylisation step -- stylize.py:98-116 (render_val, no_grad) and :143-152 (the differentiable render of the same rays) are two
 * calls of NeRFRenderer.run (instant_nsr.py:133-299) in the reference.  The two copies of a ray are neighbours in the hand-out order and meet in
 * their XCD's L2 (-11 % against two launches on the stride-4 training view); every value is bit-identical to two ac_render_rays calls.
 * rays_o, rays_d [N,3]; bg2 [2,N,3] or NULL; noise2 [2,N,num_steps]; out: image, weights_sum, depth, normal_map, eik hold 2N rows (rows [0,N) = the
 * copy rendered with noise2[0] / bg2[0], rows [N,2N) = the other); the optional per-sample arrays are written for the SECOND copy only, N rows. */
int ac_render_rays_pair(const ac_field *field, const ac_render_opts *opts, const float *rays_o, const float *rays_d,
                        const float *bg2, const float *noise2, const float *lin_z, const float *lin_u,
                        const ac_render_out *out, ac_stream_t stream);

/* gradient_error = sum(relax*err) / (sum(relax) + 1e-5) over the per-ray partials, fixed order
 * (instant_nsr.py:270-272); result: 1 float (device) */
int ac_eikonal_reduce(const float *eik, int32_t n_rays, float *result, ac_stream_t stream);
/* same, result: 2 floats (device) = { gradient_error, sum(relax) + 1e-5 } -- the denominator is what the backward of the term needs */
int ac_eikonal_reduce2(const float *eik, int32_t n_rays, float *result2, ac_stream_t stream);

/* SDF-only / field queries used by density(), extract_geometry() and the unit tests:
 * out16 [B,16] = forward_sdf(x) (instant_nsr.py:627-642), x [B,3] in [-bound,bound] */
int ac_field_sdf(const ac_field *field, const float *x, uint32_t B, float bound, float *out16, ac_stream_t stream);
/* rgb [B,3] = forward_color(x, -, n, feat) (instant_nsr.py:644-663); sdfout [B,16] as returned above */
int ac_field_color(const ac_field *field, const float *x, const float *n, const float *sdfout, uint32_t B,
                   float *rgb, ac_stream_t stream);

/* ---- SMPL-guided animation path (render_warp.py, NeRFRenderer.run with render_can=False) ---------------------
 * replaces geometry_guided_near_far_torch (utils/ray_utils.py:277-294): per ray, over V vertex spheres of radius
 * geo_threshold; near/far [N], +inf / -inf where the ray misses every sphere (the caller substitutes the cube bounds,
 * models/instant_nsr.py:152-153). */
int ac_mesh_near_far(const float *rays_o, const float *rays_d, const float *verts, uint32_t N, uint32_t V, float geo_threshold,
                     float *near, float *far, ac_stream_t stream);

/* replaces warp_samples_to_canonical (utils/ray_utils.py:62-90; CPU libigl + numpy in the reference): for every sample
 * pts[P,3] (fp32): exact closest point on the triangle mesh (verts[V,3] fp32, faces[F,3] int32), mask = dist^2 < threshold
 * (the reference compares the SQUARED distance with DEFAULT_GEO_THRESH), barycentric blend of the per-vertex transforms
 * T[V',4,4] (fp64, V' >= max vertex index + 1), 4x4 inverse and application, all in fp64.
 * Outputs: can_pts [P,3] fp64 and/or can_pts_f32 [P,3] (at least one non-NULL); optional closest [P,3] fp64, dist2 [P] fp64,
 * face_id [P]; mask [P] uint8. */
int ac_warp_samples(const float *pts, const float *verts, const int32_t *faces, const double *T, uint32_t P, uint32_t V, uint32_t F,
                    double threshold, double *can_pts, float *can_pts_f32, double *closest, double *dist2, int32_t *face_id,
                    uint8_t *mask, ac_stream_t stream);

/* Accelerated form of ac_warp_samples: identical outputs bit for bit (same fp64 point-triangle arithmetic, ties -> lowest face id).
 * ac_warp_accel_build (once per frame / mesh) sorts the faces along a space-filling curve and cuts them into tiles of 32 with
 * bounding boxes; ac_warp_samples_accel then tests only the tiles whose box can contain the closest face (exact culling).
 * F <= 16384 (SMPL: 13776); ac_warp_accel_bytes returns 0 for meshes outside that range. */
size_t ac_warp_accel_bytes(uint32_t F);
/* Measurement accessors (bench.py's posed-frame roofline; no effect on results):
 * ac_warp_accel_work: the work the searches have done on this structure since its last build -- out[0] exact fp64 point-triangle tests,
 *   out[1] bounding-disc tests, out[2] sub-box tests, out[3] tile-box tests (counted per wave, one atomic each; waits for `stream`).
 * ac_debug_warped_phases(1): ac_render_rays_warped records HIP events at its phase boundaries; ac_debug_warped_phase_ms returns the five intervals of
 *   the last call in ms: near / far + coarse points + ray cull | first search | up-sampling pass | second search | final pass. */
int ac_warp_accel_work(const void *accel, unsigned long long out[4], ac_stream_t stream);
void ac_debug_warped_phases(int enable);
int ac_debug_warped_phase_ms(float out[5]);
int ac_warp_accel_build(const float *verts, const int32_t *faces, uint32_t V, uint32_t F, void *accel, size_t accel_bytes,
                        ac_stream_t stream);
int ac_warp_samples_accel(const float *pts, const float *verts, const int32_t *faces, const double *T, uint32_t P, uint32_t V, uint32_t F,
                          double threshold, const void *accel, double *can_pts, float *can_pts_f32, double *closest, double *dist2,
                          int32_t *face_id, uint8_t *mask, ac_stream_t stream);

/* ac_hash_encode_backward with caller scratch: for D = 3, C = 2 and levels of at most 2^19 entries (the default model) the table
 * gradient goes through the binned two-pass scatter (see ac_hash_stencil_backward) instead of one float atomic per corner and
 * channel; any other configuration, a NULL / too small scratch or calc_grad_inputs falls back to ac_hash_encode_backward.
 * ac_hash_encode_backward_scratch returns the bytes needed (0 = not coverable). */
size_t ac_hash_encode_backward_scratch(const int32_t *offsets_host, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t B);
int ac_hash_encode_backward_ws(const float *grad, const float *inputs, const float *embeddings, const int32_t *offsets,
                               const int32_t *offsets_host, float *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                               uint32_t H, int calc_grad_inputs, const float *dy_dx, float *grad_inputs, void *scratch, size_t scratch_bytes,
                               ac_stream_t stream);

/* ---- hash-grid encoder on the 7-point finite-difference stencil (training path)
 * One SDF query of the render core is 7 HashEncoder calls in the reference: forward_sdf at x (models/instant_nsr.py:627-642) and at
 * clamp(x +- eps e_k) (finite_difference_normals_approximator, :687-704), each through encoder/hashencoder/hashgrid.py:11-73.
 * These two entry points evaluate / back-propagate the seven points of every sample in one launch; the backward combines the
 * table gradients of the seven points in registers before its atomics (same sums as 7 x ac_hash_encode_backward).
 * x [B,3] world space, clamped to [-bound, bound]; point order x, +x, -x, +y, -y, +z, -z; outputs / grad [7, L, B, C], C = 2, D = 3;
 * grad_embeddings is accumulated into (zero it first).
 */
int ac_hash_stencil_forward(const float *x, const float *embeddings, const int32_t *offsets_host, float *outputs, uint32_t B,
                            uint32_t C, uint32_t L, float S, uint32_t H, float eps, float bound, ac_stream_t stream);
/* scratch (optional, NULL = none = every level through hardware float atomics):
 * ac_hash_stencil_backward_scratch(offsets_host, L, S, H, n_copies, B) bytes hold (a) n_copies >= 2 private copies of the small dense
 * levels, which otherwise take bursts of same-address atomics from neighbouring rays, and (b) for B > 0 the per-bucket record queues of
 * the hashed levels (binned two-pass scatter: records are queued by destination bucket with one global atomic per flush and bucket,
 * then every bucket is summed in LDS by one workgroup and added to the table without atomics); ~4 GB for B = 524288. */
size_t ac_hash_stencil_backward_scratch(const int32_t *offsets_host, uint32_t L, float S, uint32_t H, uint32_t n_copies, uint32_t B);
int ac_hash_stencil_backward(const float *grad, const float *x, const int32_t *offsets_host, float *grad_embeddings, uint32_t B,
                             uint32_t C, uint32_t L, float S, uint32_t H, float eps, float bound, void *scratch, size_t scratch_bytes,
                             ac_stream_t stream);

/* The stencil's gradient w.r.t. the sample POSITIONS through the encodings: what the reference obtains from dy_dx (kernel_grid's derivative branch,
 * hashencoder.cu:177-220, + kernel_input_backward, :311-337) when the encoder's input requires grad -- the curvature term's perturbed points
 * (models/instant_nsr.py:276-288).  gfeat [7, L, B, C] as ac_sdf_stencil_backward writes it; the corners are gathered again, nothing is stored.
 * gx_part [(L + 3) / 4][B][3]: one partial per group of four levels (the caller sums them: a fixed order); an offset coordinate passes through its
 * clamp to [-bound, bound] with torch.clamp's rule (inclusive), an out-of-range point contributes nothing (hashencoder.cu:95-119). */
int ac_hash_stencil_input_backward(const float *gfeat, const float *x, const float *embeddings, const int32_t *offsets_host, float *gx_part,
                                   uint32_t B, uint32_t C, uint32_t L, float S, uint32_t H, float eps, float bound, ac_stream_t stream);

/* The sampling stage of run() alone: coarse z (+ jitter), coarse SDF, 4x NeuS up-sampling -> z_vals [N, num_steps + upsample_steps]
 * (models/instant_nsr.py:155-184; the reference runs it under no_grad before the differentiable render core).  Same z as
 * ac_render_rays bit for bit, at a fraction of its cost. */
int ac_sample_rays(const ac_field *field, const ac_render_opts *opts, const float *rays_o, const float *rays_d, const float *noise,
                   const float *lin_z, const float *lin_u, float *z_vals, ac_stream_t stream);

/* ---- fused SDF query of the differentiable render core (training path): forward_sdf(x) (models/instant_nsr.py:627-642) and
 * finite_difference_normals_approximator(x) (:687-704), i.e. 7 hash-encoder + MLP evaluations per sample, in one kernel each way.
 * forward : x [B,3] (clamped to the bound) -> out16 [B,16] = forward_sdf(x), grad [B,3] = the finite-difference gradient (eps > 0);
 *           bit-identical to the values ac_render_rays computes internally.
 * backward: (x, g_out16 [B,16], g_grad [B,3]) -> gfeat [7,16,B,2]: the gradient w.r.t. the hash features of the 7 stencil points in
 *           the layout ac_hash_stencil_backward consumes (the table scatter stays there), and gparams [3344] =
 *           dW1 [64][36] (column 35 = db1, columns 0..34 = the 35 input columns of sdf_net.0) | dW2 [16][64] | db2 [16]
 *           w.r.t. the EFFECTIVE (weight-normed) matrices of `field`.  The forward is recomputed per tile; no activations are stored.
 *           scratch: ac_sdf_stencil_backward_scratch(B) bytes (per-wave partial sums, reduced deterministically). */
int ac_sdf_stencil_forward(const ac_field *field, const float *x, uint32_t B, float bound, float eps, float *out16, float *grad,
                           ac_stream_t stream);
size_t ac_sdf_stencil_backward_scratch(uint32_t B);
int ac_sdf_stencil_backward(const ac_field *field, const float *x, const float *g_out16, const float *g_grad, uint32_t B, float bound,
                            float eps, float *gfeat, float *gparams, void *scratch, size_t scratch_bytes, ac_stream_t stream);

/* The same backward for positions that carry a gradient themselves (x.requires_grad in the reference: the curvature term's perturbed points,
 * models/instant_nsr.py:276-288): additionally g_x [B,3] = the share of d loss / d x that enters through the MLP's own xyz inputs (include_input,
 * :632-633; the offset coordinate of an offset evaluation through its clamp, :690-702).  The share through the encodings: ac_hash_stencil_input_backward
 * on the gfeat this call wrote; the caller adds the two. */
int ac_sdf_stencil_backward_inputs(const ac_field *field, const float *x, const float *g_out16, const float *g_grad, uint32_t B, float bound,
                                   float eps, float *gfeat, float *gparams, float *g_x, void *scratch, size_t scratch_bytes, ac_stream_t stream);

/* Liveness of the renderer's segment hand-off (a ray's final pass is cut into segments that different waves may take; a taker waits, bounded to ~1 s,
 * for the previous segment's state).  A timed-out hand-off poisons the pixel with NaN AND is counted here, per (device, stream), over the life of the
 * launch scratch of that stream: 0 on a healthy run.  Waits for `stream`.  The render launches tag their scratch with a host-side generation number
 * baked into the kernel arguments: they must NOT be captured into a hipGraph (a replay would match the flags of its previous run). */
int ac_render_handoff_timeouts(ac_stream_t stream, uint32_t *count);

/* ---- field evaluation on PACKED samples: the body of NeRFRenderer.run_cuda between raymarching.march_rays[_train] and
 * raymarching.composite_rays[_train].  The reference dispatches cuda_ray=True renders to `run_cuda` (models/instant_nsr.py:362-363) but never defines it
 * (SURVEY 0.1); the per-sample arithmetic is run()'s render core (:205-243) with the marcher's step as the section length:
 *   p = clamp(xyz, +-bound); out16 = forward_sdf(p) (:627-642); g = finite-difference gradient (:687-704, eps); n = g / (1e-5 + |g|);
 *   rgb = forward_color(p, n, out16[1:]) (:644-663); alpha = NeuS alpha of (out16[0], dot(dir, n), delta, inv_s, cos_anneal_ratio) (:219-243), clipped to [0, 1].
 * xyzs, dirs [M,3]; deltas [M * delta_stride] (column 0 is used: march_rays_train writes stride 1, march_rays stride 2).
 * inv_s_dev (optional) = forward_variance() as one float on the device, else inv_s.  sdf [M], gradient [M,3] optional.
 * Bit-identical, per sample, to what ac_render_rays computes for the same point / direction / section length (same tile code). */
int ac_field_samples(const ac_field *field, const float *xyzs, const float *dirs, const float *deltas, uint32_t delta_stride, uint32_t M,
                     float bound, float eps, float inv_s, const float *inv_s_dev, float cos_anneal_ratio, float *alpha, float *rgb, float *normal,
                     float *sdf, float *gradient, ac_stream_t stream);

/* The inference form of run_cuda as ONE launch: per ray, march through the occupancy grid (the arithmetic of ac_march_rays), evaluate the field on the
 * samples (the arithmetic of ac_field_samples, samples of a wave's 64 rays packed into tiles of 16) and composite them in order (the arithmetic of
 * ac_composite_rays: T = 1 - weights_sum, a ray stops at T < 1e-2 or at `far`) -- what the reference-shaped loop compact_rays / march_rays / field /
 * composite_rays computes in rounds with one host read-back each, bit for bit, without the rounds.  near / far = near_far_from_bound(type = 'cube')
 * (models/instant_nsr.py:58-77).  Outputs are the accumulators as composite_rays leaves them: weights_sum [N], depth [N] (sum of w * t: the caller normalises,
 * like run_cuda's last lines), image [N,3] (no background), normal_map [N,3]; n_samples (optional, device, [1]): samples evaluated, accumulated.
 * max_steps (ABI 6): a ray stops after that many samples (0 = only at `far` / T < 1e-2).  The loop of rounds stops at the first round that brings its
 * step count to >= max_steps -- after max_steps .. max_steps + 7 samples per ray, depending on how many rays were alive in its last rounds; this entry
 * stops at exactly max_steps.  The two forms are bit-identical for every ray that needs fewer than max_steps samples (all of them at the default 1024
 * unless a ray crosses > 1024 occupied cells: ~1600 steps of dt_min fit the diagonal of a bound-1.6 cube). */
int ac_render_rays_occupancy(const ac_field *field, const float *rays_o, const float *rays_d, uint32_t N, const float *grid, uint32_t H,
                             float mean_density, float bound, float eps, float inv_s, const float *inv_s_dev, float cos_anneal_ratio,
                             float *weights_sum, float *depth, float *image, float *normal_map, uint32_t *n_samples, uint32_t max_steps,
                             ac_stream_t stream);

/* The same inference render, the same bits, as rounds of march | field | composite INSIDE one launch with grid barriers between the phases (ABI 7): every lane
 * walks a ray, the samples of a round are evaluated on tiles dealt to all waves, and the rays that go on (not at `far`, T >= 1e-2, below max_steps) form the
 * next round's list -- the reference's loop without its host read-backs; faster than ac_render_rays_occupancy on whole views, where that kernel keeps a
 * quarter of a wave's lanes walking and every wave evaluating its own tiles one after the other.  n_step = 16 samples per ray and round (AC_OCC_NLOG = 1 .. 6
 * overrides its log2); results do not depend on it.  scratch: ac_render_rays_occupancy_phased_scratch(N) bytes (64 B per ray and round sample: 67 MB for a
 * 256 x 256 view), ZERO-FILLED by the caller before its first use, re-armed by every call, one buffer per stream; a call with fewer rays may reuse it.
 * A launch needs every workgroup resident: the grid is min(what the rays want, hipOccupancyMaxActiveBlocksPerMultiprocessor x compute units) and goes
 * through hipLaunchCooperativeKernel where the device supports it (AC_COOP_LAUNCH=0: a plain launch of the same grid) -- a grid that cannot be co-resident is
 * refused AT LAUNCH (AC_ERR_LAUNCH).  What remains: a foreign kernel holding compute units for longer than a barrier's bounded spin (two seconds;
 * ac_set_occupancy_barrier_ms / AC_OCC_BARRIER_MS override) -- then the phased kernel gives up, counts itself in the scratch's 32-bit word 8 (sticky) and sets
 * the launch's verdict word 9; the call has ALREADY queued the barrier-free kernel of ac_render_rays_occupancy behind it, conditional on that word: it
 * renders every ray again (the same bits) -- a few microseconds when not needed, no host round trip, and the outputs are complete whenever the stream
 * reaches the caller's next operation.  The reference's loop (raymarching/raymarching.py:136-188) never returns partial results either. */
size_t ac_render_rays_occupancy_phased_scratch(uint32_t N);
/* bound of a grid barrier's spin in the phased launches (inference and training form), milliseconds; 0 = back to the default (2000, or AC_OCC_BARRIER_MS).
 * Returns the value in force before the call.  Process-wide. */
uint32_t ac_set_occupancy_barrier_ms(uint32_t ms);
int ac_render_rays_occupancy_phased(const ac_field *field, const float *rays_o, const float *rays_d, uint32_t N, const float *grid, uint32_t H,
                                    float mean_density, float bound, float eps, float inv_s, const float *inv_s_dev, float cos_anneal_ratio,
                                    float *weights_sum, float *depth, float *image, float *normal_map, uint32_t *n_samples, uint32_t max_steps,
                                    void *scratch, size_t scratch_bytes, ac_stream_t stream);

/* The training form of run_cuda WITHOUT autograd as ONE launch (ABI 7) -- stylize.py's render_val of a cuda_ray network, which never leaves train() mode
 * (stylize.py:46-215 never calls eval()): what ac_march_rays_train (count, scan, write) -> ac_field_samples -> ac_composite_rays_train_forward twice (colour,
 * normal) -> the eikonal term and the background in torch compute, as four phases of one persistent launch separated by grid barriers: the walk of
 * kernel_march_rays_train per ray (raymarching.cu:56-222; perturb: t0 = near + dt_min * pcg32(ray).next_float()), counting and recording the samples'
 * positions; offsets in ray order, the reference's budget rule and the packed samples (replayed from the records, no second walk); the field on tiles of
 * 16 packed samples dealt to all waves (the arithmetic of ac_field_samples with the marcher's step as the section length); kernel_composite_rays_train_forward's
 * loop per ray (:232-301: T < 1e-4 ends the sums) for image and normal map on the same weights.  A BUDGETED call only -- the packed layout lives in the scratch:
 *   capacity            M of ac_march_rays_train (> 0): a ray whose samples would end at or beyond it (offset + count >= M, offsets in ray order from
 *                       counter[0]) is not marched (:133);
 *   composite_capacity  M of ac_composite_rays_train_forward (> 0; the same number in run_cuda): such a ray gets weights_sum = image = 0 (:249);
 *   counter             optional [2] int32 (device): [0] += samples of all rays, [1] += N, like the stand-alone marcher.
 * bg_mode: image += (1 - weights_sum) * bg (run_cuda's last line; torch's three operations in torch's order): 0 none, 1 the scalar bg_value, 2 bg[3],
 * 3 bg[N][3].  gradient_error [1] (device): sum(relax * (|gradient| - 1)^2) / (sum(relax) + 1e-5) over the marched samples, relax = |x| < 1.2
 * (models/instant_nsr.py:266-272), summed in double in a fixed order (run to run identical; torch's fp32 tree differs from it by ~1e-6 relative);
 * NaN if a grid barrier timed out (it cannot with one workgroup per compute unit; the bound exists so that a mis-sized launch fails instead of hanging).
 * weights_sum / image / normal_map are the bits of the chain of operators.  scratch: ac_render_rays_occupancy_train_scratch(N, capacity) bytes, ZERO-FILLED
 * by the caller before its first use (every call leaves it re-armed for calls with the SAME N and capacity: the layout depends on both), one buffer per stream. */
size_t ac_render_rays_occupancy_train_scratch(uint32_t N, uint32_t capacity);
int ac_render_rays_occupancy_train(const ac_field *field, const float *rays_o, const float *rays_d, uint32_t N, const float *grid, uint32_t H,
                                   float mean_density, float bound, float eps, float inv_s, const float *inv_s_dev, float cos_anneal_ratio,
                                   uint32_t perturb, uint32_t capacity, uint32_t composite_capacity, int32_t *counter, const float *bg,
                                   uint32_t bg_mode, float bg_value, float *weights_sum, float *image, float *normal_map, float *gradient_error,
                                   void *scratch, size_t scratch_bytes, ac_stream_t stream);

/* ---- geometry of the learned surface: mesh export and the marcher's density grid ----------------------------------------
 * ac_field_sdf_grid replaces extract_fields (models/instant_nsr.py:728-745): forward_sdf(x)[0] (:627-642) on the grid axis_x x axis_y x axis_z
 * (three DEVICE arrays of nx / ny / nz coordinates: what torch.linspace(bound_min, bound_max, resolution) holds -- the kernel forms the points itself,
 * no meshgrid / cat tensors, no 256^3 blocks, nothing goes to the host); volume [nx,ny,nz] (z fastest, like the reference's `u`), negate != 0 stores
 * -sdf (the `u = -1.0 * u` of extract_geometry :752-753).  Values are bit-identical to ac_field_sdf at the same points. */
int ac_field_sdf_grid(const ac_field *field, const float *axis_x, const float *axis_y, const float *axis_z, uint32_t nx, uint32_t ny, uint32_t nz,
                      float bound, int negate, float *volume, ac_stream_t stream);
/* Marching cubes on a device volume: replaces mcubes.marching_cubes(u, threshold) (models/instant_nsr.py:757; PyMCubes is a third-party package, not
 * vendored in the reference and absent from this image: the algorithm is restated from its definition -- Lorensen & Cline's 256 cases, table generated
 * by tools/gen_mc_table.py; corner flagged <=> u <= iso; one vertex per sign-changing grid edge at the linear zero crossing, formed in double; shared
 * by every triangle that touches the edge; triangle normals point towards u <= iso).  Two calls because the caller owns the output buffers:
 *   ac_marching_cubes_count: classify + scan; counts (DEVICE, [2]) = { vertices, triangles }.  scratch: ac_marching_cubes_scratch(nx, ny, nz) bytes
 *                            (5 bytes per grid point), kept unchanged until the emit call.
 *   ac_marching_cubes_emit : vertices [n_vertices,3] double = index / den * span + lo per axis (the reference's scaling to world units, :760-762, in its
 *                            order of operations; den = resolution - 1, span / lo HOST arrays [3]; den = 1, span = 1, lo = 0 gives PyMCubes' index space),
 *                            triangles [n_triangles,3] int32.  Deterministic order: vertices by owning grid point (linear index, z fastest) then axis
 *                            x, y, z; triangles by cell (linear index) then table position.
 * nx, ny, nz >= 2 and fewer than 2^31 grid points. */
size_t ac_marching_cubes_scratch(uint32_t nx, uint32_t ny, uint32_t nz);
int ac_marching_cubes_count(const float *volume, uint32_t nx, uint32_t ny, uint32_t nz, float iso, void *scratch, size_t scratch_bytes,
                            uint32_t *counts, ac_stream_t stream);
int ac_marching_cubes_emit(const float *volume, uint32_t nx, uint32_t ny, uint32_t nz, float iso, void *scratch, size_t scratch_bytes,
                           double den, const double span[3], const double lo[3], double *vertices, uint32_t n_vertices, int32_t *triangles,
                           uint32_t n_triangles, ac_stream_t stream);
/* ac_density_grid_update replaces the grid update of NeRFRenderer.update_extra_state (models/instant_nsr.py:303-346) in two launches (round 6: the densities on
 * ac_field_sdf_grid's x-tiles into the scratch, then one pooling / merging pass; round 5's single launch evaluated a halo per brick and was 3 x slower): forward_sdf on
 * the H^3 grid axis x axis x axis (axis [H], device: torch.linspace(-bound, bound, H)) -> density = inv_s e^(-inv_s |sdf|) / (1 + e^(-inv_s |sdf|)) in the
 * reference's two branches (:331-337; inv_s = 512) -> zero pad by one at the far ends + 2x2x2 max pool, stride 1 (:341-342) -> grid = max(grid * decay,
 * new) IN PLACE (:345) -> mean_out (DEVICE, 1 double) = mean(grid) (:346; accumulated in double, fixed order).  grid [H,H,H].
 * mean: in double where torch.mean reduces in fp32 -- equal to ~1e-7 relative, not bit for bit (a marcher threshold: values it is compared with differ by orders of magnitude).
 * scratch: ac_density_grid_update_scratch(H) bytes (4 H^3 + a few KB), ZEROED once by the caller before its first use (the launch re-arms it).  2 <= H <= 1024. */
size_t ac_density_grid_update_scratch(uint32_t H);
int ac_density_grid_update(const ac_field *field, const float *axis, uint32_t H, float bound, float inv_s, float decay, float *grid,
                           double *mean_out, void *scratch, size_t scratch_bytes, ac_stream_t stream);

/* ---- shading of PACKED samples under autograd (ABI 9, round 6): what NeRFRenderer.run_cuda's train() branch computes between the fused SDF query
 * (ac_sdf_stencil_forward: sdf16 [M,16], gradient [M,3]) and the packed compositor (ac_composite_rays_train_*), one launch each way instead of ~40 torch kernels:
 *   normal = g / (1e-5 + |g|); alpha = the cos-annealed NeuS alpha of ac_field_samples with the marcher's step (deltas, stride 1 | 2) as section length;
 *   eik [M,2] = (relax (|g| - 1)^2, relax), relax = [|xyz| < 1.2][row < *n_valid] -- the two sums of gradient_error (models/instant_nsr.py:266-272).
 * backward: g_alpha [M], g_normal [M,3] (the colour network's), g_eik [M,2] (column 0 is read), any of them NULL = zero -> g_sdf16 [M,16] (column 0 WRITTEN, the
 * caller zero-fills the rest), g_gradient [M,3], g_inv_s_rows [M] (d / d inv_s = their sum).  n_valid: a DEVICE int32. */
int ac_packed_shading_forward(const float *sdf16, const float *gradient, const float *xyzs, const float *dirs, const float *deltas, uint32_t delta_stride,
                              uint32_t M, const int32_t *n_valid, float inv_s, const float *inv_s_dev, float cos_anneal_ratio, float *alpha, float *normal,
                              float *eik, ac_stream_t stream);
int ac_packed_shading_backward(const float *sdf16, const float *gradient, const float *xyzs, const float *dirs, const float *deltas, uint32_t delta_stride,
                               uint32_t M, const int32_t *n_valid, float inv_s, const float *inv_s_dev, float cos_anneal_ratio, const float *g_alpha,
                               const float *g_normal, const float *g_eik, float *g_sdf16, float *g_gradient, float *g_inv_s_rows, ac_stream_t stream);

/* use_viewdirs: the per-ray layer-1 bias of the colour network the renderer forms in its prologue, bias[r][u] = fma chain over j = 0..15 of
 * Wc1_sh[u][j] sh_j(rays_d[r]) (sh = the degree-4 values of ac_sh_encode_forward on the raw direction), as its own launch: bias [N,64], sh [N,16] or NULL.
 * What ac_render_core_backward needs beside the forward's outputs (ac_core_saved.sh_bias) and what turns its g_sh_tiles into d Wc1_sh. */
int ac_sh_bias(const ac_field *field, const float *rays_d, uint32_t N, float *bias, float *sh, ac_stream_t stream);
/* ac_field_color with the view direction of every point: dirs [B,3] (required when field->Wc1_sh is set, ignored otherwise) */
int ac_field_color_dirs(const ac_field *field, const float *x, const float *dirs, const float *n, const float *sdfout, uint32_t B, float *rgb,
                        ac_stream_t stream);

/* ---- colour MLP of the render core (training path): forward_color (models/instant_nsr.py:644-663, use_viewdirs = False)
 * rgb = sigmoid(Wc3 relu(Wc2 relu(Wc1 [x, normal, feat]))) with feat = sdf16[:, 1:16].
 * forward : same values as ac_field_color / ac_render_rays.   backward: recomputes the forward per tile of 16 samples and returns
 * g_normal [B,3], g_sdf16 [B,16] (column 0 = 0: the sdf itself is not an input of the colour net) and
 * gparams [7168] = dWc1 [64][32] (columns 0..20 = x(3), normal(3), feat(15)) | dWc2 [64][64] | dWc3 [16][64] (rows 0..2)
 * w.r.t. the EFFECTIVE matrices of `field`.  scratch: ac_color_backward_scratch(B) bytes. */
int ac_color_forward(const ac_field *field, const float *x, const float *normal, const float *sdf16, uint32_t B, float *rgb, ac_stream_t stream);
size_t ac_color_backward_scratch(uint32_t B);
int ac_color_backward(const ac_field *field, const float *x, const float *normal, const float *sdf16, const float *g_rgb, uint32_t B,
                      float *g_normal, float *g_sdf16, float *gparams, void *scratch, size_t scratch_bytes, ac_stream_t stream);

/* ---- NeuS alpha + compositing of the render core (training path): models/instant_nsr.py:219-263,290-299 for rays in canonical space.
 * Inputs per sample: z_vals, sdf, normal (= gradient / (1e-5 + |gradient|)), colour, [N,T] / [N,T,3]; per ray: origin, direction
 * (near/far of the cube and sample_dist = (far - near) / num_steps are recomputed), optional background [N,3] (NULL = white).
 * forward : image [N,3], weights_sum [N], depth [N], normal_map [N,3], weights [N,T], alpha [N,T] -- the arithmetic of ac_render_rays.
 * backward: (d image, d weights_sum, d depth, d normal_map) -> d sdf [N,T], d normal [N,T,3], d colour [N,T,3] and the per-ray partial
 *           sums of d inv_s [N] (the caller adds them up).  T = num_steps + upsample_steps, a multiple of 16, <= 128. */
int ac_composite_forward(const float *rays_o, const float *rays_d, const float *z_vals, const float *sdf, const float *normal, const float *color,
                         const float *bg, int32_t n_rays, int32_t num_steps, int32_t T, float bound, float inv_s, float cos_anneal_ratio,
                         float *image, float *weights_sum, float *depth, float *normal_map, float *weights, float *alpha, ac_stream_t stream);
int ac_composite_backward(const float *rays_o, const float *rays_d, const float *z_vals, const float *sdf, const float *normal, const float *color,
                          const float *bg, int32_t n_rays, int32_t num_steps, int32_t T, float bound, float inv_s, float cos_anneal_ratio,
                          const float *g_image, const float *g_weights_sum, const float *g_depth, const float *g_normal_map,
                          float *g_sdf, float *g_normal, float *g_color, float *g_inv_s_per_ray, ac_stream_t stream);

/* ---- the whole differentiable render core (models/instant_nsr.py:190-299) backward in one call (training path).
 * The forward is ac_render_rays itself with the per-sample outputs kept (z_vals, pts, sdf, sdf_out16, gradient, color and the
 * eikonal denominator of ac_eikonal_reduce2): the training render and the inference render are the same launch, bit for bit.
 * The backward chains, on `stream`: normals from the finite-difference gradients -> NeuS alpha / compositing backward ->
 * colour MLP backward -> normalisation + eikonal backward -> fused SDF-query backward -> table-gradient scatter.
 *   upstream: d image [N,3], d weights_sum [N], d depth [N], d normal_map [N,3] (any may be NULL = 0), d gradient_error (1 float, device, or NULL)
 *   results : g_table [offsets[16],2] ACCUMULATED into (like hash_encode_backward); g_sdf_params [3344] and g_color_params [7168]
 *             (layouts of ac_sdf_stencil_backward / ac_color_backward, w.r.t. the EFFECTIVE matrices); g_inv_s_per_ray [N].
 *   scratch : ac_render_core_backward_scratch(field, n_rays, T) bytes (~9 KB per sample, dominated by the scatter queues). */
typedef struct ac_core_saved {
    const float *z_vals, *pts, *sdf, *sdf_out16, *gradient, *color;      /* per-sample outputs of the forward launch */
    const float *eik_den;                                                 /* result2[1] of ac_eikonal_reduce2          */
    const float *feat7;                                                   /* ac_render_out.feat7 of the forward, or NULL: gather again */
    const uint8_t *mask;                                                  /* posed space only (the forward was ac_render_rays_warped): the warp's
                                                                           * alpha mask [N,T] (instant_nsr.py:246-249), with opts->near_m / far_m =
                                                                           * the forward's mesh-guided range and `pts` = the warped points the
                                                                           * forward kept; NULL = canonical space (ABI version 4)                  */
    const float *sh_bias;                                                 /* a field with view directions (ac_field.Wc1_sh) only: ac_sh_bias of the
                                                                           * forward's rays, [N,64]; NULL otherwise (ABI version 6)                  */
} ac_core_saved;
typedef struct ac_core_upstream {
    const float *g_image, *g_weights_sum, *g_depth, *g_normal_map, *g_eik;
    /* ABI 9 (round 6): SEVERAL patches of a view in one backward (stylize.py:143-199 back-propagates a 256 x 256 view as 16 patches of 4096 rays whose
     * gradients add up before the one optimizer.step()).  The eikonal term is a RATIO per patch (instant_nsr.py:266-272), so with eik_group_rays > 0 the
     * rays [k * eik_group_rays, (k + 1) * eik_group_rays) form patch k, whose upstream is g_eik[k] and whose denominator is
     * saved->eik_den[k * eik_den_stride]; 0 = the whole launch is one patch (g_eik[0], eik_den[0]). */
    int32_t eik_group_rays, eik_den_stride;
} ac_core_upstream;
typedef struct ac_core_grads {
    float *g_table, *g_sdf_params, *g_color_params, *g_inv_s_per_ray;
    /* data-parallel training (stylize.py under torch.distributed: one all-reduce of the flat gradient per step): with side_stream != NULL the table
     * scatter finishes levels >= split_level first and makes side_stream wait for exactly that; the caller enqueues the all-reduce of that slice of
     * g_table (entries [offsets[split_level], offsets[16])) on side_stream right after this call returns, and it overlaps the rest of the backward.
     * The slice is final at that point; results are bit-identical to the unsplit call.  NULL / 0 = off (ABI version 4). */
    ac_stream_t side_stream;
    int32_t split_level;
    int32_t reserved;
    float *g_sh_tiles;        /* a field with view directions only: [N * T / 16, 64], per tile of 16 samples the sum of d (layer-1 pre-activation of the colour
                               * network) = the gradient of that tile's share of its ray's view-direction bias.  d Wc1_sh [64,16] = sum over rays r of
                               * (sum of the ray's T / 16 rows) (x) sh(rays_d[r]) (sh from ac_sh_bias): the caller's [N,64]^T x [N,16] product (ABI version 6) */
} ac_core_grads;
size_t ac_render_core_backward_scratch(const ac_field *field, int32_t n_rays, int32_t T);
int ac_render_core_backward(const ac_field *field, const ac_render_opts *opts, const float *rays_o, const float *rays_d, const float *bg,
                            const ac_core_saved *saved, const ac_core_upstream *upstream, const ac_core_grads *grads,
                            void *scratch, size_t scratch_bytes, ac_stream_t stream);

/* ---- the per-step pieces around the training render (stylize.py:95-199), so that a stylisation step runs no framework kernel between the
 * guidance gradient and the optimizer.
 * ac_weight_norm_forward: w[r, :] = v[r, :] * g[r] / ||v[r, :]|| for every layer in ONE launch -- torch.nn.utils.weight_norm(dim = 0) of the
 *   sdf_net / color_net layers (models/instant_nsr.py:557-590).  w may be a strided view (row stride w_stride >= cols). */
#define AC_WN_MAX_LAYERS 8
typedef struct ac_wn_layer { const float *v, *g; float *w; uint32_t rows, cols, w_stride, reserved_; } ac_wn_layer;
int ac_weight_norm_forward(const ac_wn_layer *layers, uint32_t n_layers, ac_stream_t stream);
/* ac_param_grads: gradients w.r.t. the EFFECTIVE matrices (g_sdf_params / g_color_params of ac_render_core_backward) -> the parameters' own
 *   gradient buffers, ACCUMULATED (+=, like autograd's), one launch for up to AC_PG_MAX_ENTRIES entries:
 *   AC_PG_WEIGHT_NORM: src = d W [rows, cols] (row stride src_stride), v, g -> dst = d weight_v [rows, cols], dst2 = d weight_g [rows]
 *   AC_PG_ADD        : dst[i] += src[i * src_stride], i < rows                              (biases)
 *   AC_PG_VARIANCE   : dst[0] += 10 * inv_s * sum_{i < rows} src[i] if 1e-6 < inv_s < 1e6;  g = device pointer to inv_s = exp(10 variance)
 *                      (SingleVarianceNetwork + clip, models/instant_nsr.py:35-45, 666-667; src = g_inv_s_per_ray) */
#define AC_PG_MAX_ENTRIES 12
enum { AC_PG_WEIGHT_NORM = 0, AC_PG_ADD = 1, AC_PG_VARIANCE = 2 };
/* optimizer.step() of the stylisation / reconstruction step (stylize.py:199 over torch.optim.Adam, :355-363; no amsgrad, weight decay or maximize): every
 * parameter tensor of the network in ONE launch, m = m + (1 - beta1)(g - m), v = beta2 v + (1 - beta2) g g, p -= step_size m / (sqrt(v) / sqrt(bc2) + eps);
 * step_size = lr / (1 - beta1^t), bias_correction2_sqrt = sqrt(1 - beta2^t), 1 - beta1 and 1 - beta2 are formed by the caller in double.  zero_grad != 0 also clears the gradients
 * (the next step's optimizer.zero_grad(), stylize.py:143) while they are in registers.  HBM-bound: 28 (32) bytes per element. */
#define AC_ADAM_MAX_TENSORS 16
typedef struct ac_adam_entry { float *param, *grad, *exp_avg, *exp_avg_sq; uint64_t n; } ac_adam_entry;
int ac_adam_step(const ac_adam_entry *tensors, uint32_t n_tensors, float step_size, float beta1, float one_minus_beta1, float beta2,
                 float one_minus_beta2, float eps, float bias_correction2_sqrt, int zero_grad, ac_stream_t stream);

/* forward_variance() of the model (models/instant_nsr.py:35-45, 666-667): inv_s[0] = clip(exp(10 * variance[0]), 1e-6, 1e6), the value
 * ac_render_opts.inv_s_dev points at -- one launch instead of torch's five (ones, mul, exp, mul, clip); same bits (the device library's expf). */
int ac_variance_forward(const float *variance, float *inv_s, ac_stream_t stream);

typedef struct ac_pg_entry { const float *src, *v, *g; float *dst, *dst2; uint32_t rows, cols, src_stride; int32_t kind; } ac_pg_entry;
int ac_param_grads(const ac_pg_entry *entries, uint32_t n_entries, ac_stream_t stream);
/* ac_sds_upstream: the opacity term of the stylisation loss (stylize.py:183-193): loss = sum_i smooth_l1(clamp(ws_i, 0, 1), clamp(ws_gt_i, 0, 1)) *
 *   scale (scale = 1e5 / n_rays for F.smooth_l1_loss(...) * 1e5); g_weights_sum [n_rays] = d loss / d ws (or NULL), loss = 1 float (device, or NULL) */
int ac_sds_upstream(const float *weights_sum, const float *weights_sum_gt, uint32_t n_rays, float scale, float *g_weights_sum, float *loss,
                    ac_stream_t stream);

/* ---- posed-space rendering: NeRFRenderer.run(render_can=False, verts, faces, Ts, use_mesh_guide)
 * models/instant_nsr.py:147-172 (mesh-guided near/far, warp of the coarse samples), :198-203 (warp of the mid points),
 * :246-249 (alpha mask).  The reference moves the samples to the CPU for libigl twice per batch; here the whole sequence
 * (near/far -> coarse points -> warp -> coarse sdf + up-sampling -> warp -> render core) is enqueued on `stream`.
 * As in the reference, the up-sampling queries the field at the UNwarped new samples (cat_z_vals, :464-469). */
typedef struct ac_warp_mesh {
    const float *verts;        /* [V,3]  posed SMPL vertices of the frame */
    const int32_t *faces;      /* [F,3] */
    const double *T;           /* [V',4,4] rest->scene transform per vertex (V' >= V; render_warp.py passes V+24) */
    uint32_t V, F;
    double threshold;          /* mask: squared distance < threshold   (DEFAULT_GEO_THRESH = 0.05) */
    float geo_threshold;       /* radius of the vertex spheres of the near/far guide (DEFAULT_GEO_THRESH) */
    int32_t use_mesh_guide;
    const void *accel;         /* ac_warp_accel_build output for (verts, faces), or NULL = brute-force search */
    /* ABI 9 (round 6) -- temporal seeds of the closest-face searches, or NULL: [n_rays][seed_stride] int32 owned by the caller and kept ACROSS FRAMES, row r =
     * ray r of the launch, columns [0, num_steps) the faces the first search found for the ray's coarse samples, [num_steps, num_steps + T) those of the
     * second search; -1 = none (fill a new buffer with -1).  A search starts every sample from the exact distance of the face stored for its (ray, slot)
     * -- the previous frame's answer, a real face of THIS frame's mesh and therefore a valid bound -- and stores what it finds.  Results are unchanged
     * bit for bit; only the culling gets tighter (an animation's body moves little between frames).  seed_stride >= num_steps + T.  Needs accel. */
    int32_t *seed_faces;
    uint32_t seed_stride;
} ac_warp_mesh;

/* bytes of device scratch ac_render_rays_warped needs for n_rays rays of T = num_steps + upsample_steps samples; if offs is not
 * NULL it receives the byte offsets of {near_m[N] f32, far_m[N] f32, posed pts[N,T,3] f32, canonical pts[N,T,3] f32,
 * mask[N,T] u8, z[N,T] f32} (valid after the call; exposed for tests). */
size_t ac_render_rays_warped_scratch(int32_t n_rays, int32_t T, size_t offs[6]);

/* same arguments and outputs as ac_render_rays, plus the mesh and the scratch buffer */
int ac_render_rays_warped(const ac_field *field, const ac_render_opts *opts, const float *rays_o, const float *rays_d,
                          const float *bg, const float *noise, const float *lin_z, const float *lin_u,
                          const ac_warp_mesh *mesh, void *scratch, size_t scratch_bytes, const ac_render_out *out,
                          ac_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AVATARCRAFT_HIP_H */
