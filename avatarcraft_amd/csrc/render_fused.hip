// avatarcraft_amd/csrc/render_fused.hip -- the fused Instant-NSR ray renderer for MI355X (gfx950).
//
// One launch = NeRFRenderer.run (reference models/instant_nsr.py:133-299, render_can=True) for a
// batch of rays; the reference runs the same work as ~11 forward_sdf calls x ~6 PyTorch kernels
// plus ~200 small sampling kernels, every intermediate through HBM.
//
// Mapping (wave64 / CDNA4 first):
//   * one wavefront owns one ray at a time; the ray's z / sdf arrays (<=128 entries) live in a
//     wave-private LDS slab, everything else in registers.
//   * a wave evaluates the field on TILES OF 16 SAMPLES: lane = (sample n = lane&15, group g = lane>>4).
//     Group g gathers hash levels {g, 4+g, 8+g, 12+g} (8 of the 32 features), so the features
//     of one sample are spread over 4 lanes exactly as the B operand of
//     v_mfma_f32_16x16x4_f32 wants them (B[k=lane>>4][j=lane&15]): NO LDS transpose between the
//     gather and the MLP.  Layers are computed transposed (D^T = W * X^T), so the D layout of
//     layer i (row = 4*(lane>>4)+reg) is directly the B operand of layer i+1 with the k-order
//     (t, r, g) -> unit 16t+4g+r: the whole 35-64-16 SDF MLP and 21-64-64-3 colour MLP chain
//     through registers.  f32-input MFMA is bit-for-bit an fp32 fma chain in k order, which is
//     what the CPU oracle evaluates (oracle/ac_oracle.c: orc_sdf_mlp / orc_color_mlp).
//   * weights are converted once per workgroup from row-major global memory into MFMA A-fragment
//     order in LDS (40 KB, shared by the 8 waves of the workgroup) and read with conflict-free
//     lane-linear ds_read_b32.
//   * per-ray cumprod / cumsum / reductions are 16-lane DPP Kogge-Stone scans (row_shr 1,2,4,8)
//     with a scalar carry between tiles = "wavefront segmented scan".
//   * NeuS up-sampling (up_sample + sample_pdf + cat_z_vals) runs inside the wave: 64-lane chunks
//     for the per-bin math, binary searches in LDS, rank-based stable merge instead of a sort.
//
// Numerics contract: see ac_devmath.hpp and DESIGN.md; every value produced here is bit-identical
// to oracle/ac_oracle.c:orc_render_rays on the same inputs.
#include <atomic>
#include <map>
#include <mutex>
#include "nsr_device.hpp"

namespace {

#ifndef AC_DYNAMIC_RAYS
#define AC_DYNAMIC_RAYS 1          // persistent workgroups (one per compute unit), rays handed out by per-XCD counters; 0: eight fixed rays per workgroup
#endif
#ifndef AC_XCD_CHUNK
#define AC_XCD_CHUNK 512
#endif
#ifndef AC_WG_TICKETS
#define AC_WG_TICKETS 0            // 1 (round 6 experiment): a workgroup's waves draw their work items from blocks of 8 CONSECUTIVE rays (one global ticket per
#endif                             // block, handed out inside the workgroup through LDS) instead of one global ticket per wave: the 8 rays a compute unit
                                   // works on at a time are neighbouring pixels, whose coarse / middle level cells share L1 lines
#ifndef AC_FAST_COLOR
#define AC_FAST_COLOR 1            // fast precision: the colour network in split bf16 too (0: only layer 1 of the finite-difference evaluations)
#endif
// EX = false: a launch that wants the per-ray results only (image, weights_sum, depth, normal_map, eik): none of the optional per-sample outputs is
// compiled in, which takes their sixteen pointers (and the address arithmetic on them) out of the register budget of the tile loop
// SH = true: a field with view directions (ac_field.Wc1_sh).  A template parameter, not a run-time branch: the tile loop runs at 256 VGPRs with a few
// spilled registers, and the live pointer / flag of a run-time switch cost the DEFAULT model six more spills (+0.7 % on the headline launch, measured).
template <int MODE, bool FAST, bool EX, bool SH = false>
__global__ __launch_bounds__(BLOCK) void render_rays_kernel(const RenderArgs a)
{
    constexpr bool FC = FAST && AC_FAST_COLOR;
    extern __shared__ __attribute__((aligned(16))) float lds[];
#if AC_WG_TICKETS && AC_DYNAMIC_RAYS
    __shared__ uint32_t wg_cnt[8], wg_tag[8][16], wg_base[8][16];          // per segment: tickets drawn inside this workgroup | block k's (k + 1, global base)
    if (threadIdx.x < 8) wg_cnt[threadIdx.x] = 0u;
    if (threadIdx.x < 128) wg_tag[threadIdx.x >> 4][threadIdx.x & 15] = 0u;
#endif
    if (a.prepared) {
        // the weights arrive in LDS order (ac_field_prepare): a linear copy, 16 bytes per lane and trip, instead of ~27 dependent
        // gather-and-place trips per thread in each of the 512 workgroups of a launch; only the per-launch sampling tables are added
        const float4 *src = reinterpret_cast<const float4 *>(a.prepared);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        // (fast precision: the colour region holds the split-bf16 fragments, which the image keeps behind the exact one)
        for (int e = threadIdx.x; e < OFF_RWAVE / 4; e += blockDim.x)
            dst[e] = src[(FC && 4 * e >= OFF_C1F && 4 * e < OFF_B1) ? e + (OFF_RWAVE - OFF_C1F) / 4 : e];
        __syncthreads();
        for (int e = threadIdx.x; e < 64; e += blockDim.x) lds[OFF_LIN + e] = e < a.T0 ? a.lin_z[e] : 0.0f;
        for (int e = threadIdx.x; e < 16; e += blockDim.x) lds[OFF_LIN + 64 + e] = a.lin_u ? a.lin_u[e] : 0.0f;
    } else {
        fill_lds_sdf(lds, a);
        if constexpr (FAST) fill_lds_fast(lds, a);
        if constexpr (FC) fill_lds_color_fast<true, true>(lds, a);
        else fill_lds_color(lds, a);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    float *zs0 = lds + OFF_RWAVE + wave * WAVE_SLAB;    // z buffer 0 [128]: where the final samples of the ray end up
    float *zs1 = zs0 + MAXT;                            // --- from here on: up-sampling state while the ray is being sampled ...
    float *sd = zs1 + MAXT;                             // sd[2][128]
    float *cdf = sd + 2 * MAXT;                         // cdf[128]
    float *znl = cdf + MAXT;                            // znew[16]
    float *fsl = zs1;                                   // ... and the features of the finite-difference points [6][8][64] afterwards
    const FieldCtx fc = make_ctx(a);
    const W2Row0 w2r0 = load_w2_row0(lds, lane);
    const float bound = a.bound;
    const float inv_s_core = a.inv_s_dev ? *a.inv_s_dev : a.inv_s;     // forward_variance(): a launch constant, or one float on the device
    const int T0 = a.T0, nup = a.nup, T = T0 + 16 * nup;

#ifdef AC_PROFILE
    unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long prof_t0 = __builtin_amdgcn_s_memtime(), prof_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // XCD-aware order: workgroup b runs on XCD b % 8 (observed dispatch rule; speed only): give every XCD a contiguous
    // slab of rays so that neighbouring pixels share one L2 instead of eight
    int bid = blockIdx.x;
#ifndef AC_NO_XCD_REMAP
    {   // XCD k runs the workgroups b with b % 8 == k: give it the k-th contiguous range (ranges differ by one when 8 does not divide the grid)
        const int k = blockIdx.x & 7, q = gridDim.x >> 3, r = gridDim.x & 7;
        bid = k * q + (k < r ? k : r) + (blockIdx.x >> 3);
    }
#endif
    // Staggered start: wave w of a workgroup begins AC_START_STAGGER x 4096 clocks (~2 us each) x w late.  All 2048 waves of a launch would otherwise walk
    // through the same phases of their first ray together (every wave gathering, then every wave in the MLPs): rays of the first round took 350 .. 420 us
    // against 280 .. 330 us for the ones fetched later, when the waves have drifted apart (tools/phase_profile.py).  Round 2 used 3 (6 us per wave index, 41 us
    // for the last wave); with quarter-ray work items (round 3) 1 = 2 us per index, 14 us for the last wave, does the same (0.762 vs 0.760 ms; none: 0.776,
    // profiles/r03_experiments.txt section 14).
#ifndef AC_START_STAGGER
#define AC_START_STAGGER 1
#endif
    // (only launches that fill the device: a small batch -- a posed frame's tail, a unit test -- has no lock-step to break and would only pay the delay)
    if (a.n_rays >= 2048)
        for (int k_ = 0; k_ < AC_START_STAGGER * wave; ++k_) __builtin_amdgcn_s_sleep(64);
#if AC_DYNAMIC_RAYS
    // Work items are (ray, segment) pairs fetched one at a time from per-XCD counters.  A ray is cut into seg_n segments of the tile loop (segment 0 =
    // the sampling stage + the first tiles); a wave that finishes a segment leaves the ray's z values and running sums in seg_state and raises the ray's
    // flag, whichever wave of the XCD fetches the next segment of that ray continues the SAME sequential arithmetic from there (bit-identical results).
    // Every wave works through all segment-0 items of its XCD first, then the segment-1 items, ...: a 4096-ray launch is only two rays per wave slot,
    // and with whole rays as work items it ended with the slowest pair (per-wave busy time: mean 664 us, max 802 us); now the last items are a
    // quarter-ray long.  A wave never waits for an item nobody has started: segment s + 1 of a ray is handed out only after every segment-s item
    // of the XCD has been fetched by a running wave.
    // The batch is dealt to the XCDs in chunks of AC_XCD_CHUNK consecutive rays (two image rows of a 256-wide view: neighbouring rays share grid
    // cells in the XCD's L2), chunk c to XCD c % 8; a batch of up to 8 chunks is cut into eight contiguous parts.  Large batches stay balanced
    // that way when the body covers only some rows of the image (posed frames, skip_masked).
    const int xper = ((a.n_rays + 7) / 8 + 7) & ~7, xchunk = xper < AC_XCD_CHUNK ? xper : AC_XCD_CHUNK, xcd = blockIdx.x & 7;
    const int seg_n = (MODE == MODE_UPSAMPLE) ? 1 : a.seg_n;
    for (int seg = 0; seg < seg_n; ++seg) {
    const int c_begin = (MODE == MODE_UPSAMPLE) ? 0 : (int)((a.seg_cb >> (4 * seg)) & 15u), c_end = (MODE == MODE_UPSAMPLE) ? 0 : (int)((a.seg_cb >> (4 * seg + 4)) & 15u);
    const bool seg_first = seg == 0, seg_last = seg + 1 == seg_n;
    for (;;) {
        int ray = 0;
#if AC_WG_TICKETS
        if (lane == 0) {
            const uint32_t t = atomicAdd(&wg_cnt[seg], 1u), blk = t >> 3, slot = t & 7u, idx = blk & 15u;
            if (slot == 0u) {                                            // this wave opens block blk: 8 consecutive tickets of the XCD's counter
                const uint32_t b = atomicAdd(a.ray_counter + xcd * 8 + seg, 8u);
                __hip_atomic_store(&wg_base[seg][idx], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&wg_tag[seg][idx], blk + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                while (__hip_atomic_load(&wg_tag[seg][idx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != blk + 1u) __builtin_amdgcn_s_sleep(1);
            }
            ray = (int)(__hip_atomic_load(&wg_base[seg][idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + slot);
        }
#else
        if (lane == 0) ray = (int)atomicAdd(a.ray_counter + xcd * 8 + seg, 1u);
#endif
        ray = __builtin_amdgcn_readfirstlane(ray);
        {
            const int k = ray / xchunk, base = (k * 8 + xcd) * xchunk;
            if (base >= a.n_rays) break;
            ray = base + (ray - k * xchunk);
            if (ray >= a.n_rays) continue;
        }
        int rin = ray;                                               // row of this ray in rays_o / rays_d / near_m / far_m
        if (a.pair_n) { rin = ray >> 1; ray = rin + ((ray & 1) ? a.pair_n : 0); }      // a0 b0 a1 b1 ...: the two copies of a ray meet in their XCD's L2
        (void)bid;
#else
    const int seg_n = 1, seg = 0, c_begin = 0, c_end = MAXT / 16;
    const bool seg_first = true, seg_last = true;
    {
    for (int ray = bid * WAVES_PER_BLOCK + wave; ray < a.n_rays; ray += gridDim.x * WAVES_PER_BLOCK) {
        const int rin = (a.pair_n && ray >= a.pair_n) ? ray - a.pair_n : ray;      // (pair launch: the copies are rows [0, N) and [N, 2N))
#endif
        const int exr = ray - a.ex_from;                             // row in the per-sample outputs (pair launches keep them for copy b only)
        const bool ex_on = exr >= 0;
        AC_T0();
        const float ox = a.rays_o[3 * rin], oy = a.rays_o[3 * rin + 1], oz = a.rays_o[3 * rin + 2];
        const float dx = a.rays_d[3 * rin], dy = a.rays_d[3 * rin + 1], dz = a.rays_d[3 * rin + 2];
        float near, far;
        cube_near_far(ox, oy, oz, dx, dy, dz, bound, near, far);
        if (a.near_m) {                                          // :148-153 mesh-guided range where the ray passes the body
            const float nm = a.near_m[rin], fm = a.far_m[rin];
            if (!is_inf(nm)) near = nm;
            if (!is_inf(fm)) far = fm;
        }
        const float span = far - near;
        const float sample_dist = span / (float)T0;
        int cur = (MODE == MODE_FINAL) ? 0 : (nup & 1), cnt = T0;      // the buffers swap once per up-sampling iteration: start so that the last lands in zs0
        float *const zs_first = cur ? zs1 : zs0;

        if constexpr (MODE == MODE_UPSAMPLE) {
            // skip_masked: no sample of this ray can be unmasked (ray_cull_kernel) -- its pixel is the background whatever the field says, so neither
            // the coarse SDF nor the up-sampling runs; the z array is the coarse one padded with its last value (sorted, finite)
            if (a.ray_dead && a.ray_dead[ray]) {
                for (int i = lane; i < T; i += 64) {
                    const int ic = i < T0 ? i : T0 - 1;
                    float zi = near + span * lds[OFF_LIN + ic];
                    if (a.perturb) zi = zi + (a.noise[(size_t)ray * T0 + ic] - 0.5f) * sample_dist;
                    const size_t si = (size_t)ray * T + i;
                    a.zbuf[si] = zi;
                    if (a.mid_pts) { a.mid_pts[3 * si] = ox + dx * zi; a.mid_pts[3 * si + 1] = oy + dy * zi; a.mid_pts[3 * si + 2] = oz + dz * zi; }
                }
                wave_sync();
                continue;
            }
        }
        // ---- coarse samples :155-180 -------------------------------------------------------------
        if constexpr (MODE == MODE_FINAL) {
            for (int i = lane; i < T; i += 64) zs0[i] = a.zbuf[(size_t)ray * T + i];
        } else if (seg_first) {
            for (int c = 0; c < T0 / 16; ++c) {
                const int i = 16 * c + n;
                float zi = near + span * lds[OFF_LIN + i];
                if (a.perturb) zi = zi + (a.noise[(size_t)ray * T0 + i] - 0.5f) * sample_dist;
                if (nup > 0) {
                    float px, py, pz;
                    if (MODE == MODE_UPSAMPLE && a.ext_pts) {       // posed space: the warped coarse points (NULL: canonical sampling only)
                        const float *e = a.ext_pts + ((size_t)ray * T0 + i) * 3;
                        px = clampf(e[0], -bound, bound); py = clampf(e[1], -bound, bound); pz = clampf(e[2], -bound, bound);
                    } else {
                        px = clampf(ox + dx * zi, -bound, bound); py = clampf(oy + dy * zi, -bound, bound);
                        pz = clampf(oz + dz * zi, -bound, bound);
                    }
                    const f32x4 o2 = sdf_tile(lds, fc, lane, px, py, pz);
                    if (g == 0) sd[cur * MAXT + i] = o2[0];
                }
                if (g == 0) zs_first[i] = zi;
            }
        }
        wave_sync();
        AC_TICK(0)

        // ---- NeuS up-sampling :182-184, :410-475 -----------------------------------------------------
        for (int it = 0; it < ((MODE == MODE_FINAL || !seg_first) ? 0 : nup); ++it) {
            const float *zc = cur ? zs1 : zs0, *sc = sd + cur * MAXT;
            float *zn_ = cur ? zs0 : zs1, *sn_ = sd + (cur ^ 1) * MAXT;
            const int m = cnt - 1;
            const float inv_s = (float)(64 << it);
            float w[2];
            float carry = 1.0f; bool first = true;
            // pass 1: alpha, transmittance scan, weights (+1e-5)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int i = 64 * ch + lane;
                float alpha = 0.0f, om = 1.0f;
                if (i < m) {
                    const float z0 = zc[i], z1 = zc[i + 1], s0 = sc[i], s1 = sc[i + 1];
                    const float p0x = ox + dx * z0, p0y = oy + dy * z0, p0z = oz + dz * z0;
                    const float p1x = ox + dx * z1, p1y = oy + dy * z1, p1z = oz + dz * z1;
                    const float r0 = __builtin_sqrtf((p0x * p0x + p0y * p0y) + p0z * p0z);
                    const float r1 = __builtin_sqrtf((p1x * p1x + p1y * p1y) + p1z * p1z);
                    const bool inside = (r0 < 1.0f) | (r1 < 1.0f);
                    const float mid = (s0 + s1) * 0.5f;
                    const float dist = z1 - z0;
                    const float cosv = (s1 - s0) / (dist + 1e-5f);
                    float prev_cos = 0.0f;
                    if (i > 0) { const float zm = zc[i - 1], sm = sc[i - 1]; prev_cos = (s0 - sm) / ((z0 - zm) + 1e-5f); }
                    float cmin = prev_cos < cosv ? prev_cos : cosv;
                    cmin = clampf(cmin, -1e3f, 0.0f) * (inside ? 1.0f : 0.0f);
                    const float half = cmin * dist * 0.5f;
                    const float pc = dv_sigmoid((mid - half) * inv_s), nc = dv_sigmoid((mid + half) * inv_s);
                    alpha = (pc - nc + 1e-5f) / (pc + 1e-5f);
                    om = 1.0f - alpha + 1e-7f;
                }
                float loc, row_in; bool row_first;
                (void)chunk_scan<true>(om, lane, carry, first, loc, row_in, row_first);
                // exclusive transmittance T_i = cp[i-1]: row-local inclusive value of lane n-1 times the row carry
                const float sh = dpp_shr<1>(1.0f, loc);                 // local[n-1], identity in lane n==0
                float Tex;
                if (n == 0) Tex = row_first ? 1.0f : row_in;
                else Tex = row_first ? sh : row_in * sh;
                w[ch] = (i < m) ? alpha * Tex + 1e-5f : 0.0f;
            }
            // pass 2: total = last element of the inclusive add tile-scan of w
            float total;
            {
                float c2 = 0.0f; bool f2 = true; float lc, ri; bool rf;
                float last = 0.0f;
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const float incl = chunk_scan<false>(w[ch], lane, c2, f2, lc, ri, rf);
                    const int il = m - 1 - 64 * ch;                    // lane holding element m-1 (wave-uniform)
                    const float cand = __shfl(incl, il & 63);
                    if (il >= 0 && il < 64) last = cand;
                }
                total = last;
            }
            // pass 3: pdf, cdf
            {
                float c3 = 0.0f; bool f3 = true; float lc, ri; bool rf;
                if (lane == 0) cdf[0] = 0.0f;
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const int i = 64 * ch + lane;
                    const float pdf = (i < m) ? w[ch] / total : 0.0f;
                    const float incl = chunk_scan<false>(pdf, lane, c3, f3, lc, ri, rf);
                    if (i < m) cdf[i + 1] = incl;
                }
            }
            wave_sync();
            // sample_pdf(det=True): one new sample per n (replicated over the 4 groups)
            float znew;
            int ind;
            {
                const float u = lds[OFF_LIN + 64 + n];
                int lo = 0, hi = cnt;
                while (lo < hi) { const int md = (lo + hi) >> 1; if (cdf[md] <= u) lo = md + 1; else hi = md; }
                ind = lo;
                const int below = lo - 1 > 0 ? lo - 1 : 0;
                const int above = lo < cnt - 1 ? lo : cnt - 1;
                const float cb = cdf[below], ca = cdf[above];
                float den = ca - cb;
                if (den < 1e-5f) den = 1.0f;
                const float t = (u - cb) / den;
                const float zb = zc[below], za = zc[above];
                znew = zb + t * (za - zb);
            }
            if (g == 0) {
                znl[n] = znew;
                if (EX && ex_on && a.out.ss_inds) a.out.ss_inds[((size_t)exr * nup + it) * 16 + n] = ind;
            }
            AC_TICK(1)
            const bool last_it = (it + 1 == nup);
            float sdf_new = 0.0f;
            if (!last_it) {
                const float px = clampf(ox + dx * znew, -bound, bound), py = clampf(oy + dy * znew, -bound, bound),
                            pz = clampf(oz + dz * znew, -bound, bound);
                const f32x4 o2 = sdf_tile(lds, fc, lane, px, py, pz);
                sdf_new = o2[0];
            }
            wave_sync();
            AC_TICK(2)
            // stable merge == torch.sort(cat([z, znew])) :466-473.  The old z are sorted except in the first iteration of a ray whose slab
            // test gave far < near (it misses the cube: its coarse z run from near DOWN to far): there the old elements are ranked too
            const bool old_sorted = !(it == 0 && span < 0.0f);         // wave-uniform
            int32_t *sidx = (EX && ex_on && a.out.sort_index) ? a.out.sort_index + ((size_t)exr * nup + it) * 128 : nullptr;
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int i = 64 * ch + lane;
                if (i < cnt) {
                    const float zi = zc[i];
                    int c = 0;
#pragma unroll
                    for (int j = 0; j < 16; ++j) c += (znl[j] < zi) ? 1 : 0;
                    int before = i;
                    if (!old_sorted) {
                        before = 0;
                        for (int k = 0; k < cnt; ++k) { const float zk = zc[k]; before += ((zk < zi) || (zk == zi && k < i)) ? 1 : 0; }
                    }
                    zn_[before + c] = zi; sn_[before + c] = sc[i];
                    if (sidx) sidx[before + c] = i;
                }
                if (sidx && i >= cnt + 16) sidx[i] = -1;
            }
            if (g == 0) {
                int lo = 0, hi = cnt;
                if (old_sorted) {
                    while (lo < hi) { const int md = (lo + hi) >> 1; if (zc[md] <= znew) lo = md + 1; else hi = md; }
                } else {
                    for (int k = 0; k < cnt; ++k) lo += (zc[k] <= znew) ? 1 : 0;
                }
                int c = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) { const float zj = znl[j]; c += ((zj < znew) || (zj == znew && j < n)) ? 1 : 0; }
                zn_[lo + c] = znew; sn_[lo + c] = sdf_new;
                if (sidx) sidx[lo + c] = cnt + n;
            }
            cnt += 16; cur ^= 1;
            wave_sync();
            AC_TICK(1)
        }

        // ---- render core :190-299 ---------------------------------------------------------------------
        const float *zf = zs0;                                  // cur == 0 here by construction
        if constexpr (MODE == MODE_UPSAMPLE) {                 // hand z and the posed-space mid points to the warp
            for (int i = lane; i < T; i += 64) {
                const float zi = zf[i];
                const float delta = (i < T - 1) ? zf[i + 1] - zi : sample_dist;
                const float zmid = (i < T - 1) ? zi + 0.5f * delta : zi;
                const size_t si = (size_t)ray * T + i;
                a.zbuf[si] = zi;
                if (a.mid_pts) { a.mid_pts[3 * si] = ox + dx * zmid; a.mid_pts[3 * si + 1] = oy + dy * zmid; a.mid_pts[3 * si + 2] = oz + dz * zmid; }
            }
            wave_sync();
            continue;
        }
        float cT = 1.0f;                                        // transmittance carry (cumprod)
        // the ten running sums of the ray (weights, colour, normal, depth, eikonal numerator / denominator) live in the wave's LDS slab, not in
        // registers: they are touched once per tile by one lane (lane 15, which holds the tile totals of the row scans), and ten registers less at the
        // peak of the stencil / MLP code is the difference between ~30 and ~10 spilled registers.  Slots: 1 s_w 2..4 rgb 5..7 normal 8 depth 9 10 eikonal
        float *const accs = zs0 + SLAB_ACC;
        // use_viewdirs: layer-1 bias of the colour network for THIS ray's direction, in the wave's slab: formed by the ray's first segment (16 sh values + 64
        // dot products of 16 terms), handed to the later ones with the segment state (64 floats: one store / one load per lane instead of the prologue again)
        if constexpr (SH && MODE != MODE_UPSAMPLE) { if (seg_first && !a.opacity_only) ray_sh_bias(zs0 + SLAB_SHB, a.Wsh, dx, dy, dz, lane); }
        if (!seg_first) {
            // continue a ray another wave (of this XCD) started: wait until its previous segment is published, then take over z and the running sums.
            // All accesses to seg_flags / seg_state are agent-scope atomics = served by the XCD's L2, past the (incoherent) vector L1 caches.
            int timed_out = 0;
            if (lane == 0) {
                int spins = 0;
                // the flag of THIS launch: generation in the upper bits (a value left by an earlier launch in the same slot never matches)
                for (;;) {
                    const uint32_t f = __hip_atomic_load(a.seg_flags + ray, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (((f >> 4) == a.gen && (f & 15u) >= (uint32_t)seg) || spins >= (1 << 21)) break;
                    __builtin_amdgcn_s_sleep(8); ++spins;       // (bounded: ~1 s; a ray's previous segment takes ~100 us)
                }
                timed_out = spins >= (1 << 21);
                if (timed_out && a.handoff_timeouts) atomicAdd(a.handoff_timeouts, 1u);     // the host can ask (ac_render_handoff_timeouts): a lost hand-off is an ERROR, not only a NaN pixel
            }
            timed_out = __builtin_amdgcn_readfirstlane(timed_out);
            // NO acquire fence here, on purpose and measured (round 4): `fence acquire, agent` is `buffer_inv sc1` on gfx950 -- it empties the compute
            // unit's vector L1, i.e. the table lines all eight resident waves are gathering from, three times per ray: the 4096-ray launch went from
            // 0.744 to 1.087 ms with the acquire / release pair in place.  What makes the hand-off correct without it: the state and the
            // flag are written and read with agent-scope ATOMIC accesses only (they bypass the non-coherent L1 on both sides and meet at the device's
            // coherence point, whichever XCD either wave runs on); the publisher drains its state stores (s_waitcnt vmcnt(0)) before it issues the flag
            // store; the taker issues its state loads only after lane 0 has observed the flag (control dependence + the wave barrier below), and the
            // memory pipeline returns a wave's loads in issue order.  No ordinary (cached) access ever touches these words.
            // COMPILER ordering is pinned, not assumed: wave_sync() is `fence acq_rel, wavefront` + wave barrier (nsr_device.hpp) -- a wavefront-scope fence
            // costs no instruction (no cache maintenance) but forbids LLVM to move the relaxed state loads below above the flag load of the loop above
            // (ADVICE round 4); the publisher's side has the same fence between its state stores and its flag store.
            wave_sync();
            const uint32_t *st = reinterpret_cast<const uint32_t *>(a.seg_state + (size_t)ray * SEG_STATE);
            if constexpr (MODE != MODE_FINAL)
                for (int i = lane; i < T; i += 64) zs0[i] = __uint_as_float(__hip_atomic_load(st + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const float cv = __uint_as_float(__hip_atomic_load(st + MAXT + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            cT = lane_bcast(cv, 0);
            if (lane < 16) accs[lane] = cv;
            if constexpr (SH) zs0[SLAB_SHB + lane] = __uint_as_float(__hip_atomic_load(st + MAXT + 16 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (timed_out) {                                     // never observed; if the previous segment was not published in ~1 s the ray's pixel must not
                cT = __builtin_nanf("");                         // look like a result: NaN, which the callers' finite checks and every parity test catch
                if (lane < 16) accs[lane] = cT;
            }
            wave_sync();
        }
        const float bxe = a.eps;
        const int c_hi = c_end < T / 16 ? c_end : T / 16;
        for (int c = c_begin; c < c_hi; ++c) {
            const int i = 16 * c + n;
            const float zi = zf[i];
            const float delta = (i < T - 1) ? zf[i + 1] - zi : sample_dist;
            const float zmid = (i < T - 1) ? zi + 0.5f * delta : zi;
            float px, py, pz;
            if constexpr (MODE == MODE_FINAL) {
                const float *e = a.ext_pts + ((size_t)ray * T + i) * 3;
                px = clampf(e[0], -bound, bound); py = clampf(e[1], -bound, bound); pz = clampf(e[2], -bound, bound);
            } else {
                px = clampf(ox + dx * zmid, -bound, bound); py = clampf(oy + dy * zmid, -bound, bound);
                pz = clampf(oz + dz * zmid, -bound, bound);
            }
            // centre + 6 finite-difference evaluations (:687-704): features of all 7 points first (shared corner
            // fetches), then 7 MLP passes as one loop body over a rotating feature register file.
            AC_TICK(7)
            // posed space: a tile whose 16 samples the warp masks out contributes alpha * 0 whatever the field says there (opt-in: skip_masked)
            bool skip = false;
            if constexpr (MODE == MODE_FINAL) {
                if (a.skip_masked) skip = __ballot(a.mask[(size_t)ray * T + i] != 0) == 0ull;       // wave-uniform
            }
            f32x4 oc = { 0.0f, 0.0f, 0.0f, 0.0f };
            float gr[3] = { 0.0f, 0.0f, 0.0f };
            if (!skip) {
            float fe0[4][2];
            encode_stencil<(FAST && AC_FACE_VALUE) ? 1 : 0>(lds, fsl, fc, lane, px, py, pz, bxe, fe0);
            if (EX && ex_on && a.out.feat7) {                                     // training render: keep the 7 x 8 features of this lane (the backward streams them back)
                // layout [tile of 16 samples][14][lane][4]: float k = 8 e + q (evaluation e, slot q = 2 j + channel) of lane (n, g) sits in group k / 4,
                // component k % 4 -- every store (and every load of the backward) is one 16-byte access per lane, 1 KB contiguous per wave
                // (the lane term enters through an opaque copy: otherwise the per-lane base pointer is hoisted out of the tile loop and, at 256 registers, spilled -- a scratch
                // reload with a full wait in every tile; formed here it is one scalar product and one 64-bit add.  Round 6, profiles/r06_experiments.txt section 14)
                int lane_x = lane;
                asm volatile("" : "+v"(lane_x));
                f32x4 *dst = reinterpret_cast<f32x4 *>(a.out.feat7) + (((size_t)exr * (T / 16) + (i >> 4)) * 14) * 64 + lane_x;
                dst[0] = f32x4{ fe0[0][0], fe0[0][1], fe0[1][0], fe0[1][1] };
                dst[64] = f32x4{ fe0[2][0], fe0[2][1], fe0[3][0], fe0[3][1] };
#pragma unroll 1
                for (int e = 0; e < 6; ++e) {
                    const float *sp = fsl + (e * 8) * 64 + lane;
                    dst[(2 * e + 2) * 64] = f32x4{ sp[0], sp[64], sp[128], sp[192] };
                    dst[(2 * e + 3) * 64] = f32x4{ sp[256], sp[320], sp[384], sp[448] };
                }
            }
            AC_TICK(3)
            const float pc0 = sel4(g, px, py, pz, 0.0f);
            float spos = 0.0f;
            if constexpr (FAST) {
                // precision 1: the centre evaluation exactly (fp32 MFMA), the six offset evaluations as corrections of its layer 1 on the
                // bf16 matrix pipe (sdf_l1_delta): 12 short MFMA + ~50 VALU per evaluation instead of 36 fp32 MFMA of 32 clocks each
                const Acc4 acc0 = sdf_l1(lds, lane, pc0, fe0);
                oc = sdf_l2(lds, lane, acc0);
#pragma unroll 1
                for (int e = 0; e < 6; ++e) {
                    const int kn = e >> 1;
                    float fe[4][2];
#pragma unroll
                    for (int q_ = 0; q_ < 8; ++q_) fe[q_ >> 1][q_ & 1] = fsl[(e * 8 + q_) * 64 + lane];
                    const float pk = kn == 0 ? px : (kn == 1 ? py : pz);
                    const float poff = clampf(pk + ((e & 1) ? -bxe : bxe), -bound, bound);
                    const Acc4 acc = sdf_l1_delta(lds, lane, acc0, fe, fe0, kn, poff - pk);
                    const float s_e = sdf_l2_sdf(lds, acc, w2r0);
                    if (!(e & 1)) spos = s_e;
                    else {
                        const float gk = 0.5f * (spos - s_e) / bxe;
                        if (kn == 0) gr[0] = gk; else if (kn == 1) gr[1] = gk; else gr[2] = gk;
                    }
                }
            } else {
            // 7 MLP passes, software-pipelined: layer 1 of evaluation e+1 (MFMA) is issued next to the softplus +
            // layer 2 of evaluation e (VALU), so the matrix and vector pipes of the SIMD overlap inside one wave.
            Acc4 acc = sdf_l1(lds, lane, pc0, fe0);
#pragma unroll 1
            for (int e = 0; e < 7; ++e) {
                Acc4 accn = acc;
                if (e < 6) {                                       // layer 1 of the next evaluation
                    const int kn = e >> 1;
                    float fe[4][2];
#pragma unroll
                    for (int q_ = 0; q_ < 8; ++q_) fe[q_ >> 1][q_ & 1] = fsl[(e * 8 + q_) * 64 + lane];
                    const float pk = kn == 0 ? px : (kn == 1 ? py : pz);
                    const float poff = clampf(pk + ((e & 1) ? -bxe : bxe), -bound, bound);
                    accn = sdf_l1(lds, lane, g == kn ? poff : pc0, fe);
                }
                if (e == 0) oc = sdf_l2(lds, lane, acc);           // the centre needs all 16 outputs
                else {                                             // the six offset points only their sdf
                    const float s_e = sdf_l2_sdf(lds, acc, w2r0);
                    const int k = (e - 1) >> 1;
                    if (e & 1) spos = s_e;
                    else {
                        const float gk = 0.5f * (spos - s_e) / bxe;
                        if (k == 0) gr[0] = gk; else if (k == 1) gr[1] = gk; else gr[2] = gk;
                    }
                }
                acc = accn;
            }
            }
            }
            AC_TICK(4)
            const float gx = gr[0], gy = gr[1], gz = gr[2];        // every lane of a sample holds the same finite-difference gradient
            const float gn = __builtin_sqrtf((gx * gx + gy * gy) + gz * gz);
            const float nx = gx / (1e-5f + gn), ny = gy / (1e-5f + gn), nz = gz / (1e-5f + gn);
            float rgb[3] = { 0.0f, 0.0f, 0.0f };
            if (!skip && !a.opacity_only) {                      // (wave-uniform)
                if constexpr (FC) color_tile_fast(lds, lane, px, py, pz, nx, ny, nz, oc, rgb, nullptr, 16, SH ? zs0 + SLAB_SHB : nullptr);
                else color_tile(lds, lane, px, py, pz, nx, ny, nz, oc, rgb, nullptr, 16, SH ? zs0 + SLAB_SHB : nullptr);
            }
            AC_TICK(5)
            // NeuS alpha :219-248
            const float sdf0 = oc[0];
            const float tc = (dx * nx + dy * ny) + dz * nz;
            const float a1 = dv_softplus100(lds + OFF_SPQ, -tc * 0.5f + 0.5f) * a.one_m_car;
            const float a2 = dv_softplus100(lds + OFF_SPQ, -tc) * a.car;
            const float iter_cos = -(a1 + a2);
            const float half = iter_cos * delta * 0.5f;
            const float pc = dv_sigmoid((sdf0 - half) * inv_s_core), nc = dv_sigmoid((sdf0 + half) * inv_s_core);
            float alpha = clampf((pc - nc + 1e-5f) / (pc + 1e-5f), 0.0f, 1.0f);
            if constexpr (MODE == MODE_FINAL) alpha = alpha * (a.mask[(size_t)ray * T + i] ? 1.0f : 0.0f);      // :246-249
            const float om = 1.0f - alpha + 1e-7f;
            // transmittance: exclusive tile scan with carry  :250
            const float loc = row_scan<true>(om);
            const float sh = dpp_shr<1>(1.0f, loc);
            float Tex;
            if (n == 0) Tex = (c == 0) ? 1.0f : cT;
            else Tex = (c == 0) ? sh : cT * sh;
            const float tot = lane_bcast(loc, 15);
            cT = (c == 0) ? tot : cT * tot;
            const float wgt = alpha * Tex;
            const float zn01 = clampf((zi - near) / span, 0.0f, 1.0f);
            const float pn = __builtin_sqrtf((px * px + py * py) + pz * pz);
            const float relax = (pn < 1.2f && !skip) ? 1.0f : 0.0f;
            const float eerr = relax * ((gn - 1.0f) * (gn - 1.0f));
            // reductions (lane 15 of row 0 holds the tile totals)
            // reductions: lane 15 holds the tile totals of the row scans and adds them to the ray's running sums (sequential over the tiles, like the oracle)
            {
                const float t1 = row_scan<false>(wgt), t2 = row_scan<false>(rgb[0] * wgt), t3 = row_scan<false>(rgb[1] * wgt), t4 = row_scan<false>(rgb[2] * wgt),
                            t5 = row_scan<false>(nx * wgt), t6 = row_scan<false>(ny * wgt), t7 = row_scan<false>(nz * wgt), t8 = row_scan<false>(wgt * zn01),
                            t9 = row_scan<false>(eerr), t10 = row_scan<false>(relax);
                if (lane == 15) {
#define AC_ACC(K, T_) accs[K] = (c == 0) ? T_ : accs[K] + T_;
                    AC_ACC(1, t1) AC_ACC(2, t2) AC_ACC(3, t3) AC_ACC(4, t4) AC_ACC(5, t5) AC_ACC(6, t6) AC_ACC(7, t7) AC_ACC(8, t8) AC_ACC(9, t9) AC_ACC(10, t10)
#undef AC_ACC
                }
            }
            AC_TICK(6)
            if (g == 0) {
                const size_t si = (size_t)exr * T + i;
                if (EX && ex_on && a.out.z_vals) a.out.z_vals[si] = zi;
                if (EX && ex_on && a.out.weights) a.out.weights[si] = wgt;
                if (EX && ex_on && a.out.alpha) a.out.alpha[si] = alpha;
                if (EX && ex_on && a.out.sdf) a.out.sdf[si] = sdf0;
                if (EX && ex_on && a.out.color) { a.out.color[3 * si] = rgb[0]; a.out.color[3 * si + 1] = rgb[1]; a.out.color[3 * si + 2] = rgb[2]; }
                if (EX && ex_on && a.out.gradient) { a.out.gradient[3 * si] = gx; a.out.gradient[3 * si + 1] = gy; a.out.gradient[3 * si + 2] = gz; }
                if (EX && ex_on && a.out.pts) { a.out.pts[3 * si] = px; a.out.pts[3 * si + 1] = py; a.out.pts[3 * si + 2] = pz; }
            }
            if (EX && ex_on && a.out.sdf_out16) {                                 // lane (n, g) holds outputs 4g..4g+3
                int g_x = g;
                asm volatile("" : "+v"(g_x));                                   // (see feat7 above)
                *reinterpret_cast<f32x4 *>(a.out.sdf_out16 + ((size_t)exr * T + i) * 16 + 4 * g_x) = oc;
            }
        }
        if (!seg_last) {
            // hand the ray to its next segment: z values (once), the running sums, then the flag -- in that order (the stores are complete in L2
            // before the flag leaves: s_waitcnt vmcnt(0); the reader's loads are issued after it has seen the flag)
            uint32_t *st = reinterpret_cast<uint32_t *>(a.seg_state + (size_t)ray * SEG_STATE);
            if (MODE != MODE_FINAL && seg_first)
                for (int i = lane; i < T; i += 64) __hip_atomic_store(st + i, __float_as_uint(zf[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wave_sync();
            if (lane < 11) {
                const float cv = lane == 0 ? cT : accs[lane];
                __hip_atomic_store(st + MAXT + lane, __float_as_uint(cv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if constexpr (SH) { if (seg_first) __hip_atomic_store(st + MAXT + 16 + lane, __float_as_uint(zs0[SLAB_SHB + lane]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every lane's state stores have left the wave ...
            wave_sync();
            // ... before the flag store is issued (relaxed, agent scope: see the taker's side for why no release / acquire pair is used)
            if (lane == 0) __hip_atomic_store(a.seg_flags + ray, (a.gen << 4) | (uint32_t)(seg + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wave_sync();
            continue;
        }
        wave_sync();
        if (lane == 0) {
            const float s_w = accs[1], s_r = accs[2], s_g = accs[3], s_b = accs[4], s_nx = accs[5], s_ny = accs[6], s_nz = accs[7], s_d = accs[8], s_en = accs[9], s_ed = accs[10];
            const float b0 = a.bg ? a.bg[3 * ray] : 1.0f, b1 = a.bg ? a.bg[3 * ray + 1] : 1.0f, b2 = a.bg ? a.bg[3 * ray + 2] : 1.0f;
            a.out.image[3 * ray] = s_r + (1.0f - s_w) * b0;
            a.out.image[3 * ray + 1] = s_g + (1.0f - s_w) * b1;
            a.out.image[3 * ray + 2] = s_b + (1.0f - s_w) * b2;
            a.out.normal_map[3 * ray] = s_nx; a.out.normal_map[3 * ray + 1] = s_ny; a.out.normal_map[3 * ray + 2] = s_nz;
            a.out.weights_sum[ray] = s_w;
            a.out.depth[ray] = s_d;
            a.out.eik[2 * ray] = s_en; a.out.eik[2 * ray + 1] = s_ed;
        }
        wave_sync();
#ifdef AC_PROFILE               // per-ray wall time (100 MHz ticks) behind the per-wave records: [n_rays * 10 + ray]
        if (a.prof && lane == 0) a.prof[(size_t)a.n_rays * 10 + ray] = __builtin_amdgcn_s_memrealtime() - ray_r0_;
#endif
    }
    }
#if AC_DYNAMIC_RAYS
    // ---- epilogue: the last workgroup to finish reduces gradient_error and re-arms the slot's work counters (RenderArgs::done_counter) --------------------
    // (the arguments used here are read from the kernel-argument segment again, behind an opaque barrier: kept in scalar registers from the start of the kernel
    //  they cost the work loops two spilled vector registers)
    const __attribute__((address_space(4))) RenderArgs *ka = (const __attribute__((address_space(4))) RenderArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    uint32_t *const k_done = ka->done_counter;
    if (k_done) {
        __syncthreads();                                         // every wave of this workgroup has left the work loops: the weights in LDS are free
        uint32_t *ldsu = reinterpret_cast<uint32_t *>(lds);
        if (threadIdx.x == 0) {
            __threadfence();                                     // this workgroup's per-ray results are visible device-wide before its ticket is
            ldsu[0] = atomicAdd(k_done, 1u) == gridDim.x - 1u ? 1u : 0u;
        }
        __syncthreads();
        const bool last = ldsu[0] != 0u;
        __syncthreads();
        if (last) {
            __threadfence();                                     // ... and every other workgroup's are visible here
            float *const k_red = ka->eik_red;
            const int k_pair = ka->pair_n, k_n = ka->n_rays;
            const float *const k_eik = ka->out.eik;
            uint32_t *const k_cnt = ka->ray_counter;
            if (k_red) {
                // gradient_error (:270-272) of the batch -- of each copy of a pair launch -- in eikonal_reduce_kernel's order (oracle: orc_eikonal_reduce):
                // 1024 strided sequential partial sums (two per thread here), then a halving tree
                const int nred = k_pair ? 2 : 1, nper = k_pair ? k_pair : k_n;
                float *pn = lds, *pd = lds + 1024;
                for (int q = 0; q < nred; ++q) {
                    const float *e = k_eik + (size_t)q * nper * 2;
                    for (int t = (int)threadIdx.x; t < 1024; t += BLOCK) {
                        float sa = 0.0f, sb = 0.0f;
                        for (int r = t; r < nper; r += 1024) { sa += e[2 * r]; sb += e[2 * r + 1]; }
                        pn[t] = sa; pd[t] = sb;
                    }
                    __syncthreads();
                    for (int st = 512; st > 0; st >>= 1) {
                        for (int t = (int)threadIdx.x; t < st; t += BLOCK) { pn[t] += pn[t + st]; pd[t] += pd[t + st]; }
                        __syncthreads();
                    }
                    if (threadIdx.x == 0) { k_red[2 * q] = pn[0] / (pd[0] + 1e-5f); k_red[2 * q + 1] = pd[0] + 1e-5f; }
                    __syncthreads();
                }
            }
            if (threadIdx.x < 64) k_cnt[threadIdx.x] = 0u;       // the slot's next launch starts from zero again
            if (threadIdx.x == 64) *k_done = 0u;
        }
    }
#endif
#ifdef AC_PROFILE
    if (a.prof && lane == 0) { const int w_ = blockIdx.x * WAVES_PER_BLOCK + wave; for (int i = 0; i < 8; ++i) a.prof[w_ * 10 + i] = prof_acc[i];
        a.prof[w_ * 10 + 8] = __builtin_amdgcn_s_memtime() - prof_t0; a.prof[w_ * 10 + 9] = __builtin_amdgcn_s_memrealtime() - prof_r0; }   // shader clock vs 100 MHz
#endif
}

// ---- stand-alone field queries (density(), extract_geometry(), unit tests) -------------------------------
__global__ __launch_bounds__(BLOCK) void field_sdf_kernel(const RenderArgs a, const float *__restrict__ x, uint32_t B,
                                                          float *__restrict__ out16)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    fill_lds(lds, a);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const FieldCtx fc = make_ctx(a);
    const uint32_t ntiles = (B + 15) / 16;
    for (uint32_t tile = blockIdx.x * WAVES_PER_BLOCK + wave; tile < ntiles; tile += gridDim.x * WAVES_PER_BLOCK) {
        const uint32_t b = tile * 16 + n, bb = b < B ? b : B - 1;
        const float px = x[3 * bb], py = x[3 * bb + 1], pz = x[3 * bb + 2];
        const f32x4 o = sdf_tile(lds, fc, lane, px, py, pz);
        if (b < B) *reinterpret_cast<f32x4 *>(out16 + (size_t)b * 16 + 4 * g) = o;
    }
}

// dirs (use_viewdirs, with a.Wsh): the view direction of every point; the per-sample bias goes through a [4][64][4] slab per wave behind the weights
__global__ __launch_bounds__(BLOCK) void field_color_kernel(const RenderArgs a, const float *__restrict__ x, const float *__restrict__ dirs,
                                                            const float *__restrict__ nrm, const float *__restrict__ sdfout,
                                                            uint32_t B, float *__restrict__ rgb_out)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    fill_lds(lds, a);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const uint32_t ntiles = (B + 15) / 16;
    float *slab = lds + OFF_WAVE + wave * 1024;
    for (uint32_t tile = blockIdx.x * WAVES_PER_BLOCK + wave; tile < ntiles; tile += gridDim.x * WAVES_PER_BLOCK) {
        const uint32_t b = tile * 16 + n, bb = b < B ? b : B - 1;
        const f32x4 so = *reinterpret_cast<const f32x4 *>(sdfout + (size_t)bb * 16 + 4 * g);
        float rgb[3];
        if (dirs) {
            wave_sync();
            sample_sh_bias(slab, a.Wsh, dirs[3 * bb], dirs[3 * bb + 1], dirs[3 * bb + 2], lane);
            color_tile(lds, lane, x[3 * bb], x[3 * bb + 1], x[3 * bb + 2], nrm[3 * bb], nrm[3 * bb + 1], nrm[3 * bb + 2], so, rgb, slab + 4 * lane, 256);
        } else
        color_tile(lds, lane, x[3 * bb], x[3 * bb + 1], x[3 * bb + 2], nrm[3 * bb], nrm[3 * bb + 1], nrm[3 * bb + 2], so, rgb);
        if (b < B && g == 0) { rgb_out[3 * b] = rgb[0]; rgb_out[3 * b + 1] = rgb[1]; rgb_out[3 * b + 2] = rgb[2]; }
    }
}

// ac_field_prepare: one workgroup lays the weights out in LDS exactly like a render workgroup would and dumps the image
__global__ __launch_bounds__(BLOCK) void field_prepare_kernel(const RenderArgs a, float *__restrict__ image)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    fill_lds(lds, a);
    fill_lds_fast(lds, a);
    fill_lds_color_fast<false, true>(lds, a);                 // the two fragments outside the overlay
    __syncthreads();
    for (int e = threadIdx.x; e < OFF_RWAVE; e += blockDim.x) image[e] = lds[e];
    __syncthreads();
    fill_lds_color_fast<true, false>(lds, a);                 // the overlay of the colour region (fast precision), kept behind the exact image
    __syncthreads();
    for (int e = threadIdx.x; e < CF_OVERLAY; e += blockDim.x) image[OFF_RWAVE + e] = lds[OFF_C1F + e];
}

// gradient_error: fixed-order reduction of per-ray partials (oracle: orc_eikonal_reduce)
__global__ __launch_bounds__(1024) void eikonal_reduce_kernel(const float *__restrict__ eik, int n_rays, float *__restrict__ result, int with_den)
{
    __shared__ float pn[1024], pd[1024];
    const int t = threadIdx.x;
    float a = 0.0f, b = 0.0f;
    for (int r = t; r < n_rays; r += 1024) { a += eik[2 * r]; b += eik[2 * r + 1]; }
    pn[t] = a; pd[t] = b;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (t < s) { pn[t] += pn[t + s]; pd[t] += pd[t + s]; }
        __syncthreads();
    }
    if (t == 0) { result[0] = pn[0] / (pd[0] + 1e-5f); if (with_den) result[1] = pd[0] + 1e-5f; }
}

}  // namespace

#ifdef AC_PROFILE
static unsigned long long *g_prof = nullptr;
AC_API void ac_debug_set_prof(unsigned long long *p) { g_prof = p; }
#endif

// posed-space coarse samples: pts[n, i] = o + d * z_i (unclamped, fp32), the input of the first warp  (:155-165)
__global__ __launch_bounds__(256) void coarse_pts_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                         const float *__restrict__ near_m, const float *__restrict__ far_m,
                                                         const float *__restrict__ lin_z, const float *__restrict__ noise, int n_rays, int T0,
                                                         float bound, int perturb, float *__restrict__ pts)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rays * T0) return;
    const int ray = idx / T0, i = idx - ray * T0;
    const float ox = rays_o[3 * ray], oy = rays_o[3 * ray + 1], oz = rays_o[3 * ray + 2];
    const float dx = rays_d[3 * ray], dy = rays_d[3 * ray + 1], dz = rays_d[3 * ray + 2];
    float near, far;
    cube_near_far(ox, oy, oz, dx, dy, dz, bound, near, far);
    if (near_m) {
        const float nm = near_m[ray], fm = far_m[ray];
        if (!is_inf(nm)) near = nm;
        if (!is_inf(fm)) far = fm;
    }
    const float span = far - near;
    const float sample_dist = span / (float)T0;
    float zi = near + span * lin_z[i];
    if (perturb) zi = zi + (noise[idx] - 0.5f) * sample_dist;
    pts[3 * (size_t)idx] = ox + dx * zi; pts[3 * (size_t)idx + 1] = oy + dy * zi; pts[3 * (size_t)idx + 2] = oz + dz * zi;
}

static int check_render_args(const char *who, const ac_render_opts *op, const float *rays_o, const float *rays_d, const float *noise,
                             const float *lin_z, const float *lin_u, const ac_render_out *out)
{
    if (op->num_steps % 16 || op->upsample_steps % 16 || op->num_steps < 16 || op->num_steps > 64 ||
        op->upsample_steps < 0 || op->num_steps + op->upsample_steps > MAXT) {
        ac::set_error("%s: num_steps=%d upsample_steps=%d unsupported (multiples of 16, num_steps<=64, sum<=128)", who,
                      op->num_steps, op->upsample_steps);
        return AC_ERR_BAD_ARG;
    }
    if (!(op->fd_eps > 0.0f)) {      // the reference stops here too: `assert (gradient == gradient).all()` after dividing by eps = 0 (instant_nsr.py:274)
        ac::set_error("%s: fd_eps = %g (normal_epsilon_ratio >= 1): the finite-difference normals divide by it, it must be positive", who, (double)op->fd_eps);
        return AC_ERR_BAD_ARG;
    }
    if (op->n_rays <= 0) return AC_OK;
    if (!rays_o || !rays_d || !lin_z || (op->upsample_steps && !lin_u) || (op->perturb && !noise) || !out->image ||
        !out->weights_sum || !out->depth || !out->normal_map || !out->eik) {
        ac::set_error("%s: NULL buffer", who); return AC_ERR_BAD_ARG;
    }
    return AC_OK;
}

static int fill_render_args(RenderArgs &a, const ac_field *field, const ac_render_opts *op, const float *rays_o, const float *rays_d,
                            const float *bg, const float *noise, const float *lin_z, const float *lin_u, const ac_render_out *out)
{
    if (int rc = fill_args(a, field, op->bound)) return rc;
    a.rays_o = rays_o; a.rays_d = rays_d; a.bg = bg; a.noise = noise; a.lin_z = lin_z; a.lin_u = lin_u;
    a.out = *out;
    a.n_rays = op->n_rays; a.T0 = op->num_steps; a.nup = op->upsample_steps / 16;
    a.pair_n = 0; a.ex_from = 0; a.ex_rows = op->n_rays;
    a.inv_s = op->inv_s; a.inv_s_dev = op->inv_s_dev; a.car = op->cos_anneal_ratio; a.one_m_car = (float)(1.0 - (double)op->cos_anneal_ratio);
    a.eps = op->fd_eps; a.perturb = op->perturb;
    if (op->precision != 0 && op->precision != 1) { ac::set_error("ac_render_opts: precision %d unknown (0 = exact, 1 = fast)", op->precision); return AC_ERR_BAD_ARG; }
    a.fast = op->precision;
    if (op->skip_masked != 0 && op->skip_masked != 1) { ac::set_error("ac_render_opts: skip_masked must be 0 or 1"); return AC_ERR_BAD_ARG; }
    a.skip_masked = op->skip_masked;
    if (op->opacity_only != 0 && op->opacity_only != 1) { ac::set_error("ac_render_opts: opacity_only must be 0 or 1"); return AC_ERR_BAD_ARG; }
    a.opacity_only = op->opacity_only;
    if ((op->near_m != nullptr) != (op->far_m != nullptr)) { ac::set_error("ac_render_opts: near_m and far_m go together"); return AC_ERR_BAD_ARG; }
    a.near_m = op->near_m; a.far_m = op->far_m;
    for (int j = 0; j < 4; ++j) {           // finite-difference reach in cells, per gather round (see encode_stencil)
        a.jfine[j] = 0;
        for (int g = 0; g < 4; ++g) {
            const double cells = (double)op->fd_eps / (double)a.two_bound * (double)a.lvl[4 * j + g].scale;
            if (!(cells * 1.001 + 1e-3 < 1.0)) a.jfine[j] = 1;
        }
    }
    return AC_OK;
}

// Per-launch scratch of the dynamic hand-out: [8 XCDs][8 segments] work counters (256 B) | finished-workgroup counter | flags [N] u32 | state [N][SEG_STATE] f32.
// One slot per (device, stream), grown to the largest batch it has served and kept for the life of the process: launches of one stream run in order, so a
// slot is never in use by two launches at once, however many streams render concurrently.
#ifndef AC_RAY_SEGMENTS
#define AC_RAY_SEGMENTS 4           // segments a ray is cut into (1 = whole rays as work items, rounds 1 - 2; at most 8)
#endif
struct SegSlot { char *p; size_t bytes; uint32_t gen; uint64_t last_use; };
constexpr size_t SEG_POOL_MAX = 32;        // (device, stream) slots kept; beyond that the least recently used one is freed (stream churn must not grow device memory without bound)
static std::mutex g_seg_mu;
static std::map<std::pair<int, hipStream_t>, SegSlot> g_seg_pool;
static uint64_t g_seg_clock = 0;
// -> the slot's memory and the generation of this launch (1 .. 2^28 - 1): a slot is zeroed when it is allocated; after that every launch leaves its
// counters at zero (the kernel's last workgroup re-arms them) and tags its per-ray flags with its generation, so nothing is cleared between launches
static char *seg_scratch(size_t need, uint32_t &gen, hipStream_t stream)
{
    std::mutex &mu = g_seg_mu;
    auto &pool = g_seg_pool;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(dev, stream);
    if (pool.find(key) == pool.end() && pool.size() >= SEG_POOL_MAX) {   // evict the least recently used slot (hipFree waits for the device: nothing can still use it)
        auto victim = pool.begin();
        for (auto it = pool.begin(); it != pool.end(); ++it) if (it->second.last_use < victim->second.last_use) victim = it;
        if (victim->second.p) (void)hipFree(victim->second.p);
        pool.erase(victim);
    }
    SegSlot &sl = pool[key];
    sl.last_use = ++g_seg_clock;
    if (sl.bytes < need) {
        if (sl.p) (void)hipFree(sl.p);                                   // (synchronises the device: no launch can still be using the slot)
        sl.p = nullptr; sl.bytes = 0; sl.gen = 0;
        const size_t want = (need + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        if (hipMalloc(reinterpret_cast<void **>(&sl.p), want) != hipSuccess) { sl.p = nullptr; return nullptr; }
        // (on the launch's own stream: ordered before the kernel that is about to use the slot -- a null-stream memset is not, for non-blocking streams)
        if (hipMemsetAsync(sl.p, 0, want, stream) != hipSuccess) { (void)hipFree(sl.p); sl.p = nullptr; return nullptr; }
        sl.bytes = want;
    }
    sl.gen = (sl.gen + 1u) & 0x0fffffffu;
    if (sl.gen == 0u) {                                                  // wrapped: flags of 2^28 launches ago could match again -- start over from zeroed memory
        if (hipMemsetAsync(sl.p, 0, sl.bytes, stream) != hipSuccess) return nullptr;
        sl.gen = 1u;
    }
    gen = sl.gen;
    return sl.p;
}

// hand-offs that timed out (a taker waited ~1 s for a segment that was never published: its pixel is NaN) on the slot of (current device, stream), over
// the life of that slot; waits for the stream.  0 on a healthy run -- the GPU tier asserts it after its soak and co-residency tests.
AC_API int ac_render_handoff_timeouts(ac_stream_t stream, uint32_t *count)
{
    if (!count) { ac::set_error("render_handoff_timeouts: NULL count"); return AC_ERR_BAD_ARG; }
    *count = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    char *p = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_seg_mu);
        auto it = g_seg_pool.find(std::make_pair(dev, (hipStream_t)stream));
        if (it != g_seg_pool.end()) p = it->second.p;
    }
    if (!p) return AC_OK;                                                // no render has run on this stream
    if (hipMemcpyAsync(count, p + 260, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { ac::set_error("render_handoff_timeouts: copy failed"); return AC_ERR_LAUNCH; }
    return AC_OK;
}

template <int MODE, bool FAST, bool EX, bool SH = false>
static void launch_render_p(const RenderArgs &a, hipStream_t stream)
{
    int blocks = (a.n_rays + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    static uint64_t seen = 0;                       // one flag per instantiation
    const size_t lds_bytes = LDS_FLOATS * sizeof(float);
    ac::allow_dynamic_lds(seen, reinterpret_cast<const void *>(render_rays_kernel<MODE, FAST, EX, SH>), lds_bytes);
#if AC_DYNAMIC_RAYS
    RenderArgs b = a;
    {
        // segments: the tiles of a ray in seg_n nearly equal runs; the sampling stage (about 1.6 tiles' worth of time) rides with the first
        const int nt = (a.T0 + 16 * a.nup) / 16;
        // (posed space: the final pass stays with whole rays -- with skip_masked most of its tiles are skipped anyway, and segments measured 8 % slower there)
        int sn = (MODE != MODE_FULL) ? 1 : (nt < AC_RAY_SEGMENTS ? nt : AC_RAY_SEGMENTS);
        if (sn < 1) sn = 1;
        b.seg_n = sn; b.seg_cb = 0;
        for (int q = 0; q <= sn; ++q) b.seg_cb |= (uint64_t)((nt * q) / sn) << (4 * q);
        const size_t N = (size_t)a.n_rays, head = 512, flags = (N * 4 + 255) & ~(size_t)255;      // head: [0, 256) work counters | [256] finished workgroups
        const size_t need = head + (sn > 1 ? flags + N * SEG_STATE * sizeof(float) : 0);
        uint32_t gen = 0;
        char *sc = seg_scratch(need, gen, stream);
        b.ray_counter = reinterpret_cast<uint32_t *>(sc);
        b.done_counter = sc ? reinterpret_cast<uint32_t *>(sc + 256) : nullptr;
        b.handoff_timeouts = sc ? reinterpret_cast<uint32_t *>(sc + 260) : nullptr;      // (never re-armed: counts over the life of the slot)
        b.gen = gen;
        b.eik_red = (MODE == MODE_UPSAMPLE) ? nullptr : a.out.eik_reduced;
        b.seg_flags = sn > 1 ? reinterpret_cast<uint32_t *>(sc + head) : nullptr;
        b.seg_state = sn > 1 ? reinterpret_cast<float *>(sc + head + flags) : nullptr;
        const int cus = (int)ac::cu_count();
        if (blocks > cus) blocks = cus;
        blocks = (blocks + 7) & ~7;                                      // every XCD gets the same number of workgroups
        if (!sc) blocks = 0;                                             // (the scratch could not be allocated: an empty grid is a launch error the caller reports)
    }
    hipLaunchKernelGGL((render_rays_kernel<MODE, FAST, EX, SH>), dim3(blocks), dim3(BLOCK), lds_bytes, stream, b);
#else
    hipLaunchKernelGGL((render_rays_kernel<MODE, FAST, EX, SH>), dim3(blocks), dim3(BLOCK), lds_bytes, stream, a);
    if (MODE != MODE_UPSAMPLE && a.out.eik_reduced) {                 // (static ray assignment builds: the reduction as its own launch(es))
        const int nred = a.pair_n ? 2 : 1, nper = a.pair_n ? a.pair_n : a.n_rays;
        for (int q = 0; q < nred; ++q)
            hipLaunchKernelGGL(eikonal_reduce_kernel, dim3(1), dim3(1024), 0, stream, a.out.eik + (size_t)q * nper * 2, nper, a.out.eik_reduced + 2 * q, 1);
    }
#endif
}
static bool wants_samples(const ac_render_out &o)
{
    return o.z_vals || o.weights || o.alpha || o.color || o.sdf || o.gradient || o.ss_inds || o.sort_index || o.sdf_out16 || o.pts || o.feat7;
}
template <int MODE>
static void launch_render(const RenderArgs &a, hipStream_t stream)
{
    const bool ex = wants_samples(a.out);
    if constexpr (MODE != MODE_UPSAMPLE) {          // (the sampling-only launch has no finite-difference stage)
        if (a.Wsh) {                                // a field with view directions: its own instantiations (see the kernel's SH parameter)
            if (a.fast) { if (ex) launch_render_p<MODE, true, true, true>(a, stream); else launch_render_p<MODE, true, false, true>(a, stream); return; }
            if (ex) launch_render_p<MODE, false, true, true>(a, stream); else launch_render_p<MODE, false, false, true>(a, stream);
            return;
        }
        if (a.fast) { if (ex) launch_render_p<MODE, true, true>(a, stream); else launch_render_p<MODE, true, false>(a, stream); return; }
        if (ex) launch_render_p<MODE, false, true>(a, stream); else launch_render_p<MODE, false, false>(a, stream);
        return;
    }
    if (a.out.ss_inds || a.out.sort_index) launch_render_p<MODE, false, true>(a, stream);      // (the sampling-only launch can export the sample indices)
    else launch_render_p<MODE, false, false>(a, stream);
}

AC_API int ac_render_rays(const ac_field *field, const ac_render_opts *op, const float *rays_o, const float *rays_d,
                          const float *bg, const float *noise, const float *lin_z, const float *lin_u,
                          const ac_render_out *out, ac_stream_t stream)
{
    if (!op || !out) { ac::set_error("render_rays: NULL opts/out"); return AC_ERR_BAD_ARG; }
    if (int rc = check_render_args("render_rays", op, rays_o, rays_d, noise, lin_z, lin_u, out)) return rc;
    if (op->n_rays <= 0) return AC_OK;
    RenderArgs a{};
    if (int rc = fill_render_args(a, field, op, rays_o, rays_d, bg, noise, lin_z, lin_u, out)) return rc;
#ifdef AC_PROFILE
    a.prof = g_prof;
#endif
    launch_render<MODE_FULL>(a, (hipStream_t)stream);
    return ac::check_launch("render_rays");
}

// The same N rays rendered twice in one launch, with two draws of the jitter noise and two backgrounds: what one stylisation step does with net_style
// (stylize.py:98-116 render_val under no_grad, then :143-152 the differentiable render of the same rays) -- two calls of run() in the reference, two launches
// of ac_render_rays before round 3.  The 2N work items are handed out as a0 b0 a1 b1 ...: the two copies of a ray walk through (nearly) the same grid
// cells at (nearly) the same time on the same XCD, so the second finds most table sectors in L2 (-11 % on the stride-4 training view, where neighbouring
// rays share little: profiles/r03_experiments.txt).  Every result is bit-identical to two separate launches (the rays are independent).
// noise [2][N][num_steps], bg [2][N][3] (or NULL); out: the per-ray arrays hold 2N rows ([0, N) copy a, [N, 2N) copy b); the optional per-sample
// arrays are written for copy b only and hold N rows.
AC_API int ac_render_rays_pair(const ac_field *field, const ac_render_opts *op, const float *rays_o, const float *rays_d,
                               const float *bg2, const float *noise2, const float *lin_z, const float *lin_u,
                               const ac_render_out *out, ac_stream_t stream)
{
    if (!op || !out) { ac::set_error("render_rays_pair: NULL opts/out"); return AC_ERR_BAD_ARG; }
    if (int rc = check_render_args("render_rays_pair", op, rays_o, rays_d, noise2, lin_z, lin_u, out)) return rc;
    if (op->n_rays <= 0) return AC_OK;
    if (op->n_rays > (1 << 29)) { ac::set_error("render_rays_pair: too many rays"); return AC_ERR_BAD_ARG; }
    RenderArgs a{};
    if (int rc = fill_render_args(a, field, op, rays_o, rays_d, bg2, noise2, lin_z, lin_u, out)) return rc;
    a.pair_n = op->n_rays; a.n_rays = 2 * op->n_rays; a.ex_from = op->n_rays; a.ex_rows = op->n_rays;
#ifdef AC_PROFILE
    a.prof = nullptr;
#endif
    launch_render<MODE_FULL>(a, (hipStream_t)stream);
    return ac::check_launch("render_rays_pair");
}

// the sampling stage alone (coarse z, coarse sdf, NeuS up-sampling): what the reference computes under no_grad before the
// differentiable render core (instant_nsr.py:155-184)
AC_API int ac_sample_rays(const ac_field *field, const ac_render_opts *op, const float *rays_o, const float *rays_d, const float *noise,
                          const float *lin_z, const float *lin_u, float *z_vals, ac_stream_t stream)
{
    if (!op || !z_vals) { ac::set_error("sample_rays: NULL opts/z_vals"); return AC_ERR_BAD_ARG; }
    ac_render_out dummy{};
    float sink = 0.0f;                       // check_render_args wants the mandatory outputs non-NULL; this mode never writes them
    dummy.image = dummy.weights_sum = dummy.depth = dummy.normal_map = dummy.eik = &sink;
    if (int rc = check_render_args("sample_rays", op, rays_o, rays_d, noise, lin_z, lin_u, &dummy)) return rc;
    if (op->n_rays <= 0) return AC_OK;
    RenderArgs a{};
    if (int rc = fill_render_args(a, field, op, rays_o, rays_d, nullptr, noise, lin_z, lin_u, &dummy)) return rc;
    a.out = ac_render_out{};
    a.zbuf = z_vals;
    launch_render<MODE_UPSAMPLE>(a, (hipStream_t)stream);
    return ac::check_launch("sample_rays");
}

// scratch layout of ac_render_rays_warped (byte offsets, 256-byte aligned): near_m, far_m [N] f32; pts, can [N,T,3] f32;
// mask [N,T] u8; zbuf [N,T] f32
AC_API size_t ac_render_rays_warped_scratch(int32_t n_rays, int32_t T, size_t offs[6])
{
    const size_t N = n_rays > 0 ? (size_t)n_rays : 0, NT = N * (size_t)(T > 0 ? T : 0);
    const size_t sz[6] = { N * 4, N * 4, NT * 12, NT * 12, NT, NT * 4 };
    size_t o = 0;
    for (int i = 0; i < 6; ++i) { if (offs) offs[i] = o; o += (sz[i] + 255) & ~(size_t)255; }
    return o + ((N + 255) & ~(size_t)255);                     // + ray_dead [N] u8 (skip_masked), behind the six documented segments
}

static int warp_any(const ac_warp_mesh *m, const float *pts, uint32_t P, float *can, uint8_t *mask, ac_stream_t stream, int skip_far = 0,
                    const uint8_t *ray_dead = nullptr, uint32_t spr = 1, uint32_t seed_off = 0)
{
    if (m->accel) {
        // temporal seeds (ac_warp_mesh.seed_faces): this search's columns [seed_off, seed_off + spr) of the caller's per-ray rows
        int32_t *ts = (m->seed_faces && m->seed_stride >= seed_off + spr && spr > 0) ? m->seed_faces : nullptr;
        return ac::warp_samples_accel_impl(pts, m->verts, m->faces, m->T, P, m->V, m->F, m->threshold, m->accel, nullptr, can, nullptr, nullptr, nullptr,
                                           mask, stream, skip_far, ray_dead, spr, ts, m->seed_stride, seed_off);
    }
    return ac_warp_samples(pts, m->verts, m->faces, m->T, P, m->V, m->F, m->threshold, nullptr, can, nullptr, nullptr, nullptr, mask, stream);
}

// Measurement hook (bench.py's posed-frame roofline): with ac_debug_warped_phases(1) every ac_render_rays_warped call records HIP events on its stream
// at the phase boundaries -- [0] start, [1] near / far + coarse points + ray cull, [2] first warp search, [3] up-sampling pass, [4] second warp search,
// [5] final pass -- and ac_debug_warped_phase_ms() returns the five intervals of the LAST call in ms (it waits for that call).  Off by default.
namespace {
hipEvent_t g_phase_ev[6];
int g_phase_on = 0, g_phase_have = 0;
void phase_mark(int k, hipStream_t st) { if (g_phase_on) { (void)hipEventRecord(g_phase_ev[k], st); if (k == 5) g_phase_have = 1; } }
}
AC_API void ac_debug_warped_phases(int enable)
{
    if (enable && !g_phase_on) for (auto &e : g_phase_ev) (void)hipEventCreate(&e);
    if (!enable && g_phase_on) { for (auto &e : g_phase_ev) (void)hipEventDestroy(e); g_phase_have = 0; }
    g_phase_on = enable ? 1 : 0;
}
AC_API int ac_debug_warped_phase_ms(float out[5])
{
    if (!g_phase_on || !g_phase_have) { ac::set_error("ac_debug_warped_phase_ms: no instrumented ac_render_rays_warped call yet"); return AC_ERR_BAD_ARG; }
    if (hipEventSynchronize(g_phase_ev[5]) != hipSuccess) { ac::set_error("ac_debug_warped_phase_ms: event wait failed"); return AC_ERR_LAUNCH; }
    for (int k = 0; k < 5; ++k) { out[k] = 0.0f; (void)hipEventElapsedTime(out + k, g_phase_ev[k], g_phase_ev[k + 1]); }
    return AC_OK;
}

AC_API int ac_render_rays_warped(const ac_field *field, const ac_render_opts *op, const float *rays_o, const float *rays_d,
                                 const float *bg, const float *noise, const float *lin_z, const float *lin_u,
                                 const ac_warp_mesh *mesh, void *scratch, size_t scratch_bytes, const ac_render_out *out,
                                 ac_stream_t stream)
{
    if (!op || !out || !mesh) { ac::set_error("render_rays_warped: NULL opts/out/mesh"); return AC_ERR_BAD_ARG; }
    if (int rc = check_render_args("render_rays_warped", op, rays_o, rays_d, noise, lin_z, lin_u, out)) return rc;
    if (op->n_rays <= 0) return AC_OK;
    if (!mesh->verts || !mesh->faces || !mesh->T || mesh->V == 0 || mesh->F == 0) {
        ac::set_error("render_rays_warped: NULL mesh buffer or empty mesh"); return AC_ERR_BAD_ARG;
    }
    const int N = op->n_rays, T0 = op->num_steps, T = T0 + op->upsample_steps;
    size_t offs[6];
    const size_t need = ac_render_rays_warped_scratch(N, T, offs);
    if (!scratch || scratch_bytes < need) {
        ac::set_error("render_rays_warped: scratch of %zu bytes needed, %zu given", need, scratch_bytes); return AC_ERR_BAD_ARG;
    }
    char *sc = static_cast<char *>(scratch);
    float *near_m = reinterpret_cast<float *>(sc + offs[0]), *far_m = reinterpret_cast<float *>(sc + offs[1]);
    float *pts = reinterpret_cast<float *>(sc + offs[2]), *can = reinterpret_cast<float *>(sc + offs[3]);
    uint8_t *mask = reinterpret_cast<uint8_t *>(sc + offs[4]);
    float *zbuf = reinterpret_cast<float *>(sc + offs[5]);
    const uint8_t *ray_dead = nullptr;
    hipStream_t st = (hipStream_t)stream;
    RenderArgs a{};
    if (int rc = fill_render_args(a, field, op, rays_o, rays_d, bg, noise, lin_z, lin_u, out)) return rc;
    phase_mark(0, st);
    if (mesh->use_mesh_guide) {
        if (int rc = ac_mesh_near_far(rays_o, rays_d, mesh->verts, (uint32_t)N, mesh->V, mesh->geo_threshold, near_m, far_m, stream)) return rc;
        a.near_m = near_m; a.far_m = far_m;
    }
    a.zbuf = zbuf; a.mid_pts = pts;
    if (op->upsample_steps > 0) {                                 // coarse samples -> canonical space (:166-172)
        hipLaunchKernelGGL(coarse_pts_kernel, dim3((N * T0 + 255) / 256), dim3(256), 0, st, rays_o, rays_d, a.near_m, a.far_m, lin_z, noise, N, T0,
                           op->bound, op->perturb, pts);
        if (int rc = ac::check_launch("render_rays_warped (coarse points)")) return rc;
        if (op->skip_masked && mesh->accel) {                     // rays that cannot hold an unmasked sample: no search, no field evaluation
            uint8_t *rdead = reinterpret_cast<uint8_t *>(sc + ac_render_rays_warped_scratch(N, T, nullptr) - (((size_t)N + 255) & ~(size_t)255));
            if (int rc = ac::warp_ray_cull(pts, (uint32_t)N, (uint32_t)T0, mesh->threshold, mesh->accel, rdead, stream)) return rc;
            ray_dead = rdead;
        }
        phase_mark(1, st);
        if (int rc = warp_any(mesh, pts, (uint32_t)(N * T0), can, mask, stream, 0, ray_dead, (uint32_t)T0, 0u)) return rc;
    } else phase_mark(1, st);
    phase_mark(2, st);
    a.ext_pts = can;
    a.ray_dead = ray_dead;
    launch_render<MODE_UPSAMPLE>(a, st);                          // coarse sdf, up-sampling, mid points (posed space)
    if (int rc = ac::check_launch("render_rays_warped (up-sampling)")) return rc;
    phase_mark(3, st);
    // (skip_masked: the final pass does not evaluate masked-out samples, so the search may leave out those the cell grids prove masked)
    if (int rc = warp_any(mesh, pts, (uint32_t)(N * T), can, mask, stream, op->skip_masked, ray_dead, (uint32_t)T, (uint32_t)T0)) return rc;     // :198-203
    phase_mark(4, st);
    a.mask = mask;
    launch_render<MODE_FINAL>(a, st);
    phase_mark(5, st);
    return ac::check_launch("render_rays_warped");
}

static_assert((OFF_RWAVE + CF_OVERLAY) * sizeof(float) <= AC_FIELD_PREPARED_BYTES && OFF_C1F % 4 == 0 && OFF_B1 % 4 == 0 && OFF_RWAVE % 4 == 0, "the prepared image fits its buffer");
AC_API int ac_field_prepare(const ac_field *field, void *prepared, ac_stream_t stream)
{
    if (!prepared) { ac::set_error("field_prepare: NULL buffer"); return AC_ERR_BAD_ARG; }
    RenderArgs a{};
    if (int rc = fill_args(a, field, 1.0f)) return rc;
    a.T0 = 0; a.lin_z = nullptr; a.lin_u = nullptr;            // the sampling tables are per launch, not part of the image
    const size_t lds_bytes = OFF_RWAVE * sizeof(float);
    static uint64_t seen = 0;
    ac::allow_dynamic_lds(seen, reinterpret_cast<const void *>(field_prepare_kernel), lds_bytes);
    hipLaunchKernelGGL(field_prepare_kernel, dim3(1), dim3(BLOCK), lds_bytes, (hipStream_t)stream, a, static_cast<float *>(prepared));
    return ac::check_launch("field_prepare");
}

AC_API int ac_eikonal_reduce(const float *eik, int32_t n_rays, float *result, ac_stream_t stream)
{
    if (!result || n_rays < 0 || (!eik && n_rays > 0)) { ac::set_error("eikonal_reduce: bad argument"); return AC_ERR_BAD_ARG; }
    hipLaunchKernelGGL(eikonal_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, eik, n_rays, result, 0);
    return ac::check_launch("eikonal_reduce");
}

AC_API int ac_eikonal_reduce2(const float *eik, int32_t n_rays, float *result2, ac_stream_t stream)
{
    if (!result2 || n_rays < 0 || (!eik && n_rays > 0)) { ac::set_error("eikonal_reduce2: bad argument"); return AC_ERR_BAD_ARG; }
    hipLaunchKernelGGL(eikonal_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, eik, n_rays, result2, 1);
    return ac::check_launch("eikonal_reduce2");
}

AC_API int ac_field_sdf(const ac_field *field, const float *x, uint32_t B, float bound, float *out16, ac_stream_t stream)
{
    if (B == 0) return AC_OK;
    if (!x || !out16) { ac::set_error("field_sdf: NULL buffer"); return AC_ERR_BAD_ARG; }
    RenderArgs a{};
    if (int rc = fill_args(a, field, bound)) return rc;
    a.T0 = 0;
    const size_t lds_bytes = OFF_WAVE * sizeof(float);
    const uint32_t ntiles = (B + 15) / 16;
    uint32_t blocks = (ntiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(field_sdf_kernel, dim3(blocks), dim3(BLOCK), lds_bytes, (hipStream_t)stream, a, x, B, out16);
    return ac::check_launch("field_sdf");
}

AC_API int ac_field_color_dirs(const ac_field *field, const float *x, const float *dirs, const float *n, const float *sdfout, uint32_t B, float *rgb,
                               ac_stream_t stream)
{
    if (B == 0) return AC_OK;
    if (!x || !n || !sdfout || !rgb) { ac::set_error("field_color: NULL buffer"); return AC_ERR_BAD_ARG; }
    RenderArgs a{};
    if (int rc = fill_args(a, field, 1.0f)) return rc;
    if (a.Wsh && !dirs) { ac::set_error("field_color: the field has view-direction weights (ac_field.Wc1_sh): pass the directions (ac_field_color_dirs)"); return AC_ERR_BAD_ARG; }
    if (!a.Wsh) dirs = nullptr;
    a.T0 = 0;
    const size_t lds_bytes = (OFF_WAVE + (dirs ? WAVES_PER_BLOCK * 1024 : 0)) * sizeof(float);
    static uint64_t seen = 0;
    ac::allow_dynamic_lds(seen, reinterpret_cast<const void *>(field_color_kernel), (OFF_WAVE + WAVES_PER_BLOCK * 1024) * sizeof(float));
    const uint32_t ntiles = (B + 15) / 16;
    uint32_t blocks = (ntiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(field_color_kernel, dim3(blocks), dim3(BLOCK), lds_bytes, (hipStream_t)stream, a, x, dirs, n, sdfout, B, rgb);
    return ac::check_launch("field_color");
}

AC_API int ac_field_color(const ac_field *field, const float *x, const float *n, const float *sdfout, uint32_t B, float *rgb,
                          ac_stream_t stream)
{
    return ac_field_color_dirs(field, x, nullptr, n, sdfout, B, rgb, stream);
}
