// avatarcraft_amd/csrc/ac_capi.hip -- library identification and error reporting of the C ABI.
#include "ac_common.hpp"

namespace ac {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
}  // namespace ac

AC_API int ac_version(void) { return 10; }
AC_API const char *ac_last_error(void) { return ac::g_err; }

AC_API void ac_hash_level_table(uint32_t L, float S, uint32_t H, float *scale_host, uint32_t *res_host)
{
    for (uint32_t l = 0; l < L; ++l) {
        float sc = ac::exp2_f32((float)l * S) * (float)H - 1.0f;
        scale_host[l] = sc;
        res_host[l] = (uint32_t)ceilf(sc) + 1u;
    }
}

// test utility (include/avatarcraft_hip.h): a foreign workload that HOLDS compute units -- every workgroup spins on the 100 MHz wall clock
__global__ __launch_bounds__(1024) void hold_cus_kernel(unsigned long long ticks, uint32_t *sink)
{
    extern __shared__ uint32_t hold_lds[];
    if (threadIdx.x == 0) hold_lds[0] = blockIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(100);
    if (sink && hold_lds[0] == 0xffffffffu) *sink = 1u;       // (keeps the LDS allocation alive)
}
AC_API int ac_debug_hold_cus(uint32_t blocks, uint32_t lds_bytes, uint32_t millis, ac_stream_t stream)
{
    if (blocks == 0 || lds_bytes > 160 * 1024 - 64) { ac::set_error("ac_debug_hold_cus: blocks == 0 or more LDS than a compute unit has"); return AC_ERR_BAD_ARG; }
    static uint64_t seen = 0;
    ac::allow_dynamic_lds(seen, reinterpret_cast<const void *>(hold_cus_kernel), 160 * 1024 - 64);
    hipLaunchKernelGGL(hold_cus_kernel, dim3(blocks), dim3(1024), lds_bytes < 4 ? 4 : lds_bytes, (hipStream_t)stream, (unsigned long long)millis * 100000ull, (uint32_t *)nullptr);
    return ac::check_launch("ac_debug_hold_cus");
}
