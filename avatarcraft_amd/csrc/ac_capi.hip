// avatarcraft_amd/csrc/ac_capi.hip -- library identification and error reporting of the C ABI.
#include "ac_common.hpp"

namespace ac {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
}  // namespace ac

AC_API int ac_version(void) { return 7; }
AC_API const char *ac_last_error(void) { return ac::g_err; }

AC_API void ac_hash_level_table(uint32_t L, float S, uint32_t H, float *scale_host, uint32_t *res_host)
{
    for (uint32_t l = 0; l < L; ++l) {
        float sc = ac::exp2_f32((float)l * S) * (float)H - 1.0f;
        scale_host[l] = sc;
        res_host[l] = (uint32_t)ceilf(sc) + 1u;
    }
}
