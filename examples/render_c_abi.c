/* examples/render_c_abi.c -- the drop-in boundary used from plain C: no Python, no torch, only the HIP runtime's C API and include/avatarcraft_hip.h.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/render_c_abi.c avatarcraft_amd/libavatarcraft_hip.so \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$PWD/avatarcraft_amd -o render_c_abi
 *   ./render_c_abi field_and_rays.bin image.bin
 *
 * Renders N rays of NeRFRenderer.run (models/instant_nsr.py:133-299, eval mode, white background) with ONE ac_render_rays launch -- what a maintainer's C / C++
 * / cgo / JNI binding of the reference's renderer would call.  Input blob (little endian; written by tests/test_gpu_c_abi.py):
 *   int32 n_rays, num_steps, upsample_steps, H;  int32 offsets[17];  float S, bound, inv_s;
 *   float table[offsets[16] * 2], W1[64*35], b1[64], W2[16*64], b2[16], Wc1[64*21], Wc2[64*64], Wc3[3*64];
 *   float rays_o[n*3], rays_d[n*3], lin_z[num_steps], lin_u[16]
 * Output blob: float image[n*3], weights_sum[n], depth[n], normal_map[n*3], eik[n*2]. */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "avatarcraft_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_AC(x) do { int r_ = (x); if (r_ != AC_OK) { fprintf(stderr, "%s: %d %s\n", #x, r_, ac_last_error()); return 3; } } while (0)

static float *upload(FILE *f, size_t n)
{
    float *h = (float *)malloc(n * sizeof(float)), *d = NULL;
    if (!h || fread(h, sizeof(float), n, f) != n) { fprintf(stderr, "short read (%zu floats)\n", n); exit(4); }
    if (hipMalloc((void **)&d, n * sizeof(float)) != hipSuccess || hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        fprintf(stderr, "upload of %zu floats failed\n", n); exit(5);
    }
    free(h);
    return d;
}

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s input.bin output.bin\n", argv[0]); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t hdr[4], offsets[17];
    float fl[3];
    if (fread(hdr, 4, 4, f) != 4 || fread(offsets, 4, 17, f) != 17 || fread(fl, 4, 3, f) != 3) { fprintf(stderr, "short header\n"); return 4; }
    const int32_t n = hdr[0], num_steps = hdr[1], upsample_steps = hdr[2];

    ac_field field = { 0 };
    field.table = upload(f, (size_t)offsets[16] * 2);
    for (int i = 0; i < 17; ++i) field.offsets[i] = offsets[i];
    field.S = fl[0];
    field.H = (uint32_t)hdr[3];
    field.W1 = upload(f, 64 * 35); field.b1 = upload(f, 64); field.W2 = upload(f, 16 * 64); field.b2 = upload(f, 16);
    field.Wc1 = upload(f, 64 * 21); field.Wc2 = upload(f, 64 * 64); field.Wc3 = upload(f, 3 * 64);
    const float *rays_o = upload(f, (size_t)n * 3), *rays_d = upload(f, (size_t)n * 3);
    const float *lin_z = upload(f, (size_t)num_steps), *lin_u = upload(f, 16);
    fclose(f);

    void *prepared = NULL;                                   /* the weights in the renderer's LDS order, once per parameter version */
    CHECK_HIP(hipMalloc(&prepared, AC_FIELD_PREPARED_BYTES));
    CHECK_AC(ac_field_prepare(&field, prepared, NULL));
    field.prepared = prepared;

    ac_render_opts opts = { 0 };
    opts.n_rays = n; opts.num_steps = num_steps; opts.upsample_steps = upsample_steps;
    opts.bound = fl[1]; opts.inv_s = fl[2]; opts.cos_anneal_ratio = 1.0f; opts.fd_eps = 0.005f;      /* normal_epsilon_ratio = 0 */

    ac_render_out out = { 0 };
    const size_t sizes[5] = { (size_t)n * 3, (size_t)n, (size_t)n, (size_t)n * 3, (size_t)n * 2 };
    float *dev[5];
    for (int i = 0; i < 5; ++i) CHECK_HIP(hipMalloc((void **)&dev[i], sizes[i] * sizeof(float)));
    out.image = dev[0]; out.weights_sum = dev[1]; out.depth = dev[2]; out.normal_map = dev[3]; out.eik = dev[4];

    CHECK_AC(ac_render_rays(&field, &opts, rays_o, rays_d, NULL /* white */, NULL /* no jitter */, lin_z, lin_u, &out, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    uint32_t timeouts = 0;
    CHECK_AC(ac_render_handoff_timeouts(NULL, &timeouts));
    if (timeouts) { fprintf(stderr, "%u segment hand-offs timed out\n", timeouts); return 6; }

    FILE *g = fopen(argv[2], "wb");
    if (!g) { perror(argv[2]); return 1; }
    for (int i = 0; i < 5; ++i) {
        float *h = (float *)malloc(sizes[i] * sizeof(float));
        CHECK_HIP(hipMemcpy(h, dev[i], sizes[i] * sizeof(float), hipMemcpyDeviceToHost));
        fwrite(h, sizeof(float), sizes[i], g);
        free(h);
    }
    fclose(g);
    printf("ac_version %d: %d rays x (%d + %d) samples rendered through the C ABI\n", ac_version(), n, num_steps, upsample_steps);
    return 0;
}
