/* TEST INFRASTRUCTURE -- a plain-C host of the C ABI (no Python, no torch): what a non-PyTorch embedder of the hot path binds.
 * Links libartiboost_hip.so + the HIP runtime only; allocates with hipMalloc, launches on its own stream, checks against a C
 * restatement of the reference arithmetic (anakin/models/simplebaseline.py:16-71,183-189: softmax over D*H*W per class, confidence =
 * max probability, /(sum + 1e-7), 3-D integral) written here from the formulas, and exercises the fp32 -> (hi, lo) bf16 split used by
 * the bf16x3 convolutions (hi + lo == v to 2^-17).
 *   hipcc -x c ... is not needed: gcc tests/c_host/abi_host_test.c -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ \
 *        -L artiboost_amd -lartiboost_hip -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,... -o abi_host_test                       */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "artiboost_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define AB(x) do { int r_ = (x); if (r_ != 0) { printf("FAIL %s -> %d\n", #x, r_); return 3; } } while (0)

static float bf16_to_f32(uint16_t h) { union { uint32_t u; float f; } c; c.u = (uint32_t)h << 16; return c.f; }

int main(void) {
    const int B = 3, C = 5, D = 7, DP = 8, H = 6, W = 4;       /* depth pitch 8 > depth 7: padded slots must be ignored */
    const int CD = C * DP, npix = H * W;
    const size_t n = (size_t)B * npix * CD;
    float* h_logits = (float*)malloc(n * sizeof(float));
    uint32_t s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h_logits[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * 6.0f; }
    printf("abi version %d\n", ab_abi_version());
    hipStream_t st; CK(hipStreamCreate(&st));
    float *d_logits, *d_part, *d_uvd, *d_conf, *d_stat;
    const int nt = ab_softargmax3d_ntiles(H, W);
    CK(hipMalloc((void**)&d_logits, n * 4)); CK(hipMalloc((void**)&d_part, (size_t)B * nt * C * 8 * 4));
    CK(hipMalloc((void**)&d_uvd, (size_t)B * C * 3 * 4)); CK(hipMalloc((void**)&d_conf, (size_t)B * C * 4)); CK(hipMalloc((void**)&d_stat, (size_t)B * C * 2 * 4));
    CK(hipMemcpyAsync(d_logits, h_logits, n * 4, hipMemcpyHostToDevice, st));
    AB(ab_softargmax3d_fwd(d_logits, 0 /* AB_DT_F32 */, B, C, D, DP, H, W, d_part, d_uvd, d_conf, d_stat, st));
    float uvd[3 * 5 * 3], conf[3 * 5];
    CK(hipMemcpyAsync(uvd, d_uvd, sizeof(uvd), hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(conf, d_conf, sizeof(conf), hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    double worst = 0;
    for (int b = 0; b < B; ++b) for (int c = 0; c < C; ++c) {
        double m = -1e30, sum = 0, su = 0, sv = 0, sd = 0;
        for (int p = 0; p < npix; ++p) for (int d = 0; d < D; ++d) { double x = h_logits[((size_t)b * npix + p) * CD + c * DP + d]; if (x > m) m = x; }
        for (int p = 0; p < npix; ++p) for (int d = 0; d < D; ++d) {
            double e = exp((double)h_logits[((size_t)b * npix + p) * CD + c * DP + d] - m);
            sum += e; su += e * ((double)(p % W) / W); sv += e * ((double)(p / W) / H); sd += e * ((double)d / D);
        }
        const double z = 1.0 + 1e-7, ref[3] = {su / sum / z, sv / sum / z, sd / sum / z};
        for (int k = 0; k < 3; ++k) { double e = fabs(uvd[(b * C + c) * 3 + k] - ref[k]); if (e > worst) worst = e; }
        { double e = fabs(conf[b * C + c] - 1.0 / sum); if (e > worst) worst = e; }
    }
    printf("softargmax3d_fwd: worst abs deviation from the C restatement %.3g\n", worst);
    if (!(worst < 2e-6)) { printf("FAIL softargmax parity\n"); return 4; }
    /* ---- ab_split_f32: hi = bf16(v), lo = bf16(v - hi) */
    const long ns = (long)(n / 8) * 8;
    uint16_t *d_hi, *d_lo; CK(hipMalloc((void**)&d_hi, ns * 2)); CK(hipMalloc((void**)&d_lo, ns * 2));
    AB(ab_split_f32(d_logits, ns, d_hi, d_lo, st));
    uint16_t* hi = (uint16_t*)malloc(ns * 2); uint16_t* lo = (uint16_t*)malloc(ns * 2);
    CK(hipMemcpyAsync(hi, d_hi, ns * 2, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(lo, d_lo, ns * 2, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    double wrel = 0;
    for (long i = 0; i < ns; ++i) {
        double v = h_logits[i], r = (double)bf16_to_f32(hi[i]) + (double)bf16_to_f32(lo[i]);
        double e = fabs(r - v) / (fabs(v) + 1e-30); if (e > wrel) wrel = e;
    }
    printf("split_f32: worst relative |hi + lo - v| / |v| = %.3g (2^-17 = 7.6e-6)\n", wrel);
    if (!(wrel < 1.6e-5)) { printf("FAIL split precision\n"); return 5; }
    /* ---- argument errors are reported, not crashed on */
    if (ab_softargmax3d_fwd(NULL, 0, B, C, D, DP, H, W, d_part, d_uvd, d_conf, d_stat, st) == 0) { printf("FAIL: NULL input accepted\n"); return 6; }
    printf("C_ABI_HOST_OK\n");
    return 0;
}
