/* artiboost_hip.h -- C ABI of libartiboost_hip.so (hand-written HIP kernels for gfx950 / MI355X).
 *
 * The reference (lixiny/ArtiBoost) is 100 % Python and has no FFI of its own; every "kernel" it runs lives in a
 * third-party wheel (cuDNN via torch, pyrender/OpenGL, manotorch).  Each entry point below therefore names the
 * reference *call site* whose arithmetic it replaces (paths relative to the reference root).  INTEGRATION.md shows
 * the ctypes stub a maintainer would add on the reference side.
 *
 * Conventions
 *  - plain pointers + sizes only; all pointers are DEVICE pointers unless the name ends in _host
 *  - every function enqueues work on `stream` (a hipStream_t passed as void*) and returns immediately
 *  - return value: 0 on success, >0 = hipError_t from the launch, <0 = argument error (AB_E*)
 *  - no global state; re-entrant per stream
 *  - activations are NHWC ("pixels x channels"); `dtype`: 0 = float32, 1 = bfloat16 (raw uint16 bits)
 */
#ifndef ARTIBOOST_HIP_H
#define ARTIBOOST_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AB_DT_F32 0
#define AB_DT_BF16 1
#define AB_EINVAL (-1)
#define AB_ESHAPE (-2)
#define AB_EALIGN (-3)

/* library / device info: returns ABI version (integer, bumped on any signature change) */
int ab_abi_version(void);

/* ---- M3: fused softmax + 3-D integral (soft-argmax) head ------------------------------------------------------
 * replaces: norm_heatmap('softmax') + max + renorm + view_to_bcdhw + integral_heatmap3d
 *           anakin/models/simplebaseline.py:16-40, 43-71, 183-189
 * logits : [B, H, W, C*D] (NHWC of the reference's (B, C*D, H, W); channel = c*D + d), dtype f32|bf16
 * part   : workspace, float [B, ntile, C, 8]  (ntile from ab_softargmax3d_ntiles)
 * uvd    : float [B, C, 3]  (u = width, v = height, d = depth, each in [0,1))
 * conf   : float [B, C]     (max softmax probability)
 * stat   : float [B, C, 2]  (global max, sum exp(x - max)) kept for backward                                  */
int ab_softargmax3d_ntiles(int H, int W);
int ab_softargmax3d_fwd(const void* logits, int dtype, int B, int C, int D, int H, int W,
                        float* part, float* uvd, float* conf, float* stat, void* stream);
/* dlogits[b,h,w,c*D+d] = p * ( g_uvd . (coord - uvd) ) / (1+1e-7) + g_conf * conf * (argmax? 1 : 0 - p)
 * g_conf may be NULL.  dlogits has the dtype/layout of logits (may alias logits: in-place is allowed).        */
int ab_softargmax3d_bwd(const void* logits, int dtype, int B, int C, int D, int H, int W,
                        const float* uvd, const float* conf, const float* stat,
                        const float* g_uvd, const float* g_conf, void* dlogits, void* stream);

#ifdef __cplusplus
}
#endif
#endif
