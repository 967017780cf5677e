// Global-norm gradient clip + Adam on the flat parameter buffer (train_artiboost.py:91-96: clip_grad_norm_(0.001)
// then torch.optim.Adam(lr, wd=0).step(); utils/netutils.py:26-33).  Two HBM passes: (1) sum of squares -> per-block
// partials (deterministic order), (2) every thread re-reduces the (few hundred) partials, derives the clip
// coefficient and applies Adam; the same pass refreshes the bf16 copy of the weights used by the conv kernels.
#include "common.h"

#define OPT_THREADS 256
#define OPT_MAX_PARTS 1024

__global__ __launch_bounds__(OPT_THREADS) void sqnorm_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
    double s = 0.0;
    long nvec = n / 4;
    for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < nvec; i += (long)gridDim.x * OPT_THREADS) {
        float4 v = ((const float4*)g)[i];
        s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    __shared__ double sm[OPT_THREADS];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = OPT_THREADS / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = (float)sm[0];
}

// total_norm_out[0] = sqrt(sum partials)
__global__ void norm_finalize_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) out[0] = (float)sqrt(s);
}

__global__ __launch_bounds__(OPT_THREADS) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                float* __restrict__ m, float* __restrict__ v, long n,
                                                                const float* __restrict__ total_norm, float max_norm,
                                                                float lr, float b1, float b2, float eps, float bc1,
                                                                float bc2_sqrt, const float* __restrict__ hyper,
                                                                bf16_t* __restrict__ lp, bf16_t* __restrict__ lp_lo = nullptr) {
    if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2_sqrt = hyper[2]; }   // device-side step state (hipGraph replay)
    float coef = 1.f;
    if (max_norm > 0.f) { coef = max_norm / (total_norm[0] + 1e-6f); coef = coef > 1.f ? 1.f : coef; }
    const float step = lr / bc1;
    long nvec = n / 4;
    for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < nvec; i += (long)gridDim.x * OPT_THREADS) {
        float4 pv = ((float4*)p)[i], gv = ((const float4*)g)[i], mv = ((float4*)m)[i], vv = ((float4*)v)[i];
        float pp[4] = {pv.x, pv.y, pv.z, pv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
        float mm[4] = {mv.x, mv.y, mv.z, mv.w}, vvv[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gk = gg[k] * coef;
            mm[k] = mm[k] * b1 + (1.f - b1) * gk;
            vvv[k] = vvv[k] * b2 + (1.f - b2) * gk * gk;
            float denom = sqrtf(vvv[k]) / bc2_sqrt + eps;
            pp[k] = pp[k] - step * (mm[k] / denom);
        }
        ((float4*)p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        ((float4*)m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        ((float4*)v)[i] = make_float4(vvv[0], vvv[1], vvv[2], vvv[3]);
        if (lp) {
            uint2 o;
            o.x = pack_bf16x2(pp[0], pp[1]);
            o.y = pack_bf16x2(pp[2], pp[3]);
            ((uint2*)lp)[i] = o;
            if (lp_lo) {      // split-bf16 weights: lo = bf16(p - hi) (conv_x3.hip)
                uint2 l;
                l.x = pack_bf16x2(pp[0] - __uint_as_float(o.x << 16), pp[1] - __uint_as_float(o.x & 0xffff0000u));
                l.y = pack_bf16x2(pp[2] - __uint_as_float(o.y << 16), pp[3] - __uint_as_float(o.y & 0xffff0000u));
                ((uint2*)lp_lo)[i] = l;
            }
        }
    }
}

// total_norm: float[1] device scalar (output).  part: float[OPT_MAX_PARTS] workspace.  n % 4 == 0.
extern "C" int ab_grad_norm(const float* grad, long n, float* part, float* total_norm, void* stream) {
    if (!grad || !part || !total_norm) return AB_EINVAL;
    if (n % 4) return AB_ESHAPE;
    long b = (n / 4 + OPT_THREADS - 1) / OPT_THREADS;
    int nb = (int)(b > OPT_MAX_PARTS ? OPT_MAX_PARTS : (b < 1 ? 1 : b));
    sqnorm_kernel<<<nb, OPT_THREADS, 0, as_stream(stream)>>>(grad, n, part);
    AB_LAUNCH_CHECK();
    norm_finalize_kernel<<<1, 64, 0, as_stream(stream)>>>(part, nb, total_norm);
    AB_LAUNCH_CHECK();
    return 0;
}

// x *= s, in place: the 1 / world_size of the gradient average after a SUM all-reduce (train.allreduce_flat_).  RCCL's ReduceOp.AVG is
// PreMulSum, whose gfx950 ring kernels multiply with v_pk_mul_f32 -- and run on the comm stream beside this build's MFMA kernels (build.py on
// -packed-fp32-ops); its plain Sum ring kernels hold no packed fp32, and this library is compiled without it.
__global__ __launch_bounds__(OPT_THREADS) void scale_kernel(float* __restrict__ x, long n, float s, int head) {
    // [0, head) scalar up to the first 16-byte boundary, float4 body, scalar tail
    long nvec = (n - head) / 4;
    float4* xv = (float4*)(x + head);
    for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < nvec; i += (long)gridDim.x * OPT_THREADS) {
        float4 v = xv[i];
        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
        xv[i] = v;
    }
    if (blockIdx.x == 0) {
        if ((int)threadIdx.x < head) x[threadIdx.x] *= s;
        long t = head + nvec * 4 + threadIdx.x;
        if (threadIdx.x < 4 && t < n) x[t] *= s;
    }
}

extern "C" int ab_scale_f32(float* x, long n, float s, void* stream) {
    if (!x || n < 0 || (uintptr_t)x % 4) return AB_EINVAL;
    if (n == 0) return 0;
    int head = (int)(((16 - (uintptr_t)x % 16) % 16) / 4);
    if (head > n) head = (int)n;
    long b = ((n - head) / 4 + OPT_THREADS - 1) / OPT_THREADS;
    int nb = (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
    scale_kernel<<<nb, OPT_THREADS, 0, as_stream(stream)>>>(x, n, s, head);
    AB_LAUNCH_CHECK();
    return 0;
}

// step: 1-based.  max_norm <= 0 disables clipping.  lp (optional): bf16 copy of the updated params.
// hyper (optional, device float[3] = {lr, 1-beta1^step, sqrt(1-beta2^step)}) overrides lr/step so that a captured
// hipGraph can be replayed with per-step values.
static int clip_adam_impl(float* param, const float* grad, float* m, float* v, long n, const float* total_norm, float max_norm, float lr,
                          float beta1, float beta2, float eps, int step, const float* hyper, void* lp, void* lp_lo, void* stream);

extern "C" int ab_clip_adam(float* param, const float* grad, float* m, float* v, long n, const float* total_norm,
                            float max_norm, float lr, float beta1, float beta2, float eps, int step,
                            const float* hyper, void* lp, void* stream) {
    return clip_adam_impl(param, grad, m, v, n, total_norm, max_norm, lr, beta1, beta2, eps, step, hyper, lp, nullptr, stream);
}

// as ab_clip_adam, refreshing the split-bf16 weight planes (hi, lo) of the updated parameters in the same pass
extern "C" int ab_clip_adam_x3(float* param, const float* grad, float* m, float* v, long n, const float* total_norm,
                               float max_norm, float lr, float beta1, float beta2, float eps, int step,
                               const float* hyper, void* lp_hi, void* lp_lo, void* stream) {
    if (!lp_hi || !lp_lo) return AB_EINVAL;
    return clip_adam_impl(param, grad, m, v, n, total_norm, max_norm, lr, beta1, beta2, eps, step, hyper, lp_hi, lp_lo, stream);
}

static int clip_adam_impl(float* param, const float* grad, float* m, float* v, long n, const float* total_norm, float max_norm, float lr,
                          float beta1, float beta2, float eps, int step, const float* hyper, void* lp, void* lp_lo, void* stream) {
    if (!param || !grad || !m || !v || (max_norm > 0.f && !total_norm)) return AB_EINVAL;
    if (n % 4 || step < 1) return AB_ESHAPE;
    double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    long b = (n / 4 + OPT_THREADS - 1) / OPT_THREADS;
    int nb = (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
    clip_adam_kernel<<<nb, OPT_THREADS, 0, as_stream(stream)>>>(param, grad, m, v, n, total_norm, max_norm, lr, beta1,
                                                                beta2, eps, (float)bc1, (float)sqrt(bc2), hyper, (bf16_t*)lp, (bf16_t*)lp_lo);
    AB_LAUNCH_CHECK();
    return 0;
}


// ---- measurement aid (bench.py's roofline leg): the device's constant-rate wall clock written to a slot by a one-thread launch.  Captured
// into the step's hipGraph around every conv-stack call, the slot differences are those launches' durations INSIDE the graph replay
// (rocprofv3 is not available in the driver's bench run; torch's external timing events are disallowed on ROCm).
__global__ void wall_stamp_kernel(int64_t* slot) { *slot = (int64_t)wall_clock64(); }
extern "C" int ab_wall_stamp(int64_t* slot, void* stream) {
    if (!slot) return AB_EINVAL;
    wall_stamp_kernel<<<1, 1, 0, as_stream(stream)>>>(slot);
    AB_LAUNCH_CHECK();
    return 0;
}
extern "C" int ab_wall_clock_khz(void) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
    return khz;
}
