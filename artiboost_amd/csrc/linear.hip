// Small-batch fp32 linear layers (the 512 -> 256 -> 128 -> 6 box-rotation head, anakin/models/mlp.py:11-25): M = batch
// rows (64) is far too small for the tiled conv kernels (a 1x1 conv GEMM with a 32-step K loop costs ~14 us of pure
// latency per layer here).  Plain FMA kernels, exact fp32, one launch per product:
//   fwd   y[m][n]  = act( sum_k x[m][k] * W[n][k] + b[n] )
//   dgrad gx[m][k] = ( sum_n g[m][n] * W[n][k] ) * (act_out[m][k] > 0 if masked)
//   wgrad dW[n][k] = sum_m g[m][n] * x[m][k],   db[n] = sum_m g[m][n]
// W is [N][K] row-major (the OHWI layout of a 1x1 conv).  K % 4 == 0.
#include "common.h"

// grid (ceil(N/4), ceil(M/64)); block 256: lane -> row (64 rows), wave -> one of 4 output columns.  W[n][k..k+3] is a
// wave-uniform (broadcast) load, x rows are L1-resident after the first touch of each 128-byte line.
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ bias, int M, int N, int K, int relu,
                                                         float* __restrict__ y) {
    const int m = blockIdx.y * 64 + (threadIdx.x & 63), n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M || n >= N) return;
    const float4* xr = (const float4*)(x + (long)m * K);
    const float4* wr = (const float4*)(W + (long)n * K);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int k = 0; k < K / 4; ++k) {
        float4 xv = xr[k], wv = wr[k];
        a0 += xv.x * wv.x; a1 += xv.y * wv.y; a2 += xv.z * wv.z; a3 += xv.w * wv.w;
    }
    float v = (a0 + a1) + (a2 + a3) + (bias ? bias[n] : 0.f);
    if (relu) v = fmaxf(v, 0.f);
    y[(long)m * N + n] = v;
}

// grid (ceil(K/256), M); block 64 lanes x 4 k each: W rows read coalesced along k, g[m][n] broadcast
__global__ __launch_bounds__(64) void linear_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ W,
                                                          const float* __restrict__ act_out, int M, int N, int K,
                                                          float* __restrict__ gx) {
    const int m = blockIdx.y, k4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (k4 >= K) return;
    const float* gr = g + (long)m * N;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int n = 0; n < N; ++n) {
        const float gv = gr[n];
        const float4 wv = *(const float4*)(W + (long)n * K + k4);
        a.x += gv * wv.x; a.y += gv * wv.y; a.z += gv * wv.z; a.w += gv * wv.w;
    }
    if (act_out) {
        const float4 o = *(const float4*)(act_out + (long)m * K + k4);
        if (!(o.x > 0.f)) a.x = 0.f; if (!(o.y > 0.f)) a.y = 0.f; if (!(o.z > 0.f)) a.z = 0.f; if (!(o.w > 0.f)) a.w = 0.f;
    }
    *(float4*)(gx + (long)m * K + k4) = a;
}

// grid (ceil(K/256), N); block 64 lanes x 4 k each; rows summed in order (deterministic).  Lane 0 of block x == 0 also
// writes db[n].
__global__ __launch_bounds__(64) void linear_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ x, int M, int N,
                                                          int K, float* __restrict__ dW, float* __restrict__ db) {
    const int n = blockIdx.y, k4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (k4 < K) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int m = 0; m < M; ++m) {
            const float gv = g[(long)m * N + n];
            const float4 xv = *(const float4*)(x + (long)m * K + k4);
            a.x += gv * xv.x; a.y += gv * xv.y; a.z += gv * xv.z; a.w += gv * xv.w;
        }
        *(float4*)(dW + (long)n * K + k4) = a;
    }
    if (db && blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int m = 0; m < M; ++m) s += g[(long)m * N + n];
        db[n] = s;
    }
}

extern "C" int ab_linear_fwd(const float* x, const float* w, const float* bias, int M, int N, int K, int relu, float* y,
                             void* stream) {
    if (!x || !w || !y || M < 1 || N < 1 || K < 4) return AB_EINVAL;
    if (K % 4) return AB_ESHAPE;
    linear_fwd_kernel<<<dim3((N + 3) / 4, (M + 63) / 64), 256, 0, as_stream(stream)>>>(x, w, bias, M, N, K, relu, y);
    AB_LAUNCH_CHECK(); return 0;
}
extern "C" int ab_linear_dgrad(const float* g, const float* w, const float* act_out, int M, int N, int K, float* gx,
                               void* stream) {
    if (!g || !w || !gx || M < 1 || N < 1 || K < 4) return AB_EINVAL;
    if (K % 4) return AB_ESHAPE;
    linear_dgrad_kernel<<<dim3((K + 255) / 256, M), 64, 0, as_stream(stream)>>>(g, w, act_out, M, N, K, gx);
    AB_LAUNCH_CHECK(); return 0;
}
extern "C" int ab_linear_wgrad(const float* g, const float* x, int M, int N, int K, float* dw, float* db, void* stream) {
    if (!g || !x || !dw || M < 1 || N < 1 || K < 4) return AB_EINVAL;
    if (K % 4) return AB_ESHAPE;
    linear_wgrad_kernel<<<dim3((K + 255) / 256, N), 64, 0, as_stream(stream)>>>(g, x, M, N, K, dw, db);
    AB_LAUNCH_CHECK(); return 0;
}
