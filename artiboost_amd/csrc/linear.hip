// Small-batch fp32 linear layers (the 512 -> 256 -> 128 -> 6 box-rotation head, anakin/models/mlp.py:11-25): M = batch
// rows (64) is far too small for the tiled conv kernels (a 1x1 conv GEMM with a 32-step K loop costs ~14 us of pure
// latency per layer here).  Plain FMA kernels, exact fp32, one launch per product:
//   fwd   y[m][n]  = act( sum_k x[m][k] * W[n][k] + b[n] )
//   dgrad gx[m][k] = ( sum_n g[m][n] * Wt[k][n] ) * (act_out[m][k] > 0 if masked)      (Wt = W transposed, [K][N])
//   wgrad dW[n][k] = sum_m g[m][n] * x[m][k],   db[n] = sum_m g[m][n]
// W is [N][K] row-major (the OHWI layout of a 1x1 conv).  Reduction lengths are multiples of 4.
#include "common.h"

// C[i][j] = epilogue( sum_r A[i][r] * B[j][r] ), both operands reduction-contiguous.  One wave per 4 x 4 micro-tile: the
// lanes split the reduction (float4 slices, fully coalesced 1-KiB row loads, all 16+ loads of a lane independent), sixteen
// wave reductions finish it.  Block = 4 waves = 4 rows x 16 columns; grid (ceil(J/16), ceil(I/4)).  Epilogue: + bias[j],
// per-column affine (an eval-mode BatchNorm1d), + residual[i][j], ReLU / leaky ReLU, or zero where mask[i][j] <= 0.
struct LinEpi { const float* scale; const float* shift; const float* residual; int ldr; float slope; int ldc; };
__global__ __launch_bounds__(256) void linear_nt_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                        const float* __restrict__ bias, const float* __restrict__ mask,
                                                        int I, int J, int R, int relu, float* __restrict__ C,
                                                        LinEpi ep = LinEpi{nullptr, nullptr, nullptr, 0, 0.f, 0}) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = blockIdx.y * 4, j0 = (blockIdx.x * 4 + wave) * 4;
    if (j0 >= J) return;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    for (int q = lane * 4; q < R; q += 256) {
        float4 av[4], bv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) av[a] = *(const float4*)(A + (long)min(i0 + a, I - 1) * R + q);
#pragma unroll
        for (int b = 0; b < 4; ++b) bv[b] = *(const float4*)(B + (long)min(j0 + b, J - 1) * R + q);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[a][b] += (av[a].x * bv[b].x + av[a].y * bv[b].y) + (av[a].z * bv[b].z + av[a].w * bv[b].w);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = wave_sum(acc[a][b]);
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int i = i0 + a, j = j0 + b;
                if (i >= I || j >= J) continue;
                float v = acc[a][b] + (bias ? bias[j] : 0.f);
                if (ep.scale) v = v * ep.scale[j] + ep.shift[j];
                if (ep.residual) v += ep.residual[(long)i * ep.ldr + j];
                if (relu == 1) v = fmaxf(v, 0.f);
                else if (relu == 2) v = v > 0.f ? v : v * ep.slope;
                if (mask && !(mask[(long)i * J + j] > 0.f)) v = 0.f;
                C[(long)i * (ep.ldc ? ep.ldc : J) + j] = v;
            }
    }
}

// grid (ceil(K/256), N); block 64 lanes x 4 k each; the M rows are summed in order (deterministic), 16 rows of loads in
// flight.  Lane 0 of block x == 0 also writes db[n].
__global__ __launch_bounds__(64) void linear_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ x, int M, int N,
                                                          int K, float* __restrict__ dW, float* __restrict__ db) {
    const int n = blockIdx.y, k4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (k4 < K) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int m0 = 0; m0 < M; m0 += 16) {          // 16 rows of loads issued before any is consumed
            float gv[16]; float4 xv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int m = min(m0 + u, M - 1);
                gv[u] = (m0 + u < M) ? g[(long)m * N + n] : 0.f;
                xv[u] = *(const float4*)(x + (long)m * K + k4);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) { a.x += gv[u] * xv[u].x; a.y += gv[u] * xv[u].y; a.z += gv[u] * xv[u].z; a.w += gv[u] * xv[u].w; }
        }
        *(float4*)(dW + (long)n * K + k4) = a;
    }
    if (db && blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll 16
        for (int m = 0; m < M; ++m) s += g[(long)m * N + n];
        db[n] = s;
    }
}

extern "C" int ab_linear_fwd(const float* x, const float* w, const float* bias, int M, int N, int K, int relu, float* y,
                             void* stream) {
    if (!x || !w || !y || M < 1 || N < 1 || K < 4) return AB_EINVAL;
    if (K % 4) return AB_ESHAPE;
    linear_nt_kernel<<<dim3((N + 15) / 16, (M + 3) / 4), 256, 0, as_stream(stream)>>>(x, w, bias, nullptr, M, N, K, relu, y);
    AB_LAUNCH_CHECK(); return 0;
}
extern "C" int ab_linear_dgrad(const float* g, const float* wt, const float* act_out, int M, int N, int K, float* gx,
                               void* stream) {
    if (!g || !wt || !gx || M < 1 || N < 4 || K < 1) return AB_EINVAL;
    if (N % 4) return AB_ESHAPE;
    linear_nt_kernel<<<dim3((K + 15) / 16, (M + 3) / 4), 256, 0, as_stream(stream)>>>(g, wt, nullptr, act_out, M, K, N, 0, gx);
    AB_LAUNCH_CHECK(); return 0;
}
extern "C" int ab_linear_wgrad(const float* g, const float* x, int M, int N, int K, float* dw, float* db, void* stream) {
    if (!g || !x || !dw || M < 1 || N < 1 || K < 4) return AB_EINVAL;
    if (K % 4) return AB_ESHAPE;
    linear_wgrad_kernel<<<dim3((K + 255) / 256, N), 64, 0, as_stream(stream)>>>(g, x, M, N, K, dw, db);
    AB_LAUNCH_CHECK(); return 0;
}

// y[m][n] = act(((x W^T + bias) * scale[n] + shift[n]) + residual[m][n]); act 0 none, 1 ReLU, 2 leaky ReLU(slope).
// scale/shift (both or neither) and residual (row pitch ldr) may be NULL; ldy = row pitch of y (>= N), so a layer can
// write a column block of a wider feature matrix.  The MLP of the grasp refiner (anakin/artiboost/refiner.py:227-319).
extern "C" int ab_linear_fused(const float* x, const float* w, const float* bias, const float* scale, const float* shift,
                               const float* residual, int ldr, int M, int N, int K, int act, float slope, float* y, int ldy,
                               void* stream) {
    if (!x || !w || !y || M < 1 || N < 1 || K < 4 || ldy < N || act < 0 || act > 2) return AB_EINVAL;
    if ((scale == nullptr) != (shift == nullptr) || (residual && ldr < N)) return AB_EINVAL;
    if (K % 4) return AB_ESHAPE;
    LinEpi ep{scale, shift, residual, ldr, slope, ldy};
    linear_nt_kernel<<<dim3((N + 15) / 16, (M + 3) / 4), 256, 0, as_stream(stream)>>>(x, w, bias, nullptr, M, N, K, act, y, ep);
    AB_LAUNCH_CHECK(); return 0;
}
