// HBM-bound NHWC elementwise / reduction kernels around the conv stack: training-mode BatchNorm (statistics, apply,
// backward), ReLU and residual fusion, 3x3/2 max-pool, global average pool, precision packing.
// All tensors are [M = N*H*W pixels][C channels] with C % 8 == 0; every access is a 16-byte vector per lane.
#include "common.h"

template <typename T> struct Vec;
template <> struct Vec<float> { static constexpr int N = 4; };
template <> struct Vec<bf16_t> { static constexpr int N = 8; };

template <typename T> __device__ __forceinline__ void vload(const T* p, float* f);
template <> __device__ __forceinline__ void vload<float>(const float* p, float* f) {
    float4 v = *(const float4*)p; f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}
template <> __device__ __forceinline__ void vload<bf16_t>(const bf16_t* p, float* f) {
    uint4 v = *(const uint4*)p;
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
template <typename T> __device__ __forceinline__ void vstore(T* p, const float* f);
template <> __device__ __forceinline__ void vstore<float>(float* p, const float* f) {
    *(float4*)p = make_float4(f[0], f[1], f[2], f[3]);
}
template <> __device__ __forceinline__ void vstore<bf16_t>(bf16_t* p, const float* f) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
    *(uint4*)p = make_uint4(w[0], w[1], w[2], w[3]);
}

template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }

// rows of the [M][C] matrix reduced by one workgroup: ~1024 workgroups for large M, never fewer than 16 rows (the per-lane
// row loop is a chain of dependent-latency loads; at small M short chains in more workgroups win)
static inline int red_rows(long M) { long r = (M + 1023) / 1024; if (r < 16) r = 16; return (int)((r + 15) / 16 * 16); }

// ---------------------------------------------------------------- column partial sums: (sum x, sum x^2) per block
template <typename T>
__global__ __launch_bounds__(256) void col_stats_kernel(const T* __restrict__ x, long M, int C, int rows_per_block, float* __restrict__ part) {
    constexpr int V = Vec<T>::N;
    const int vc = C / V;                  // vector lanes along channels
    const int rl = 256 / vc;               // row lanes
    const int cv = threadIdx.x % vc, rr = threadIdx.x / vc;
    long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
    float s[V], q[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { s[i] = 0.f; q[i] = 0.f; }
    if (rr < rl)
        for (long r = r0 + rr; r < r1; r += rl) {
            float f[V]; vload<T>(x + r * C + cv * V, f);
#pragma unroll
            for (int i = 0; i < V; ++i) { s[i] += f[i]; q[i] += f[i] * f[i]; }
        }
    extern __shared__ float sm[];          // [rl][C][2]
    if (rr < rl)
#pragma unroll
        for (int i = 0; i < V; ++i) { sm[(rr * C + cv * V + i) * 2] = s[i]; sm[(rr * C + cv * V + i) * 2 + 1] = q[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < rl; ++k) { a += sm[(k * C + c) * 2]; b += sm[(k * C + c) * 2 + 1]; }
        part[((long)blockIdx.x * C + c) * 2] = a; part[((long)blockIdx.x * C + c) * 2 + 1] = b;
    }
}

// Fixed-order reduction of per-workgroup partials part[nparts][C][2] for CPW channels per workgroup of 1024 threads
// (CPW channel lanes x 1024/CPW partial lanes; eight independent loads per lane in flight, double accumulators).  Fewer
// channels per workgroup = more workgroups and fewer dependent load rounds per lane: these launches sit on the step's
// dependency chain, their latency -- not their bandwidth -- is what counts.  Returns the totals in (s, q) for pl == 0.
template <int CPW>
__device__ __forceinline__ void reduce_parts_1024(const float* __restrict__ part, int nparts, int C, int c, double& s, double& q,
                                                  double* sm /* [16][CPW][2] */) {
    constexpr int NPL = 1024 / CPW;
    const int cl = threadIdx.x % CPW, pl = threadIdx.x / CPW;
    double sa[4] = {0, 0, 0, 0}, qa[4] = {0, 0, 0, 0};
    if (c < C) {
        int k = pl;
        for (; k + 7 * NPL < nparts; k += 8 * NPL) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const float2*)(part + ((long)(k + u * NPL) * C + c) * 2);
#pragma unroll
            for (int u = 0; u < 8; ++u) { sa[u & 3] += v[u].x; qa[u & 3] += v[u].y; }
        }
        for (; k + NPL < nparts; k += 2 * NPL) {
            float2 a = *(const float2*)(part + ((long)k * C + c) * 2), b = *(const float2*)(part + ((long)(k + NPL) * C + c) * 2);
            sa[0] += a.x; qa[0] += a.y; sa[1] += b.x; qa[1] += b.y;
        }
        for (; k < nparts; k += NPL) { float2 a = *(const float2*)(part + ((long)k * C + c) * 2); sa[0] += a.x; qa[0] += a.y; }
    }
    // part-lanes of a wave by butterfly (lane = pl * CPW + cl: xor offsets CPW .. 32), the 16 waves through LDS in wave order
    double a = (sa[0] + sa[1]) + (sa[2] + sa[3]), b = (qa[0] + qa[1]) + (qa[2] + qa[3]);
#pragma unroll
    for (int o = 32; o >= CPW; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < CPW) { sm[(wave * CPW + cl) * 2] = a; sm[(wave * CPW + cl) * 2 + 1] = b; }
    __syncthreads();
    s = 0; q = 0;
    if (pl == 0) for (int k = 0; k < 16; ++k) { s += sm[(k * CPW + cl) * 2]; q += sm[(k * CPW + cl) * 2 + 1]; }
}
static inline int finalize_cpw(int C) {
    // channels per workgroup of the finalize launches: 2, 4 and 8 measure the same end to end (6.25-6.28 ms/step; the
    // launches sit at their ~5 us launch + cross-XCD read latency floor whatever their shape) -- 8 unless AB_BNFIN_CPW says so
    static const int force = getenv("AB_BNFIN_CPW") ? atoi(getenv("AB_BNFIN_CPW")) : 0;
    (void)C;
    return (force == 2 || force == 4) ? force : 8;
}

// ---------------------------------------------------------------- BN finalize: partials -> scale/shift, saved stats
// bnp: float [4][C] = scale (gamma*invstd), shift, mean, invstd.  running stats updated in place when non-null.
// block = CPW channels x 1024/CPW part-lanes, grid = ceil(C/CPW)
template <int CPW>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ part, int nparts, int C, long count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float momentum, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float* __restrict__ bnp) {
    const int cl = threadIdx.x % CPW, pl = threadIdx.x / CPW;
    const int c = blockIdx.x * CPW + cl;
    __shared__ double sm[16 * CPW * 2];
    // the finishing thread's parameters are requested before the reduction so their latency hides under it
    float g = 0.f, bt = 0.f, rm = 0.f, rv = 0.f;
    const bool fin = pl == 0 && c < C;
    if (fin) { g = gamma[c]; bt = beta[c]; if (running_mean) { rm = running_mean[c]; rv = running_var[c]; } }
    double s, q; reduce_parts_1024<CPW>(part, nparts, C, c, s, q, sm);
    if (fin) {
        double mean = s / (double)count;
        double var = q / (double)count - mean * mean; if (var < 0) var = 0;
        float invstd = (float)(1.0 / sqrt(var + (double)eps));
        float sc = g * invstd;
        bnp[c] = sc; bnp[C + c] = bt - (float)mean * sc; bnp[2 * C + c] = (float)mean; bnp[3 * C + c] = invstd;
        if (running_mean) {
            double unb = count > 1 ? var * (double)count / (double)(count - 1) : var;
            running_mean[c] = (1.f - momentum) * rm + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * rv + momentum * (float)unb;
        }
    }
}

// eval-mode: scale/shift from running statistics
__global__ void bn_eval_params_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, float* bnp) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float invstd = 1.f / sqrtf(rv[c] + eps);
    float sc = gamma[c] * invstd;
    bnp[c] = sc; bnp[C + c] = beta[c] - rm[c] * sc; bnp[2 * C + c] = rm[c]; bnp[3 * C + c] = invstd;
}

// every BatchNorm of a network in ONE launch: desc[n][5] = (gamma offset, beta offset in `flat`; running_mean, running_var offset in
// `stats`; output offset in `out`), C = desc[n][5 * .. ] -- see ab_bn_eval_params_batch
__global__ void bn_eval_params_batch_kernel(const float* __restrict__ flat, const float* __restrict__ stats, const int* __restrict__ desc,
                                            float eps, float* __restrict__ out) {
    const int* d = desc + blockIdx.y * 6;
    const int C = d[5];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float* gamma = flat + d[0]; const float* beta = flat + d[1]; const float* rm = stats + d[2]; const float* rv = stats + d[3];
    float* bnp = out + d[4];
    float invstd = 1.f / sqrtf(rv[c] + eps);
    float sc = gamma[c] * invstd;
    bnp[c] = sc; bnp[C + c] = beta[c] - rm[c] * sc; bnp[2 * C + c] = rm[c]; bnp[3 * C + c] = invstd;
}

// ---------------------------------------------------------------- BN apply (+residual) (+ReLU)
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ y, const T* __restrict__ res,
                                                       const float* __restrict__ bnp, long nvec, int C, int relu,
                                                       T* __restrict__ out) {
    constexpr int V = Vec<T>::N;
    // grid stride (gridDim*256 vectors) is a multiple of C/V whenever 256*V % C == 0: the thread's channels are fixed
    const bool fixed = ((256 * V) % C) == 0;
    float sc[V], sh[V];
    if (fixed) {
        int c = (int)(((long)threadIdx.x * V) % C);
#pragma unroll
        for (int k = 0; k < V; ++k) { sc[k] = bnp[c + k]; sh[k] = bnp[C + c + k]; }
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        long e = i * V;
        if (!fixed) {
            int c = (int)(e % C);
#pragma unroll
            for (int k = 0; k < V; ++k) { sc[k] = bnp[c + k]; sh[k] = bnp[C + c + k]; }
        }
        float f[V]; vload<T>(y + e, f);
        float r[V];
        if (res) vload<T>(res + e, r);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float v = f[k] * sc[k] + sh[k];
            if (res) v += r[k];
            if (relu) v = fmaxf(v, 0.f);
            f[k] = v;
        }
        vstore<T>(out + e, f);
    }
}

// gradient of the 3x3/2 max-pool input at pixel (n,h,w), channels cv*V..: sum over the <= 4 windows that cover the pixel
// of dpool where the recorded winner is this pixel (what maxpool_bwd_kernel stores, rounded to T the same way)
template <typename T>
__device__ __forceinline__ void pool_gather(const uint8_t* __restrict__ idx, const T* __restrict__ dpool, int n, int h, int w,
                                            int H, int W, int C, int cv, float* acc) {
    constexpr int V = Vec<T>::N;
    const int Ho = H / 2, Wo = W / 2;
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    for (int ho = h / 2; ho <= (h + 1) / 2; ++ho) {
        if (ho >= Ho) continue;
        int dh = h - (ho * 2 - 1);
        for (int wo = w / 2; wo <= (w + 1) / 2; ++wo) {
            if (wo >= Wo) continue;
            int code = dh * 3 + (w - (wo * 2 - 1));
            long o = (((long)n * Ho + ho) * Wo + wo) * C + cv * V;
            uint8_t am[V];
            if constexpr (V == 8) {
                uint2 q = *(const uint2*)(idx + o);
#pragma unroll
                for (int k = 0; k < 4; ++k) { am[k] = (q.x >> (8 * k)) & 0xff; am[4 + k] = (q.y >> (8 * k)) & 0xff; }
            } else {
                uint32_t q = *(const uint32_t*)(idx + o);
#pragma unroll
                for (int k = 0; k < 4; ++k) am[k] = (q >> (8 * k)) & 0xff;
            }
            float g[V]; vload<T>(dpool + o, g);
#pragma unroll
            for (int k = 0; k < V; ++k) if (am[k] == code) acc[k] += g[k];
        }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = round_to<T>(acc[k]);
}

// ---------------------------------------------------------------- BN backward
// pass 1: dz = dout * (out > 0 if relu); partial sums of dz and dz * xhat per channel
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ out,
                                                            const T* __restrict__ y, const float* __restrict__ bnp,
                                                            long M, int C, int relu, int rows_per_block, float* __restrict__ part,
                                                            const uint8_t* __restrict__ pool_idx = nullptr, int pH = 0, int pW = 0,
                                                            const bf16_t* __restrict__ out_hi = nullptr) {
    constexpr int V = Vec<T>::N;
    // wide tensors (C / V > 256: the 2048-channel stage of the Bottleneck ResNets): gridDim.y slices of CS = C / gridDim.y channels
    const int CS = C / (int)gridDim.y, vcs = CS / V, rl = 256 / vcs;
    const int cv = (int)blockIdx.y * vcs + (int)(threadIdx.x % vcs), rr = threadIdx.x / vcs;
    long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
    float s[V], q[V], mean[V], istd[V], sc[V], sh[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        s[i] = 0.f; q[i] = 0.f; mean[i] = bnp[2 * C + cv * V + i]; istd[i] = bnp[3 * C + cv * V + i];
        sc[i] = bnp[cv * V + i]; sh[i] = bnp[C + cv * V + i];
    }
    if (rr < rl)
        for (long r = r0 + rr; r < r1; r += rl) {
            long e = r * C + cv * V;
            float g[V], o[V], yy[V];
            if (pool_idx) {       // dout is the POOLED gradient: gather this pixel's share (fused max-pool backward)
                int w = (int)(r % pW); long t = r / pW; int h = (int)(t % pH); int n = (int)(t / pH);
                pool_gather<T>(pool_idx, dout, n, h, w, pH, pW, C, cv, g);
            } else vload<T>(dout + e, g);
            vload<T>(y + e, yy);
            if (relu == 1) {
                // out_hi (fp32 tensors only, V == 4): the ReLU mask from the hi plane of the split activation (same sign,
                // half the bytes of the fp32 copy)
                if (out_hi) {
                    const uint2 h = *(const uint2*)(out_hi + e);
                    o[0] = __uint_as_float(h.x << 16); o[1] = __uint_as_float(h.x & 0xffff0000u);
                    if constexpr (V >= 4) { o[2] = __uint_as_float(h.y << 16); o[3] = __uint_as_float(h.y & 0xffff0000u); }
                } else vload<T>(out + e, o);
            }
#pragma unroll
            for (int i = 0; i < V; ++i) {
                // relu == 2: the mask is recomputed from y with bn_apply's own expression (same f32 value that was
                // clamped and stored), which saves reading the activation
                const bool dead = relu == 1 ? !(o[i] > 0.f) : relu == 2 ? !(yy[i] * sc[i] + sh[i] > 0.f) : false;
                float dz = dead ? 0.f : g[i];
                s[i] += dz; q[i] += dz * ((yy[i] - mean[i]) * istd[i]);
            }
        }
    extern __shared__ float sm[];
    const int cl0 = (int)blockIdx.y * CS;                 // first channel of this slice
    if (rr < rl)
#pragma unroll
        for (int i = 0; i < V; ++i) { sm[(rr * CS + cv * V + i - cl0) * 2] = s[i]; sm[(rr * CS + cv * V + i - cl0) * 2 + 1] = q[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < CS; c += 256) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < rl; ++k) { a += sm[(k * CS + c) * 2]; b += sm[(k * CS + c) * 2 + 1]; }
        part[((long)blockIdx.x * C + cl0 + c) * 2] = a; part[((long)blockIdx.x * C + cl0 + c) * 2 + 1] = b;
    }
}

// reduce partials -> dgamma, dbeta (written to the flat grad buffer) and bwdp[2][C] = (sum dz, sum dz*xhat)
template <int CPW>
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nparts, int C,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ bwdp) {
    const int cl = threadIdx.x % CPW, pl = threadIdx.x / CPW;
    const int c = blockIdx.x * CPW + cl;
    __shared__ double sm[16 * CPW * 2];
    double s, q; reduce_parts_1024<CPW>(part, nparts, C, c, s, q, sm);
    if (pl == 0 && c < C) {
        dbeta[c] = (float)s; dgamma[c] = (float)q;
        bwdp[c] = (float)s; bwdp[C + c] = (float)q;
    }
}

// pass 2: dy = gamma*invstd * (dz - mean(dz) - xhat * mean(dz*xhat));  optionally also writes dz (residual gradient)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ out,
                                                           const T* __restrict__ y, const float* __restrict__ bnp,
                                                           const float* __restrict__ bwdp, long nvec, int C, long M,
                                                           int relu, T* __restrict__ dy, T* __restrict__ dz_out,
                                                           const uint8_t* __restrict__ pool_idx = nullptr, int pH = 0, int pW = 0) {
    constexpr int V = Vec<T>::N;
    const float invM = 1.f / (float)M;
    const bool fixed = ((256 * V) % C) == 0;
    float ga[V], sh[V], mu[V], is[V], k1[V], k2[V];    // gamma*invstd, shift, mean, invstd, mean(dz), mean(dz*xhat)
    auto loadp = [&](int c) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            ga[k] = bnp[c + k]; sh[k] = bnp[C + c + k]; mu[k] = bnp[2 * C + c + k]; is[k] = bnp[3 * C + c + k];
            k1[k] = bwdp[c + k] * invM; k2[k] = bwdp[C + c + k] * invM;
        }
    };
    if (fixed) loadp((int)(((long)threadIdx.x * V) % C));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        long e = i * V;
        if (!fixed) loadp((int)(e % C));
        float g[V], o[V], yy[V], d[V];
        if (pool_idx) {
            long r = e / C; int cvv = (int)(e - r * C) / V;
            int w = (int)(r % pW); long t = r / pW; int h = (int)(t % pH); int n = (int)(t / pH);
            pool_gather<T>(pool_idx, dout, n, h, w, pH, pW, C, cvv, g);
        } else vload<T>(dout + e, g);
        vload<T>(y + e, yy);
        if (relu == 1) vload<T>(out + e, o);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const bool dead = relu == 1 ? !(o[k] > 0.f) : relu == 2 ? !(yy[k] * ga[k] + sh[k] > 0.f) : false;
            float dz = dead ? 0.f : g[k];
            float xhat = (yy[k] - mu[k]) * is[k];
            d[k] = ga[k] * (dz - k1[k] - xhat * k2[k]);
            g[k] = dz;
        }
        vstore<T>(dy + e, d);
        if (dz_out) vstore<T>(dz_out + e, g);
    }
}

// ---------------------------------------------------------------- elementwise add (gradient accumulation)
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, long nvec, T* __restrict__ out) {
    constexpr int V = Vec<T>::N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        float x[V], y[V]; vload<T>(a + i * V, x); vload<T>(b + i * V, y);
#pragma unroll
        for (int k = 0; k < V; ++k) x[k] += y[k];
        vstore<T>(out + i * V, x);
    }
}

// ---------------------------------------------------------------- 3x3 / stride 2 / pad 1 max-pool (resnet.py:157)
// forward also records WHICH of the 9 taps won (first maximum in row-major scan order == torch semantics), one byte
// per output element; backward then gathers from the <= 4 windows covering an input pixel without re-reading x.
// bnp != nullptr: the input is a raw conv output and relu(x*scale+shift), rounded to T exactly as ab_bn_apply would store
// it, is pooled instead (the activation tensor is never materialised).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ bnp, int N, int H,
                                                          int W, int C, T* __restrict__ out, uint8_t* __restrict__ idx,
                                                          bf16_t* __restrict__ out_hi = nullptr, bf16_t* __restrict__ out_lo = nullptr) {
    constexpr int V = Vec<T>::N;
    const int Ho = H / 2, Wo = W / 2, vc = C / V;
    // grid (ceil(Wo*vc / 256), Ho, N): one output row per (blockIdx.y, blockIdx.z) -- no 64-bit div/mod per vector (they dominated the
    // flat-index version of this HBM-bound kernel)
    const unsigned col = blockIdx.x * 256 + threadIdx.x;
    if (col < (unsigned)(Wo * vc)) {
        const int wo = (int)(col / (unsigned)vc), cv = (int)(col - (unsigned)wo * vc);
        const int n = blockIdx.z, ho = blockIdx.y;
        const long i = (((long)n * Ho + ho) * Wo + wo) * vc + cv;
        float m[V]; uint8_t am[V];
        float sc[V], sh[V];
        if (bnp) {
#pragma unroll
            for (int k = 0; k < V; ++k) { sc[k] = bnp[cv * V + k]; sh[k] = bnp[C + cv * V + k]; }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) { m[k] = -INFINITY; am[k] = 0; }
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            int h = ho * 2 + dh - 1; if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                int w = wo * 2 + dw - 1; if ((unsigned)w >= (unsigned)W) continue;
                float f[V]; vload<T>(x + (((long)n * H + h) * W + w) * C + cv * V, f);
                if (bnp) {
#pragma unroll
                    for (int k = 0; k < V; ++k) f[k] = round_to<T>(fmaxf(f[k] * sc[k] + sh[k], 0.f));
                }
#pragma unroll
                for (int k = 0; k < V; ++k) if (f[k] > m[k]) { m[k] = f[k]; am[k] = (uint8_t)(dh * 3 + dw); }
            }
        }
        vstore<T>(out + i * V, m);
        if constexpr (V == 4) {
            if (out_hi) {        // fp32 path: the pooled tensor also as (hi, lo) planes for the split-bf16 convolutions that read it
                const uint32_t h0 = pack_bf16x2(m[0], m[1]), h1 = pack_bf16x2(m[2], m[3]);
                const uint32_t l0 = pack_bf16x2(m[0] - __uint_as_float(h0 << 16), m[1] - __uint_as_float(h0 & 0xffff0000u));
                const uint32_t l1 = pack_bf16x2(m[2] - __uint_as_float(h1 << 16), m[3] - __uint_as_float(h1 & 0xffff0000u));
                *(uint2*)(out_hi + i * 4) = make_uint2(h0, h1);
                *(uint2*)(out_lo + i * 4) = make_uint2(l0, l1);
            }
        }
        if (idx) {
            if constexpr (V == 8) {
                uint2 o; o.x = am[0] | (am[1] << 8) | (am[2] << 16) | ((uint32_t)am[3] << 24);
                o.y = am[4] | (am[5] << 8) | (am[6] << 16) | ((uint32_t)am[7] << 24);
                *(uint2*)(idx + i * V) = o;
            } else {
                *(uint32_t*)(idx + i * V) = am[0] | (am[1] << 8) | (am[2] << 16) | ((uint32_t)am[3] << 24);
            }
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const uint8_t* __restrict__ idx, const T* __restrict__ dout,
                                                          int N, int H, int W, int C, T* __restrict__ dx) {
    constexpr int V = Vec<T>::N;
    const int Ho = H / 2, Wo = W / 2, vc = C / V;
    // grid (ceil(W*vc / 256), H/2, N): a thread owns one channel vector of the input row PAIR (2p, 2p+1): row 2p lies in pooled
    // row p only, row 2p+1 in p and p+1 -- the winners / gradients of pooled row p are loaded once for both (4 loads of each kind
    // instead of 6, half the workgroups); 32-bit index math only
    const unsigned col = blockIdx.x * 256 + threadIdx.x;
    if (col < (unsigned)(W * vc)) {
        const int w = (int)(col / (unsigned)vc), cv = (int)(col - (unsigned)w * vc);
        const int n = blockIdx.z, p = blockIdx.y;
        float acc0[V], acc1[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { acc0[k] = 0.f; acc1[k] = 0.f; }
#pragma unroll
        for (int dp = 0; dp < 2; ++dp) {
            const int ho = p + dp;
            if (ho >= Ho) continue;
            for (int wo = w / 2; wo <= (w + 1) / 2; ++wo) {
                if (wo >= Wo) continue;
                const int cw = w - (wo * 2 - 1);
                const long o = (((long)n * Ho + ho) * Wo + wo) * C + cv * V;
                uint8_t am[V];
                if constexpr (V == 8) {
                    uint2 q = *(const uint2*)(idx + o);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { am[k] = (q.x >> (8 * k)) & 0xff; am[4 + k] = (q.y >> (8 * k)) & 0xff; }
                } else {
                    uint32_t q = *(const uint32_t*)(idx + o);
#pragma unroll
                    for (int k = 0; k < 4; ++k) am[k] = (q >> (8 * k)) & 0xff;
                }
                float g[V]; vload<T>(dout + o, g);
                // window rows of pooled row ho: 2ho-1 .. 2ho+1.  dp = 0: input row 2p is its row 1, row 2p+1 its row 2;
                // dp = 1: input row 2p+1 is row 0 of pooled row p+1
                if (dp == 0) {
#pragma unroll
                    for (int k = 0; k < V; ++k) { if (am[k] == 3 + cw) acc0[k] += g[k]; if (am[k] == 6 + cw) acc1[k] += g[k]; }
                } else {
#pragma unroll
                    for (int k = 0; k < V; ++k) if (am[k] == cw) acc1[k] += g[k];
                }
            }
        }
        const long i = (((long)n * H + 2 * p) * W + w) * vc + cv;
        vstore<T>(dx + i * V, acc0);
        vstore<T>(dx + (i + (long)W * vc) * V, acc1);
    }
}

// Backward of maxpool3x3/2(relu(bn(y))) for the fp32 path, first half: the row-pair max-pool backward above with the BatchNorm
// backward's mask + reduction riding on it.  A thread owns one channel vector of RP input row pairs; per row it gathers the
// pooled gradient, recomputes the ReLU mask from y (bn_apply's own expression), stores the MASKED gradient dz and keeps
// sum(dz), sum(dz * xhat) for its four channels; the 16 threads of a workgroup that share a channel vector combine through LDS:
// one partial row per workgroup.  Replaces the separate reduction pass over (da, y) -- 0.54 GB at the benchmark size.
template <int RP, bool WRITE_DZ = true>
__global__ __launch_bounds__(256) void pool_bwd_bn_reduce_kernel(const uint8_t* __restrict__ idx, const float* __restrict__ dout,
                                                                  const float* __restrict__ y, const float* __restrict__ bnp,
                                                                  int N, int H, int W, int C, float* __restrict__ dz,
                                                                  float* __restrict__ part) {
    const int Ho = H / 2, Wo = W / 2, vc = C / 4;
    const unsigned col = blockIdx.x * 256 + threadIdx.x;
    const bool ok = col < (unsigned)(W * vc);
    const int w = ok ? (int)(col / (unsigned)vc) : 0, cv = ok ? (int)(col - (unsigned)w * vc) : 0;
    const int n = blockIdx.z;
    float sc[4], sh[4], mean[4], istd[4], s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sc[k] = bnp[cv * 4 + k]; sh[k] = bnp[C + cv * 4 + k]; mean[k] = bnp[2 * C + cv * 4 + k]; istd[k] = bnp[3 * C + cv * 4 + k];
    }
    if (ok)
        for (int rp = 0; rp < RP; ++rp) {
            const int p = blockIdx.y * RP + rp;
            if (p >= Ho) break;
            float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dp = 0; dp < 2; ++dp) {
                const int ho = p + dp;
                if (ho >= Ho) continue;
                for (int wo = w / 2; wo <= (w + 1) / 2; ++wo) {
                    if (wo >= Wo) continue;
                    const int cw = w - (wo * 2 - 1);
                    const long o = (((long)n * Ho + ho) * Wo + wo) * C + cv * 4;
                    const uint32_t qd = *(const uint32_t*)(idx + o);
                    const float4 g4 = *(const float4*)(dout + o);
                    const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int am = (qd >> (8 * k)) & 0xff;
                        if (dp == 0) { if (am == 3 + cw) acc0[k] += g[k]; if (am == 6 + cw) acc1[k] += g[k]; }
                        else if (am == cw) acc1[k] += g[k];
                    }
                }
            }
            const long i = ((((long)n * H + 2 * p) * W + w) * vc + cv) * 4;
            const float4 y0 = *(const float4*)(y + i), y1 = *(const float4*)(y + i + (long)W * C);
            const float ya[4] = {y0.x, y0.y, y0.z, y0.w}, yb[4] = {y1.x, y1.y, y1.z, y1.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc0[k] = (ya[k] * sc[k] + sh[k] > 0.f) ? acc0[k] : 0.f;
                acc1[k] = (yb[k] * sc[k] + sh[k] > 0.f) ? acc1[k] : 0.f;
                s[k] += acc0[k]; q[k] += acc0[k] * ((ya[k] - mean[k]) * istd[k]);
                s[k] += acc1[k]; q[k] += acc1[k] * ((yb[k] - mean[k]) * istd[k]);
            }
            if constexpr (WRITE_DZ) {
                *(float4*)(dz + i) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
                *(float4*)(dz + i + (long)W * C) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
            }
        }
    // threads sharing cv: tid % vc (vc divides 256 for C = 64 .. 1024); fixed-order combine
    extern __shared__ float sm[];                          // [256][8]
#pragma unroll
    for (int k = 0; k < 4; ++k) { sm[threadIdx.x * 8 + k] = ok ? s[k] : 0.f; sm[threadIdx.x * 8 + 4 + k] = ok ? q[k] : 0.f; }
    __syncthreads();
    const long prow = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        const int cvv = c >> 2, k = c & 3;
        float a = 0.f, b = 0.f;
        // this workgroup's threads with channel vector cvv: t = cvv - first_cv (mod vc) + j * vc
        const unsigned col0 = blockIdx.x * 256;
        int t0 = (int)((cvv + vc - (int)(col0 % (unsigned)vc)) % vc);
        for (int t = t0; t < 256; t += vc) { a += sm[t * 8 + k]; b += sm[t * 8 + 4 + k]; }
        part[(prow * C + c) * 2] = a; part[(prow * C + c) * 2 + 1] = b;
    }
}

// ---------------------------------------------------------------- global average pool (resnet.py:219) fwd / bwd
// grid (N, ceil(C / (32*V))): 32 channel-vector lanes x 8 pixel lanes per workgroup, pixel-lane partials combined in
// fixed order through LDS
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, int HW, int C, float* __restrict__ out) {
    constexpr int V = Vec<T>::N;
    __shared__ float sm[8][32 * V + 1];
    const int n = blockIdx.x, cv = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int c0 = (blockIdx.y * 32 + cv) * V;
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    if (c0 < C)
        for (int p = pl; p < HW; p += 8) {
            float f[V]; vload<T>(x + ((long)n * HW + p) * C + c0, f);
#pragma unroll
            for (int k = 0; k < V; ++k) acc[k] += f[k];
        }
#pragma unroll
    for (int k = 0; k < V; ++k) sm[pl][cv * V + k] = acc[k];
    __syncthreads();
    for (int c = threadIdx.x; c < 32 * V; c += 256) {
        int cg = blockIdx.y * 32 * V + c;
        if (cg >= C) continue;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sm[k][c];
        out[(long)n * C + cg] = t / (float)HW;
    }
}
// dx[n,p,c] (+)= g[n,c] / HW  (one 16-byte vector per thread)
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ g, int HW, int C, long nvec,
                                                          T* __restrict__ dx, int accumulate) {
    constexpr int V = Vec<T>::N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        long e = i * V;
        int c = (int)(e % C); long n = e / ((long)HW * C);
        float v[V];
        if (accumulate) vload<T>(dx + e, v);
#pragma unroll
        for (int k = 0; k < V; ++k) { float t = g[n * C + c + k] / (float)HW; v[k] = accumulate ? v[k] + t : t; }
        vstore<T>(dx + e, v);
    }
}

// ---------------------------------------------------------------- precision / layout packing
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, long n, bf16_t* __restrict__ dst) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = f32_to_bf16(src[i]);
}
// src [O][K][I] (OHWI, K = kh*kw) -> dst [I][K][O], optionally converting to bf16
template <typename T>
__global__ void transpose_oki_kernel(const float* __restrict__ src, int O, int K, int I, T* __restrict__ dst) {
    long n = (long)O * K * I;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        int o = (int)(e % O); long t = e / O; int k = (int)(t % K); int i = (int)(t / K);
        st_f32(dst + e, src[((long)o * K + k) * I + i]);
    }
}
// All dgrad weight copies in one launch: tensor t = desc[t] (see ab_transpose_desc), a workgroup moves one 64(o) x 64(i)
// tile of one tap k through LDS: 256-byte coalesced row reads of src [O][K][I], 16-byte stores into dst [I][K][O].
// LO_OFF != 0 (T = bf16 only): also write the lo plane bf16(v - hi) at dst + lo_off elements (split-bf16 IHWO copies)
template <typename T>
__global__ __launch_bounds__(256) void transpose_oki_batch_kernel(const ab_transpose_desc* __restrict__ desc, int ntensors, long lo_off = 0) {
    __shared__ float sm[64][65];
    int t = 0;
    while (t + 1 < ntensors && (long)blockIdx.x >= desc[t + 1].tile_begin) ++t;
    const ab_transpose_desc d = desc[t];
    const int O = d.O, K = d.K, I = d.I;
    const int ti = (I + 63) / 64;
    long tile = (long)blockIdx.x - d.tile_begin;
    const int it = (int)(tile % ti); tile /= ti;
    const int k = (int)(tile % K); const int ot = (int)(tile / K);
    const int o0 = ot * 64, i0 = it * 64;
    const float* __restrict__ src = (const float*)d.src;
    T* __restrict__ dst = (T*)d.dst;
    {
        const int c4 = (threadIdx.x & 15) * 4, r = threadIdx.x >> 4;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            int o = o0 + r + pass * 16, i = i0 + c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (o < O && i < I) v = *(const float4*)(src + ((long)o * K + k) * I + i);
            sm[r + pass * 16][c4] = v.x; sm[r + pass * 16][c4 + 1] = v.y; sm[r + pass * 16][c4 + 2] = v.z; sm[r + pass * 16][c4 + 3] = v.w;
        }
    }
    __syncthreads();
    {
        constexpr int V = Vec<T>::N;                 // 8 bf16 / 4 f32 per 16-byte store
        constexpr int CL = 64 / V, RP = 256 / CL;    // column lanes, rows per pass
        const int cv = (threadIdx.x % CL) * V, r = threadIdx.x / CL;
#pragma unroll
        for (int pass = 0; pass < 64 / RP; ++pass) {
            int il = r + pass * RP;
            int i = i0 + il, o = o0 + cv;
            if (i < I && o < O) {
                float f[V];
#pragma unroll
                for (int q = 0; q < V; ++q) f[q] = sm[cv + q][il];
                vstore<T>(dst + ((long)i * K + k) * O + o, f);
                if (lo_off) {
#pragma unroll
                    for (int q = 0; q < V; ++q) f[q] -= round_to<T>(f[q]);
                    vstore<T>(dst + lo_off + ((long)i * K + k) * O + o, f);
                }
            }
        }
    }
}

// image NCHW float [N,3,H,W] -> zero-bordered NHWC4 [N, H+6, W+8, 4] in T (border 3 px, channel 3 = 0)
template <typename T>
__global__ void image_pad_kernel(const float* __restrict__ img, int N, int H, int W, T* __restrict__ out) {
    const int Hp = H + 6, Wp = W + 8;
    long n = (long)N * Hp * Wp;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        int wp = (int)(e % Wp); long t = e / Wp; int hp = (int)(t % Hp); int b = (int)(t / Hp);
        int h = hp - 3, w = wp - 3;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W)
            for (int c = 0; c < 3; ++c) v[c] = img[(((long)b * 3 + c) * H + h) * W + w];
        for (int c = 0; c < 4; ++c) st_f32(out + e * 4 + c, v[c]);
    }
}

// ---------------------------------------------------------------- ReLU backward, column-sum finalize (bias grads)
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ out, long nvec, T* __restrict__ dz) {
    constexpr int V = Vec<T>::N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        float g[V], o[V]; vload<T>(dout + i * V, g); vload<T>(out + i * V, o);
#pragma unroll
        for (int k = 0; k < V; ++k) g[k] = (o[k] > 0.f) ? g[k] : 0.f;
        vstore<T>(dz + i * V, g);
    }
}
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ out) {
    const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    double s = 0.0;
    if (c < C) for (int k = pl; k < nparts; k += 32) s += part[((long)k * C + c) * 2];
    __shared__ double sm[32][8];
    sm[pl][cl] = s;
    __syncthreads();
    if (pl == 0 && c < C) { for (int k = 1; k < 32; ++k) s += sm[k][cl]; out[c] = (float)s; }
}

// ================================================================ split-bf16 ("bf16x3") producers
// The fp32 elementwise passes that feed a convolution write its operand directly as (hi, lo) bf16 planes (conv_x3.hip):
// same bytes as the fp32 tensor they replace, and no separate split pass.  8 channels per thread.
__device__ __forceinline__ void store_split8(bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, long e, const float* f) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        h[k] = pack_bf16x2(f[2 * k], f[2 * k + 1]);
        l[k] = pack_bf16x2(f[2 * k] - __uint_as_float(h[k] << 16), f[2 * k + 1] - __uint_as_float(h[k] & 0xffff0000u));
    }
    *(uint4*)(hi + e) = make_uint4(h[0], h[1], h[2], h[3]);
    *(uint4*)(lo + e) = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ void load8(const float* __restrict__ p, float* f) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
__device__ __forceinline__ void store8(float* __restrict__ p, const float* f) {
    *(float4*)p = make_float4(f[0], f[1], f[2], f[3]); *(float4*)(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}

// out = [relu]( y*scale + shift [+ res] ) -> fp32 `out` (optional) and split planes
// res_bnp != NULL: `res` is itself a raw conv output and the residual is res*scale2 + shift2 (the downsample branch,
// resnet.py:95-99: its BatchNorm output is never materialised)
// (RESBN is a template parameter: as a run-time branch the second parameter set doubled the time of EVERY launch of this kernel,
// 540 -> 1 005 us per step over its 34 launches)
// RESPL: the residual arrives as its (hi, lo) planes (res_hi, res_lo; `res` unused) -- the block input need not exist in fp32.
template <bool RESBN, bool RESPL = false>
__global__ __launch_bounds__(256) void bn_apply_x3_kernel(const float* __restrict__ y, const float* __restrict__ res,
                                                          const float* __restrict__ bnp, long nvec, int C, int relu,
                                                          float* __restrict__ out, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo,
                                                          const float* __restrict__ res_bnp = nullptr,
                                                          const bf16_t* __restrict__ res_hi = nullptr,
                                                          const bf16_t* __restrict__ res_lo = nullptr) {
    const bool fixed = ((256 * 8) % C) == 0;
    float sc[8], sh[8], sc2[RESBN ? 8 : 1], sh2[RESBN ? 8 : 1];
    if (fixed) {
        const int c = (int)(((long)threadIdx.x * 8) % C);
#pragma unroll
        for (int k = 0; k < 8; ++k) { sc[k] = bnp[c + k]; sh[k] = bnp[C + c + k]; }
        if constexpr (RESBN) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { sc2[k] = res_bnp[c + k]; sh2[k] = res_bnp[C + c + k]; }
        }
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const long e = i * 8;
        if (!fixed) {
            const int c = (int)(e % C);
#pragma unroll
            for (int k = 0; k < 8; ++k) { sc[k] = bnp[c + k]; sh[k] = bnp[C + c + k]; }
            if constexpr (RESBN) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { sc2[k] = res_bnp[c + k]; sh2[k] = res_bnp[C + c + k]; }
            }
        }
        float f[8], r[8];
        load8(y + e, f);
        if constexpr (RESPL) {
            const uint4 h4 = *(const uint4*)(res_hi + e), l4 = *(const uint4*)(res_lo + e);
            const uint32_t hw[4] = {h4.x, h4.y, h4.z, h4.w}, lw[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r[2 * k] = __uint_as_float(hw[k] << 16) + __uint_as_float(lw[k] << 16);
                r[2 * k + 1] = __uint_as_float(hw[k] & 0xffff0000u) + __uint_as_float(lw[k] & 0xffff0000u);
            }
        } else if (res) load8(res + e, r);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = f[k] * sc[k] + sh[k];
            if constexpr (RESBN) r[k] = r[k] * sc2[k] + sh2[k];
            if (RESPL || res) v += r[k];
            if (relu) v = fmaxf(v, 0.f);
            f[k] = v;
        }
        if (out) store8(out + e, f);
        store_split8(hi, lo, e, f);
    }
}

// BatchNorm-backward pass 2 with dy written as split planes (its only consumers are the data- and weight-gradient convs)
__global__ __launch_bounds__(256) void bn_bwd_apply_x3_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                              const float* __restrict__ y, const float* __restrict__ bnp,
                                                              const float* __restrict__ bwdp, long nvec, int C, long M, int relu,
                                                              bf16_t* __restrict__ dy_hi, bf16_t* __restrict__ dy_lo,
                                                              float* __restrict__ dz_out, const bf16_t* __restrict__ out_hi) {
    const float invM = 1.f / (float)M;
    const bool fixed = ((256 * 8) % C) == 0;
    float ga[8], sh[8], mu[8], is[8], k1[8], k2[8];
    auto loadp = [&](int c) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            ga[k] = bnp[c + k]; sh[k] = bnp[C + c + k]; mu[k] = bnp[2 * C + c + k]; is[k] = bnp[3 * C + c + k];
            k1[k] = bwdp[c + k] * invM; k2[k] = bwdp[C + c + k] * invM;
        }
    };
    if (fixed) loadp((int)(((long)threadIdx.x * 8) % C));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const long e = i * 8;
        if (!fixed) loadp((int)(e % C));
        float g[8], o[8], yy[8], d[8];
        load8(dout + e, g);
        load8(y + e, yy);
        if (relu == 1) {
            if (out_hi) {
                const uint4 h = *(const uint4*)(out_hi + e);
                const uint32_t hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) { o[2 * k] = __uint_as_float(hw[k] << 16); o[2 * k + 1] = __uint_as_float(hw[k] & 0xffff0000u); }
            } else load8(out + e, o);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool dead = relu == 1 ? !(o[k] > 0.f) : relu == 2 ? !(yy[k] * ga[k] + sh[k] > 0.f) : false;
            const float dz = dead ? 0.f : g[k];
            const float xhat = (yy[k] - mu[k]) * is[k];
            d[k] = ga[k] * (dz - k1[k] - xhat * k2[k]);
            g[k] = dz;
        }
        store_split8(dy_hi, dy_lo, e, d);
        if (dz_out) store8(dz_out + e, g);
    }
}

// ---------------------------------------------------------------- BatchNorm finalize INSIDE the apply passes (round 4)
// The 76 bn_finalize / bn_bwd_finalize launches of a step (~5 us each: a launch + one dependent reduction over the partial rows) sit on the
// step's dependency chain between the convolution that leaves the per-tile sums and the elementwise pass that needs (scale, shift).  Where
// the partial rows are few (<= BNFIN_MAX_ROWS = 64: layers 3 - 4; measured per step at B = 64: off 9.72 ms, rows <= 32 9.69, <= 64 9.66, <= 256 9.73 --
// at 256 rows the prologue's 64 dependent row loads per lane cost more than the launch they replace) the apply pass reduces them itself: a workgroup owns a SLICE of 64
// channels (256 B of every pixel's fp32 row, 128 B of each plane -- whole cache lines) over a strip of pixels, so its prologue reads only
// rows x 64 x 8 bytes (16 - 128 KB from L2) with 4 row-lanes per channel, fixed summation order (rows r = lane, lane + 4, ... in a double
// each, then the four lanes in order): deterministic.  The workgroup of pixel strip 0 writes bnp / the running statistics / dgamma, dbeta
// for its slice.  Arithmetic of the parameters and of the elementwise body: bn_finalize_kernel / bn_apply_x3_kernel / bn_bwd_apply_x3_kernel.
#define BNFIN_MAX_ROWS 64
#define BNFIN_SLICE 64

// (s, q) = column sums of part[nparts][C][2] for channel c0 + (tid & 63); valid in every thread after the call
__device__ __forceinline__ void bnfin_reduce(const float* __restrict__ part, int nparts, int C, int c0, double* sm /* [4][64][2] */, double& s, double& q) {
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    double a = 0, b = 0;
#pragma unroll 4
    for (int r = rl; r < nparts; r += 4) { const float2 v = *(const float2*)(part + ((long)r * C + c0 + cl) * 2); a += v.x; b += v.y; }
    sm[(rl * 64 + cl) * 2] = a; sm[(rl * 64 + cl) * 2 + 1] = b;
    __syncthreads();
    s = ((sm[cl * 2] + sm[(64 + cl) * 2]) + sm[(128 + cl) * 2]) + sm[(192 + cl) * 2];
    q = ((sm[cl * 2 + 1] + sm[(64 + cl) * 2 + 1]) + sm[(128 + cl) * 2 + 1]) + sm[(192 + cl) * 2 + 1];
}

struct BnFinArgs {
    const float* part; int nparts; long count; const float* gamma; const float* beta; float eps, momentum;
    float* running_mean; float* running_var; float* bnp;      // bnp [4][C]: written by the strip-0 workgroups
};

// RES: 0 no residual / fp32 residual `res`, 1 residual as (hi, lo) planes, 2 `res` is a raw conv output normalised by res_bnp on the fly
template <int RES>
__global__ __launch_bounds__(256) void bn_fin_apply_x3_kernel(BnFinArgs fa, const float* __restrict__ y, const float* __restrict__ res,
                                                              const bf16_t* __restrict__ res_hi, const bf16_t* __restrict__ res_lo,
                                                              const float* __restrict__ res_bnp, long M, int C, int relu,
                                                              float* __restrict__ out, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo) {
    __shared__ double sm[4 * 64 * 2];
    __shared__ float par[2][BNFIN_SLICE];
    const int nsl = C / BNFIN_SLICE, slice = blockIdx.x % nsl, strip = blockIdx.x / nsl, nstrips = gridDim.x / nsl;
    const int c0 = slice * BNFIN_SLICE;
    double s, q; bnfin_reduce(fa.part, fa.nparts, C, c0, sm, s, q);
    if (threadIdx.x < 64) {
        const int c = c0 + threadIdx.x;
        const float g = fa.gamma[c], bt = fa.beta[c];
        double mean = s / (double)fa.count;
        double var = q / (double)fa.count - mean * mean; if (var < 0) var = 0;
        float invstd = (float)(1.0 / sqrt(var + (double)fa.eps));
        float sc = g * invstd, sh = bt - (float)mean * sc;
        par[0][threadIdx.x] = sc; par[1][threadIdx.x] = sh;
        if (strip == 0) {
            fa.bnp[c] = sc; fa.bnp[C + c] = sh; fa.bnp[2 * C + c] = (float)mean; fa.bnp[3 * C + c] = invstd;
            if (fa.running_mean) {
                double unb = fa.count > 1 ? var * (double)fa.count / (double)(fa.count - 1) : var;
                fa.running_mean[c] = (1.f - fa.momentum) * fa.running_mean[c] + fa.momentum * (float)mean;
                fa.running_var[c] = (1.f - fa.momentum) * fa.running_var[c] + fa.momentum * (float)unb;
            }
        }
    }
    __syncthreads();
    const int j = threadIdx.x & 7, pl = threadIdx.x >> 3;             // 8 channel groups of a pixel, 32 pixels per pass
    float sc[8], sh[8], sc2[RES == 2 ? 8 : 1], sh2[RES == 2 ? 8 : 1];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = par[0][j * 8 + k]; sh[k] = par[1][j * 8 + k]; }
    if constexpr (RES == 2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { sc2[k] = res_bnp[c0 + j * 8 + k]; sh2[k] = res_bnp[C + c0 + j * 8 + k]; }
    }
    for (long p = (long)strip * 32 + pl; p < M; p += (long)nstrips * 32) {
        const long e = p * C + c0 + j * 8;
        float f[8], r[8];
        load8(y + e, f);
        if constexpr (RES == 1) {
            const uint4 h4 = *(const uint4*)(res_hi + e), l4 = *(const uint4*)(res_lo + e);
            const uint32_t hw[4] = {h4.x, h4.y, h4.z, h4.w}, lw[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r[2 * k] = __uint_as_float(hw[k] << 16) + __uint_as_float(lw[k] << 16);
                r[2 * k + 1] = __uint_as_float(hw[k] & 0xffff0000u) + __uint_as_float(lw[k] & 0xffff0000u);
            }
        } else if (res) load8(res + e, r);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = f[k] * sc[k] + sh[k];
            if constexpr (RES == 2) r[k] = r[k] * sc2[k] + sh2[k];
            if (RES == 1 || res) v += r[k];
            if (relu) v = fmaxf(v, 0.f);
            f[k] = v;
        }
        if (out) store8(out + e, f);
        store_split8(hi, lo, e, f);
    }
}

// BatchNorm backward, pass 2 with its finalize inside: part[nparts][C][2] = per-tile (sum dz, sum dz * xhat)
__global__ __launch_bounds__(256) void bn_fin_bwd_apply_x3_kernel(const float* __restrict__ part, int nparts, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta, float* __restrict__ bwdp,
                                                                  const float* __restrict__ dout, const float* __restrict__ out,
                                                                  const float* __restrict__ y, const float* __restrict__ bnp, int C, long M,
                                                                  int relu, bf16_t* __restrict__ dy_hi, bf16_t* __restrict__ dy_lo,
                                                                  float* __restrict__ dz_out, const bf16_t* __restrict__ out_hi) {
    __shared__ double sm[4 * 64 * 2];
    __shared__ float par[2][BNFIN_SLICE];
    const int nsl = C / BNFIN_SLICE, slice = blockIdx.x % nsl, strip = blockIdx.x / nsl, nstrips = gridDim.x / nsl;
    const int c0 = slice * BNFIN_SLICE;
    double s, q; bnfin_reduce(part, nparts, C, c0, sm, s, q);
    if (threadIdx.x < 64) {
        par[0][threadIdx.x] = (float)s; par[1][threadIdx.x] = (float)q;
        if (strip == 0) {
            const int c = c0 + threadIdx.x;
            dbeta[c] = (float)s; dgamma[c] = (float)q; bwdp[c] = (float)s; bwdp[C + c] = (float)q;
        }
    }
    __syncthreads();
    const float invM = 1.f / (float)M;
    const int j = threadIdx.x & 7, pl = threadIdx.x >> 3;
    float ga[8], sh[8], mu[8], is[8], k1[8], k2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = c0 + j * 8 + k;
        ga[k] = bnp[c]; sh[k] = bnp[C + c]; mu[k] = bnp[2 * C + c]; is[k] = bnp[3 * C + c];
        k1[k] = par[0][j * 8 + k] * invM; k2[k] = par[1][j * 8 + k] * invM;
    }
    for (long p = (long)strip * 32 + pl; p < M; p += (long)nstrips * 32) {
        const long e = p * C + c0 + j * 8;
        float g[8], o[8], yy[8], d[8];
        load8(dout + e, g);
        load8(y + e, yy);
        if (relu == 1) {
            if (out_hi) {
                const uint4 h = *(const uint4*)(out_hi + e);
                const uint32_t hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) { o[2 * k] = __uint_as_float(hw[k] << 16); o[2 * k + 1] = __uint_as_float(hw[k] & 0xffff0000u); }
            } else load8(out + e, o);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool dead = relu == 1 ? !(o[k] > 0.f) : relu == 2 ? !(yy[k] * ga[k] + sh[k] > 0.f) : false;
            const float dz = dead ? 0.f : g[k];
            const float xhat = (yy[k] - mu[k]) * is[k];
            d[k] = ga[k] * (dz - k1[k] - xhat * k2[k]);
            g[k] = dz;
        }
        store_split8(dy_hi, dy_lo, e, d);
        if (dz_out) store8(dz_out + e, g);
    }
}

static bool bnfin_ok(int nparts, int C) {
    static const int on = getenv("AB_BNFIN_FUSE") ? atoi(getenv("AB_BNFIN_FUSE")) : 1;
    static const int maxrows = getenv("AB_BNFIN_ROWS") ? atoi(getenv("AB_BNFIN_ROWS")) : BNFIN_MAX_ROWS;
    return on && nparts > 0 && nparts <= maxrows && nparts <= BNFIN_MAX_ROWS && C % BNFIN_SLICE == 0;
}
// workgroups: every channel slice x enough pixel strips for ~4 workgroups per CU (each pays the prologue once)
static int bnfin_grid(long M, int C) {
    const int nsl = C / BNFIN_SLICE;
    long strips = (M + 31) / 32;
    const long want = (1024 + nsl - 1) / nsl;
    if (strips > want) strips = want;
    if (strips < 1) strips = 1;
    return (int)(strips * nsl);
}

// maxpool3x3/2(relu(bn(y))) forward of the split-bf16 path (the stem): maxpool_fwd_kernel<float>'s arithmetic and scan order with 8
// channels per thread and PFX_ROWS output rows per thread: consecutive output rows share an input row (2ho+1 = 2(ho+1)-1), which the
// one-row-per-workgroup version read twice (268 MB of input became ~400 MB of reads); here the bottom taps of a window are carried
// in registers as the top taps of the next -- six tap loads per output row instead of nine.
#define PFX_ROWS 4
__global__ __launch_bounds__(256) void maxpool_fwd_x3_kernel(const float* __restrict__ y, const float* __restrict__ bnp, int N, int H,
                                                             int W, int C, float* __restrict__ out, uint8_t* __restrict__ idx,
                                                             bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                                             float* __restrict__ ywin) {
    const int Ho = H / 2, Wo = W / 2, vc = C / 8;
    const unsigned col = blockIdx.x * 256 + threadIdx.x;
    if (col >= (unsigned)(Wo * vc)) return;
    const int wo = (int)(col / (unsigned)vc), cv = (int)(col - (unsigned)wo * vc);
    const int n = blockIdx.z, ho0 = blockIdx.y * PFX_ROWS;
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = bnp[cv * 8 + k]; sh[k] = bnp[C + cv * 8 + k]; }
    const bool w0ok = wo > 0;                                // H, W even: only h = -1 / w = -1 can fall outside
    auto load_row = [&](int h, float (*r)[8]) {              // the three taps of input row h
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
            if (dw == 0 && !w0ok) continue;
            load8(y + (((long)n * H + h) * W + (wo * 2 + dw - 1)) * C + cv * 8, r[dw]);
        }
    };      // (RAW values are kept: relu(bn(.)) is applied at the comparison, so the winner's raw value is at hand for `ywin`)
    float top[3][8], mid[3][8], bot[3][8];
    if (ho0 > 0) load_row(ho0 * 2 - 1, top);
#pragma unroll
    for (int rr = 0; rr < PFX_ROWS; ++rr) {
        const int ho = ho0 + rr;
        if (ho >= Ho) break;
        load_row(ho * 2, mid);
        load_row(ho * 2 + 1, bot);
        float m[8], mraw[8]; uint32_t am[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { m[k] = -INFINITY; mraw[k] = 0.f; am[k] = 0; }
        auto scan = [&](const float (*r)[8], const int t0) {        // (inlined per row: the tap arrays stay in registers)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                if (dw == 0 && !w0ok) continue;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float v = fmaxf(r[dw][k] * sc[k] + sh[k], 0.f);
                    if (v > m[k]) { m[k] = v; mraw[k] = r[dw][k]; am[k] = t0 + dw; }
                }
            }
        };
        if (ho > 0) scan(top, 0);
        scan(mid, 3);
        scan(bot, 6);
        const long e = ((((long)n * Ho + ho) * Wo + wo) * vc + cv) * 8;
        if (out) store8(out + e, m);           // (NULL: the pooled tensor is wanted as planes only)
        *(uint2*)(idx + e) = make_uint2(am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24), am[4] | (am[5] << 8) | (am[6] << 16) | (am[7] << 24));
        store_split8(out_hi, out_lo, e, m);
        if (ywin) store8(ywin + e, mraw);      // the winner's RAW conv output: what the backward's reduction needs of y
#pragma unroll
        for (int dw = 0; dw < 3; ++dw)
#pragma unroll
            for (int k = 0; k < 8; ++k) top[dw][k] = bot[dw][k];
    }
}

// Second half of maxpool3x3/2(relu(bn(y))) backward (pool_bwd_bn_reduce_kernel) for the split-bf16 path without the fp32 dz round trip: the same gather and mask again (winners + pooled gradient are
// a quarter of the map), then BatchNorm-backward pass 2 on the spot, dy written as planes.  Against (reduce writes dz, apply reads
// dz): 0.27 GB less written and 0.27 GB less read at the benchmark size for 84 MB of winners / pooled gradient read twice.
__global__ __launch_bounds__(256) void pool_bwd_bn_apply_x3_kernel(const uint8_t* __restrict__ idx, const float* __restrict__ dout,
                                                                    const float* __restrict__ y, const float* __restrict__ bnp,
                                                                    const float* __restrict__ bwdp, int N, int H, int W, int C,
                                                                    float invM, bf16_t* __restrict__ dy_hi, bf16_t* __restrict__ dy_lo) {
    const int Ho = H / 2, Wo = W / 2, vc = C / 8;
    const unsigned col = blockIdx.x * 256 + threadIdx.x;
    if (col >= (unsigned)(W * vc)) return;
    const int w = (int)(col / (unsigned)vc), cv = (int)(col - (unsigned)w * vc);
    const int n = blockIdx.z, p = blockIdx.y;
    float acc0[8], acc1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { acc0[k] = 0.f; acc1[k] = 0.f; }
#pragma unroll
    for (int dp = 0; dp < 2; ++dp) {
        const int ho = p + dp;
        if (ho >= Ho) continue;
        for (int wo = w / 2; wo <= (w + 1) / 2; ++wo) {
            if (wo >= Wo) continue;
            const int cw = w - (wo * 2 - 1);
            const long o = (((long)n * Ho + ho) * Wo + wo) * C + cv * 8;
            const uint2 qd = *(const uint2*)(idx + o);
            float g[8]; load8(dout + o, g);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int am = ((k < 4 ? qd.x : qd.y) >> (8 * (k & 3))) & 0xff;
                if (dp == 0) { if (am == 3 + cw) acc0[k] += g[k]; if (am == 6 + cw) acc1[k] += g[k]; }
                else if (am == cw) acc1[k] += g[k];
            }
        }
    }
    const long e = ((((long)n * H + 2 * p) * W + w) * vc + cv) * 8;
    float ya[8], yb[8], d0[8], d1[8];
    load8(y + e, ya); load8(y + e + (long)W * C, yb);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = cv * 8 + k;
        const float ga = bnp[c], sh = bnp[C + c], mu = bnp[2 * C + c], is = bnp[3 * C + c];
        const float k1 = bwdp[c] * invM, k2 = bwdp[C + c] * invM;
        const float z0 = (ya[k] * ga + sh > 0.f) ? acc0[k] : 0.f, z1 = (yb[k] * ga + sh > 0.f) ? acc1[k] : 0.f;
        d0[k] = ga * (z0 - k1 - ((ya[k] - mu) * is) * k2);
        d1[k] = ga * (z1 - k1 - ((yb[k] - mu) * is) * k2);
    }
    store_split8(dy_hi, dy_lo, e, d0);
    store_split8(dy_hi, dy_lo, e + (long)W * C, d1);
}

// column sums of a split tensor (hi + lo): the bias gradient of the final layer from the dlogits planes
__global__ __launch_bounds__(256) void col_stats_x3_kernel(const bf16_t* __restrict__ hi, const bf16_t* __restrict__ lo, long M, int C,
                                                          int rows_per_block, float* __restrict__ part) {
    const int vc = C / 8, rl = 256 / vc;
    const int cv = threadIdx.x % vc, rr = threadIdx.x / vc;
    long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = 0.f; q[i] = 0.f; }
    if (rr < rl)
        for (long r = r0 + rr; r < r1; r += rl) {
            float a[8], b[8];
            vload<bf16_t>(hi + r * C + cv * 8, a);
            vload<bf16_t>(lo + r * C + cv * 8, b);
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float v = a[i] + b[i]; s[i] += v; q[i] += v * v; }
        }
    extern __shared__ float sm[];
    if (rr < rl)
#pragma unroll
        for (int i = 0; i < 8; ++i) { sm[(rr * C + cv * 8 + i) * 2] = s[i]; sm[(rr * C + cv * 8 + i) * 2 + 1] = q[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < rl; ++k) { a += sm[(k * C + c) * 2]; b += sm[(k * C + c) * 2 + 1]; }
        part[((long)blockIdx.x * C + c) * 2] = a; part[((long)blockIdx.x * C + c) * 2 + 1] = b;
    }
}

// BatchNorm-backward reduction of  maxpool3x3/2(relu(bn(y)))  over the POOLED elements only: the masked gradient is non-zero at
// window winners, so  sum dz = sum_w [relu'] dpool_w  and  sum dz*xhat = sum_w [relu'] dpool_w * xhat(ywin_w)  (an input position that
// wins several windows contributes once per window: the sums are linear).  Reads 2 x [N,H/2,W/2,C] fp32 instead of the full-resolution
// y + winners + pooled gradient; same mask expression as the forward.  part rows: ab_col_stats_nparts(N*H/2*W/2).
__global__ __launch_bounds__(256) void pool_win_bn_reduce_kernel(const float* __restrict__ dpool, const float* __restrict__ ywin,
                                                                  const float* __restrict__ bnp, long M, int C, int rows_per_block,
                                                                  float* __restrict__ part) {
    const int vc = C / 8, rl = 256 / vc;
    const int cv = threadIdx.x % vc, rr = threadIdx.x / vc;
    long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
    float s[8], q[8], sc[8], sh[8], mu[8], is[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s[i] = 0.f; q[i] = 0.f;
        sc[i] = bnp[cv * 8 + i]; sh[i] = bnp[C + cv * 8 + i]; mu[i] = bnp[2 * C + cv * 8 + i]; is[i] = bnp[3 * C + cv * 8 + i];
    }
    if (rr < rl)
        for (long r = r0 + rr; r < r1; r += rl) {
            float g[8], yw[8];
            load8(dpool + r * C + cv * 8, g);
            load8(ywin + r * C + cv * 8, yw);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float dz = (yw[i] * sc[i] + sh[i] > 0.f) ? g[i] : 0.f;
                s[i] += dz; q[i] += dz * ((yw[i] - mu[i]) * is[i]);
            }
        }
    extern __shared__ float sm[];
    if (rr < rl)
#pragma unroll
        for (int i = 0; i < 8; ++i) { sm[(rr * C + cv * 8 + i) * 2] = s[i]; sm[(rr * C + cv * 8 + i) * 2 + 1] = q[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < rl; ++k) { a += sm[(k * C + c) * 2]; b += sm[(k * C + c) * 2 + 1]; }
        part[((long)blockIdx.x * C + c) * 2] = a; part[((long)blockIdx.x * C + c) * 2 + 1] = b;
    }
}

// ================================================================ C ABI
static inline int grid_for(long nvec) { long b = (nvec + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }
#define DISPATCH(dtype, CALL_F, CALL_B) do { if ((dtype) == AB_DT_F32) { CALL_F; } else if ((dtype) == AB_DT_BF16) { CALL_B; } else return AB_EINVAL; } while (0)

extern "C" int ab_col_stats_nparts(long M) { int r = red_rows(M); return (int)((M + r - 1) / r); }

extern "C" int ab_col_stats(const void* x, int dtype, long M, int C, float* part, void* stream) {
    if (!x || !part) return AB_EINVAL;
    int V = dtype == AB_DT_F32 ? 4 : 8;
    if (C % V || C / V > 256) return AB_ESHAPE;
    int np = ab_col_stats_nparts(M); int rl = 256 / (C / V); size_t sh = (size_t)rl * C * 2 * 4;
    DISPATCH(dtype, (col_stats_kernel<float><<<np, 256, sh, as_stream(stream)>>>((const float*)x, M, C, red_rows(M), part)),
             (col_stats_kernel<bf16_t><<<np, 256, sh, as_stream(stream)>>>((const bf16_t*)x, M, C, red_rows(M), part)));
    AB_LAUNCH_CHECK(); return 0;
}

extern "C" int ab_bn_finalize(const float* part, int nparts, int C, long count, const float* gamma, const float* beta,
                              float eps, float momentum, float* running_mean, float* running_var, float* bnp,
                              void* stream) {
    if (!part || !gamma || !beta || !bnp) return AB_EINVAL;
    hipStream_t st = as_stream(stream);
    switch (finalize_cpw(C)) {
    case 2: bn_finalize_kernel<2><<<(C + 1) / 2, 1024, 0, st>>>(part, nparts, C, count, gamma, beta, eps, momentum, running_mean, running_var, bnp); break;
    case 4: bn_finalize_kernel<4><<<(C + 3) / 4, 1024, 0, st>>>(part, nparts, C, count, gamma, beta, eps, momentum, running_mean, running_var, bnp); break;
    default: bn_finalize_kernel<8><<<(C + 7) / 8, 1024, 0, st>>>(part, nparts, C, count, gamma, beta, eps, momentum, running_mean, running_var, bnp); break;
    }
    AB_LAUNCH_CHECK(); return 0;
}

extern "C" int ab_bn_eval_params(int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                                 float eps, float* bnp, void* stream) {
    if (!gamma || !beta || !rm || !rv || !bnp) return AB_EINVAL;
    bn_eval_params_kernel<<<(C + 255) / 256, 256, 0, as_stream(stream)>>>(C, gamma, beta, rm, rv, eps, bnp);
    AB_LAUNCH_CHECK(); return 0;
}

extern "C" int ab_bn_eval_params_batch(const float* flat, const float* stats, const int32_t* desc_dev, int n, int max_c, float eps,
                                       float* out, void* stream) {
    if (!flat || !stats || !desc_dev || !out) return AB_EINVAL;
    if (n <= 0 || max_c <= 0) return AB_ESHAPE;
    bn_eval_params_batch_kernel<<<dim3((max_c + 255) / 256, n), 256, 0, as_stream(stream)>>>(flat, stats, (const int*)desc_dev, eps, out);
    AB_LAUNCH_CHECK(); return 0;
}

extern "C" int ab_bn_apply(const void* y, const void* res, const float* bnp, int dtype, long M, int C, int relu,
                           void* out, void* stream) {
    if (!y || !bnp || !out) return AB_EINVAL;
    int V = dtype == AB_DT_F32 ? 4 : 8; if (C % V) return AB_ESHAPE;
    long nvec = M * C / V;
    DISPATCH(dtype, (bn_apply_kernel<float><<<grid_for(nvec), 256, 0, as_stream(stream)>>>((const float*)y, (const float*)res, bnp, nvec, C, relu, (float*)out)),
             (bn_apply_kernel<bf16_t><<<grid_for(nvec), 256, 0, as_stream(stream)>>>((const bf16_t*)y, (const bf16_t*)res, bnp, nvec, C, relu, (bf16_t*)out)));
    AB_LAUNCH_CHECK(); return 0;
}

static void launch_bn_bwd_finalize(const float* part, int np, int C, float* dgamma, float* dbeta, float* bwdp, hipStream_t st) {
    switch (finalize_cpw(C)) {
    case 2: bn_bwd_finalize_kernel<2><<<(C + 1) / 2, 1024, 0, st>>>(part, np, C, dgamma, dbeta, bwdp); break;
    case 4: bn_bwd_finalize_kernel<4><<<(C + 3) / 4, 1024, 0, st>>>(part, np, C, dgamma, dbeta, bwdp); break;
    default: bn_bwd_finalize_kernel<8><<<(C + 7) / 8, 1024, 0, st>>>(part, np, C, dgamma, dbeta, bwdp); break;
    }
}

static int bn_bwd_impl(const void* dout, const void* out, const void* y, const float* bnp, int dtype, long M, int C,
                       int relu, float* part, float* bwdp, float* dgamma, float* dbeta, void* dy, void* dz_out,
                       const uint8_t* pool_idx, int pH, int pW, hipStream_t st, int given_parts = 0) {
    int V = dtype == AB_DT_F32 ? 4 : 8;
    int ysl = 1;                                  // channel slices of the reduction (<= 256 V-channel groups each)
    while (C / V / ysl > 256) ysl *= 2;
    if (C % V || C % (V * ysl)) return AB_ESHAPE;
    int np = given_parts > 0 ? given_parts : ab_col_stats_nparts(M); const int cs = C / ysl; int rl = 256 / (cs / V); size_t sh = (size_t)rl * cs * 2 * 4;
    if (given_parts <= 0)
    DISPATCH(dtype, (bn_bwd_reduce_kernel<float><<<dim3(np, ysl), 256, sh, st>>>((const float*)dout, (const float*)out, (const float*)y, bnp, M, C, relu, red_rows(M), part, pool_idx, pH, pW)),
             (bn_bwd_reduce_kernel<bf16_t><<<dim3(np, ysl), 256, sh, st>>>((const bf16_t*)dout, (const bf16_t*)out, (const bf16_t*)y, bnp, M, C, relu, red_rows(M), part, pool_idx, pH, pW)));
    AB_LAUNCH_CHECK();
    launch_bn_bwd_finalize(part, np, C, dgamma, dbeta, bwdp, st);
    AB_LAUNCH_CHECK();
    long nvec = M * C / V;
    DISPATCH(dtype, (bn_bwd_apply_kernel<float><<<grid_for(nvec), 256, 0, st>>>((const float*)dout, (const float*)out, (const float*)y, bnp, bwdp, nvec, C, M, relu, (float*)dy, (float*)dz_out, pool_idx, pH, pW)),
             (bn_bwd_apply_kernel<bf16_t><<<grid_for(nvec), 256, 0, st>>>((const bf16_t*)dout, (const bf16_t*)out, (const bf16_t*)y, bnp, bwdp, nvec, C, M, relu, (bf16_t*)dy, (bf16_t*)dz_out, pool_idx, pH, pW)));
    AB_LAUNCH_CHECK(); return 0;
}

extern "C" int ab_bn_bwd(const void* dout, const void* out, const void* y, const float* bnp, int dtype, long M, int C,
                         int relu, float* part, float* bwdp, float* dgamma, float* dbeta, void* dy, void* dz_out,
                         void* stream) {
    if (!dout || !y || !bnp || !part || !bwdp || !dgamma || !dbeta || !dy || (relu == 1 && !out) || relu < 0 || relu > 2) return AB_EINVAL;
    return bn_bwd_impl(dout, out, y, bnp, dtype, M, C, relu, part, bwdp, dgamma, dbeta, dy, dz_out, nullptr, 0, 0, as_stream(stream));
}

// Second half of ab_bn_bwd only (finalize + apply): `part` [nparts][C][2] already holds the per-tile sums (sum dz,
// sum dz*xhat), written by ab_conv2d_dgrad_bnstats.
extern "C" int ab_bn_bwd_apply(const void* dout, const void* out, const void* y, const float* bnp, int dtype, long M, int C,
                               int relu, const float* part, int nparts, float* bwdp, float* dgamma, float* dbeta, void* dy,
                               void* dz_out, void* stream) {
    if (!dout || !y || !bnp || !part || nparts < 1 || !bwdp || !dgamma || !dbeta || !dy || (relu == 1 && !out) || relu < 0 || relu > 2) return AB_EINVAL;
    return bn_bwd_impl(dout, out, y, bnp, dtype, M, C, relu, (float*)part, bwdp, dgamma, dbeta, dy, dz_out, nullptr, 0, 0,
                       as_stream(stream), nparts);
}

// Backward of  maxpool3x3/2( relu( bn(y) ) )  in the two BN passes: the pooled gradient dpool [N,H/2,W/2,C] is scattered
// on the fly through the recorded winners (idx of ab_bn_relu_maxpool3x3s2_fwd), the ReLU mask is recomputed from y.
extern "C" int ab_bn_relu_maxpool_bwd(const void* dpool, const void* idx, const void* y, const float* bnp, int dtype, int N,
                                      int H, int W, int C, float* part, float* bwdp, float* dgamma, float* dbeta, void* dy,
                                      void* stream) {
    if (!dpool || !idx || !y || !bnp || !part || !bwdp || !dgamma || !dbeta || !dy) return AB_EINVAL;
    if ((H & 1) || (W & 1)) return AB_ESHAPE;
    return bn_bwd_impl(dpool, nullptr, y, bnp, dtype, (long)N * H * W, C, 2, part, bwdp, dgamma, dbeta, dy, nullptr,
                       (const uint8_t*)idx, H, W, as_stream(stream));
}

extern "C" int ab_add(const void* a, const void* b, int dtype, long n, void* out, void* stream) {
    int V = dtype == AB_DT_F32 ? 4 : 8; if (n % V) return AB_ESHAPE;
    long nvec = n / V;
    DISPATCH(dtype, (add_kernel<float><<<grid_for(nvec), 256, 0, as_stream(stream)>>>((const float*)a, (const float*)b, nvec, (float*)out)),
             (add_kernel<bf16_t><<<grid_for(nvec), 256, 0, as_stream(stream)>>>((const bf16_t*)a, (const bf16_t*)b, nvec, (bf16_t*)out)));
    AB_LAUNCH_CHECK(); return 0;
}

extern "C" int ab_maxpool3x3s2_fwd(const void* x, int dtype, int N, int H, int W, int C, void* out, void* idx, void* stream) {
    int V = dtype == AB_DT_F32 ? 4 : 8; if (C % V || (H & 1) || (W & 1)) return AB_ESHAPE;
    dim3 pgrid((unsigned)(((long)(W / 2) * (C / V) + 255) / 256), (unsigned)(H / 2), (unsigned)N);
    DISPATCH(dtype, (maxpool_fwd_kernel<float><<<pgrid, 256, 0, as_stream(stream)>>>((const float*)x, nullptr, N, H, W, C, (float*)out, (uint8_t*)idx)),
             (maxpool_fwd_kernel<bf16_t><<<pgrid, 256, 0, as_stream(stream)>>>((const bf16_t*)x, nullptr, N, H, W, C, (bf16_t*)out, (uint8_t*)idx)));
    AB_LAUNCH_CHECK(); return 0;
}
extern "C" int ab_bn_relu_maxpool3x3s2_fwd(const void* y, const float* bnp, int dtype, int N, int H, int W, int C, void* out,
                                           void* idx, void* stream) {
    if (!y || !bnp || !out) return AB_EINVAL;
    int V = dtype == AB_DT_F32 ? 4 : 8; if (C % V || (H & 1) || (W & 1)) return AB_ESHAPE;
    dim3 pgrid((unsigned)(((long)(W / 2) * (C / V) + 255) / 256), (unsigned)(H / 2), (unsigned)N);
    DISPATCH(dtype, (maxpool_fwd_kernel<float><<<pgrid, 256, 0, as_stream(stream)>>>((const float*)y, bnp, N, H, W, C, (float*)out, (uint8_t*)idx)),
             (maxpool_fwd_kernel<bf16_t><<<pgrid, 256, 0, as_stream(stream)>>>((const bf16_t*)y, bnp, N, H, W, C, (bf16_t*)out, (uint8_t*)idx)));
    AB_LAUNCH_CHECK(); return 0;
}
// fp32 in, fp32 + (hi, lo) bf16 planes out: the pooled tensor is read by three split-bf16 convolutions (layer1.0 conv1, its
// weight gradient, the residual stays fp32)
extern "C" int ab_bn_relu_maxpool3x3s2_fwd_x3(const float* y, const float* bnp, int N, int H, int W, int C, float* out, void* out_hi,
                                              void* out_lo, void* idx, void* stream) {
    if (!y || !bnp || !out || !out_hi || !out_lo) return AB_EINVAL;
    if (C % 4 || (H & 1) || (W & 1)) return AB_ESHAPE;
    static const int v8 = getenv("AB_POOL_FWD_V8") ? atoi(getenv("AB_POOL_FWD_V8")) : 1;
    if (v8 && C % 8 == 0 && idx) {
        dim3 g8((unsigned)(((long)(W / 2) * (C / 8) + 255) / 256), (unsigned)((H / 2 + PFX_ROWS - 1) / PFX_ROWS), (unsigned)N);
        maxpool_fwd_x3_kernel<<<g8, 256, 0, as_stream(stream)>>>(y, bnp, N, H, W, C, out, (uint8_t*)idx, (bf16_t*)out_hi, (bf16_t*)out_lo, nullptr);
        AB_LAUNCH_CHECK(); return 0;
    }
    dim3 pgrid((unsigned)(((long)(W / 2) * (C / 4) + 255) / 256), (unsigned)(H / 2), (unsigned)N);
    maxpool_fwd_kernel<float><<<pgrid, 256, 0, as_stream(stream)>>>(y, bnp, N, H, W, C, out, (uint8_t*)idx, (bf16_t*)out_hi, (bf16_t*)out_lo);
    AB_LAUNCH_CHECK(); return 0;
}
// ... also writing ywin [N,H/2,W/2,C] fp32: the RAW conv output at each window's winner, for ab_bn_relu_maxpool_bwd_x3w
extern "C" int ab_bn_relu_maxpool3x3s2_fwd_x3w(const float* y, const float* bnp, int N, int H, int W, int C, float* out, void* out_hi,
                                               void* out_lo, void* idx, float* ywin, void* stream) {
    if (!y || !bnp || !out_hi || !out_lo || !idx || !ywin) return AB_EINVAL;          // out may be NULL: planes only
    if (C % 8 || (H & 1) || (W & 1)) return AB_ESHAPE;
    dim3 g8((unsigned)(((long)(W / 2) * (C / 8) + 255) / 256), (unsigned)((H / 2 + PFX_ROWS - 1) / PFX_ROWS), (unsigned)N);
    maxpool_fwd_x3_kernel<<<g8, 256, 0, as_stream(stream)>>>(y, bnp, N, H, W, C, out, (uint8_t*)idx, (bf16_t*)out_hi, (bf16_t*)out_lo, ywin);
    AB_LAUNCH_CHECK(); return 0;
}
extern "C" int ab_maxpool3x3s2_bwd(const void* idx, const void* dout, int dtype, int N, int H, int W, int C, void* dx,
                                   void* stream) {
    int V = dtype == AB_DT_F32 ? 4 : 8; if (C % V || (H & 1) || (W & 1)) return AB_ESHAPE;
    dim3 pgrid((unsigned)(((long)W * (C / V) + 255) / 256), (unsigned)(H / 2), (unsigned)N);
    DISPATCH(dtype, (maxpool_bwd_kernel<float><<<pgrid, 256, 0, as_stream(stream)>>>((const uint8_t*)idx, (const float*)dout, N, H, W, C, (float*)dx)),
             (maxpool_bwd_kernel<bf16_t><<<pgrid, 256, 0, as_stream(stream)>>>((const uint8_t*)idx, (const bf16_t*)dout, N, H, W, C, (bf16_t*)dx)));
    AB_LAUNCH_CHECK(); return 0;
}

extern "C" int ab_avgpool_fwd(const void* x, int dtype, int N, int HW, int C, float* out, void* stream) {
    if (!x || !out) return AB_EINVAL;
    if (C % (dtype == AB_DT_F32 ? 4 : 8)) return AB_ESHAPE;
    DISPATCH(dtype, (avgpool_fwd_kernel<float><<<dim3(N, (C + 127) / 128), 256, 0, as_stream(stream)>>>((const float*)x, HW, C, out)),
             (avgpool_fwd_kernel<bf16_t><<<dim3(N, (C + 255) / 256), 256, 0, as_stream(stream)>>>((const bf16_t*)x, HW, C, out)));
    AB_LAUNCH_CHECK(); return 0;
}
extern "C" int ab_avgpool_bwd(const float* g, int dtype, int N, int HW, int C, void* dx, int accumulate, void* stream) {
    if (!g || !dx) return AB_EINVAL;
    if (C % (dtype == AB_DT_F32 ? 4 : 8)) return AB_ESHAPE;
    DISPATCH(dtype, (avgpool_bwd_kernel<float><<<grid_for((long)N * HW * C / 4), 256, 0, as_stream(stream)>>>(g, HW, C, (long)N * HW * C / 4, (float*)dx, accumulate)),
             (avgpool_bwd_kernel<bf16_t><<<grid_for((long)N * HW * C / 8), 256, 0, as_stream(stream)>>>(g, HW, C, (long)N * HW * C / 8, (bf16_t*)dx, accumulate)));
    AB_LAUNCH_CHECK(); return 0;
}

extern "C" int ab_cast_f32_bf16(const float* src, long n, void* dst, void* stream) {
    cast_f32_bf16_kernel<<<grid_for(n), 256, 0, as_stream(stream)>>>(src, n, (bf16_t*)dst);
    AB_LAUNCH_CHECK(); return 0;
}
extern "C" int ab_transpose_oki(const float* src, int O, int K, int I, int dtype, void* dst, void* stream) {
    long n = (long)O * K * I;
    DISPATCH(dtype, (transpose_oki_kernel<float><<<grid_for(n), 256, 0, as_stream(stream)>>>(src, O, K, I, (float*)dst)),
             (transpose_oki_kernel<bf16_t><<<grid_for(n), 256, 0, as_stream(stream)>>>(src, O, K, I, (bf16_t*)dst)));
    AB_LAUNCH_CHECK(); return 0;
}
extern "C" int ab_transpose_oki_batch(const ab_transpose_desc* desc_dev, int ntensors, long total_tiles, int dtype,
                                      void* stream) {
    if (!desc_dev || ntensors < 1 || total_tiles < 1 || total_tiles > 0x7fffffffL) return AB_EINVAL;
    DISPATCH(dtype, (transpose_oki_batch_kernel<float><<<(unsigned)total_tiles, 256, 0, as_stream(stream)>>>(desc_dev, ntensors)),
             (transpose_oki_batch_kernel<bf16_t><<<(unsigned)total_tiles, 256, 0, as_stream(stream)>>>(desc_dev, ntensors)));
    AB_LAUNCH_CHECK(); return 0;
}
extern "C" int ab_image_pad_nhwc4(const float* img_nchw, int dtype, int N, int H, int W, void* out, void* stream) {
    long n = (long)N * (H + 6) * (W + 8);
    DISPATCH(dtype, (image_pad_kernel<float><<<grid_for(n), 256, 0, as_stream(stream)>>>(img_nchw, N, H, W, (float*)out)),
             (image_pad_kernel<bf16_t><<<grid_for(n), 256, 0, as_stream(stream)>>>(img_nchw, N, H, W, (bf16_t*)out)));
    AB_LAUNCH_CHECK(); return 0;
}

extern "C" int ab_relu_bwd(const void* dout, const void* out, int dtype, long n, void* dz, void* stream) {
    int V = dtype == AB_DT_F32 ? 4 : 8; if (n % V) return AB_ESHAPE;
    long nvec = n / V;
    DISPATCH(dtype, (relu_bwd_kernel<float><<<grid_for(nvec), 256, 0, as_stream(stream)>>>((const float*)dout, (const float*)out, nvec, (float*)dz)),
             (relu_bwd_kernel<bf16_t><<<grid_for(nvec), 256, 0, as_stream(stream)>>>((const bf16_t*)dout, (const bf16_t*)out, nvec, (bf16_t*)dz)));
    AB_LAUNCH_CHECK(); return 0;
}
// column sums (bias gradient) of x [M][C]: part = float [ab_col_stats_nparts(M)][C][2] workspace, out float [C]
extern "C" int ab_col_sum(const void* x, int dtype, long M, int C, float* part, float* out, void* stream) {
    int rc = ab_col_stats(x, dtype, M, C, part, stream);
    if (rc) return rc;
    colsum_finalize_kernel<<<(C + 7) / 8, 256, 0, as_stream(stream)>>>(part, ab_col_stats_nparts(M), C, out);
    AB_LAUNCH_CHECK(); return 0;
}

// ---- split-bf16 producers (fp32 in; see the kernels above).  out (bn_apply) / dz_out may be NULL.
extern "C" int ab_bn_apply_x3(const float* y, const float* res, const float* bnp, long M, int C, int relu, float* out,
                              void* out_hi, void* out_lo, void* stream) {
    if (!y || !bnp || !out_hi || !out_lo) return AB_EINVAL;
    if (C % 8) return AB_ESHAPE;
    const long nvec = M * C / 8;
    bn_apply_x3_kernel<false><<<grid_for(nvec), 256, 0, as_stream(stream)>>>(y, res, bnp, nvec, C, relu, out, (bf16_t*)out_hi, (bf16_t*)out_lo);
    AB_LAUNCH_CHECK(); return 0;
}

// ... with the residual given as the raw output of the downsample conv + its BatchNorm parameters
extern "C" int ab_bn_apply_x3_resbn(const float* y, const float* res_y, const float* bnp, const float* res_bnp, long M, int C,
                                    int relu, float* out, void* out_hi, void* out_lo, void* stream) {
    if (!y || !res_y || !bnp || !res_bnp || !out_hi || !out_lo) return AB_EINVAL;
    if (C % 8) return AB_ESHAPE;
    const long nvec = M * C / 8;
    bn_apply_x3_kernel<true><<<grid_for(nvec), 256, 0, as_stream(stream)>>>(y, res_y, bnp, nvec, C, relu, out, (bf16_t*)out_hi, (bf16_t*)out_lo, res_bnp);
    AB_LAUNCH_CHECK(); return 0;
}

// ... with the residual given as its (hi, lo) planes
extern "C" int ab_bn_apply_x3_respl(const float* y, const void* res_hi, const void* res_lo, const float* bnp, long M, int C, int relu,
                                    float* out, void* out_hi, void* out_lo, void* stream) {
    if (!y || !res_hi || !res_lo || !bnp || !out_hi || !out_lo) return AB_EINVAL;
    if (C % 8) return AB_ESHAPE;
    const long nvec = M * C / 8;
    bn_apply_x3_kernel<false, true><<<grid_for(nvec), 256, 0, as_stream(stream)>>>(y, nullptr, bnp, nvec, C, relu, out, (bf16_t*)out_hi, (bf16_t*)out_lo,
                                                                                 nullptr, (const bf16_t*)res_hi, (const bf16_t*)res_lo);
    AB_LAUNCH_CHECK(); return 0;
}

// BatchNorm finalize + apply in ONE launch (training forward): part [nparts][C][2] are the per-tile (sum, sum of squares) a convolution's
// epilogue left, count = elements per channel; writes bnp [4][C] and updates the running statistics like ab_bn_finalize, then applies like
// ab_bn_apply_x3 / _respl (res_hi != NULL) / _resbn (res_bnp != NULL).  AB_ESHAPE when the shape is not taken (nparts > 64 or C % 64:
// ab_bn_fin_apply_x3_ok says so beforehand) -- the caller then runs ab_bn_finalize + ab_bn_apply_x3*.
extern "C" int ab_bn_fin_apply_x3_ok(int nparts, int C) { return bnfin_ok(nparts, C) ? 1 : 0; }

extern "C" int ab_bn_fin_apply_x3(const float* part, int nparts, long count, const float* gamma, const float* beta, float eps, float momentum,
                                  float* running_mean, float* running_var, float* bnp, const float* y, const float* res,
                                  const void* res_hi, const void* res_lo, const float* res_bnp, long M, int C, int relu, float* out,
                                  void* out_hi, void* out_lo, void* stream) {
    if (!part || !gamma || !beta || !bnp || !y || !out_hi || !out_lo || (!res_hi != !res_lo) || (res_bnp && !res) || (res_hi && (res || res_bnp)))
        return AB_EINVAL;
    if (!bnfin_ok(nparts, C)) return AB_ESHAPE;
    BnFinArgs fa = {part, nparts, count, gamma, beta, eps, momentum, running_mean, running_var, bnp};
    const int grid = bnfin_grid(M, C);
    hipStream_t st = as_stream(stream);
    if (res_hi) bn_fin_apply_x3_kernel<1><<<grid, 256, 0, st>>>(fa, y, nullptr, (const bf16_t*)res_hi, (const bf16_t*)res_lo, nullptr, M, C, relu, out, (bf16_t*)out_hi, (bf16_t*)out_lo);
    else if (res_bnp) bn_fin_apply_x3_kernel<2><<<grid, 256, 0, st>>>(fa, y, res, nullptr, nullptr, res_bnp, M, C, relu, out, (bf16_t*)out_hi, (bf16_t*)out_lo);
    else bn_fin_apply_x3_kernel<0><<<grid, 256, 0, st>>>(fa, y, res, nullptr, nullptr, nullptr, M, C, relu, out, (bf16_t*)out_hi, (bf16_t*)out_lo);
    AB_LAUNCH_CHECK(); return 0;
}

// ab_bn_bwd / ab_bn_bwd_apply with dy as split planes: nparts_given = 0 runs the reduction pass into `part`
// ([ab_col_stats_nparts(M)][C][2]); > 0 takes `part` as already reduced per-tile sums (see ab_bn_bwd_apply).
extern "C" int ab_bn_bwd_x3(const float* dout, const void* out, int out_is_hi_plane, const float* y, const float* bnp, long M, int C,
                            int relu, float* part, int nparts_given, float* bwdp, float* dgamma, float* dbeta, void* dy_hi,
                            void* dy_lo, float* dz_out, void* stream) {
    if (!dout || !y || !bnp || !part || !bwdp || !dgamma || !dbeta || !dy_hi || !dy_lo || (relu == 1 && !out) || relu < 0 || relu > 2)
        return AB_EINVAL;
    const float* out_f = out_is_hi_plane ? nullptr : (const float*)out;
    const bf16_t* out_h = out_is_hi_plane ? (const bf16_t*)out : nullptr;
    int ysl = 1;                                  // channel slices of the reduction (<= 256 four-channel groups each)
    while (C / 4 / ysl > 256) ysl *= 2;
    if (C % 8 || C % (4 * ysl)) return AB_ESHAPE;
    hipStream_t st = as_stream(stream);
    int np = nparts_given > 0 ? nparts_given : ab_col_stats_nparts(M);
    if (nparts_given <= 0) {
        const int cs = C / ysl, rl = 256 / (cs / 4); const size_t sh = (size_t)rl * cs * 2 * 4;
        bn_bwd_reduce_kernel<float><<<dim3(np, ysl), 256, sh, st>>>(dout, out_f, y, bnp, M, C, relu, red_rows(M), part, nullptr, 0, 0, out_h);
        AB_LAUNCH_CHECK();
    }
    if (bnfin_ok(np, C)) {          // few partial rows: the apply pass reduces them itself (no finalize launch)
        bn_fin_bwd_apply_x3_kernel<<<bnfin_grid(M, C), 256, 0, st>>>(part, np, dgamma, dbeta, bwdp, dout, out_f, y, bnp, C, M, relu, (bf16_t*)dy_hi,
                                                                     (bf16_t*)dy_lo, dz_out, out_h);
        AB_LAUNCH_CHECK(); return 0;
    }
    launch_bn_bwd_finalize(part, np, C, dgamma, dbeta, bwdp, st);
    AB_LAUNCH_CHECK();
    const long nvec = M * C / 8;
    bn_bwd_apply_x3_kernel<<<grid_for(nvec), 256, 0, st>>>(dout, out_f, y, bnp, bwdp, nvec, C, M, relu, (bf16_t*)dy_hi, (bf16_t*)dy_lo, dz_out, out_h);
    AB_LAUNCH_CHECK(); return 0;
}

// Backward of  maxpool3x3/2( relu( bn(y) ) )  for the split-bf16 path (resnet.py:154-157 backwards; the stem).  One pass scatters
// the pooled gradient to the full resolution, masks it and reduces it (pool_bwd_bn_reduce_kernel, part rows =
// ab_bn_relu_maxpool_bwd_x3_nparts); finalize; the apply pass repeats the gather and writes dy as planes (pool_bwd_bn_apply_x3_kernel;
// AB_POOL_BWD_REGATHER=0: the reduce pass stores the masked gradient in the fp32 scratch `dz` and the generic apply pass reads it).
// row pairs per workgroup of the reduction: 4 / 8 / 16 -> 91 + 18 / 92 + 9 / 110 + 6 us for the pass + its finalize over 2 048 / 1 024 /
// 512 partial rows
static int pbr_rp() { static const int v = getenv("AB_PBR_RP") ? atoi(getenv("AB_PBR_RP")) : 8; return (v == 4 || v == 16) ? v : 8; }
extern "C" int ab_bn_relu_maxpool_bwd_x3_nparts(int N, int H, int W, int C) {
    if ((H & 1) || (W & 1) || C % 8 || C / 4 > 256 || 256 % (C / 4)) return 0;
    const int gx = (int)(((long)W * (C / 4) + 255) / 256), gy = (H / 2 + pbr_rp() - 1) / pbr_rp();
    return gx * gy * N;
}
extern "C" int ab_bn_relu_maxpool_bwd_x3(const float* dpool, const void* idx, const float* y, const float* bnp, int N, int H, int W,
                                         int C, float* part, float* bwdp, float* dgamma, float* dbeta, float* dz, void* dy_hi,
                                         void* dy_lo, void* stream) {
    if (!dpool || !idx || !y || !bnp || !part || !bwdp || !dgamma || !dbeta || !dz || !dy_hi || !dy_lo) return AB_EINVAL;
    const int np = ab_bn_relu_maxpool_bwd_x3_nparts(N, H, W, C);
    if (!np) return AB_ESHAPE;
    hipStream_t st = as_stream(stream);
    dim3 grid((unsigned)(((long)W * (C / 4) + 255) / 256), (unsigned)((H / 2 + pbr_rp() - 1) / pbr_rp()), (unsigned)N);
    static const int regather = getenv("AB_POOL_BWD_REGATHER") ? atoi(getenv("AB_POOL_BWD_REGATHER")) : 1;
    const long M = (long)N * H * W, nvec = M * C / 8;
#define PBR_LAUNCH(RP, WR) pool_bwd_bn_reduce_kernel<RP, WR><<<grid, 256, 256 * 8 * 4, st>>>((const uint8_t*)idx, dpool, y, bnp, N, H, W, C, dz, part)
    const int rp = pbr_rp();
    if (regather) { if (rp == 4) PBR_LAUNCH(4, false); else if (rp == 8) PBR_LAUNCH(8, false); else PBR_LAUNCH(16, false); }
    else { if (rp == 4) PBR_LAUNCH(4, true); else if (rp == 8) PBR_LAUNCH(8, true); else PBR_LAUNCH(16, true); }
#undef PBR_LAUNCH
    AB_LAUNCH_CHECK();
    launch_bn_bwd_finalize(part, np, C, dgamma, dbeta, bwdp, st);
    AB_LAUNCH_CHECK();
    if (regather) {
        dim3 g2((unsigned)(((long)W * (C / 8) + 255) / 256), (unsigned)(H / 2), (unsigned)N);
        pool_bwd_bn_apply_x3_kernel<<<g2, 256, 0, st>>>((const uint8_t*)idx, dpool, y, bnp, bwdp, N, H, W, C, 1.f / (float)M,
                                                       (bf16_t*)dy_hi, (bf16_t*)dy_lo);
    } else
        bn_bwd_apply_x3_kernel<<<grid_for(nvec), 256, 0, st>>>(dz, nullptr, y, bnp, bwdp, nvec, C, M, 0, (bf16_t*)dy_hi, (bf16_t*)dy_lo, nullptr, nullptr);
    AB_LAUNCH_CHECK(); return 0;
}

// ab_bn_relu_maxpool_bwd_x3 with the reduction over the pooled elements (pool_win_bn_reduce_kernel: dpool + ywin, 134 MB at the
// benchmark size instead of 352); part: ab_col_stats_nparts(N*H/2*W/2) rows.
extern "C" int ab_bn_relu_maxpool_bwd_x3w(const float* dpool, const void* idx, const float* ywin, const float* y, const float* bnp,
                                          int N, int H, int W, int C, float* part, float* bwdp, float* dgamma, float* dbeta,
                                          void* dy_hi, void* dy_lo, void* stream) {
    if (!dpool || !idx || !ywin || !y || !bnp || !part || !bwdp || !dgamma || !dbeta || !dy_hi || !dy_lo) return AB_EINVAL;
    if ((H & 1) || (W & 1) || C % 8 || C / 8 > 256) return AB_ESHAPE;
    hipStream_t st = as_stream(stream);
    const long Mp = (long)N * (H / 2) * (W / 2), M = (long)N * H * W;
    const int np = ab_col_stats_nparts(Mp), rl = 256 / (C / 8);
    pool_win_bn_reduce_kernel<<<np, 256, (size_t)rl * C * 2 * 4, st>>>(dpool, ywin, bnp, Mp, C, red_rows(Mp), part);
    AB_LAUNCH_CHECK();
    launch_bn_bwd_finalize(part, np, C, dgamma, dbeta, bwdp, st);
    AB_LAUNCH_CHECK();
    dim3 g2((unsigned)(((long)W * (C / 8) + 255) / 256), (unsigned)(H / 2), (unsigned)N);
    pool_bwd_bn_apply_x3_kernel<<<g2, 256, 0, st>>>((const uint8_t*)idx, dpool, y, bnp, bwdp, N, H, W, C, 1.f / (float)M,
                                                   (bf16_t*)dy_hi, (bf16_t*)dy_lo);
    AB_LAUNCH_CHECK(); return 0;
}

// ab_transpose_oki_batch with bf16 outputs as split planes: hi at desc.dst, lo at desc.dst + lo_offset_elems
extern "C" int ab_transpose_oki_batch_x3(const ab_transpose_desc* desc_dev, int ntensors, long total_tiles, long lo_offset_elems,
                                         void* stream) {
    if (!desc_dev || ntensors < 1 || total_tiles < 1 || total_tiles > 0x7fffffffL || lo_offset_elems <= 0) return AB_EINVAL;
    transpose_oki_batch_kernel<bf16_t><<<(unsigned)total_tiles, 256, 0, as_stream(stream)>>>(desc_dev, ntensors, lo_offset_elems);
    AB_LAUNCH_CHECK(); return 0;
}

// ab_col_sum over a split tensor (planes hi, lo: bf16 [M][C]); part as in ab_col_sum
extern "C" int ab_col_sum_x3(const void* hi, const void* lo, long M, int C, float* part, float* out, void* stream) {
    if (!hi || !lo || !part || !out) return AB_EINVAL;
    if (C % 8 || C / 8 > 256) return AB_ESHAPE;
    const int np = ab_col_stats_nparts(M), rl = 256 / (C / 8);
    col_stats_x3_kernel<<<np, 256, (size_t)rl * C * 2 * 4, as_stream(stream)>>>((const bf16_t*)hi, (const bf16_t*)lo, M, C, red_rows(M), part);
    AB_LAUNCH_CHECK();
    colsum_finalize_kernel<<<(C + 7) / 8, 256, 0, as_stream(stream)>>>(part, np, C, out);
    AB_LAUNCH_CHECK(); return 0;
}
