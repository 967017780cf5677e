// Fused softmax + 3-D integral (soft-argmax) head, forward and backward.  HBM-bound: the logits are read exactly
// once forward (reference: ~5 passes, simplebaseline.py:183-189 + 43-71) and once backward.
//
// Layout: logits NHWC [B, H, W, C*D]; a pixel's C*D channel vector is contiguous, so a wave reads fully coalesced
// rows.  Stage 1: one workgroup per (batch, pixel tile); each lane owns channel slots ch = lane + 64*k and runs an
// online (max, sum, moments) softmax over the tile's pixels for those channels; depth slots of one class are then
// merged through LDS.  Stage 2: merge tiles per (b, c) with max-rescaling and emit uvd / conf / stat.
#include "common.h"

#define SAM_TILE_PIX 64      // pixels per workgroup
#define SAM_THREADS 256
#define SAM_MAXCH 1024       // C*D upper bound handled (22*28 = 616)

struct Acc { float m, s, su, sv, sd; };

// NORM_TYPE of IntegralDeconvHead (simplebaseline.py:16-40): 0 = softmax, 1 = sigmoid.  The sigmoid head is
//   w = sigmoid(x),  conf = max w,  uvd = sum(w * coord) / (sum(w) + 1e-7)
// which is the softmax machinery below applied to x' = log w = log sigmoid(x): exp(x' - m') = w / w_max, so the online (max, sum,
// moments) accumulation, the tile merge and the backward's exp(x' - m') factor carry over unchanged; stage 2 rescales by
// w_max = exp(m') for the 1e-7 and the confidence, the backward multiplies by d x'/d x = sigmoid(-x).
// ("divide_sum" -- raw, possibly negative weights -- has no log form; the reference itself warns against it.  Not implemented.)
template <int NORM> __device__ __forceinline__ float sam_pre(float x) {
    if (NORM == 1) return fminf(x, 0.f) - log1pf(__expf(-fabsf(x)));       // log sigmoid(x), stable on both sides
    return x;
}
template <int NORM> __device__ __forceinline__ float sam_dpre(float x) {  // d sam_pre / dx
    if (NORM == 1) return 1.f / (1.f + __expf(x));                          // sigmoid(-x)
    return 1.f;
}

__device__ __forceinline__ void acc_merge(Acc& a, const Acc& b) {
    float m = fmaxf(a.m, b.m);
    float fa = (a.m == -INFINITY) ? 0.f : __expf(a.m - m);
    float fb = (b.m == -INFINITY) ? 0.f : __expf(b.m - m);
    a.s = a.s * fa + b.s * fb;
    a.su = a.su * fa + b.su * fb;
    a.sv = a.sv * fa + b.sv * fb;
    a.sd = a.sd * fa + b.sd * fb;
    a.m = m;
}

// ---- generic scalar kernels (any C*DP; used when C*DP is odd, where rows are not 4-byte aligned)
template <typename T, int NORM = 0>
__global__ __launch_bounds__(SAM_THREADS) void sam_stage1_scalar(const T* __restrict__ logits, int C, int D, int DP, int H, int W,
                                                          int ntile, float* __restrict__ part) {
    // grid: (ntile, B).  4 waves; wave w handles pixels w, w+4, ... of the tile; lane handles channels lane+64k.
    const int CD = C * DP;   // channel = c*DP + d, d < D valid (DP >= D: padded depth pitch)
    const int b = blockIdx.y, tile = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int npix = H * W;
    const int p0 = tile * SAM_TILE_PIX;
    constexpr int KMAX = SAM_MAXCH / 64;
    Acc acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] = {-INFINITY, 0.f, 0.f, 0.f, 0.f};
    const float invW = 1.f / W, invH = 1.f / H;
    for (int pi = wave; pi < SAM_TILE_PIX; pi += 4) {
        int p = p0 + pi;
        if (p >= npix) break;
        int h = p / W, w = p - h * W;
        const T* row = logits + ((size_t)b * npix + p) * CD;
        float cu = w * invW, cv = h * invH;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            int ch = lane + 64 * k;
            if (ch < CD && (ch % DP) < D) {
                float x = sam_pre<NORM>(ld_f32(row + ch));
                float m = fmaxf(acc[k].m, x);
                float f = (acc[k].m == -INFINITY) ? 0.f : __expf(acc[k].m - m);
                float e = __expf(x - m);
                acc[k].s = acc[k].s * f + e;
                acc[k].su = acc[k].su * f + e * cu;
                acc[k].sv = acc[k].sv * f + e * cv;
                acc[k].m = m;
            }
        }
    }
    // per-channel accumulators -> LDS [wave][ch], then reduce over waves and over the D channels of each class
    __shared__ float sm[4][SAM_MAXCH][4];  // m, s, su, sv   (64 KiB)
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        int ch = lane + 64 * k;
        if (ch < CD) {
            sm[wave][ch][0] = acc[k].m; sm[wave][ch][1] = acc[k].s;
            sm[wave][ch][2] = acc[k].su; sm[wave][ch][3] = acc[k].sv;
        }
    }
    __syncthreads();
    // one thread per class
    for (int c = threadIdx.x; c < C; c += SAM_THREADS) {
        Acc r = {-INFINITY, 0.f, 0.f, 0.f, 0.f};
        const float invD = 1.f / D;
        for (int d = 0; d < D; ++d) {
            int ch = c * DP + d;
            for (int wv = 0; wv < 4; ++wv) {
                Acc t = {sm[wv][ch][0], sm[wv][ch][1], sm[wv][ch][2], sm[wv][ch][3], 0.f};
                t.sd = t.s * (d * invD);
                acc_merge(r, t);
            }
        }
        float* o = part + (((size_t)b * ntile + tile) * C + c) * 8;
        o[0] = r.m; o[1] = r.s; o[2] = r.su; o[3] = r.sv; o[4] = r.sd;
    }
}


// ---- pair kernels (C*DP even): a lane owns adjacent channel pairs and moves them with one 4/8-byte access
template <typename T> __device__ __forceinline__ void ld_pair(const T* p, float& a, float& b);
template <> __device__ __forceinline__ void ld_pair<float>(const float* p, float& a, float& b) {
    float2 v = *(const float2*)p; a = v.x; b = v.y;
}
template <> __device__ __forceinline__ void ld_pair<bf16_t>(const bf16_t* p, float& a, float& b) {
    uint32_t v = *(const uint32_t*)p; a = __uint_as_float(v << 16); b = __uint_as_float(v & 0xffff0000u);
}

// KP: channel-pair groups of 128 channels a lane walks (ceil(C*DP / 128) rounded up to an instantiated value): the launch of the
// 22 x 32 head (704 channels) runs with 6 instead of the maximal 8 -- a quarter fewer loads and exps, 48 instead of 64 KB of LDS.
template <typename T, int KP = SAM_MAXCH / 128, int NORM = 0>
__global__ __launch_bounds__(SAM_THREADS) void sam_stage1(const T* __restrict__ logits, int C, int D, int DP, int H, int W,
                                                          int ntile, float* __restrict__ part) {
    // grid: (ntile, B).  4 waves; wave w handles pixels w, w+4, ... of the tile, four at a time (all their loads in flight,
    // one max-rescale per four pixels); a lane owns the channel pairs 2*(lane + 64k), +1.  Branch-free: pixels past the
    // end contribute x = -inf (exp -> 0); padded / out-of-range channel slots are computed and never read back.
    const int CD = C * DP;   // channel = c*DP + d, d < D valid (DP >= D: padded depth pitch); even here
    const int b = blockIdx.y, tile = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int npix = H * W;
    const int p0 = tile * SAM_TILE_PIX;
    constexpr int NCH = 2 * KP;
    constexpr float NEG = -3.0e38f;          // finite stand-in for -inf: exp(NEG - m) == 0 without inf - inf NaNs
    float am[NCH], as[NCH], asu[NCH], asv[NCH];
    int choff[KP];
#pragma unroll
    for (int k = 0; k < NCH; ++k) { am[k] = NEG; as[k] = 0.f; asu[k] = 0.f; asv[k] = 0.f; }
#pragma unroll
    for (int k = 0; k < KP; ++k) choff[k] = min(2 * (lane + 64 * k), CD - 2);
    const float invW = 1.f / W, invH = 1.f / H;
    const T* base = logits + (size_t)b * npix * CD;
#pragma unroll 1
    for (int pi = wave; pi < SAM_TILE_PIX; pi += 16) {
        float x[4][NCH], cu[4], cv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int p = p0 + pi + 4 * j;
            const bool pv = (pi + 4 * j) < SAM_TILE_PIX && p < npix;
            int pc = pv ? p : 0;
            int h = pc / W, w = pc - h * W;
            cu[j] = w * invW; cv[j] = h * invH;
            const T* row = base + (size_t)pc * CD;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                float a, c2;
                ld_pair<T>(row + choff[k], a, c2);
                x[j][2 * k] = pv ? sam_pre<NORM>(a) : -INFINITY; x[j][2 * k + 1] = pv ? sam_pre<NORM>(c2) : -INFINITY;
            }
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            float m = fmaxf(fmaxf(fmaxf(x[0][k], x[1][k]), fmaxf(x[2][k], x[3][k])), am[k]);
            float f = __expf(am[k] - m);
            float s = as[k] * f, su = asu[k] * f, sv = asv[k] * f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { float e = __expf(x[j][k] - m); s += e; su += e * cu[j]; sv += e * cv[j]; }
            am[k] = m; as[k] = s; asu[k] = su; asv[k] = sv;
        }
    }
    // per-channel accumulators -> LDS [wave][ch], then reduce over waves and over the D channels of each class
    __shared__ float sm[4][KP * 128][4];  // m, s, su, sv
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        int ch = 2 * (lane + 64 * (k >> 1)) + (k & 1);
        if (ch < CD) {
            sm[wave][ch][0] = am[k]; sm[wave][ch][1] = as[k];
            sm[wave][ch][2] = asu[k]; sm[wave][ch][3] = asv[k];
        }
    }
    __syncthreads();
    if (D <= 32) {
        // 32 lanes per class: lane d merges the four wave entries of channel (c, d), then a 5-step butterfly over the depth
        // bins (fixed order).  The former one-thread-per-class loop was 4*D dependent merges on 22 lanes while the other
        // 234 idled -- ~2 us at the end of each of the 8 workgroup rounds of a B = 64 launch.
        const int d = threadIdx.x & 31;
        const float invD = 1.f / D;
        for (int c0 = 0; c0 < C; c0 += SAM_THREADS / 32) {
            const int c = c0 + (threadIdx.x >> 5);
            Acc r = {-INFINITY, 0.f, 0.f, 0.f, 0.f};
            if (c < C && d < D) {
                const int ch = c * DP + d;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) {
                    Acc t = {sm[wv][ch][0], sm[wv][ch][1], sm[wv][ch][2], sm[wv][ch][3], 0.f};
                    t.sd = t.s * (d * invD);
                    acc_merge(r, t);
                }
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                Acc t = {__shfl_xor(r.m, o, 64), __shfl_xor(r.s, o, 64), __shfl_xor(r.su, o, 64), __shfl_xor(r.sv, o, 64), __shfl_xor(r.sd, o, 64)};
                acc_merge(r, t);
            }
            if (d == 0 && c < C) {
                float* o = part + (((size_t)b * ntile + tile) * C + c) * 8;
                o[0] = r.m; o[1] = r.s; o[2] = r.su; o[3] = r.sv; o[4] = r.sd;
            }
        }
        return;
    }
    // one thread per class
    for (int c = threadIdx.x; c < C; c += SAM_THREADS) {
        Acc r = {-INFINITY, 0.f, 0.f, 0.f, 0.f};
        const float invD = 1.f / D;
        for (int d = 0; d < D; ++d) {
            int ch = c * DP + d;
            for (int wv = 0; wv < 4; ++wv) {
                Acc t = {sm[wv][ch][0], sm[wv][ch][1], sm[wv][ch][2], sm[wv][ch][3], 0.f};
                t.sd = t.s * (d * invD);
                acc_merge(r, t);
            }
        }
        float* o = part + (((size_t)b * ntile + tile) * C + c) * 8;
        o[0] = r.m; o[1] = r.s; o[2] = r.su; o[3] = r.sv; o[4] = r.sd;
    }
}

__global__ void sam_stage2(const float* __restrict__ part, int C, int ntile, float* __restrict__ uvd,
                           float* __restrict__ conf, float* __restrict__ stat, int norm = 0) {
    // one wave per (b, c)
    const int bc = blockIdx.x;
    const int b = bc / C, c = bc - b * C;
    const int lane = threadIdx.x;
    Acc r = {-INFINITY, 0.f, 0.f, 0.f, 0.f};
    for (int t = lane; t < ntile; t += 64) {
        const float* o = part + (((size_t)b * ntile + t) * C + c) * 8;
        Acc a = {o[0], o[1], o[2], o[3], o[4]};
        acc_merge(r, a);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Acc o;
        o.m = __shfl_xor(r.m, off, 64); o.s = __shfl_xor(r.s, off, 64); o.su = __shfl_xor(r.su, off, 64);
        o.sv = __shfl_xor(r.sv, off, 64); o.sd = __shfl_xor(r.sd, off, 64);
        acc_merge(r, o);
    }
    if (lane == 0) {
        // softmax sums to 1 (up to rounding); the reference then divides by (sum + 1e-7): simplebaseline.py:187
        const float z = 1.0f + 1e-7f;
        float inv = 1.f / (r.s * z);
        float cf = 1.f / r.s;  // max p = exp(max - max) / sum
        if (norm == 1) {       // sigmoid: sums are in units of w_max = exp(m): uvd = su w_max / (s w_max + 1e-7), conf = w_max
            const float wmax = __expf(r.m);
            inv = wmax / (r.s * wmax + 1e-7f);
            cf = wmax;
        }
        uvd[bc * 3 + 0] = r.su * inv;
        uvd[bc * 3 + 1] = r.sv * inv;
        uvd[bc * 3 + 2] = r.sd * inv;
        conf[bc] = cf;
        stat[bc * 2 + 0] = r.m;
        stat[bc * 2 + 1] = r.s;
    }
}

template <typename T, int NORM = 0>
__global__ __launch_bounds__(256) void sam_bwd_scalar(const T* __restrict__ logits, int C, int D, int DP, int H, int W,
                                               const float* __restrict__ uvd, const float* __restrict__ conf,
                                               const float* __restrict__ stat, const float* __restrict__ g_uvd,
                                               const float* __restrict__ g_conf, T* __restrict__ dlogits) {
    // grid: (ceil(npix / 4), B); each wave handles one pixel row of C*D channels
    const int CD = C * DP, npix = H * W;
    const int b = blockIdx.y;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= npix) return;
    const int h = p / W, w = p - h * W;
    const float cu = (float)w / W, cv = (float)h / H;
    const float z = 1.0f + 1e-7f;
    const T* row = logits + ((size_t)b * npix + p) * CD;
    T* drow = dlogits + ((size_t)b * npix + p) * CD;
    for (int ch = lane; ch < CD; ch += 64) {
        int c = ch / DP, d = ch - c * DP;
        if (d >= D) { st_f32(drow + ch, 0.f); continue; }
        int bc = b * C + c;
        const float xraw = ld_f32(row + ch);
        float x = sam_pre<NORM>(xraw);
        float m = stat[bc * 2], s = stat[bc * 2 + 1];
        float pr = __expf(x - m) / s;
        float u = uvd[bc * 3] * z, v = uvd[bc * 3 + 1] * z, dd = uvd[bc * 3 + 2] * z;  // sum p*coord
        float g = g_uvd[bc * 3] * (cu - u) + g_uvd[bc * 3 + 1] * (cv - v) + g_uvd[bc * 3 + 2] * ((float)d / D - dd);
        float out = pr * g / z;
        if (g_conf) {
            float cf = conf[bc];
            out += g_conf[bc] * cf * ((x == m ? 1.f : 0.f) - pr);
        }
        out *= sam_dpre<NORM>(xraw);
        st_f32(drow + ch, out);
    }
}


template <typename T> __device__ __forceinline__ void st_pair(T* p, float a, float b);
template <> __device__ __forceinline__ void st_pair<float>(float* p, float a, float b) { *(float2*)p = make_float2(a, b); }
template <> __device__ __forceinline__ void st_pair<bf16_t>(bf16_t* p, float a, float b) {
    *(uint32_t*)p = pack_bf16x2(a, b);
}

#define SAM_BWD_PIX 16      // pixels per workgroup (4 per wave)

// d logits = p * (g_u*(cu-u) + g_v*(cv-v) + g_d*(d/D-dd)) / z  [+ g_conf * conf * ([x == max] - p)],  p = exp(x-m)/s.
// Everything that depends only on (b, channel) is folded into per-lane constants once per workgroup:
//   out = exp(x - m) * (A*cu + Bv*cv + K),  A = g_u/(s z), Bv = g_v/(s z), K = (g_d*(d/D-dd) - g_u*u - g_v*v)/(s z)
// SPLIT (T = float): dlogits leaves as the split-bf16 planes the final layer's data / weight gradients consume
// (dl_hi, dl_lo: bf16 [B, H, W, C*DP]) instead of fp32 -- no separate split pass over the largest activation of the net.
// colpart != NULL (PIX pixels per workgroup): the workgroup also writes the column sums of its dlogits rows (fixed order: a
// lane over its pixels, then the four waves), colpart[(b * gridDim.x + blockIdx.x)][CD] -- the final layer's bias gradient
// without another pass over the planes.
// V, KV: a lane walks KV groups of V consecutive channels (group k: channels V * (lane + 64 k) ..): V = 2 covers any even C*DP with 8-byte loads
// and 4-byte plane stores; V = 4 (C*DP % 4 == 0) moves 16 bytes per load and 8 per plane store, and KV = 3 covers the 22 x 32 head's 704 channels
// with 768 slots where the V = 2 walk always took SAM_MAXCH = 1024 (a quarter of its loads and exps were padding): 98 -> see DESIGN 13.8.
template <typename T, bool SPLIT = false, int PIX = SAM_BWD_PIX, int NORM = 0, int V = 2, int KV = SAM_MAXCH / 128>
__global__ __launch_bounds__(256) void sam_bwd(const T* __restrict__ logits, int C, int D, int DP, int H, int W,
                                               const float* __restrict__ uvd, const float* __restrict__ conf,
                                               const float* __restrict__ stat, const float* __restrict__ g_uvd,
                                               const float* __restrict__ g_conf, T* __restrict__ dlogits,
                                               bf16_t* __restrict__ dl_hi = nullptr, bf16_t* __restrict__ dl_lo = nullptr,
                                               float* __restrict__ colpart = nullptr) {
    const int CD = C * DP, npix = H * W;      // CD even
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int NCH = V * KV;
    static_assert((V == 2 || V == 4) && NCH * 64 <= SAM_MAXCH, "channel walk");
    const float z = 1.0f + 1e-7f;
    // per-channel constants: computed once per workgroup (rolled loop), exchanged through LDS
    __shared__ float cst[6][SAM_MAXCH];
#pragma unroll 1
    for (int ch = threadIdx.x; ch < SAM_MAXCH; ch += 256) {
        int c = ch / DP, d = ch - c * DP;
        float m = INFINITY, a = 0.f, bb = 0.f, k0 = 0.f, is = 0.f, gc = 0.f;   // exp(x - inf) = 0: padded slots get zeros
        if (ch < CD && d < D) {
            int bc = b * C + c;
            float s = stat[bc * 2];
            m = s; s = stat[bc * 2 + 1];
            float u = uvd[bc * 3] * z, v = uvd[bc * 3 + 1] * z, dd = uvd[bc * 3 + 2] * z;  // sum p*coord
            float gu = g_uvd[bc * 3], gv = g_uvd[bc * 3 + 1], gd = g_uvd[bc * 3 + 2];
            float sc = 1.f / (s * z);
            a = gu * sc; bb = gv * sc;
            k0 = (gd * ((float)d / D - dd) - gu * u - gv * v) * sc;
            is = 1.f / s;
            gc = g_conf ? g_conf[bc] * conf[bc] : 0.f;
        }
        cst[0][ch] = m; cst[1][ch] = a; cst[2][ch] = bb; cst[3][ch] = k0; cst[4][ch] = is; cst[5][ch] = gc;
    }
    __syncthreads();
    float cm[NCH], ca[NCH], cb[NCH], ck[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        int ch = V * (lane + 64 * (k / V)) + (k % V);
        cm[k] = cst[0][ch]; ca[k] = cst[1][ch]; cb[k] = cst[2][ch]; ck[k] = cst[3][ch];
    }
    float csum[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) csum[k] = 0.f;
#pragma unroll 1
    for (int j = 0; j < PIX / 4; ++j) {
        const int p = blockIdx.x * PIX + j * 4 + wave;
        if (p >= npix) break;
        const int h = p / W, w = p - h * W;
        const float cu = (float)w / W, cv = (float)h / H;
        const T* row = logits + ((size_t)b * npix + p) * CD;
        T* drow = SPLIT ? nullptr : dlogits + ((size_t)b * npix + p) * CD;
        const size_t prow = ((size_t)b * npix + p) * CD;
        float x[NCH];
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const T* src = row + min(V * (lane + 64 * k), CD - V);
            ld_pair<T>(src, x[V * k], x[V * k + 1]);
            if constexpr (V == 4) ld_pair<T>(src + 2, x[V * k + 2], x[V * k + 3]);      // (merged with the first into one 16-byte load)
        }
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            int ch = V * (lane + 64 * k);
            float o[V];
#pragma unroll
            for (int t = 0; t < V; ++t) {
                int kk = V * k + t;
                const float xp = sam_pre<NORM>(x[kk]);
                float e = __expf(xp - cm[kk]);
                float out = e * ((ca[kk] * cu + cb[kk] * cv) + ck[kk]);
                if (g_conf) out += cst[5][ch + t] * ((xp == cm[kk] ? 1.f : 0.f) - e * cst[4][ch + t]);
                if (NORM != 0) out *= sam_dpre<NORM>(x[kk]);
                o[t] = out;
                csum[kk] += out;
            }
            if (ch < CD) {
                if constexpr (SPLIT) {
                    const uint32_t hw = pack_bf16x2(o[0], o[1]);
                    const uint32_t lw = pack_bf16x2(o[0] - __uint_as_float(hw << 16), o[1] - __uint_as_float(hw & 0xffff0000u));
                    if constexpr (V == 4) {
                        const uint32_t hw2 = pack_bf16x2(o[2], o[3]);
                        const uint32_t lw2 = pack_bf16x2(o[2] - __uint_as_float(hw2 << 16), o[3] - __uint_as_float(hw2 & 0xffff0000u));
                        *(uint2*)(dl_hi + prow + ch) = make_uint2(hw, hw2);
                        *(uint2*)(dl_lo + prow + ch) = make_uint2(lw, lw2);
                    } else {
                        *(uint32_t*)(dl_hi + prow + ch) = hw;
                        *(uint32_t*)(dl_lo + prow + ch) = lw;
                    }
                } else {
                    st_pair<T>(drow + ch, o[0], o[1]);
                    if constexpr (V == 4) st_pair<T>(drow + ch + 2, o[2], o[3]);
                }
            }
        }
    }
    if (colpart) {
        __syncthreads();                                     // cst is free: rows 0..3 take the four waves' sums
#pragma unroll
        for (int k = 0; k < NCH; ++k) cst[wave][V * (lane + 64 * (k / V)) + (k % V)] = csum[k];
        __syncthreads();
        float* dst = colpart + ((size_t)b * gridDim.x + blockIdx.x) * CD;
        for (int ch = threadIdx.x; ch < CD; ch += 256) dst[ch] = ((cst[0][ch] + cst[1][ch]) + cst[2][ch]) + cst[3][ch];
    }
}

// column sums of colpart [rows][CD] in fixed order -> out [CD] (8 channels x 32 row lanes per workgroup, as colsum_finalize_kernel)
__global__ __launch_bounds__(256) void sam_bias_finalize(const float* __restrict__ part, int rows, int CD, float* __restrict__ out) {
    const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    double s = 0.0;
    if (c < CD) for (int k = pl; k < rows; k += 32) s += part[(size_t)k * CD + c];
    __shared__ double sm[32][8];
    sm[pl][cl] = s;
    __syncthreads();
    if (pl == 0 && c < CD) { for (int k = 1; k < 32; ++k) s += sm[k][cl]; out[c] = (float)s; }
}

extern "C" int ab_softargmax3d_ntiles(int H, int W) { return (H * W + SAM_TILE_PIX - 1) / SAM_TILE_PIX; }

static int sam_fwd_impl(const void* logits, int dtype, int B, int C, int D, int DP, int H, int W, float* part,
                        float* uvd, float* conf, float* stat, int norm, void* stream) {
    if (!logits || !part || !uvd || !conf || !stat) return AB_EINVAL;
    if (B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || DP < D || C * DP > SAM_MAXCH || norm < 0 || norm > 1) return AB_ESHAPE;
    int ntile = ab_softargmax3d_ntiles(H, W);
    dim3 grid(ntile, B);
    const bool pair = ((C * DP) & 1) == 0;
    hipStream_t st = as_stream(stream);
    const int kp = (C * DP + 127) / 128;
#define SAM_S1(TT, NN) \
    do { \
        if (!pair) sam_stage1_scalar<TT, NN><<<grid, SAM_THREADS, 0, st>>>((const TT*)logits, C, D, DP, H, W, ntile, part); \
        else if (kp <= 2) sam_stage1<TT, 2, NN><<<grid, SAM_THREADS, 0, st>>>((const TT*)logits, C, D, DP, H, W, ntile, part); \
        else if (kp <= 4) sam_stage1<TT, 4, NN><<<grid, SAM_THREADS, 0, st>>>((const TT*)logits, C, D, DP, H, W, ntile, part); \
        else if (kp <= 6) sam_stage1<TT, 6, NN><<<grid, SAM_THREADS, 0, st>>>((const TT*)logits, C, D, DP, H, W, ntile, part); \
        else sam_stage1<TT, 8, NN><<<grid, SAM_THREADS, 0, st>>>((const TT*)logits, C, D, DP, H, W, ntile, part); \
    } while (0)
    if (dtype == AB_DT_F32) { if (norm) SAM_S1(float, 1); else SAM_S1(float, 0); }
    else if (dtype == AB_DT_BF16) { if (norm) SAM_S1(bf16_t, 1); else SAM_S1(bf16_t, 0); }
    else return AB_EINVAL;
#undef SAM_S1
    AB_LAUNCH_CHECK();
    sam_stage2<<<B * C, 64, 0, st>>>(part, C, ntile, uvd, conf, stat, norm);
    AB_LAUNCH_CHECK();
    return 0;
}

extern "C" int ab_softargmax3d_fwd(const void* logits, int dtype, int B, int C, int D, int DP, int H, int W, float* part,
                                   float* uvd, float* conf, float* stat, void* stream) {
    return sam_fwd_impl(logits, dtype, B, C, D, DP, H, W, part, uvd, conf, stat, 0, stream);
}
extern "C" int ab_softargmax3d_fwd_norm(const void* logits, int dtype, int B, int C, int D, int DP, int H, int W, int norm_type, float* part,
                                        float* uvd, float* conf, float* stat, void* stream) {
    return sam_fwd_impl(logits, dtype, B, C, D, DP, H, W, part, uvd, conf, stat, norm_type, stream);
}

// Second stage alone: `part` [B][ntile][C][8] was written by whoever produced the logits (ab_conv1x1_sam_fwd_x3: the final layer's
// GEMM epilogue), ntile = ab_softargmax3d_ntiles(H, W).  Softmax head only.
extern "C" int ab_softargmax3d_stage2(const float* part, int B, int C, int ntile, float* uvd, float* conf, float* stat, void* stream) {
    if (!part || !uvd || !conf || !stat) return AB_EINVAL;
    if (B <= 0 || C <= 0 || ntile <= 0) return AB_ESHAPE;
    sam_stage2<<<B * C, 64, 0, as_stream(stream)>>>(part, C, ntile, uvd, conf, stat, 0);
    AB_LAUNCH_CHECK();
    return 0;
}

static int sam_bwd_impl(const void* logits, int dtype, int B, int C, int D, int DP, int H, int W, const float* uvd,
                        const float* conf, const float* stat, const float* g_uvd, const float* g_conf, void* dlogits, int norm, void* stream) {
    if (!logits || !uvd || !conf || !stat || !g_uvd || !dlogits) return AB_EINVAL;
    if (B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || DP < D || C * DP > SAM_MAXCH || norm < 0 || norm > 1) return AB_ESHAPE;
    if (norm && g_conf) return AB_EINVAL;          // the sigmoid head's confidence (max w) is not differentiated here
    const bool pair = ((C * DP) & 1) == 0;
    hipStream_t st = as_stream(stream);
    dim3 grid(pair ? (H * W + SAM_BWD_PIX - 1) / SAM_BWD_PIX : (H * W + 3) / 4, B);
#define SAM_BWD_ARGS(TT) (const TT*)logits, C, D, DP, H, W, uvd, conf, stat, g_uvd, g_conf, (TT*)dlogits
#define SAM_BW(TT, NN) \
    do { \
        if (pair) sam_bwd<TT, false, SAM_BWD_PIX, NN><<<grid, 256, 0, st>>>(SAM_BWD_ARGS(TT)); \
        else sam_bwd_scalar<TT, NN><<<grid, 256, 0, st>>>(SAM_BWD_ARGS(TT)); \
    } while (0)
    if (dtype == AB_DT_F32) { if (norm) SAM_BW(float, 1); else SAM_BW(float, 0); }
    else if (dtype == AB_DT_BF16) { if (norm) SAM_BW(bf16_t, 1); else SAM_BW(bf16_t, 0); }
    else return AB_EINVAL;
#undef SAM_BW
#undef SAM_BWD_ARGS
    AB_LAUNCH_CHECK();
    return 0;
}

extern "C" int ab_softargmax3d_bwd(const void* logits, int dtype, int B, int C, int D, int DP, int H, int W, const float* uvd,
                                   const float* conf, const float* stat, const float* g_uvd, const float* g_conf,
                                   void* dlogits, void* stream) {
    return sam_bwd_impl(logits, dtype, B, C, D, DP, H, W, uvd, conf, stat, g_uvd, g_conf, dlogits, 0, stream);
}
extern "C" int ab_softargmax3d_bwd_norm(const void* logits, int dtype, int B, int C, int D, int DP, int H, int W, int norm_type, const float* uvd,
                                        const float* conf, const float* stat, const float* g_uvd, const float* g_conf,
                                        void* dlogits, void* stream) {
    return sam_bwd_impl(logits, dtype, B, C, D, DP, H, W, uvd, conf, stat, g_uvd, g_conf, dlogits, norm_type, stream);
}

// the split-plane launches: 4 channels per lane and group where C*DP allows (3 groups up to 768 channels, else 4), otherwise pairs
#define SAM_BWX_ARGS logits, C, D, DP, H, W, uvd, conf, stat, g_uvd, g_conf, nullptr, (bf16_t*)dl_hi, (bf16_t*)dl_lo, colpart
#define SAM_BWX_N(PIXV, NN) \
    do { \
        static const int quad = getenv("AB_SAM_BWD_QUAD") ? atoi(getenv("AB_SAM_BWD_QUAD")) : 1; \
        if (quad && (C * DP) % 4 == 0 && C * DP <= 768) sam_bwd<float, true, PIXV, NN, 4, 3><<<grid, 256, 0, st>>>(SAM_BWX_ARGS); \
        else if (quad && (C * DP) % 4 == 0) sam_bwd<float, true, PIXV, NN, 4, 4><<<grid, 256, 0, st>>>(SAM_BWX_ARGS); \
        else sam_bwd<float, true, PIXV, NN><<<grid, 256, 0, st>>>(SAM_BWX_ARGS); \
    } while (0)
#define SAM_BWX(PIXV) do { if (norm) SAM_BWX_N(PIXV, 1); else SAM_BWX_N(PIXV, 0); } while (0)

// fp32 logits in, dlogits out as split-bf16 planes (C * DP even)
static int sam_bwd_x3_impl(const float* logits, int B, int C, int D, int DP, int H, int W, const float* uvd,
                           const float* conf, const float* stat, const float* g_uvd, const float* g_conf,
                           void* dl_hi, void* dl_lo, int norm, void* stream) {
    if (!logits || !uvd || !conf || !stat || !g_uvd || !dl_hi || !dl_lo) return AB_EINVAL;
    if (B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || DP < D || C * DP > SAM_MAXCH || ((C * DP) & 1) || norm < 0 || norm > 1) return AB_ESHAPE;
    if (norm && g_conf) return AB_EINVAL;
    dim3 grid((H * W + SAM_BWD_PIX - 1) / SAM_BWD_PIX, B);
    hipStream_t st = as_stream(stream);
    float* colpart = nullptr;
    SAM_BWX(SAM_BWD_PIX);
    AB_LAUNCH_CHECK();
    return 0;
}
extern "C" int ab_softargmax3d_bwd_x3(const float* logits, int B, int C, int D, int DP, int H, int W, const float* uvd,
                                      const float* conf, const float* stat, const float* g_uvd, const float* g_conf,
                                      void* dl_hi, void* dl_lo, void* stream) {
    return sam_bwd_x3_impl(logits, B, C, D, DP, H, W, uvd, conf, stat, g_uvd, g_conf, dl_hi, dl_lo, 0, stream);
}

// ... and the column sums of dlogits over all B*H*W rows (the bias gradient of the layer that produced the logits) -> dbias [C*DP]
// fp32.  colpart: scratch of ab_softargmax3d_bwd_x3_bias_rows(B, H, W) x C*DP floats.
#define SAM_BWD_PIX_BIAS 64
extern "C" int ab_softargmax3d_bwd_x3_bias_rows(int B, int H, int W) { return B * ((H * W + SAM_BWD_PIX_BIAS - 1) / SAM_BWD_PIX_BIAS); }
static int sam_bwd_x3_bias_impl(const float* logits, int B, int C, int D, int DP, int H, int W, const float* uvd,
                                const float* conf, const float* stat, const float* g_uvd, const float* g_conf,
                                void* dl_hi, void* dl_lo, float* colpart, float* dbias, int norm, void* stream) {
    if (!logits || !uvd || !conf || !stat || !g_uvd || !dl_hi || !dl_lo || !colpart || !dbias) return AB_EINVAL;
    if (B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || DP < D || C * DP > SAM_MAXCH || ((C * DP) & 1) || norm < 0 || norm > 1) return AB_ESHAPE;
    if (norm && g_conf) return AB_EINVAL;
    dim3 grid((H * W + SAM_BWD_PIX_BIAS - 1) / SAM_BWD_PIX_BIAS, B);
    hipStream_t st = as_stream(stream);
    SAM_BWX(SAM_BWD_PIX_BIAS);
    AB_LAUNCH_CHECK();
    sam_bias_finalize<<<(C * DP + 7) / 8, 256, 0, st>>>(colpart, (int)(grid.x * grid.y), C * DP, dbias);
    AB_LAUNCH_CHECK();
    return 0;
}
extern "C" int ab_softargmax3d_bwd_x3_bias(const float* logits, int B, int C, int D, int DP, int H, int W, const float* uvd,
                                           const float* conf, const float* stat, const float* g_uvd, const float* g_conf,
                                           void* dl_hi, void* dl_lo, float* colpart, float* dbias, void* stream) {
    return sam_bwd_x3_bias_impl(logits, B, C, D, DP, H, W, uvd, conf, stat, g_uvd, g_conf, dl_hi, dl_lo, colpart, dbias, 0, stream);
}
// NORM_TYPE-aware form of the two split-plane backwards (colpart / dbias NULL: no bias gradient)
extern "C" int ab_softargmax3d_bwd_x3_norm(const float* logits, int B, int C, int D, int DP, int H, int W, int norm_type, const float* uvd,
                                           const float* conf, const float* stat, const float* g_uvd, void* dl_hi, void* dl_lo,
                                           float* colpart, float* dbias, void* stream) {
    if (colpart && dbias)
        return sam_bwd_x3_bias_impl(logits, B, C, D, DP, H, W, uvd, conf, stat, g_uvd, nullptr, dl_hi, dl_lo, colpart, dbias, norm_type, stream);
    return sam_bwd_x3_impl(logits, B, C, D, DP, H, W, uvd, conf, stat, g_uvd, nullptr, dl_hi, dl_lo, norm_type, stream);
}

extern "C" int ab_abi_version(void) { return 2; }
