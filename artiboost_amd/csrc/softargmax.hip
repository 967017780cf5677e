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

template <typename T>
__global__ __launch_bounds__(SAM_THREADS) void sam_stage1(const T* __restrict__ logits, int C, int D, int DP, int H, int W,
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
                float x = ld_f32(row + ch);
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

__global__ void sam_stage2(const float* __restrict__ part, int C, int ntile, float* __restrict__ uvd,
                           float* __restrict__ conf, float* __restrict__ stat) {
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
        uvd[bc * 3 + 0] = r.su * inv;
        uvd[bc * 3 + 1] = r.sv * inv;
        uvd[bc * 3 + 2] = r.sd * inv;
        conf[bc] = 1.f / r.s;  // max p = exp(max - max) / sum
        stat[bc * 2 + 0] = r.m;
        stat[bc * 2 + 1] = r.s;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sam_bwd(const T* __restrict__ logits, int C, int D, int DP, int H, int W,
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
        float x = ld_f32(row + ch);
        float m = stat[bc * 2], s = stat[bc * 2 + 1];
        float pr = __expf(x - m) / s;
        float u = uvd[bc * 3] * z, v = uvd[bc * 3 + 1] * z, dd = uvd[bc * 3 + 2] * z;  // sum p*coord
        float g = g_uvd[bc * 3] * (cu - u) + g_uvd[bc * 3 + 1] * (cv - v) + g_uvd[bc * 3 + 2] * ((float)d / D - dd);
        float out = pr * g / z;
        if (g_conf) {
            float cf = conf[bc];
            out += g_conf[bc] * cf * ((x == m ? 1.f : 0.f) - pr);
        }
        st_f32(drow + ch, out);
    }
}

extern "C" int ab_softargmax3d_ntiles(int H, int W) { return (H * W + SAM_TILE_PIX - 1) / SAM_TILE_PIX; }

extern "C" int ab_softargmax3d_fwd(const void* logits, int dtype, int B, int C, int D, int DP, int H, int W, float* part,
                                   float* uvd, float* conf, float* stat, void* stream) {
    if (!logits || !part || !uvd || !conf || !stat) return AB_EINVAL;
    if (B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || DP < D || C * DP > SAM_MAXCH) return AB_ESHAPE;
    int ntile = ab_softargmax3d_ntiles(H, W);
    dim3 grid(ntile, B);
    if (dtype == AB_DT_F32)
        sam_stage1<float><<<grid, SAM_THREADS, 0, as_stream(stream)>>>((const float*)logits, C, D, DP, H, W, ntile, part);
    else if (dtype == AB_DT_BF16)
        sam_stage1<bf16_t><<<grid, SAM_THREADS, 0, as_stream(stream)>>>((const bf16_t*)logits, C, D, DP, H, W, ntile, part);
    else return AB_EINVAL;
    AB_LAUNCH_CHECK();
    sam_stage2<<<B * C, 64, 0, as_stream(stream)>>>(part, C, ntile, uvd, conf, stat);
    AB_LAUNCH_CHECK();
    return 0;
}

extern "C" int ab_softargmax3d_bwd(const void* logits, int dtype, int B, int C, int D, int DP, int H, int W, const float* uvd,
                                   const float* conf, const float* stat, const float* g_uvd, const float* g_conf,
                                   void* dlogits, void* stream) {
    if (!logits || !uvd || !conf || !stat || !g_uvd || !dlogits) return AB_EINVAL;
    if (B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || DP < D || C * DP > SAM_MAXCH) return AB_ESHAPE;
    dim3 grid((H * W + 3) / 4, B);
    if (dtype == AB_DT_F32)
        sam_bwd<float><<<grid, 256, 0, as_stream(stream)>>>((const float*)logits, C, D, DP, H, W, uvd, conf, stat, g_uvd,
                                                           g_conf, (float*)dlogits);
    else if (dtype == AB_DT_BF16)
        sam_bwd<bf16_t><<<grid, 256, 0, as_stream(stream)>>>((const bf16_t*)logits, C, D, DP, H, W, uvd, conf, stat, g_uvd,
                                                            g_conf, (bf16_t*)dlogits);
    else return AB_EINVAL;
    AB_LAUNCH_CHECK();
    return 0;
}

extern "C" int ab_abi_version(void) { return 1; }
