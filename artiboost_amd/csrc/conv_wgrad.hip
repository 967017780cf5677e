// Weight gradient of conv2d on MFMA (gfx950):  dW[co][t][ci] = sum_m dy[m][co] * x[apix(m,t)][ci]
//
// GEMM view: I = Cout, J = (tap, Cin), reduction R = N*P*Q output pixels.  Both operands are reduction-major in
// memory (NHWC: a pixel's channels are contiguous, consecutive pixels are a channel-pitch apart), while an MFMA lane
// needs 8 consecutive reduction indices for ONE channel.  The tiles are therefore staged in LDS as [pixel][channel]
// exactly as they sit in HBM (coalesced 16-byte loads) and the MFMA fragments are fetched with the gfx950 hardware
// transpose read ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group; semantics pinned by tools/probe_tr16).
// The f32 parity path uses v_mfma_f32_32x32x2_f32 whose fragments are single dwords, read directly.
//
// Each workgroup owns a BI x BJ tile of dW and one slice of the pixel range (split-R); slices are written to
// separate slabs and summed in a fixed order by wgrad_reduce (deterministic, no atomics).
//
// This register-staged kernel is the f32 / odd-shape path (and the reference the full-size check compares against);
// bf16 problems are routed to wgrad3x3.hip (3x3 / stride 1: all nine taps per workgroup) or wgrad_gemm2.hip (everything
// else, LDS-DMA operands) by ab_conv2d_wgrad / ab_conv2d_stem_wgrad below.  wgrad_reduce serves all three.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short short4_t __attribute__((ext_vector_type(4)));

struct WgradArgs {
    const void* X; const void* DY; float* slabs;
    int N, Ha, Wa, Ca;        // x tensor (Ca = channel pitch)
    int P, Q, Cout;           // dy tensor [N,P,Q,Cout]
    int a_sh, a_sw;
    int ntaps, seglen;        // seglen = J-elements contributed by one tap (Cin, or 32 for the padded stem rows)
    int jtot;                 // ntaps * seglen  (row length of dW)
    int M, rows_per_slice, nslices;
    int8_t dh[16], dw[16];
};

template <typename T> struct WG;
template <> struct WG<bf16_t> { static constexpr int BR = 32; };
template <> struct WG<float> { static constexpr int BR = 16; };

template <typename T, int BI, int BJ>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs g) {
    constexpr int BR = WG<T>::BR;
    constexpr int ES = sizeof(T);
    constexpr int CE = 16 / ES;                   // elements per 16-byte chunk
    constexpr int PI = BI * ES + 64, PJ = BJ * ES + 64;   // LDS row pitches (bytes): +64 keeps 4 consecutive rows on distinct banks
    constexpr int CI = BI / CE, CJ = BJ / CE;     // chunks per row
    constexpr int NI = (BR * CI + 255) / 256, NJ = (BR * CJ + 255) / 256;   // chunks per thread
    constexpr int TI = BI / 64, TJ = BJ / 64;     // 32x32 MFMA tiles per wave
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BR * (PI + PJ)];
    constexpr int BUFSZ = BR * (PI + PJ);
#define SI(buf) (smem + (buf) * BUFSZ)
#define SJ(buf) (smem + (buf) * BUFSZ + BR * PI)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_i = wave >> 1, wave_j = wave & 1;
    const int tiles_j = g.jtot / BJ;
    const int tile_i = blockIdx.x / tiles_j, tile_j = blockIdx.x - tile_i * tiles_j;
    const int slice = blockIdx.y;
    const int i0 = tile_i * BI, j0 = tile_j * BJ;
    const T* __restrict__ X = (const T*)g.X;
    const T* __restrict__ DY = (const T*)g.DY;
    const int PQ = g.P * g.Q;
    const int r_begin = slice * g.rows_per_slice;
    const int r_end = min(g.M, r_begin + g.rows_per_slice);

    // static per-thread chunk assignment
    int ir[NI], ic[NI], jr[NJ], jc[NJ], jtap[NJ], jel[NJ];
#pragma unroll
    for (int s = 0; s < NI; ++s) { int id = tid + 256 * s; ir[s] = id / CI; ic[s] = id - ir[s] * CI; }
#pragma unroll
    for (int s = 0; s < NJ; ++s) {
        int id = tid + 256 * s; jr[s] = id / CJ; jc[s] = id - jr[s] * CJ;
        int jj = j0 + jc[s] * CE;                 // global J index of the chunk
        jtap[s] = jj / g.seglen; jel[s] = jj - jtap[s] * g.seglen;
    }
    uint4 ri[NI], rj[NJ];
    auto gload = [&](int rbase) {
#pragma unroll
        for (int s = 0; s < NI; ++s) {
            ri[s] = make_uint4(0, 0, 0, 0);
            int m = rbase + ir[s];
            if (ir[s] < BR && m < r_end) ri[s] = *(const uint4*)(DY + ((long)m * g.Cout + i0 + ic[s] * CE));
        }
#pragma unroll
        for (int s = 0; s < NJ; ++s) {
            rj[s] = make_uint4(0, 0, 0, 0);
            int m = rbase + jr[s];
            if (jr[s] < BR && m < r_end) {
                int n = m / PQ, r = m - n * PQ;
                int p = r / g.Q, q = r - p * g.Q;
                int hi = p * g.a_sh + g.dh[jtap[s]], wi = q * g.a_sw + g.dw[jtap[s]];
                if ((unsigned)hi < (unsigned)g.Ha && (unsigned)wi < (unsigned)g.Wa)
                    rj[s] = *(const uint4*)(X + (((long)n * g.Ha + hi) * g.Wa + wi) * g.Ca + jel[s]);
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int s = 0; s < NI; ++s) if (ir[s] < BR) *(uint4*)(SI(buf) + ir[s] * PI + ic[s] * 16) = ri[s];
#pragma unroll
        for (int s = 0; s < NJ; ++s) if (jr[s] < BR) *(uint4*)(SJ(buf) + jr[s] * PJ + jc[s] * 16) = rj[s];
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nsteps = (r_end - r_begin + BR - 1) / BR;
    if (nsteps > 0) { gload(r_begin); lstore(0); }
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        if (step + 1 < nsteps) gload(r_begin + (step + 1) * BR);
        if constexpr (sizeof(T) == 2) {
            // 16 reduction rows per MFMA; lane group grp = lane>>4: columns 16*(grp&1).., rows 8*(grp>>1) + {0..3 | 4..7}
            const int grp = lane >> 4, l16 = lane & 15;
            const int rsub = (grp >> 1) * 8 + (l16 >> 2);
            const int csub = (grp & 1) * 16 + (l16 & 3) * 4;
#pragma unroll
            for (int kk = 0; kk < BR / 16; ++kk) {
                uint4 fa[TI], fb[TJ];
#pragma unroll
                for (int a = 0; a < TI; ++a) {
                    const unsigned char* p = SI(cur) + (kk * 16 + rsub) * PI + (wave_i * TI * 32 + a * 32 + csub) * 2;
                    short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)p);
                    short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(p + 4 * PI));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fa[a] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
#pragma unroll
                for (int b = 0; b < TJ; ++b) {
                    const unsigned char* p = SJ(cur) + (kk * 16 + rsub) * PJ + (wave_j * TJ * 32 + b * 32 + csub) * 2;
                    short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)p);
                    short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(p + 4 * PJ));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fb[b] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
#pragma unroll
                for (int a = 0; a < TI; ++a)
#pragma unroll
                    for (int b = 0; b < TJ; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[a]),
                                                                           __builtin_bit_cast(bf16x8, fb[b]), acc[a][b], 0, 0, 0);
            }
        } else {
            const int l32 = lane & 31, half = lane >> 5;
#pragma unroll
            for (int kk = 0; kk < BR / 2; ++kk) {
                float fa[TI], fb[TJ];
#pragma unroll
                for (int a = 0; a < TI; ++a)
                    fa[a] = *(const float*)(SI(cur) + (kk * 2 + half) * PI + (wave_i * TI * 32 + a * 32 + l32) * 4);
#pragma unroll
                for (int b = 0; b < TJ; ++b)
                    fb[b] = *(const float*)(SJ(cur) + (kk * 2 + half) * PJ + (wave_j * TJ * 32 + b * 32 + l32) * 4);
#pragma unroll
                for (int a = 0; a < TI; ++a)
#pragma unroll
                    for (int b = 0; b < TJ; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
            }
        }
        if (step + 1 < nsteps) lstore(cur ^ 1);
        __syncthreads();
    }
    // write the slab tile: rows = cout, cols = J
    float* out = g.slabs + (long)slice * g.Cout * g.jtot;
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = i0 + wave_i * TI * 32 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                int col = j0 + wave_j * TJ * 32 + b * 32 + (lane & 31);
                out[(long)row * g.jtot + col] = acc[a][b][r];
            }
}

// dW[e] (+)= sum_s slabs[s][e]; optionally drops padded taps: dst row layout [Cout][dst_j], src [Cout][src_j].
// (256/KY) element quads x KY slice lanes per workgroup (KY = 4 for few slabs, 16 otherwise): lane ky sums slices ky, ky+KY,
// ... in order, the KY partials are then combined in fixed order through LDS (deterministic; every load is a 16-byte
// vector, contiguous along the row).
template <int KY>
__global__ __launch_bounds__(256) void wgrad_reduce(const float* __restrict__ slabs, int nslices, long slab_elems, int src_j,
                                                    int dst_j, float* __restrict__ dst, int accumulate, int stem_mask) {
    constexpr int QX = 256 / KY;
    __shared__ float4 part[KY][QX + 1];
    const int qx = threadIdx.x % QX, ky = threadIdx.x / QX;
    const long quad = (long)blockIdx.x * QX + qx;
    const long total = (slab_elems / src_j) * dst_j;
    const long e = quad * 4;
    const bool live = e < total;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    long row = 0; int col = 0;
    if (live) {
        row = e / dst_j; col = (int)(e - row * dst_j);
        const float* src = slabs + row * src_j + col;
        for (int k = ky; k < nslices; k += KY) {
            float4 v = *(const float4*)(src + (long)k * slab_elems);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    part[ky][qx] = s;
    __syncthreads();
    if (ky == 0 && live) {
        float4 t = part[0][qx];
#pragma unroll
        for (int k = 1; k < KY; ++k) { float4 v = part[k][qx]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        if (stem_mask) {                                      // padding taps of the [7][8][4] stem layout
            if ((col & 31) >= 28) t = make_float4(0.f, 0.f, 0.f, 0.f);
            t.w = 0.f;
        }
        float4* d = (float4*)(dst + e);
        if (accumulate) { float4 o = *d; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
        *d = t;
    }
}

// ---- deferred slab reductions: ab_conv2d_wgrad_deferred / ab_conv2d_stem_wgrad_deferred record the reduction they would
// launch, ab_wgrad_reduce_batch runs up to AB_WGRAD_BATCH_MAX recorded reductions in ONE launch (descriptor table in the
// kernel arguments).  39 reductions per step of 4-10 us each cost more as graph nodes than as bandwidth.
static thread_local ab_wgrad_reduce_desc* g_reduce_sink = nullptr;

struct ReduceBatch {
    ab_wgrad_reduce_desc d[AB_WGRAD_BATCH_MAX];
    int block0[AB_WGRAD_BATCH_MAX + 1];          // first workgroup of descriptor i
    int n;
};
static inline int reduce_ky(int ns) { return ns <= 8 ? 4 : 16; }
static inline long reduce_blocks(const ab_wgrad_reduce_desc& d) {
    const long total = (d.slab_elems / d.src_j) * d.dst_j;
    const int qx = 256 / reduce_ky(d.nslices);
    return (total / 4 + qx - 1) / qx;
}

// same arithmetic, in the same order, as wgrad_reduce<KY> of the descriptor's slice count: bit-identical results
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const ReduceBatch b) {
    __shared__ float4 part[16 * 17 + 4];
    int di = 0;
    while (di + 1 < b.n && (int)blockIdx.x >= b.block0[di + 1]) ++di;
    const ab_wgrad_reduce_desc& d = b.d[di];
    const int KY = d.nslices <= 8 ? 4 : 16, QX = 256 / KY;
    const int qx = threadIdx.x % QX, ky = threadIdx.x / QX;
    const long quad = (long)((int)blockIdx.x - b.block0[di]) * QX + qx;
    const long total = (d.slab_elems / d.src_j) * d.dst_j;
    const long e = quad * 4;
    const bool live = e < total;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int col = 0;
    if (live) {
        const long row = e / d.dst_j; col = (int)(e - row * d.dst_j);
        const float* src = d.slabs + row * d.src_j + col;
        for (int k = ky; k < d.nslices; k += KY) {
            float4 v = *(const float4*)(src + (long)k * d.slab_elems);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    part[ky * (QX + 1) + qx] = s;
    __syncthreads();
    if (ky == 0 && live) {
        float4 t = part[qx];
        for (int k = 1; k < KY; ++k) { float4 v = part[k * (QX + 1) + qx]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        if (d.stem_mask) {
            if ((col & 31) >= 28) t = make_float4(0.f, 0.f, 0.f, 0.f);
            t.w = 0.f;
        }
        float4* o = (float4*)(d.dst + e);
        if (d.accumulate) { float4 v = *o; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *o = t;
    }
}

extern "C" int ab_wgrad_reduce_batch(const ab_wgrad_reduce_desc* desc, int n, void* stream) {
    if (!desc || n < 0) return AB_EINVAL;
    for (int i0 = 0; i0 < n; i0 += AB_WGRAD_BATCH_MAX) {
        ReduceBatch b;
        long blocks = 0;
        b.n = 0;
        for (int i = i0; i < n && b.n < AB_WGRAD_BATCH_MAX; ++i) {
            const ab_wgrad_reduce_desc& d = desc[i];
            if (d.nslices <= 0) continue;
            if (!d.slabs || !d.dst || d.src_j <= 0 || d.dst_j <= 0 || d.dst_j % 4 || d.src_j % 4) return AB_EINVAL;
            b.d[b.n] = d; b.block0[b.n] = (int)blocks; ++b.n;
            blocks += reduce_blocks(d);
            if (blocks > 0x7fffffffL) return AB_ESHAPE;
        }
        if (!b.n) continue;
        b.block0[b.n] = (int)blocks;
        wgrad_reduce_batch_kernel<<<(unsigned)blocks, 256, 0, as_stream(stream)>>>(b);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

static int launch_reduce(const float* slabs, int ns, long slab_elems, int src_j, int dst_j, float* dst, int accumulate,
                         int stem_mask, hipStream_t st) {
    if (g_reduce_sink) {
        ab_wgrad_reduce_desc* d = g_reduce_sink;
        d->slabs = slabs; d->dst = dst; d->slab_elems = slab_elems; d->nslices = ns; d->src_j = src_j; d->dst_j = dst_j;
        d->accumulate = accumulate; d->stem_mask = stem_mask;
        return 0;
    }
    const long total = (slab_elems / src_j) * dst_j;
    if (ns <= 8) wgrad_reduce<4><<<(unsigned)((total / 4 + 63) / 64), 256, 0, st>>>(slabs, ns, slab_elems, src_j, dst_j, dst, accumulate, stem_mask);
    else wgrad_reduce<16><<<(unsigned)((total / 4 + 15) / 16), 256, 0, st>>>(slabs, ns, slab_elems, src_j, dst_j, dst, accumulate, stem_mask);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// the fixed-order slab reduction for the weight-gradient kernels of other translation units (conv_x3.hip)
int wgrad_launch_reduce(const float* slabs, int ns, long slab_elems, int src_j, int dst_j, float* dst, int accumulate,
                        int stem_mask, hipStream_t st) {
    return launch_reduce(slabs, ns, slab_elems, src_j, dst_j, dst, accumulate, stem_mask, st);
}

static void pick_wgrad(int M, int Cout, int jtot, int* bi, int* bj, int* nslices, int* rows) {
    *bi = (Cout % 128 == 0) ? 128 : 64;
    *bj = (jtot % 128 == 0) ? 128 : 64;
    long tiles = (long)(Cout / *bi) * (jtot / *bj);
    int want = (int)((640 + tiles - 1) / tiles);            // ~2.5 workgroups per CU
    int maxs = (M + 1023) / 1024;                            // at least 1024 pixels per slice
    int ns = want < 1 ? 1 : want; if (ns > maxs) ns = maxs; if (ns < 1) ns = 1; if (ns > 512) ns = 512;
    int r = (M + ns - 1) / ns; r = (r + 31) / 32 * 32;
    *nslices = (M + r - 1) / r; *rows = r;
}

int wgrad3x3_slices(int N, int H, int W, int Cin, int Cout);
int wgrad3x3_run(const void* x, const void* dy, float* slabs, int N, int H, int W, int Cin, int Cout, hipStream_t st);
int wgrad_gemm2_slices(int M, int Cout, int Cin, int ntaps);
int wgrad_gemm2_max_slices(int M, int Cout, int jtot);
int wgrad_gemm2_stem_slices(int N, int H, int W, int Cout);
int wgrad_gemm2_stem_run(const void* xpad, const void* dy, float* slabs, int N, int H, int W, int Cout, hipStream_t st);
int wgrad_gemm2_run(const void* x, const void* dy, float* slabs, int N, int H, int W, int Cin, int Cout, int kh, int kw,
                    int stride, int pad, hipStream_t st);

extern "C" long ab_conv2d_wgrad_workspace(int M, int Cout, int jtot) {
    int bi, bj, ns, rows; pick_wgrad(M, Cout, jtot, &bi, &bj, &ns, &rows);
    long bytes = (long)ns * Cout * jtot * 4;
    if (jtot % 576 == 0 && Cout % 64 == 0) {                 // the all-taps 3x3 kernel uses up to 512/tiles + 1 slabs
        long tiles = (long)(Cout / 64) * (jtot / 576);
        long alt = (512 + tiles) * 64 * 576 * 4;
        if (alt > bytes) bytes = alt;
    }
    long alt2 = (long)wgrad_gemm2_max_slices(M, Cout, jtot) * Cout * jtot * 4;
    if (alt2 > bytes) bytes = alt2;
    return bytes;
}

template <typename T>
static int launch_wgrad(const WgradArgs& g, int bi, int bj, hipStream_t st) {
    dim3 grid((g.Cout / bi) * (g.jtot / bj), g.nslices);
    if (bi == 128 && bj == 128) wgrad_kernel<T, 128, 128><<<grid, 256, 0, st>>>(g);
    else if (bi == 128 && bj == 64) wgrad_kernel<T, 128, 64><<<grid, 256, 0, st>>>(g);
    else if (bi == 64 && bj == 128) wgrad_kernel<T, 64, 128><<<grid, 256, 0, st>>>(g);
    else wgrad_kernel<T, 64, 64><<<grid, 256, 0, st>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

static int run_wgrad(WgradArgs& g, int dtype, float* dw, int dst_j, int accumulate, hipStream_t st, int stem_mask = 0) {
    int bi, bj, ns, rows; pick_wgrad(g.M, g.Cout, g.jtot, &bi, &bj, &ns, &rows);
    if (g.Cout % bi || g.jtot % bj) return AB_ESHAPE;
    g.nslices = ns; g.rows_per_slice = rows;
    int rc = dtype == AB_DT_BF16 ? launch_wgrad<bf16_t>(g, bi, bj, st) : dtype == AB_DT_F32 ? launch_wgrad<float>(g, bi, bj, st) : AB_EINVAL;
    if (rc) return rc;
    long slab = (long)g.Cout * g.jtot;
    return launch_reduce(g.slabs, ns, slab, g.jtot, dst_j, dw, accumulate, stem_mask, st);
}

// dw: float [Cout][kh][kw][Cin] (OHWI).  workspace: ab_conv2d_wgrad_workspace(N*Ho*Wo, Cout, kh*kw*Cin) bytes.
extern "C" int ab_conv2d_wgrad(const void* x, const void* dy, float* dw, int dtype, int N, int H, int W, int Cin,
                               int Cout, int kh, int kw, int stride, int pad, void* workspace, int accumulate,
                               void* stream) {
    if (!x || !dy || !dw || !workspace) return AB_EINVAL;
    if (kh * kw > 16 || Cin % 64 || Cout % 64) return AB_ESHAPE;
    if (dtype == AB_DT_BF16 && kh == 3 && kw == 3 && stride == 1 && pad == 1) {
        int ns = wgrad3x3_slices(N, H, W, Cin, Cout);
        if (ns > 0) {
            int rc = wgrad3x3_run(x, dy, (float*)workspace, N, H, W, Cin, Cout, as_stream(stream));
            if (rc) return rc;
            long slab = (long)Cout * 9 * Cin;
            return launch_reduce((float*)workspace, ns, slab, 9 * Cin, 9 * Cin, dw, accumulate, 0, as_stream(stream));
        }
    }
    if (dtype == AB_DT_BF16) {
        const int M = N * ((H + 2 * pad - kh) / stride + 1) * ((W + 2 * pad - kw) / stride + 1);
        int ns = wgrad_gemm2_slices(M, Cout, Cin, kh * kw);
        if (ns > 0) {
            int rc = wgrad_gemm2_run(x, dy, (float*)workspace, N, H, W, Cin, Cout, kh, kw, stride, pad, as_stream(stream));
            if (rc) return rc;
            long slab = (long)Cout * kh * kw * Cin;
            return launch_reduce((float*)workspace, ns, slab, kh * kw * Cin, kh * kw * Cin, dw, accumulate, 0, as_stream(stream));
        }
    }
    WgradArgs g = {};
    g.X = x; g.DY = dy; g.slabs = (float*)workspace;
    g.N = N; g.Ha = H; g.Wa = W; g.Ca = Cin;
    g.P = (H + 2 * pad - kh) / stride + 1; g.Q = (W + 2 * pad - kw) / stride + 1; g.Cout = Cout;
    g.a_sh = g.a_sw = stride; g.ntaps = kh * kw; g.seglen = Cin; g.jtot = kh * kw * Cin; g.M = N * g.P * g.Q;
    for (int i = 0; i < kh; ++i) for (int j = 0; j < kw; ++j) { g.dh[i * kw + j] = (int8_t)(i - pad); g.dw[i * kw + j] = (int8_t)(j - pad); }
    return run_wgrad(g, dtype, dw, g.jtot, accumulate, as_stream(stream));
}

// Stem (see ab_conv2d_stem_fwd): x is the zero-bordered NHWC4 image; dw: float [Cout][7][8][4].
extern "C" int ab_conv2d_stem_wgrad(const void* xpad, const void* dy, float* dw, int dtype, int N, int H, int W,
                                    int Cout, void* workspace, void* stream) {
    if (!xpad || !dy || !dw || !workspace) return AB_EINVAL;
    if ((H & 1) || (W & 1) || Cout % 64) return AB_ESHAPE;
    if (dtype == AB_DT_BF16) {
        int ns = wgrad_gemm2_stem_slices(N, H, W, Cout);
        if (ns > 0) {
            int rc = wgrad_gemm2_stem_run(xpad, dy, (float*)workspace, N, H, W, Cout, as_stream(stream));
            if (rc) return rc;
            return launch_reduce((float*)workspace, ns, (long)Cout * 256, 256, 7 * 32, dw, 0, 1, as_stream(stream));
        }
    }
    WgradArgs g = {};
    g.X = xpad; g.DY = dy; g.slabs = (float*)workspace;
    g.N = N; g.Ha = H + 6; g.Wa = W + 8; g.Ca = 4;
    g.P = H / 2; g.Q = W / 2; g.Cout = Cout; g.a_sh = g.a_sw = 2;
    g.ntaps = 8; g.seglen = 32; g.jtot = 256; g.M = N * g.P * g.Q;      // tap 7 is padding (in-bounds row, dropped below)
    for (int i = 0; i < 8; ++i) { g.dh[i] = (int8_t)i; g.dw[i] = 0; }
    return run_wgrad(g, dtype, dw, 7 * 32, 0, as_stream(stream), 1);
}

extern "C" long ab_conv2d_stem_wgrad_workspace(int N, int H, int W, int Cout) {
    long a = ab_conv2d_wgrad_workspace(N * (H / 2) * (W / 2), Cout, 256);
    int n = N;                                               // the split-bf16 stem splits batches beyond 2^21 output pixels in halves
    while ((long)n * (H / 2) * (W / 2) >= (1L << 21) && n > 1) n -= n / 2;
    long b = (long)(wgrad_gemm2_stem_slices(n, H, W, Cout) + 1) * Cout * 256 * 4;
    return a > b ? a : b;
}

// As ab_conv2d_wgrad / ab_conv2d_stem_wgrad, but the final slab reduction is recorded in *pending instead of launched
// (pending->nslices == 0 if the path had nothing to reduce).  `workspace` must stay untouched until ab_wgrad_reduce_batch
// has consumed the descriptor -- one workspace per deferred call.
extern "C" int ab_conv2d_wgrad_deferred(const void* x, const void* dy, float* dw, int dtype, int N, int H, int W, int Cin,
                                        int Cout, int kh, int kw, int stride, int pad, void* workspace, int accumulate,
                                        ab_wgrad_reduce_desc* pending, void* stream) {
    if (!pending) return AB_EINVAL;
    *pending = ab_wgrad_reduce_desc{};
    g_reduce_sink = pending;
    int rc = ab_conv2d_wgrad(x, dy, dw, dtype, N, H, W, Cin, Cout, kh, kw, stride, pad, workspace, accumulate, stream);
    g_reduce_sink = nullptr;
    return rc;
}
// ... and of the split-bf16 weight gradients (conv_x3.hip)
extern "C" int ab_conv2d_wgrad_x3(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, float* dw, int N,
                                  int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, void* workspace,
                                  int accumulate, void* stream);
extern "C" int ab_conv2d_stem_wgrad_x3(const void* xpad_hi, const void* xpad_lo, const void* dy_hi, const void* dy_lo, float* dw,
                                       int N, int H, int W, int Cout, void* workspace, void* stream);
extern "C" int ab_conv2d_wgrad_x3_deferred(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, float* dw, int N,
                                           int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, void* workspace,
                                           int accumulate, ab_wgrad_reduce_desc* pending, void* stream) {
    if (!pending) return AB_EINVAL;
    *pending = ab_wgrad_reduce_desc{};
    g_reduce_sink = pending;
    int rc = ab_conv2d_wgrad_x3(x_hi, x_lo, dy_hi, dy_lo, dw, N, H, W, Cin, Cout, kh, kw, stride, pad, workspace, accumulate, stream);
    g_reduce_sink = nullptr;
    return rc;
}
extern "C" int ab_conv2d_stem_wgrad_x3_deferred(const void* xpad_hi, const void* xpad_lo, const void* dy_hi, const void* dy_lo,
                                                float* dw, int N, int H, int W, int Cout, void* workspace,
                                                ab_wgrad_reduce_desc* pending, void* stream) {
    if (!pending) return AB_EINVAL;
    *pending = ab_wgrad_reduce_desc{};
    g_reduce_sink = pending;
    int rc = ab_conv2d_stem_wgrad_x3(xpad_hi, xpad_lo, dy_hi, dy_lo, dw, N, H, W, Cout, workspace, stream);
    g_reduce_sink = nullptr;
    return rc;
}
extern "C" int ab_conv2d_stem_wgrad_deferred(const void* xpad, const void* dy, float* dw, int dtype, int N, int H, int W,
                                             int Cout, void* workspace, ab_wgrad_reduce_desc* pending, void* stream) {
    if (!pending) return AB_EINVAL;
    *pending = ab_wgrad_reduce_desc{};
    g_reduce_sink = pending;
    int rc = ab_conv2d_stem_wgrad(xpad, dy, dw, dtype, N, H, W, Cout, workspace, stream);
    g_reduce_sink = nullptr;
    return rc;
}
