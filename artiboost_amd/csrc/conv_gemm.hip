// Implicit-GEMM convolution on MFMA (gfx950), NHWC activations, K-contiguous ("OHWI") weights.
//
// One kernel covers: conv forward (any k/stride/pad), conv data-gradient (stride 1 and 2, by output parity class),
// ConvTranspose forward (== data-gradient of the mirrored conv) and 1x1 / linear layers.  A launch computes
//     Out[pix(m), j] = sum_{t < ntaps} sum_{c < Ca} A[apix(m, t), c] * Bw[j, koff[t] + c]   (+bias[j]) (+addend)
// with m = (n, p, q) enumerating an output sub-grid.  M = N*P*Q rows, Cn columns, K = ntaps*Ca.
//
// Tiling: BM x BN output tile per 256-thread workgroup (4 waves as 2x2), 64 bytes of K per step (32 bf16 / 16 f32),
// register-staged double-buffered LDS with 80-byte row pitch (conflict-free ds_read_b128 for the MFMA fragments),
// v_mfma_f32_32x32x16_bf16 (bf16) or v_mfma_f32_32x32x2_f32 (exact f32 path used by the parity tests).
// Epilogue: optional bias, optional addend (residual gradient), optional per-column (sum, sum^2) partials for the
// following training-mode BatchNorm (deterministic: one partial row per M-tile, reduced by bn_finalize).
#include <stdlib.h>
#include "conv_common.h"

// (template parameter KB; the 7x7 stem keeps 64-byte steps because one kernel row of the NHWC4 image is 32 elements)

template <typename T>
__device__ __forceinline__ void mma_chunk(f32x16& acc, const uint4& a, const uint4& b);
template <>
__device__ __forceinline__ void mma_chunk<bf16_t>(f32x16& acc, const uint4& a, const uint4& b) {
    bf16x8 av = __builtin_bit_cast(bf16x8, a), bv = __builtin_bit_cast(bf16x8, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_chunk<float>(f32x16& acc, const uint4& a, const uint4& b) {
    // lanes 0-31 hold k = 0..3 of the chunk pair, lanes 32-63 hold k = 4..7: four K=2 MFMAs cover all eight
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

template <typename T, int BM, int BN, int KB>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvGemmArgs g) {
    constexpr int BK = KB / (int)sizeof(T);
    constexpr int CG_PITCH = KB + 16;
    constexpr int CPR = KB / 16;                // 16-byte chunks per row
    constexpr int RPP = 256 / CPR;              // rows staged per pass of the 256 threads
    constexpr int TM = BM / 64, TN = BN / 64;   // 32x32 MFMA tiles per wave in each direction
    constexpr int RA = BM / RPP, RB = BN / RPP; // staging rows per thread
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * (KB + 16) + BM * 4 + 2 * 2 * BN * 4];
    constexpr int BUFSZ = (BM + BN) * CG_PITCH;
#define SA(buf) (smem + (buf) * BUFSZ)
#define SB(buf) (smem + (buf) * BUFSZ + BM * CG_PITCH)
    int* s_outpix = (int*)(smem + 2 * (BM + BN) * CG_PITCH);
    float* s_stat = (float*)(s_outpix + BM);    // [2 (wave_m)][BN][2]

    // XCD-friendly tile order: consecutive workgroups walk the N dimension first so they share the A tile in L2
    const int tiles_n = (g.Cn + BN - 1) / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int chunk = tid % CPR, srow = tid / CPR;
    const T* __restrict__ A = (const T*)g.A;
    const T* __restrict__ Bw = (const T*)g.Bw;
    const int PQ = g.P * g.Q;

    // per-thread staging rows
    int a_h[RA], a_w[RA]; long a_base[RA]; bool a_ok[RA];
#pragma unroll
    for (int s = 0; s < RA; ++s) {
        int m = m0 + srow + RPP * s;
        a_ok[s] = m < g.M;
        int mm = a_ok[s] ? m : 0;
        int n = mm / PQ, r = mm - n * PQ;
        int p = r / g.Q, q = r - p * g.Q;
        a_h[s] = p * g.a_sh; a_w[s] = q * g.a_sw;
        a_base[s] = (long)n * g.Ha * g.Wa;
        if (chunk == 0) {
            int op = (n * g.Ho + p * g.out_sh + g.out_oh) * g.Wo + q * g.out_sw + g.out_ow;
            s_outpix[srow + RPP * s] = a_ok[s] ? op : -1;
        }
    }
    long b_off[RB]; bool b_ok[RB];
#pragma unroll
    for (int s = 0; s < RB; ++s) {
        int j = n0 + srow + RPP * s;
        b_ok[s] = j < g.Cn;
        b_off[s] = (long)(b_ok[s] ? j : 0) * g.ktot;
    }
    const int nsteps = g.ntaps * g.cpt;
    constexpr int CE = 16 / sizeof(T);   // elements per 16-byte chunk

    uint4 ra[RA], rb[RB];
    auto gload = [&](int step) {
        int t = step / g.cpt, c0 = (step - t * g.cpt) * BK + chunk * CE;
        int dh = g.dh[t], dw = g.dw[t], ko = g.koff[t];
#pragma unroll
        for (int s = 0; s < RA; ++s) {
            int hi = a_h[s] + dh, wi = a_w[s] + dw;
            bool ok = a_ok[s] && (unsigned)hi < (unsigned)g.Ha && (unsigned)wi < (unsigned)g.Wa;
            ra[s] = make_uint4(0, 0, 0, 0);
            if (ok) ra[s] = *(const uint4*)(A + ((a_base[s] + (long)hi * g.Wa + wi) * g.Ca + c0));
        }
#pragma unroll
        for (int s = 0; s < RB; ++s) {
            rb[s] = make_uint4(0, 0, 0, 0);
            if (b_ok[s]) rb[s] = *(const uint4*)(Bw + (b_off[s] + ko + c0));
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int s = 0; s < RA; ++s) *(uint4*)(SA(buf) + (srow + RPP * s) * CG_PITCH + chunk * 16) = ra[s];
#pragma unroll
        for (int s = 0; s < RB; ++s) *(uint4*)(SB(buf) + (srow + RPP * s) * CG_PITCH + chunk * 16) = rb[s];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gload(0);
    lstore(0);
    __syncthreads();
    const int frow = lane & 31, fhalf = lane >> 5;
    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        if (step + 1 < nsteps) gload(step + 1);
#pragma unroll
        for (int kk = 0; kk < CPR / 2; ++kk) {
            uint4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *(const uint4*)(SA(cur) + (wave_m * TM * 32 + i * 32 + frow) * CG_PITCH + (kk * 2 + fhalf) * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *(const uint4*)(SB(cur) + (wave_n * TN * 32 + j * 32 + frow) * CG_PITCH + (kk * 2 + fhalf) * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mma_chunk<T>(acc[i][j], fa[i], fb[j]);
        }
        if (step + 1 < nsteps) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue
    T* __restrict__ Out = (T*)g.Out;
    const T* __restrict__ Add = (const T*)g.addend;
    float csum[TN], csq[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { csum[j] = 0.f; csq[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wave_n * TN * 32 + j * 32 + (lane & 31);
        const bool cok = col < g.Cn;
        const float bj = (g.bias && cok) ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = wave_m * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                int op = s_outpix[row];
                float v = acc[i][j][r] + bj;
                if (op >= 0 && cok) {
                    long o = (long)op * g.Cn + col;
                    if (Add) v += ld_f32(Add + o);
                    if (g.relu) v = fmaxf(v, 0.f);
                    st_f32(Out + o, v);
                    csum[j] += v; csq[j] += v * v;
                }
            }
        }
    }
    if (g.stats) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = csum[j] + __shfl_xor(csum[j], 32, 64);
            float q = csq[j] + __shfl_xor(csq[j], 32, 64);
            if (lane < 32) {
                int cl = wave_n * TN * 32 + j * 32 + lane;
                s_stat[(wave_m * BN + cl) * 2] = s;
                s_stat[(wave_m * BN + cl) * 2 + 1] = q;
            }
        }
        __syncthreads();
        for (int c = tid; c < BN; c += 256) {
            int col = n0 + c;
            if (col < g.Cn) {
                float s = s_stat[c * 2] + s_stat[(BN + c) * 2];
                float q = s_stat[c * 2 + 1] + s_stat[(BN + c) * 2 + 1];
                g.stats[((long)tile_m * g.Cn + col) * 2] = s;
                g.stats[((long)tile_m * g.Cn + col) * 2 + 1] = q;
            }
        }
    }
}

template <typename T, int KB>
static int launch_conv_gemm(const ConvGemmArgs& g, int bm, int bn, hipStream_t st) {
    int tiles = ((g.M + bm - 1) / bm) * ((g.Cn + bn - 1) / bn);
    if (bm == 128 && bn == 128) conv_gemm_kernel<T, 128, 128, KB><<<tiles, 256, 0, st>>>(g);
    else if (bm == 128 && bn == 64) conv_gemm_kernel<T, 128, 64, KB><<<tiles, 256, 0, st>>>(g);
    else if (bm == 64 && bn == 128) conv_gemm_kernel<T, 64, 128, KB><<<tiles, 256, 0, st>>>(g);
    else conv_gemm_kernel<T, 64, 64, KB><<<tiles, 256, 0, st>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

static void pick_tile(int M, int Cn, int* bm, int* bn) {
    // fill >= ~2 waves of the 256 CUs when the problem allows; smaller tiles for the small late layers
    *bn = (Cn >= 128) ? 128 : 64;
    *bm = 128;
    long tiles = (long)((M + 127) / 128) * ((Cn + *bn - 1) / *bn);
    if (tiles < 512) { *bm = 64; tiles = (long)((M + 63) / 64) * ((Cn + *bn - 1) / *bn); }
    if (tiles < 512 && *bn == 128) { *bn = 64; }
}

int conv_gemm2_mtiles(int M, int Cn, int nsteps);
int conv_gemm2_stem_mtiles(int M);
int conv_gemm2_stem_run(ConvGemmArgs& g, hipStream_t st);
int stem_halo_tiles(int N, int H, int W);
int stem_halo_run(const void* xpad, const void* w, void* y, int N, int H, int W, int Cout, float* stats, hipStream_t st);
static bool use_stem_halo(int dtype, int Cout) {
    const char* e = getenv("AB_STEM_HALO");
    return !(e && atoi(e) == 0) && dtype == AB_DT_BF16 && Cout == 64 && !getenv("AB_STEM_V1") && !getenv("AB_CONV_V1");
}
static bool use_stem2(int dtype, int Cout) { return dtype == AB_DT_BF16 && Cout == 64 && !getenv("AB_STEM_V1") && !getenv("AB_CONV_V1"); }
int conv3x3_tiles(int N, int H, int W, int C, int Cn);
int conv3x3_run(const void* x, const void* wt, void* out, int N, int H, int W, int C, int Cn, int flip,
                const void* addend, float* stats, hipStream_t st, const void* bn_y = nullptr, const void* bn_out = nullptr,
                const float* bnp = nullptr, float* bn_part = nullptr);
static bool use_v2(int dtype, bool stem, int Ca) { return dtype == AB_DT_BF16 && !stem && Ca % 64 == 0 && !getenv("AB_CONV_V1"); }

static bool use_c3(int dtype, int kh, int kw, int stride, int pad) {
    return dtype == AB_DT_BF16 && kh == 3 && kw == 3 && stride == 1 && pad == 1 && !getenv("AB_CONV_V1");
}

// number of BN-partial rows ab_conv2d_fwd / ab_conv2d_stem_fwd (stem != 0) write into `stats` for this problem
extern "C" int ab_conv2d_stat_rows(int dtype, int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride,
                                   int pad, int stem) {
    const int Ho = stem ? H / 2 : (H + 2 * pad - kh) / stride + 1, Wo = stem ? W / 2 : (W + 2 * pad - kw) / stride + 1;
    const int M = N * Ho * Wo;
    if (!stem && use_c3(dtype, kh, kw, stride, pad)) { int t = conv3x3_tiles(N, H, W, Cin, Cout); if (t) return t; }
    if (stem && use_stem_halo(dtype, Cout)) { int t = stem_halo_tiles(N, H, W); if (t) return t; }
    if (stem && use_stem2(dtype, Cout)) return conv_gemm2_stem_mtiles(M);
    if (use_v2(dtype, stem != 0, stem ? 0 : Cin)) return conv_gemm2_mtiles(M, Cout, kh * kw * (Cin / 64));
    int bm, bn; pick_tile(M, Cout, &bm, &bn);
    return (M + bm - 1) / bm;
}

static int run(ConvGemmArgs& g, int dtype, hipStream_t st, bool stem = false) {
    if (use_v2(dtype, stem, g.Ca)) {
        int rc = conv_gemm2_run(g, st);
        if (rc != AB_ESHAPE) return rc;
    }
    int bm, bn; pick_tile(g.M, g.Cn, &bm, &bn);
    if (dtype == AB_DT_BF16) return stem ? launch_conv_gemm<bf16_t, 64>(g, bm, bn, st) : launch_conv_gemm<bf16_t, 128>(g, bm, bn, st);
    if (dtype == AB_DT_F32) return launch_conv_gemm<float, 64>(g, bm, bn, st);
    return AB_EINVAL;
}

static int bk_of(int dtype) { return dtype == AB_DT_BF16 ? 64 : 16; }

extern "C" int ab_conv2d_fwd(const void* x, const void* w, void* y, int dtype, int N, int H, int W, int Cin, int Cout,
                             int kh, int kw, int stride, int pad, const float* bias, float* stats, int relu,
                             void* stream) {
    if (!x || !w || !y) return AB_EINVAL;
    if (kh * kw > CG_MAXTAPS || Cin % bk_of(dtype)) return AB_ESHAPE;
    if (use_c3(dtype, kh, kw, stride, pad) && !bias && !relu) {
        int rc = conv3x3_run(x, w, y, N, H, W, Cin, Cout, 0, nullptr, stats, as_stream(stream));
        if (rc != AB_ESHAPE) return rc;
    }
    ConvGemmArgs g = {};
    g.A = x; g.Bw = w; g.Out = y; g.bias = bias; g.stats = stats; g.relu = relu;
    g.N = N; g.Ha = H; g.Wa = W; g.Ca = Cin;
    g.Ho = (H + 2 * pad - kh) / stride + 1; g.Wo = (W + 2 * pad - kw) / stride + 1; g.Cn = Cout;
    g.P = g.Ho; g.Q = g.Wo; g.out_sh = g.out_sw = 1; g.a_sh = g.a_sw = stride;
    g.ntaps = kh * kw; g.cpt = Cin / bk_of(dtype); g.ktot = kh * kw * Cin; g.M = N * g.P * g.Q;
    for (int i = 0; i < kh; ++i) for (int j = 0; j < kw; ++j) {
        g.dh[i * kw + j] = (int8_t)(i - pad); g.dw[i * kw + j] = (int8_t)(j - pad); g.koff[i * kw + j] = (i * kw + j) * Cin;
    }
    return run(g, dtype, as_stream(stream));
}

// Stem: 7x7 / stride 2 / pad 3 on a zero-bordered NHWC4 image [N, H+6, W+8, 4] (3 px border, +2 px right slack):
// each kh contributes one contiguous run of 8 px * 4 ch = 32 elements, so no bounds checks and K = 7*32.
// w: [Cout][7][8][4] (px 7 and ch 3 zero).  Output NHWC [N, H/2, W/2, Cout].
extern "C" int ab_conv2d_stem_fwd(const void* xpad, const void* w, void* y, int dtype, int N, int H, int W, int Cout,
                                  float* stats, void* stream) {
    if (!xpad || !w || !y) return AB_EINVAL;
    if ((H & 1) || (W & 1)) return AB_ESHAPE;
    ConvGemmArgs g = {};
    g.A = xpad; g.Bw = w; g.Out = y; g.stats = stats;
    g.N = N; g.Ha = H + 6; g.Wa = W + 8; g.Ca = 4;
    g.Ho = H / 2; g.Wo = W / 2; g.Cn = Cout; g.P = g.Ho; g.Q = g.Wo; g.out_sh = g.out_sw = 1; g.a_sh = g.a_sw = 2;
    const int bk = dtype == AB_DT_BF16 ? 32 : 16, per = 32 / bk;     // K-steps per kh row (1 for bf16, 2 for f32)
    g.ntaps = 7 * per; g.cpt = 1; g.ktot = 7 * 32; g.M = N * g.P * g.Q;
    if (g.ntaps > CG_MAXTAPS) return AB_ESHAPE;
    if (use_stem_halo(dtype, Cout)) {
        int rc = stem_halo_run(xpad, w, y, N, H, W, Cout, stats, as_stream(stream));
        if (rc != AB_ESHAPE) return rc;
    }
    if (use_stem2(dtype, Cout)) {
        ConvGemmArgs g2 = g;
        int rc = conv_gemm2_stem_run(g2, as_stream(stream));
        if (rc != AB_ESHAPE) return rc;
    }
    for (int i = 0; i < 7; ++i) for (int s = 0; s < per; ++s) {
        g.dh[i * per + s] = (int8_t)i; g.dw[i * per + s] = (int8_t)(s * (bk / 4)); g.koff[i * per + s] = i * 32 + s * bk;
    }
    return run(g, dtype, as_stream(stream), true);
}

// rows of BN partials ab_conv2d_dgrad writes into `stats` for this problem; 0 = not supported (stats must be NULL)
extern "C" int ab_conv2d_dgrad_stat_rows(int dtype, int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad) {
    if (stride != 2 || kh > 4 || kw > 4 || (H & 1) || (W & 1) || Cout % 64 || !use_v2(dtype, false, Cout)) return 0;
    int maxt = 0;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) {
        int nt = 0;
        for (int i = 0; i < kh; ++i) for (int j = 0; j < kw; ++j) if (!((a + pad - i) % 2) && !((b + pad - j) % 2)) ++nt;
        if (nt > maxt) maxt = nt;
    }
    return 4 * conv_gemm2_mtiles(N * (H / 2) * (W / 2), Cin, maxt * (Cout / 64));
}

// Data gradient of conv2d(x, w, stride, pad): dx[N,H,W,Cin] from dy[N,Ho,Wo,Cout] and wt = [Cin][kh][kw][Cout].
// Also == ConvTranspose2d forward (x:=dy).  stride 1 or 2; for stride 2 one launch per output parity class.
// addend (same shape as dx, may be NULL) is added in the epilogue (residual-branch gradient).
extern "C" int ab_conv2d_dgrad(const void* dy, const void* wt, void* dx, int dtype, int N, int H, int W, int Cin,
                               int Cout, int kh, int kw, int stride, int pad, const void* addend, float* stats,
                               void* stream) {
    if (!dy || !wt || !dx) return AB_EINVAL;
    if (Cout % bk_of(dtype) || (stride != 1 && stride != 2)) return AB_ESHAPE;
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if (stride == 2 && ((H & 1) || (W & 1))) return AB_ESHAPE;
    // BN partials of the output (sum, sum of squares per channel; a transposed convolution's forward IS this data gradient):
    // only on the one-grid stride-2 path and without an addend -- ab_conv2d_dgrad_stat_rows() says how many rows, 0 = use
    // ab_col_stats on the result instead
    if (stats && !ab_conv2d_dgrad_stat_rows(dtype, N, H, W, Cin, Cout, kh, kw, stride, pad)) return AB_ESHAPE;
    if (stats && addend) return AB_ESHAPE;
    if (use_c3(dtype, kh, kw, stride, pad)) {
        int rc = conv3x3_run(dy, wt, dx, N, H, W, Cout, Cin, 1, addend, nullptr, as_stream(stream));
        if (rc != AB_ESHAPE) return rc;
    }
    if (stride == 2 && kh <= 4 && kw <= 4 && use_v2(dtype, false, Cout)) {
        // all four output-parity classes in one grid (each alone fills half the chip or less)
        ConvGemmArgs g = {};
        g.A = dy; g.Bw = wt; g.Out = dx; g.addend = addend; g.stats = stats;
        g.N = N; g.Ha = Ho; g.Wa = Wo; g.Ca = Cout;
        g.Ho = H; g.Wo = W; g.Cn = Cin;
        g.P = H / 2; g.Q = W / 2; g.out_sh = g.out_sw = 2;
        g.a_sh = g.a_sw = 1; g.cpt = Cout / 64; g.ktot = kh * kw * Cout; g.M = N * g.P * g.Q;
        g.nclass = 4;
        int maxt = 0;
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) {
            const int c = a * 2 + b;
            int nt = 0;
            for (int i = 0; i < kh; ++i) {
                if ((a + pad - i) % 2) continue;
                for (int j = 0; j < kw; ++j) {
                    if ((b + pad - j) % 2) continue;
                    g.dh[c * 4 + nt] = (int8_t)((a + pad - i) / 2); g.dw[c * 4 + nt] = (int8_t)((b + pad - j) / 2);
                    g.koff[c * 4 + nt] = (i * kw + j) * Cout; ++nt;
                }
            }
            g.cls_ntaps[c] = nt; g.cls_oh[c] = a; g.cls_ow[c] = b;
            if (nt > maxt) maxt = nt;
        }
        g.ntaps = maxt;
        int rc = conv_gemm2_run(g, as_stream(stream));
        if (rc != AB_ESHAPE) return rc;
    }
    if (stats) return AB_ESHAPE;
    for (int a = 0; a < stride; ++a) for (int b = 0; b < stride; ++b) {
        ConvGemmArgs g = {};
        g.A = dy; g.Bw = wt; g.Out = dx; g.addend = addend; g.stats = stats;
        g.N = N; g.Ha = Ho; g.Wa = Wo; g.Ca = Cout;
        g.Ho = H; g.Wo = W; g.Cn = Cin;
        g.P = H / stride; g.Q = W / stride; g.out_sh = g.out_sw = stride; g.out_oh = a; g.out_ow = b;
        g.a_sh = g.a_sw = 1; g.cpt = Cout / bk_of(dtype); g.ktot = kh * kw * Cout; g.M = N * g.P * g.Q;
        int nt = 0;
        for (int i = 0; i < kh; ++i) {
            if ((a + pad - i) % stride) continue;       // (hi + pad - i) must be divisible by stride
            for (int j = 0; j < kw; ++j) {
                if ((b + pad - j) % stride) continue;
                if (nt >= CG_MAXTAPS) return AB_ESHAPE;
                // floor division also for negative numerators (they are multiples of stride here)
                g.dh[nt] = (int8_t)((a + pad - i) / stride); g.dw[nt] = (int8_t)((b + pad - j) / stride);
                g.koff[nt] = (i * kw + j) * Cout; ++nt;
            }
        }
        g.ntaps = nt;
        if (nt == 0) {   // this parity class receives no gradient (1x1 stride-2): write zeros / the addend
            g.ntaps = 1; g.cpt = 1; g.dh[0] = 127; g.dw[0] = 127; g.koff[0] = 0;   // always out of bounds -> zero rows
        }
        int rc = run(g, dtype, as_stream(stream));
        if (rc) return rc;
    }
    return 0;
}

// ---- data gradient with the BatchNorm-backward reduction of the layer below fused into the epilogue (3x3/s1 halo kernel)
extern "C" int ab_conv2d_dgrad_bnstats_rows(int dtype, int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride,
                                            int pad) {
    if (!use_c3(dtype, kh, kw, stride, pad) || getenv("AB_BNFUSE_OFF")) return 0;
    // Opt-in (AB_BNFUSE_MIN=<tiles>): measured on MI355X the fused epilogue LOSES to the separate reduction pass
    // (9343 vs 9440 samples/s with every 3x3 data gradient fused, 8990 vs 9090 with only the 2048-tile layer1 launches):
    // the standalone pass streams at 5+ TB/s, while the extra reads at the end of a conv workgroup are exposed latency.
    static const int min_tiles = getenv("AB_BNFUSE_MIN") ? atoi(getenv("AB_BNFUSE_MIN")) : 0x7fffffff;
    int t = conv3x3_tiles(N, H, W, Cout, Cin);
    return t >= min_tiles ? t : 0;
}

extern "C" int ab_conv2d_dgrad_bnstats(const void* dy, const void* wt, void* dx, int dtype, int N, int H, int W, int Cin,
                                       int Cout, int kh, int kw, int stride, int pad, const void* addend, const void* bn_y,
                                       const void* bn_out, const float* bnp, float* bn_part, void* stream) {
    if (!dy || !wt || !dx || !bn_y || !bnp || !bn_part) return AB_EINVAL;
    if (!ab_conv2d_dgrad_bnstats_rows(dtype, N, H, W, Cin, Cout, kh, kw, stride, pad)) return AB_ESHAPE;
    return conv3x3_run(dy, wt, dx, N, H, W, Cout, Cin, 1, addend, nullptr, as_stream(stream), bn_y, bn_out, bnp, bn_part);
}
