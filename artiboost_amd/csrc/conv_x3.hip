// C ABI of the split-bf16 ("bf16x3") convolution path: fp32-grade convolutions on the bf16 matrix cores.
//
// The reference trains in fp32 (train/train_artiboost.py:39-41,91-96: no autocast anywhere); the f32-input MFMA of
// gfx950 runs at 1/16 of the bf16 rate (157 vs 2500 TFLOP/s).  Here every fp32 operand v is carried as two bf16 planes
//     hi = bf16(v),  lo = bf16(v - hi)          (v = hi + lo up to 2^-17 |v|)
// and a product a*b is evaluated as  a.hi*b.hi + a.hi*b.lo + a.lo*b.hi  on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation (the dropped lo*lo term is 2^-18 relative): three MFMA passes, i.e. a 833 TFLOP/s roof, with outputs,
// residual addends, BatchNorm partials and weight-gradient slabs in fp32.  The kernels are the X3 instantiations of
// conv3x3.hip / conv_gemm2.hip (forward, data gradient) and wgrad3x3.hip / wgrad_gemm2.hip (weight gradient).
#include "conv_common.h"

int conv3x3_x3_tiles(int N, int H, int W, int C, int Cn);
int conv3x3_x3_tiles_bnr(int N, int H, int W, int C, int Cn);
struct C3EvalBn { void* out_hi; void* out_lo; float* out_f32; const void* res_hi; const void* res_lo; int relu; };
int conv3x3_x3_run(const void* x_hi, const void* x_lo, const void* wt_hi, const void* wt_lo, float* out, int N, int H, int W,
                   int C, int Cn, int flip, const float* addend, float* stats, hipStream_t st, const float* bn_y = nullptr,
                   const void* bn_out_hi = nullptr, const float* bnp = nullptr, float* bn_part = nullptr, const C3EvalBn* ev = nullptr);
int conv_gemm2_x3_mtiles(int M, int Cn, int nsteps, int nclass);
int conv_gemm2_x3_run(ConvGemmArgs& g, hipStream_t st);
int wgrad3x3_x3_slices(int N, int H, int W, int Cin, int Cout);
int wgrad3x3_x3_run(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, float* slabs, int N, int H, int W,
                    int Cin, int Cout, hipStream_t st);
int wgrad3x3_x3_group_slices(int G, int N, int H, int W, int Cin, int Cout);
int wgrad3x3_x3_group_run(int G, const void* const* x_hi, const void* const* x_lo, const void* const* dy_hi, const void* const* dy_lo,
                          float* const* slabs, int N, int H, int W, int Cin, int Cout, hipStream_t st);
int wgrad_gemm2_x3_slices(int M, int Cout, int Cin, int ntaps);
int wgrad_gemm2_x3_run(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, float* slabs, int N, int H, int W,
                       int Cin, int Cout, int kh, int kw, int stride, int pad, hipStream_t st);
int conv_gemm2_x3_stem_mtiles(int M);
int conv_gemm2_x3_stem_run(ConvGemmArgs& g, hipStream_t st);
int wgrad_gemm2_stem_slices(int N, int H, int W, int Cout);
int wgrad_gemm2_x3_stem_run(const void* xpad_hi, const void* xpad_lo, const void* dy_hi, const void* dy_lo, float* slabs, int N,
                            int H, int W, int Cout, hipStream_t st);
int stem_halo_x3_tiles(int N, int H, int W);                   // stem_halo.hip
int stem_halo_x3_run(const void* xpad_hi, const void* xpad_lo, const void* w_hi, const void* w_lo, float* y, int N, int H, int W,
                     int Cout, float* stats, hipStream_t st);
int conv2x2_tfwd_rows(int N, int H, int W, int Cn, int K);       // conv2x2.hip
int conv2x2_tfwd_run(const void* x_hi, const void* x_lo, const void* wt_hi, const void* wt_lo, float* out, int N, int H, int W, int Cn, int K,
                     float* stats, hipStream_t st, const float* ep_scale = nullptr, const float* ep_shift = nullptr, int ep_relu = 0,
                     void* out_hi = nullptr, void* out_lo = nullptr);
int conv2x2_s2fwd_ok(int N, int H, int W, int C, int Cn);
int convp_s2fwd_rows(int N, int H, int W, int C, int Cn);           // convp.hip
int convp_s2fwd_run(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, float* out, int N, int H, int W, int C, int Cn,
                    float* stats, hipStream_t st, const float* ep_scale = nullptr, const float* ep_shift = nullptr, int ep_relu = 0,
                    void* out_hi = nullptr, void* out_lo = nullptr);
int convp_s2dgrad_ok(int N, int H, int W, int Cn, int K);
int convp_s2dgrad_bn_rows(int N, int H, int W, int Cn, int K);
int convp_s2dgrad_run(const void* dy_hi, const void* dy_lo, const void* wt_hi, const void* wt_lo, const void* dy2_hi, const void* dy2_lo,
                      const void* wt2_hi, const void* wt2_lo, float* dx, int N, int H, int W, int Cn, int K, hipStream_t st,
                      const float* bn_y = nullptr, const void* bn_out = nullptr, const float* bnp = nullptr, float* bn_part = nullptr);
int conv2x2_s2fwd_run(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, float* out, int N, int H, int W, int C, int Cn,
                      float* stats, hipStream_t st);
struct GemmRwSam { float* part; int C, D, H, W; };              // gemm_rw.hip
int gemm_rw_ok(long M, int N, int K);
int gemm_rw_run(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias, float* out, long M, int N,
                int K, const GemmRwSam* sam, hipStream_t st);
int wgrad_launch_reduce(const float* slabs, int ns, long slab_elems, int src_j, int dst_j, float* dst, int accumulate,
                           int stem_mask, hipStream_t st);      // conv_wgrad.hip

// ---------------------------------------------------------------- fp32 -> (hi, lo) bf16 planes
__global__ __launch_bounds__(256) void split_f32_kernel(const float* __restrict__ src, long nvec, bf16_t* __restrict__ hi,
                                                        bf16_t* __restrict__ lo) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const float4 a = *(const float4*)(src + i * 8), b = *(const float4*)(src + i * 8 + 4);
        const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h[k] = pack_bf16x2(f[2 * k], f[2 * k + 1]);
            const float r0 = f[2 * k] - __uint_as_float(h[k] << 16), r1 = f[2 * k + 1] - __uint_as_float(h[k] & 0xffff0000u);
            l[k] = pack_bf16x2(r0, r1);
        }
        *(uint4*)(hi + i * 8) = make_uint4(h[0], h[1], h[2], h[3]);
        *(uint4*)(lo + i * 8) = make_uint4(l[0], l[1], l[2], l[3]);
    }
}

extern "C" int ab_split_f32(const float* src, long n, void* hi, void* lo, void* stream) {
    if (!src || !hi || !lo) return AB_EINVAL;
    if (n % 8) return AB_ESHAPE;
    const long nvec = n / 8;
    long b = (nvec + 255) / 256; if (b > 8192) b = 8192; if (b < 1) b = 1;
    split_f32_kernel<<<(int)b, 256, 0, as_stream(stream)>>>(src, nvec, (bf16_t*)hi, (bf16_t*)lo);
    AB_LAUNCH_CHECK(); return 0;
}

// ---------------------------------------------------------------- forward
static bool x3_is_c3(int kh, int kw, int stride, int pad) { return kh == 3 && kw == 3 && stride == 1 && pad == 1; }

extern "C" int ab_conv2d_x3_stat_rows(int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad) {
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if (x3_is_c3(kh, kw, stride, pad)) { int t = conv3x3_x3_tiles(N, H, W, Cin, Cout); if (t) return t; }
    if (kh == 4 && kw == 4 && stride == 2 && pad == 1) { const int r = conv2x2_s2fwd_ok(N, H, W, Cin, Cout); if (r) return r; }      // conv2x2.hip: one row per tile
    if (kh == 3 && kw == 3 && stride == 2 && pad == 1) { const int r = convp_s2fwd_rows(N, H, W, Cin, Cout); if (r) return r; }      // convp.hip: one row per tile
    return conv_gemm2_x3_mtiles(N * Ho * Wo, Cout, kh * kw * (Cin / 32), 0);
}

static int fwd_x3_impl(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, float* y, int N, int H,
                       int W, int Cin, int Cout, int kh, int kw, int stride, int pad, const float* bias, float* stats,
                       int relu, void* stream, const float* ep_scale = nullptr, void* out_hi = nullptr, void* out_lo = nullptr) {
    if (!x_hi || !x_lo || !w_hi || !w_lo || (!y && !out_hi)) return AB_EINVAL;
    if (kh * kw > CG_MAXTAPS || Cin % 32) return AB_ESHAPE;
    if (x3_is_c3(kh, kw, stride, pad) && !bias && !relu && !ep_scale) {
        int rc = conv3x3_x3_run(x_hi, x_lo, w_hi, w_lo, y, N, H, W, Cin, Cout, 0, nullptr, stats, as_stream(stream));
        if (rc != AB_ESHAPE) return rc;
    }
    if (kh == 1 && kw == 1 && stride == 1 && pad == 0 && !stats && !relu && !ep_scale && y && gemm_rw_ok((long)N * H * W, Cout, Cin)) {
        // plain GEMM with the weights resident in registers (gemm_rw.hip): the final layer of the head
        int rc = gemm_rw_run(x_hi, x_lo, w_hi, w_lo, bias, y, (long)N * H * W, Cout, Cin, nullptr, as_stream(stream));
        if (rc != AB_ESHAPE) return rc;
    }
    if (kh == 4 && kw == 4 && stride == 2 && pad == 1 && !bias && !relu && !ep_scale && y && conv2x2_s2fwd_ok(N, H, W, Cin, Cout)) {
        // the data gradient of ConvTranspose2d(4x4, s2, p1): four 2x2-tap convolutions over the parity sub-grids (conv2x2.hip)
        int rc = conv2x2_s2fwd_run(x_hi, x_lo, w_hi, w_lo, y, N, H, W, Cin, Cout, stats, as_stream(stream));
        if (rc != AB_ESHAPE) return rc;
    }
    if (kh == 3 && kw == 3 && stride == 2 && pad == 1 && (ep_scale ? (bias && !stats) : (!bias && !relu && y)) && convp_s2fwd_rows(N, H, W, Cin, Cout)) {
        // a stage's first convolution: the nine taps over four parity sub-grid patches (convp.hip); training forward with BatchNorm partials, or
        // eval mode with the BatchNorm that follows as the epilogue's affine (bit-identical to its own plain launch + ab_bn_apply_x3)
        int rc = convp_s2fwd_run(x_hi, x_lo, w_hi, w_lo, y, N, H, W, Cin, Cout, stats, as_stream(stream), ep_scale, bias, relu, out_hi, out_lo);
        if (rc != AB_ESHAPE) return rc;
    }
    // The generic kernel writes one partial row per M tile.  `stats` was sized by ab_conv2d_x3_stat_rows, which describes the kernel a launch
    // WITHOUT bias / relu takes; a launch that asks for statistics AND a bias or ReLU on a shape the specialised kernels own (3x3/s1, 3x3/s2,
    // 4x4/s2) lands here with another row count -- refuse instead of writing past the caller's buffer (no model path issues such a launch)
    if (stats && conv_gemm2_x3_mtiles(N * ((H + 2 * pad - kh) / stride + 1) * ((W + 2 * pad - kw) / stride + 1), Cout, kh * kw * (Cin / 32), 0) !=
                     ab_conv2d_x3_stat_rows(N, H, W, Cin, Cout, kh, kw, stride, pad))
        return AB_EINVAL;
    ConvGemmArgs g = {};
    g.A = x_hi; g.A_lo = x_lo; g.Bw = w_hi; g.Bw_lo = w_lo; g.Out = y; g.bias = bias; g.stats = stats; g.relu = relu;
    g.ep_scale = ep_scale; g.out_hi = out_hi; g.out_lo = out_lo;
    g.N = N; g.Ha = H; g.Wa = W; g.Ca = Cin;
    g.Ho = (H + 2 * pad - kh) / stride + 1; g.Wo = (W + 2 * pad - kw) / stride + 1; g.Cn = Cout;
    g.P = g.Ho; g.Q = g.Wo; g.out_sh = g.out_sw = 1; g.a_sh = g.a_sw = stride;
    g.ntaps = kh * kw; g.cpt = Cin / 32; g.ktot = kh * kw * Cin; g.M = N * g.P * g.Q;
    for (int i = 0; i < kh; ++i) for (int j = 0; j < kw; ++j) {
        g.dh[i * kw + j] = (int8_t)(i - pad); g.dw[i * kw + j] = (int8_t)(j - pad); g.koff[i * kw + j] = (i * kw + j) * Cin;
    }
    return conv_gemm2_x3_run(g, as_stream(stream));
}

extern "C" int ab_conv2d_fwd_x3(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, float* y, int N, int H,
                                int W, int Cin, int Cout, int kh, int kw, int stride, int pad, const float* bias, float* stats,
                                int relu, void* stream) {
    return fwd_x3_impl(x_hi, x_lo, w_hi, w_lo, y, N, H, W, Cin, Cout, kh, kw, stride, pad, bias, stats, relu, stream);
}

// Final layer of IntegralDeconvHead + the first stage of its soft-argmax in one launch (simplebaseline.py:95-101,173-175 then 183-189 with
// 43-71): logits = conv1x1(x, w) + bias as fp32 [B, H, W, C * 32] AND the per-(image, 64-pixel tile, class) softmax statistics
// `part` [B, H*W/64, C, 8] that ab_softargmax3d_stage2 merges -- the logits are not read back by a statistics pass.  Channel = c * 32 + d,
// d < D valid (DEPTH_PITCH 32).  AB_ESHAPE when the register-resident GEMM does not take the shape (Cin % 64, Cin > 256, H * W % 64):
// the caller then runs ab_conv2d_fwd_x3 + ab_softargmax3d_fwd.
extern "C" int ab_conv1x1_sam_fwd_x3_ok(int B, int H, int W, int Cin, int C, int D) {
    return D > 0 && D <= 32 && (H * W) % 64 == 0 && gemm_rw_ok((long)B * H * W, C * 32, Cin);
}
extern "C" int ab_conv1x1_sam_fwd_x3(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias, float* logits,
                                     int B, int H, int W, int Cin, int C, int D, float* part, void* stream) {
    if (!x_hi || !x_lo || !w_hi || !w_lo || !logits || !part) return AB_EINVAL;
    if (!ab_conv1x1_sam_fwd_x3_ok(B, H, W, Cin, C, D)) return AB_ESHAPE;
    GemmRwSam sam = {part, C, D, H, W};
    return gemm_rw_run(x_hi, x_lo, w_hi, w_lo, bias, logits, (long)B * H * W, C * 32, Cin, &sam, as_stream(stream));
}

// Eval-mode forms of the GENERIC convolution and of the transposed convolution (the strided 3x3, the 1x1 downsample and the two
// ConvTranspose2d of the head: resnet.py:85-101,181-184, simplebaseline.py:161-172 under model.eval()): the BatchNorm that follows as a
// per-channel affine in the epilogue, out = relu?(conv * scale + shift), written as fp32 (out_f32) or as (hi, lo) planes -- one of the two.
extern "C" int ab_conv2d_fwd_x3_affine(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, int N, int H, int W, int Cin,
                                       int Cout, int kh, int kw, int stride, int pad, const float* scale, const float* shift, int relu,
                                       float* out_f32, void* out_hi, void* out_lo, void* stream) {
    if (!scale || !shift || (!out_f32 == !out_hi) || (out_hi && !out_lo)) return AB_EINVAL;
    if (Cout % 4) return AB_ESHAPE;
    return fwd_x3_impl(x_hi, x_lo, w_hi, w_lo, out_f32, N, H, W, Cin, Cout, kh, kw, stride, pad, shift, nullptr, relu, stream, scale, out_hi, out_lo);
}

// Eval-mode 3x3 / stride 1 / pad 1 convolution with the BatchNorm that follows it folded into the epilogue (resnet.py:85-101 in
// eval(): running statistics, so (scale, shift) = bnp[0..Cout) | bnp[Cout..2 Cout) are launch constants):
//   out = relu?(conv(x, w) * scale + shift + residual)   written as (hi, lo) planes [N,H,W,Cout] (+ fp32 when out_f32 != NULL)
// residual: (res_hi, res_lo) planes, or the fp32 tensor res_f32, or none.  AB_ESHAPE when the 3x3 kernel does not take the shape
// (the caller then runs ab_conv2d_fwd_x3 + ab_bn_apply_x3; results are bit-identical either way).
extern "C" int ab_conv2d_fwd_x3_evalbn_ok(int N, int H, int W, int Cin, int Cout) {
    return conv3x3_x3_tiles(N, H, W, Cin, Cout) > 0 && Cout % 8 == 0;
}

extern "C" int ab_conv2d_fwd_x3_evalbn(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, int N, int H, int W,
                                       int Cin, int Cout, const float* bnp, const void* res_hi, const void* res_lo,
                                       const float* res_f32, int relu, void* out_hi, void* out_lo, float* out_f32, void* stream) {
    if (!x_hi || !x_lo || !w_hi || !w_lo || !bnp || !out_hi || !out_lo) return AB_EINVAL;
    if (Cin % 32) return AB_ESHAPE;
    C3EvalBn ev = {out_hi, out_lo, out_f32, res_hi, res_lo, relu};
    return conv3x3_x3_run(x_hi, x_lo, w_hi, w_lo, nullptr, N, H, W, Cin, Cout, 0, res_f32, nullptr, as_stream(stream), nullptr, nullptr,
                          bnp, nullptr, &ev);
}

// ---------------------------------------------------------------- data gradient (== transposed-convolution forward)
static bool x3_is_tconv(int kh, int kw, int stride, int pad) { return kh == 4 && kw == 4 && stride == 2 && pad == 1; }

extern "C" int ab_conv2d_dgrad_x3_stat_rows(int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad) {
    if (stride != 2 || kh > 4 || kw > 4 || (H & 1) || (W & 1) || Cout % 32) return 0;
    if (x3_is_tconv(kh, kw, stride, pad)) { const int r = conv2x2_tfwd_rows(N, H, W, Cin, Cout); if (r) return r; }
    int maxt = 0;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) {
        int nt = 0;
        for (int i = 0; i < kh; ++i) for (int j = 0; j < kw; ++j) if (!((a + pad - i) % 2) && !((b + pad - j) % 2)) ++nt;
        if (nt > maxt) maxt = nt;
    }
    return 4 * conv_gemm2_x3_mtiles(N * (H / 2) * (W / 2), Cin, maxt * (Cout / 32), 4);
}

static int dgrad_x3_impl(const void* dy_hi, const void* dy_lo, const void* wt_hi, const void* wt_lo, float* dx, int N,
                         int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, const float* addend,
                         float* stats, void* stream, const void* dy2_hi = nullptr, const void* dy2_lo = nullptr,
                         const void* wt2_hi = nullptr, const void* wt2_lo = nullptr, const float* ep_scale = nullptr,
                         const float* ep_shift = nullptr, int ep_relu = 0, void* out_hi = nullptr, void* out_lo = nullptr) {
    if (!dy_hi || !dy_lo || !wt_hi || !wt_lo || (!dx && !out_hi)) return AB_EINVAL;
    if (Cout % 32 || (stride != 1 && stride != 2)) return AB_ESHAPE;
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if (stride == 2 && ((H & 1) || (W & 1) || kh > 4 || kw > 4)) return AB_ESHAPE;
    if (stats && (addend || !ab_conv2d_dgrad_x3_stat_rows(N, H, W, Cin, Cout, kh, kw, stride, pad))) return AB_ESHAPE;
    if (x3_is_c3(kh, kw, stride, pad) && !ep_scale) {
        int rc = conv3x3_x3_run(dy_hi, dy_lo, wt_hi, wt_lo, dx, N, H, W, Cout, Cin, 1, addend, nullptr, as_stream(stream));
        if (rc != AB_ESHAPE) return rc;
    }
    if (x3_is_tconv(kh, kw, stride, pad) && !addend && !dy2_hi && (dx || out_hi) && conv2x2_tfwd_rows(N, H, W, Cin, Cout)) {
        // ConvTranspose2d(4x4, s2, p1) forward: the four output-parity classes as 2x2-tap convolutions on a resident patch (conv2x2.hip)
        int rc = conv2x2_tfwd_run(dy_hi, dy_lo, wt_hi, wt_lo, dx, N, H, W, Cin, Cout, stats, as_stream(stream), ep_scale, ep_shift, ep_relu, out_hi, out_lo);
        if (rc != AB_ESHAPE) return rc;
    }
    if (kh == 3 && kw == 3 && stride == 2 && pad == 1 && !addend && !stats && !ep_scale && dx && convp_s2dgrad_ok(N, H, W, Cin, Cout)) {
        // data gradient of a stage's first convolution (+ its downsample branch): four output-parity classes over one dy patch (convp.hip)
        int rc = convp_s2dgrad_run(dy_hi, dy_lo, wt_hi, wt_lo, dy2_hi, dy2_lo, wt2_hi, wt2_lo, dx, N, H, W, Cin, Cout, as_stream(stream));
        if (rc != AB_ESHAPE) return rc;
    }
    ConvGemmArgs g = {};
    g.A = dy_hi; g.A_lo = dy_lo; g.Bw = wt_hi; g.Bw_lo = wt_lo; g.Out = dx; g.addend = addend; g.stats = stats;
    g.ep_scale = ep_scale; g.bias = ep_shift; g.relu = ep_relu; g.out_hi = out_hi; g.out_lo = out_lo;
    g.N = N; g.Ha = Ho; g.Wa = Wo; g.Ca = Cout;
    g.Ho = H; g.Wo = W; g.Cn = Cin;
    g.a_sh = g.a_sw = 1; g.cpt = Cout / 32; g.ktot = kh * kw * Cout;
    if (stride == 2) {      // the four output-parity classes in one grid (conv_gemm2.hip, nclass)
        g.P = H / 2; g.Q = W / 2; g.out_sh = g.out_sw = 2; g.M = N * g.P * g.Q;
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
            if (c == 0 && dy2_hi) {         // the 1x1/s2/pad-0 branch: input pixel (2p, 2q) <- dy2[p, q], one more tap of class (even, even)
                if (nt >= 4) return AB_ESHAPE;
                g.dh[nt] = 0; g.dw[nt] = 0; g.koff[nt] = 0;
                g.A2 = dy2_hi; g.A2_lo = dy2_lo; g.Bw2 = wt2_hi; g.Bw2_lo = wt2_lo; g.alt_tap1 = nt + 1; g.ktot2 = Cout;
                ++nt;
            }
            g.cls_ntaps[c] = nt; g.cls_oh[c] = a; g.cls_ow[c] = b;
            if (nt > maxt) maxt = nt;
        }
        g.ntaps = maxt;
        return conv_gemm2_x3_run(g, as_stream(stream));
    }
    if (dy2_hi) return AB_ESHAPE;
    if (kh * kw > CG_MAXTAPS) return AB_ESHAPE;
    g.P = H; g.Q = W; g.out_sh = g.out_sw = 1; g.M = N * H * W;
    int nt = 0;
    for (int i = 0; i < kh; ++i) for (int j = 0; j < kw; ++j) {
        g.dh[nt] = (int8_t)(pad - i); g.dw[nt] = (int8_t)(pad - j); g.koff[nt] = (i * kw + j) * Cout; ++nt;
    }
    g.ntaps = nt;
    return conv_gemm2_x3_run(g, as_stream(stream));
}

extern "C" int ab_conv2d_dgrad_x3(const void* dy_hi, const void* dy_lo, const void* wt_hi, const void* wt_lo, float* dx, int N,
                                  int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, const float* addend,
                                  float* stats, void* stream) {
    return dgrad_x3_impl(dy_hi, dy_lo, wt_hi, wt_lo, dx, N, H, W, Cin, Cout, kh, kw, stride, pad, addend, stats, stream);
}

// ConvTranspose2d forward (== data gradient of the mirrored convolution) in eval mode with the following BatchNorm (+ ReLU) folded in
extern "C" int ab_conv2d_dgrad_x3_affine(const void* dy_hi, const void* dy_lo, const void* wt_hi, const void* wt_lo, int N, int H, int W,
                                         int Cin, int Cout, int kh, int kw, int stride, int pad, const float* scale, const float* shift,
                                         int relu, float* out_f32, void* out_hi, void* out_lo, void* stream) {
    if (!scale || !shift || (!out_f32 == !out_hi) || (out_hi && !out_lo)) return AB_EINVAL;
    if (Cin % 4) return AB_ESHAPE;
    return dgrad_x3_impl(dy_hi, dy_lo, wt_hi, wt_lo, out_f32, N, H, W, Cin, Cout, kh, kw, stride, pad, nullptr, nullptr, stream, nullptr, nullptr,
                         nullptr, nullptr, scale, shift, relu, out_hi, out_lo);
}

// dx = dgrad(dy, wt; kh x kw / stride 2) + dgrad(dy2, wt2; 1x1 / stride 2 / pad 0) [+ addend]: the two branches that leave a
// down-sampling residual block's input (resnet.py:85-101 backwards: conv1 and downsample.0 read the same x) in ONE launch --
// the 1x1 branch is one more tap of the (even, even) parity class, reading its own gradient / weight planes.  dy2: planes
// [N,Ho,Wo,Cout] of the same geometry as dy, wt2 planes [Cin][1][1][Cout].
extern "C" int ab_conv2d_dgrad_x3_pair(const void* dy_hi, const void* dy_lo, const void* wt_hi, const void* wt_lo, const void* dy2_hi,
                                       const void* dy2_lo, const void* wt2_hi, const void* wt2_lo, float* dx, int N, int H, int W,
                                       int Cin, int Cout, int kh, int kw, int pad, const float* addend, void* stream) {
    if (!dy2_hi || !dy2_lo || !wt2_hi || !wt2_lo) return AB_EINVAL;
    if (kh > 3 || kw > 3 || (H & 1) || (W & 1) || (H + 2 * pad - kh) / 2 + 1 != H / 2 || (W + 2 * pad - kw) / 2 + 1 != W / 2) return AB_ESHAPE;
    return dgrad_x3_impl(dy_hi, dy_lo, wt_hi, wt_lo, dx, N, H, W, Cin, Cout, kh, kw, 2, pad, addend, nullptr, stream, dy2_hi, dy2_lo,
                         wt2_hi, wt2_lo);
}

// ab_conv2d_dgrad_x3_pair whose result arrives at relu(bn(bn_y) [+ residual]) -- the last block of the stage below (resnet.py:85-101, 178-192
// backwards): dz receives the MASKED gradient, bn_part [rows][Cin][2] the per-tile (sum dz, sum dz * xhat) that ab_bn_bwd_x3 takes as `part`.
// rows = ab_conv2d_dgrad_x3_pair_bn_rows(...); 0: shape not handled (use ab_conv2d_dgrad_x3_pair + the full ab_bn_bwd_x3).
extern "C" int ab_conv2d_dgrad_x3_pair_bn_rows(int N, int H, int W, int Cin, int Cout, int kh, int kw, int pad) {
    if (kh != 3 || kw != 3 || pad != 1 || getenv("AB_X3_BNFUSE_OFF")) return 0;
    return convp_s2dgrad_bn_rows(N, H, W, Cin, Cout);
}
extern "C" int ab_conv2d_dgrad_x3_pair_bn(const void* dy_hi, const void* dy_lo, const void* wt_hi, const void* wt_lo, const void* dy2_hi,
                                          const void* dy2_lo, const void* wt2_hi, const void* wt2_lo, float* dz, int N, int H, int W,
                                          int Cin, int Cout, int kh, int kw, int pad, const float* bn_y, const void* bn_out_hi,
                                          const float* bnp, float* bn_part, void* stream) {
    if (!dy_hi || !dy_lo || !wt_hi || !wt_lo || !dy2_hi || !dy2_lo || !wt2_hi || !wt2_lo || !dz || !bn_y || !bnp || !bn_part) return AB_EINVAL;
    if (!ab_conv2d_dgrad_x3_pair_bn_rows(N, H, W, Cin, Cout, kh, kw, pad)) return AB_ESHAPE;
    return convp_s2dgrad_run(dy_hi, dy_lo, wt_hi, wt_lo, dy2_hi, dy2_lo, wt2_hi, wt2_lo, dz, N, H, W, Cin, Cout, as_stream(stream), bn_y,
                             bn_out_hi, bnp, bn_part);
}

// Data gradient whose result is the gradient arriving at  relu(bn(bn_y) [+ residual])  (resnet.py:85-101 backwards): dx receives
// the MASKED gradient dz (mask from bn_out_hi, the hi plane of the stored activation, or recomputed from bn_y when NULL) and
// bn_part [rows][Cin][2] the per-tile sums (sum dz, sum dz*xhat) that ab_bn_bwd_x3 takes as `part` -- the reduction pass
// of that BatchNorm backward is gone.  rows = ab_conv2d_dgrad_x3_bn_rows(...); 0: shape not handled (use ab_conv2d_dgrad_x3).
// (1x1 / s1 / p0 -- the final layer of the head, whose data gradient arrives at relu(bn(deconv output)): the generic kernel's epilogue, mask
// recomputed from bn_y only, Cin a multiple of the 64- or 128-channel tile)
static int x3_bn_1x1_rows(int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad) {
    static const int off = getenv("AB_X3_BN1X1_OFF") ? atoi(getenv("AB_X3_BN1X1_OFF")) : 0;
    if (off || kh != 1 || kw != 1 || stride != 1 || pad != 0 || Cin % 64 || (Cin > 64 && Cin % 128) || Cout % 32) return 0;
    return conv_gemm2_x3_mtiles(N * H * W, Cin, Cout / 32, 0);
}
extern "C" int ab_conv2d_dgrad_x3_bn_rows(int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad) {
    if (getenv("AB_X3_BNFUSE_OFF")) return 0;
    if (!x3_is_c3(kh, kw, stride, pad)) return x3_bn_1x1_rows(N, H, W, Cin, Cout, kh, kw, stride, pad);
    return conv3x3_x3_tiles_bnr(N, H, W, Cout, Cin);
}

extern "C" int ab_conv2d_dgrad_x3_bn(const void* dy_hi, const void* dy_lo, const void* wt_hi, const void* wt_lo, float* dz, int N,
                                     int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, const float* addend,
                                     const float* bn_y, const void* bn_out_hi, const float* bnp, float* bn_part, void* stream) {
    if (!dy_hi || !dy_lo || !wt_hi || !wt_lo || !dz || !bn_y || !bnp || !bn_part) return AB_EINVAL;
    if (!ab_conv2d_dgrad_x3_bn_rows(N, H, W, Cin, Cout, kh, kw, stride, pad)) return AB_ESHAPE;
    if (!x3_is_c3(kh, kw, stride, pad)) {
        if (addend || bn_out_hi) return AB_ESHAPE;          // the generic kernel's form: no residual below, mask from bn_y
        ConvGemmArgs g = {};
        g.A = dy_hi; g.A_lo = dy_lo; g.Bw = wt_hi; g.Bw_lo = wt_lo; g.Out = dz;
        g.bn_y = bn_y; g.bnp = bnp; g.bn_part = bn_part;
        g.N = N; g.Ha = H; g.Wa = W; g.Ca = Cout; g.Ho = H; g.Wo = W; g.Cn = Cin;
        g.a_sh = g.a_sw = 1; g.cpt = Cout / 32; g.ktot = Cout;
        g.P = H; g.Q = W; g.out_sh = g.out_sw = 1; g.M = N * H * W; g.ntaps = 1;
        return conv_gemm2_x3_run(g, as_stream(stream));
    }
    return conv3x3_x3_run(dy_hi, dy_lo, wt_hi, wt_lo, dz, N, H, W, Cout, Cin, 1, addend, nullptr, as_stream(stream), bn_y,
                          bn_out_hi, bnp, bn_part);
}

// ---------------------------------------------------------------- weight gradient
static const long X3_WGRAD_MAX_M_FWD = 1L << 21;
// bytes of slab workspace ab_conv2d_wgrad_x3 writes for this shape (its own slice counts: not those of the bf16 kernels)
extern "C" long ab_conv2d_wgrad_x3_workspace(int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad) {
    if (kh * kw > 16 || Cin % 64 || Cout % 64) return 0;
    const long slab = (long)Cout * kh * kw * Cin * 4;
    if (x3_is_c3(kh, kw, stride, pad)) {
        int ns = wgrad3x3_x3_slices(N, H, W, Cin, Cout);
        if (ns > 0) return ns * slab;
    }
    const long Mi = (long)((H + 2 * pad - kh) / stride + 1) * ((W + 2 * pad - kw) / stride + 1);
    int n = N;
    while ((long)n * Mi >= X3_WGRAD_MAX_M_FWD && n > 1) n = n - n / 2;       // the larger half of ab_conv2d_wgrad_x3's batch split
    int ns = wgrad_gemm2_x3_slices((int)(n * Mi), Cout, Cin, kh * kw);
    return ns > 0 ? ns * slab : 0;
}

// ---- G same-shape 3x3 / stride 1 / pad 1 weight gradients in ONE slab launch + ONE reduction launch (round 6).  The results are those of G
// ab_conv2d_wgrad_x3 calls up to the summation order over pixel slices (each problem has 1 / G of the slices of its own launch).
extern "C" long ab_conv2d_wgrad_x3_group_workspace(int G, int N, int H, int W, int Cin, int Cout) {
    const int ns = wgrad3x3_x3_group_slices(G, N, H, W, Cin, Cout);
    return ns > 0 ? (long)G * ns * Cout * 9 * Cin * 4 : 0;
}
extern "C" int ab_conv2d_wgrad_x3_group(const ab_wgrad_group_item* items_host, int G, int N, int H, int W, int Cin, int Cout, void* workspace,
                                        int accumulate, void* stream) {
    if (!items_host || !workspace || G < 1 || G > AB_WGRAD_GROUP_MAX) return AB_EINVAL;
    const int ns = wgrad3x3_x3_group_slices(G, N, H, W, Cin, Cout);
    if (!ns) return AB_ESHAPE;
    const long slab = (long)Cout * 9 * Cin;
    const void* xh[AB_WGRAD_GROUP_MAX]; const void* xl[AB_WGRAD_GROUP_MAX]; const void* dh[AB_WGRAD_GROUP_MAX]; const void* dl[AB_WGRAD_GROUP_MAX];
    float* sl[AB_WGRAD_GROUP_MAX];
    ab_wgrad_reduce_desc d[AB_WGRAD_GROUP_MAX];
    for (int p = 0; p < G; ++p) {
        const ab_wgrad_group_item& it = items_host[p];
        if (!it.x_hi || !it.x_lo || !it.dy_hi || !it.dy_lo || !it.dw) return AB_EINVAL;
        xh[p] = it.x_hi; xl[p] = it.x_lo; dh[p] = it.dy_hi; dl[p] = it.dy_lo; sl[p] = (float*)workspace + (long)p * ns * slab;
        if (ns == 1 && !accumulate) sl[p] = it.dw;      // one slice per problem (layer 4 in groups of four): its "slab" IS dW [Cout][9][Cin]
        d[p] = ab_wgrad_reduce_desc{};
        d[p].slabs = sl[p]; d[p].dst = it.dw; d[p].slab_elems = slab; d[p].nslices = ns; d[p].src_j = 9 * Cin; d[p].dst_j = 9 * Cin;
        d[p].accumulate = accumulate; d[p].stem_mask = 0;
    }
    int rc = wgrad3x3_x3_group_run(G, xh, xl, dh, dl, sl, N, H, W, Cin, Cout, as_stream(stream));
    if (rc) return rc;
    if (ns == 1 && !accumulate) return 0;               // (the reduction of one slice was a 38 MB copy: 32 us per group of four on layer 4)
    return ab_wgrad_reduce_batch(d, G, stream);
}

// The generic weight-gradient kernels index pixels with 21-bit magic divisions: larger launches (a stem at batch 128 x 256^2) run
// as two half-batches, the second accumulating onto the first.
static const long X3_WGRAD_MAX_M = 1L << 21;

extern "C" int ab_conv2d_wgrad_x3(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, float* dw, int N,
                                  int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, void* workspace,
                                  int accumulate, void* stream) {
    if (!x_hi || !x_lo || !dy_hi || !dy_lo || !dw || !workspace) return AB_EINVAL;
    if (kh * kw > 16 || Cin % 64 || Cout % 64) return AB_ESHAPE;
    // (same branch order as ab_conv2d_wgrad_x3_workspace: the all-taps 3x3 kernel where it applies, else the generic kernel --
    // split into half-batches beyond 2^21 output pixels, 3x3/s1 shapes the all-taps kernel declined included)
    hipStream_t st = as_stream(stream);
    const long slab = (long)Cout * kh * kw * Cin;
    if (x3_is_c3(kh, kw, stride, pad)) {
        int ns = wgrad3x3_x3_slices(N, H, W, Cin, Cout);
        if (ns > 0) {
            int rc = wgrad3x3_x3_run(x_hi, x_lo, dy_hi, dy_lo, (float*)workspace, N, H, W, Cin, Cout, st);
            if (rc) return rc;
            return wgrad_launch_reduce((float*)workspace, ns, slab, 9 * Cin, 9 * Cin, dw, accumulate, 0, st);
        }
    }
    {
        const long Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
        if ((long)N * Ho * Wo >= X3_WGRAD_MAX_M && N > 1) {
            const int n1 = N / 2;
            const long xo = (long)n1 * H * W * Cin, yo = (long)n1 * Ho * Wo * Cout;
            int rc = ab_conv2d_wgrad_x3(x_hi, x_lo, dy_hi, dy_lo, dw, n1, H, W, Cin, Cout, kh, kw, stride, pad, workspace, accumulate, stream);
            if (rc) return rc;
            return ab_conv2d_wgrad_x3((const bf16_t*)x_hi + xo, (const bf16_t*)x_lo + xo, (const bf16_t*)dy_hi + yo, (const bf16_t*)dy_lo + yo,
                                      dw, N - n1, H, W, Cin, Cout, kh, kw, stride, pad, workspace, 1, stream);
        }
    }
    const int M = N * ((H + 2 * pad - kh) / stride + 1) * ((W + 2 * pad - kw) / stride + 1);
    int ns = wgrad_gemm2_x3_slices(M, Cout, Cin, kh * kw);
    if (ns <= 0) return AB_ESHAPE;
    int rc = wgrad_gemm2_x3_run(x_hi, x_lo, dy_hi, dy_lo, (float*)workspace, N, H, W, Cin, Cout, kh, kw, stride, pad, st);
    if (rc) return rc;
    return wgrad_launch_reduce((float*)workspace, ns, slab, kh * kw * Cin, kh * kw * Cin, dw, accumulate, 0, st);
}

// ---------------------------------------------------------------- stem 7x7/2 (resnet.py:154) on split planes
// xpad planes: zero-bordered NHWC4 image [N, H+6, W+8, 4]; w planes [64][7][8][4]; y fp32 [N, H/2, W/2, 64]
extern "C" int ab_conv2d_stem_x3_stat_rows(int N, int H, int W) {
    if (int t = stem_halo_x3_tiles(N, H, W)) return t;          // persistent halo kernel: one partial row per workgroup
    return conv_gemm2_x3_stem_mtiles(N * (H / 2) * (W / 2));
}

extern "C" int ab_conv2d_stem_fwd_x3(const void* xpad_hi, const void* xpad_lo, const void* w_hi, const void* w_lo, float* y, int N,
                                     int H, int W, int Cout, float* stats, void* stream) {
    if (!xpad_hi || !w_hi || !w_lo || !y) return AB_EINVAL;
    if ((H & 1) || (W & 1) || Cout != 64) return AB_ESHAPE;
    if (stem_halo_x3_tiles(N, H, W)) return stem_halo_x3_run(xpad_hi, xpad_lo, w_hi, w_lo, y, N, H, W, Cout, stats, as_stream(stream));
    if (!xpad_lo) return AB_ESHAPE;       // the integer image plane (xpad_lo == NULL: AB_DT_U8N) runs on the persistent halo kernel only
    ConvGemmArgs g = {};
    g.A = xpad_hi; g.A_lo = xpad_lo; g.Bw = w_hi; g.Bw_lo = w_lo; g.Out = y; g.stats = stats;
    g.N = N; g.Ha = H + 6; g.Wa = W + 8; g.Ca = 4;
    g.Ho = H / 2; g.Wo = W / 2; g.Cn = Cout; g.P = g.Ho; g.Q = g.Wo; g.out_sh = g.out_sw = 1; g.a_sh = g.a_sw = 2;
    g.ktot = 7 * 32; g.M = N * g.P * g.Q;
    return conv_gemm2_x3_stem_run(g, as_stream(stream));
}

/* dw fp32 [Cout][7][8][4]; workspace: ab_conv2d_stem_wgrad_workspace(N, H, W, Cout) bytes */
static int stem_wgrad_x3_impl(const bf16_t* xpad_hi, const bf16_t* xpad_lo, const bf16_t* dy_hi, const bf16_t* dy_lo, float* dw, int N,
                              int H, int W, int Cout, void* workspace, int accumulate, hipStream_t st) {
    if ((long)N * (H / 2) * (W / 2) >= X3_WGRAD_MAX_M && N > 1) {         // two half-batches (see X3_WGRAD_MAX_M)
        const int n1 = N / 2;
        const long xo = (long)n1 * (H + 6) * (W + 8) * 4, yo = (long)n1 * (H / 2) * (W / 2) * Cout;
        int rc = stem_wgrad_x3_impl(xpad_hi, xpad_lo, dy_hi, dy_lo, dw, n1, H, W, Cout, workspace, accumulate, st);
        if (rc) return rc;
        return stem_wgrad_x3_impl(xpad_hi + xo, xpad_lo ? xpad_lo + xo : nullptr, dy_hi + yo, dy_lo + yo, dw, N - n1, H, W, Cout, workspace, 1, st);
    }
    int ns = wgrad_gemm2_stem_slices(N, H, W, Cout);
    if (ns <= 0) return AB_ESHAPE;
    int rc = wgrad_gemm2_x3_stem_run(xpad_hi, xpad_lo, dy_hi, dy_lo, (float*)workspace, N, H, W, Cout, st);
    if (rc) return rc;
    return wgrad_launch_reduce((float*)workspace, ns, (long)Cout * 256, 256, 7 * 32, dw, accumulate, 1, st);
}

extern "C" int ab_conv2d_stem_wgrad_x3(const void* xpad_hi, const void* xpad_lo, const void* dy_hi, const void* dy_lo, float* dw,
                                       int N, int H, int W, int Cout, void* workspace, void* stream) {
    if (!xpad_hi || !dy_hi || !dy_lo || !dw || !workspace) return AB_EINVAL;          // xpad_lo == NULL: xpad_hi is the integer image plane (AB_DT_U8N)
    if ((H & 1) || (W & 1) || Cout % 64) return AB_ESHAPE;
    return stem_wgrad_x3_impl((const bf16_t*)xpad_hi, (const bf16_t*)xpad_lo, (const bf16_t*)dy_hi, (const bf16_t*)dy_lo, dw, N, H, W, Cout,
                              workspace, 0, as_stream(stream));
}
