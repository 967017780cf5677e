// 2x2-tap convolutions on 16 x 16 base grids with an LDS-resident input patch: the two faces of ConvTranspose2d(4x4, stride 2, pad 1) --
// the transposed convolutions of IntegralDeconvHead (anakin/models/simplebaseline.py:152-172: deconv_layers) -- in split-bf16.
//
//   forward  (NSUB = 1): out[n, 2p + a, 2q + b, :] = sum over the 2 x 2 taps of class (a, b) of in[n, p + dy, q + dx, :] . W[tap]; the four
//            output-parity classes share one grid, a workgroup owns ONE class of a whole 16 x 16 image x BN output channels.
//   backward (NSUB = 4): the data gradient of that layer is the 4x4 / stride 2 / pad 1 convolution of the incoming gradient: the 16 taps are
//            four 2 x 2-tap convolutions over the four parity sub-grids of the (32 x 32) input, summed into ONE accumulator tile: per
//            32-channel chunk the workgroup walks the four sub-grids, each with its own patch.
//
// Why not conv_gemm2.hip's tap-by-tap implicit GEMM (what ran these four launches: 118 + 77 us forward, 142 + 80 us backward, 26 - 35 % of the
// split-bf16 roof): there every tap re-fetches its input rows from L2 and one 32-channel K step sits between two barriers with the tap's
// addresses rebuilt each time.  Here -- the loop of conv3x3.hip with four taps -- a (sub-)patch of 17 x 18 pixels is DMA'd once per
// 32-channel chunk and serves its four taps by shifting the fragment address (a quarter of the activation fill), only the weights stream
// per tap (ring of four stages, two steps ahead, counted vmcnt + raw s_barrier), and every address in the K loop is a lane constant plus
// an immediate.  LDS image, swizzles and the fp32 epilogue are those of conv3x3.hip (TW = 16, patch pitch 18).
#include "conv3x3.h"

static __device__ uint4 c22_zero_page[2];

struct C22Unit { int oy0, ox0, sy, sx; int koff[4]; };      // patch origin (base units), input parity (stride-2 input), K offset of tap (t >> 1, t & 1)
struct C22Args {
    const void* X; const void* X_lo; const void* Wt; unsigned wlo_delta;      // input planes [N, Hi, Wi, C]; weight rows [Cn][ktot], lo plane at + delta bytes
    float* Out; float* stats;                                                 // [N, Ho, Wo, Cn]; BatchNorm partials [rows][Cn][2] or NULL
    int N, Hi, Wi, C, Cn, ktot;
    int in_stride;                   // 1: input pixel = base + origin + tap;  2: = 2 * (base + origin + tap) + (sy, sx)
    int Ho, Wo, out_stride;          // output pixel = base * out_stride + (cls_oy, cls_ox)
    int nclass;                      // output classes in the grid (1 | 4)
    C22Unit unit[4];                 // NSUB == 1: by output class;  NSUB == 4: by input sub-grid
    int cls_oy[4], cls_ox[4];
    // eval-mode fold of the BatchNorm (+ ReLU) that follows: out = relu?(acc * ep_scale[c] + ep_shift[c]) (conv_gemm2.hip's expression: bit-identical
    // to this kernel + ab_bn_apply_x3), written as fp32 `Out` or, when out_hi != NULL, as the (hi, lo) planes the next convolution reads
    const float* ep_scale; const float* ep_shift; int ep_relu; void* out_hi; void* out_lo;
};

// G8 = false: base grid 16 x 16, a tile is one image (BM = 256), patch 17 rows x 18 (17 used), swizzle key (px >> 1) & 7.
// G8 = true : base grid 8 x 8, a tile is NI = BM / 64 consecutive images, each with its own 9 x 9 patch (pitch 9: consecutive rows alternate
//             the 128-byte half), swizzle key ((px >> 1) & 3) | ((py & 1) << 2) -- conv3x3.hip's TW = 8 layout: a ds_read_b128 lane group
//             spans four patch rows there.
template <int BM, int BN, int NSUB, bool G8>
__global__ __launch_bounds__(512) void conv2x2_kernel(C22Args g) {
    constexpr int TW = G8 ? 8 : 16, TH = G8 ? 8 : 16, NI = BM / (TW * TH), PW = G8 ? 9 : 18, PH = G8 ? 9 : 17, IPIX = PH * PW, NPIX = NI * IPIX;
    static_assert(BM == NI * TW * TH && (G8 || NI == 1), "tile = whole images");
    constexpr int WM = 4, WN = 2, NW = 8, NT = 512;
    constexpr int PI = (NPIX + 7) / 8, LP = (PI + NW - 1) / NW, PATCH_BYTES = LP * NW * 1024;
    constexpr int IB = BN / 8, LB = IB / NW, BBYTES = BN * 128;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int PATCH0 = 4 * BBYTES;                        // LDS: [weight ring x4][patch 0][patch 1]
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WN, wave_n = wave % WN;
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, within = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const int tiles_n = g.Cn / BN;
    const int tile_sp = logical / tiles_n, tile_n = logical - tile_sp * tiles_n;      // tile_sp = image group * nclass + class: the BatchNorm partial row
    const int img = (tile_sp / g.nclass) * NI, cls = tile_sp % g.nclass;                 // first image of the tile
    const int n0 = tile_n * BN;
    const bf16_t* __restrict__ X = (const bf16_t*)g.X;
    const bf16_t* __restrict__ Xlo = (const bf16_t*)g.X_lo;
    const bf16_t* __restrict__ Wt = (const bf16_t*)g.Wt;
    const bf16_t* zp = (const bf16_t*)c22_zero_page;
    const int nchunks = g.C / 32;
    const unsigned lds0 = lds_addr_of(smem);

    // ---- the units this workgroup walks per chunk (scalars: the tables are read once, here)
    int u_oy[NSUB], u_ox[NSUB], u_sy[NSUB], u_sx[NSUB], u_koff[NSUB][4];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
        const C22Unit& u = g.unit[NSUB == 1 ? cls : s];
        u_oy[s] = u.oy0; u_ox[s] = u.ox0; u_sy[s] = u.sy; u_sx[s] = u.sx;
#pragma unroll
        for (int t = 0; t < 4; ++t) u_koff[s][t] = u.koff[t];
    }

    // ---- per-lane patch fill assignment: instruction ii covers patch pixels ii * 8 .. + 7, lane & 7 the 16-byte slot
    int p_py[LP], p_px[LP], p_il[LP]; unsigned p_coff[LP]; bool p_lo[LP], p_in[LP];
#pragma unroll
    for (int j = 0; j < LP; ++j) {
        const int ii = wave * LP + j, pp = ii * 8 + (lane >> 3);
        p_il[j] = pp / IPIX;
        const int rem = pp - p_il[j] * IPIX;
        p_py[j] = rem / PW; p_px[j] = rem - p_py[j] * PW;
        p_in[j] = ii < PI && pp < NPIX && p_px[j] < TW + 1;
        const int c = (lane & 7) ^ (G8 ? (((p_px[j] >> 1) & 3) | ((p_py[j] & 1) << 2)) : ((p_px[j] >> 1) & 7));
        p_lo[j] = (c & 4) != 0; p_coff[j] = (unsigned)((c & 3) * 8);
    }
    unsigned b_voff[LB];
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        const int ii = wave * LB + j, r = ii * 8 + (lane >> 3);
        int c = (lane & 7) ^ ((r >> 1) & 7);
        const unsigned pl = (c & 4) ? g.wlo_delta : 0u;
        c &= 3;
        b_voff[j] = (unsigned)(((long)(n0 + r) * g.ktot + c * 8) * 2) + pl;
    }
    // ---- per-lane fragment addresses
    const int l32 = lane & 31, fhalf = lane >> 5;
    unsigned b_rel[TN][4], a_rel[TM][2][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = (wave_n * TN + j) * 32 + l32;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) b_rel[j][kk] = lds0 + r * 128 + (((kk * 2 + fhalf) ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wave_m * TM + i) * 32 + l32, il = row / (TW * TH), rr = row - il * (TW * TH), oy = rr / TW, ox = rr - oy * TW;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int px = ox + d;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)      // (G8: the patch row's bit of the key is flipped below for the second tap row)
                a_rel[i][d][kk] = lds0 + PATCH0 + (il * IPIX + oy * PW + px) * 128 +
                                  (((kk * 2 + fhalf) ^ (G8 ? (((px >> 1) & 3) | ((oy & 1) << 2)) : ((px >> 1) & 7))) << 4);
        }
    }

    auto issue_patch = [&](int sc, int pbuf) {          // super-chunk sc = chunk * NSUB + sub
        const int chunk = sc / NSUB, s = sc - chunk * NSUB;
        int oy0 = u_oy[0], ox0 = u_ox[0], sy = u_sy[0], sx = u_sx[0];
#pragma unroll
        for (int k = 1; k < NSUB; ++k) if (s == k) { oy0 = u_oy[k]; ox0 = u_ox[k]; sy = u_sy[k]; sx = u_sx[k]; }
#pragma unroll
        for (int j = 0; j < LP; ++j) {
            const int ii = wave * LP + j;
            const int r = (oy0 + p_py[j]) * g.in_stride + sy, c = (ox0 + p_px[j]) * g.in_stride + sx;
            const bool ok = p_in[j] && (unsigned)r < (unsigned)g.Hi && (unsigned)c < (unsigned)g.Wi;
            const bf16_t* src = ok ? (p_lo[j] ? Xlo : X) + ((((long)(img + p_il[j]) * g.Hi + r) * g.Wi + c) * g.C + chunk * 32 + p_coff[j]) : zp;
            glds16(src, __builtin_amdgcn_readfirstlane(lds0 + PATCH0 + pbuf * PATCH_BYTES + ii * 1024));
        }
    };
    auto issue_b = [&](int chunk, int koff, int slot) {
        const bf16_t* base = Wt + (koff + chunk * 32);
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            const int ii = wave * LB + j;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(b_voff[j]), "s"(base), "s"(__builtin_amdgcn_readfirstlane(lds0 + slot * BBYTES + ii * 1024)) : "memory");
        }
    };

    f32x16 acc[TM][TN], accx[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accx[i][j][r] = 0.f; }

    // ---- software pipeline (conv3x3.hip's, with four taps per patch): P(0) B(0) B(1) | per step after its barrier: B(step + 2), and at tap 0
    // of super-chunk sc also P(sc + 1).  All loads are inline asm: the waits below are exact.
    const int nsc = nchunks * NSUB;
    issue_patch(0, 0);
    issue_b(0, u_koff[0][0], 0);
    issue_b(0, u_koff[0][1], 1);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            const int sc = chunk * NSUB + s;
            const bool more = sc + 1 < nsc;
            const unsigned pbase = (sc & 1) * PATCH_BYTES;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (!more && t == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if ((t == 1 || t == 2) && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LB + LP) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LB) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // WAR on the ring stage restaged below: see conv_gemm2.hip
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (t + 2 < 4) issue_b(chunk, u_koff[s][t + 2], t + 2);
                else if (more) {          // taps 0, 1 of the next super-chunk
                    const int s2 = (s + 1) % NSUB, ch2 = s + 1 < NSUB ? chunk : chunk + 1;
                    issue_b(ch2, u_koff[s2][t - 2], t - 2);
                }
                if (t == 0 && more) issue_patch(sc + 1, (sc + 1) & 1);
                const int dh = t >> 1, dw = t & 1;
                const unsigned aoff = pbase + dh * PW * 128;
                const unsigned aflip = (G8 && dh) ? 64u : 0u;      // the (py & 1) bit of the G8 key: patch row = oy + dh (bases are 128-byte aligned)
                u32x4 fa[4][TM], fb[4][TN];
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) fa[k2 + 2 * h][i] = *(const lds_u32x4*)((a_rel[i][dw][k2 + 2 * h] ^ aflip) + aoff);
#pragma unroll
                        for (int j = 0; j < TN; ++j) fb[k2 + 2 * h][j] = *(const lds_u32x4*)(b_rel[j][k2 + 2 * h] + t * BBYTES);
                    }
                }
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const bf16x8 bh = __builtin_bit_cast(bf16x8, fb[k2][j]), bl = __builtin_bit_cast(bf16x8, fb[k2 + 2][j]);
                            const bf16x8 ah = __builtin_bit_cast(bf16x8, fa[k2][i]), al = __builtin_bit_cast(bf16x8, fa[k2 + 2][i]);
                            // weights as the first operand: the accumulator is the TRANSPOSED tile (a lane owns one pixel, see the epilogue)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, acc[i][j], 0, 0, 0);
                            accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al, accx[i][j], 0, 0, 0);
                            accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah, accx[i][j], 0, 0, 0);
                        }
                }
            }
        }
    }
    __syncthreads();

    // ---- fp32 epilogue (conv3x3.hip's): a lane owns ONE pixel and per register quad four consecutive channels = one 16-byte LDS store into
    // the pixel-major staging tile; rows leave as 16-byte vectors to their (strided) output pixels; BatchNorm partials of the tile as stored.
    constexpr int SPF = BN * 4 + 16;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wave_m * TM + i) * 32 + l32;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int cl = (wave_n * TN + j) * 32 + 8 * q4 + 4 * fhalf;
                float4 w;
                w.x = acc[i][j][q4 * 4] + accx[i][j][q4 * 4]; w.y = acc[i][j][q4 * 4 + 1] + accx[i][j][q4 * 4 + 1];
                w.z = acc[i][j][q4 * 4 + 2] + accx[i][j][q4 * 4 + 2]; w.w = acc[i][j][q4 * 4 + 3] + accx[i][j][q4 * 4 + 3];
                *(float4*)(smem + row * SPF + cl * 4) = w;
            }
        }
    }
    __syncthreads();
    constexpr int CPRF = BN / 4;
    static_assert(NT % CPRF == 0, "a thread keeps one channel group over all its rows");
    const int coy = g.cls_oy[cls], cox = g.cls_ox[cls];
    float fs[4] = {0.f, 0.f, 0.f, 0.f}, fq[4] = {0.f, 0.f, 0.f, 0.f};
    for (int id = tid; id < BM * CPRF; id += NT) {
        const int row = id / CPRF, c4 = id - row * CPRF;
        const int il = row / (TW * TH), rr = row - il * (TW * TH);
        const int yy = (rr / TW) * g.out_stride + coy, xx = (rr % TW) * g.out_stride + cox, col = n0 + c4 * 4;
        float4 v = *(const float4*)(smem + row * SPF + c4 * 16);
        const long o = (((long)(img + il) * g.Ho + yy) * g.Wo + xx) * g.Cn + col;
        if (g.ep_scale) {
            const float4 sc = *(const float4*)(g.ep_scale + col), sh = *(const float4*)(g.ep_shift + col);
            v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
            if (g.ep_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        }
        if (g.out_hi) {
            uint2 h, l;
            h.x = pack_bf16x2(v.x, v.y); h.y = pack_bf16x2(v.z, v.w);
            l.x = pack_bf16x2(v.x - __uint_as_float(h.x << 16), v.y - __uint_as_float(h.x & 0xffff0000u));
            l.y = pack_bf16x2(v.z - __uint_as_float(h.y << 16), v.w - __uint_as_float(h.y & 0xffff0000u));
            *(uint2*)((bf16_t*)g.out_hi + o) = h; *(uint2*)((bf16_t*)g.out_lo + o) = l;
        } else *(float4*)(g.Out + o) = v;
        fs[0] += v.x; fq[0] += v.x * v.x; fs[1] += v.y; fq[1] += v.y * v.y;
        fs[2] += v.z; fq[2] += v.z * v.z; fs[3] += v.w; fq[3] += v.w * v.w;
    }
    __syncthreads();
    if (g.stats) {
        float* sp = (float*)smem;                          // [NT / CPRF][BN][2], over the consumed staging tile
        const int rg = tid / CPRF, cb = (tid % CPRF) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) { sp[(rg * BN + cb + k) * 2] = fs[k]; sp[(rg * BN + cb + k) * 2 + 1] = fq[k]; }
        __syncthreads();
        for (int c = tid; c < BN; c += NT) {
            float s2 = 0.f, q2 = 0.f;
            for (int r = 0; r < NT / CPRF; ++r) { s2 += sp[(r * BN + c) * 2]; q2 += sp[(r * BN + c) * 2 + 1]; }
            g.stats[((long)tile_sp * g.Cn + n0 + c) * 2] = s2;
            g.stats[((long)tile_sp * g.Cn + n0 + c) * 2 + 1] = q2;
        }
    }
}

template <int BM, int BN, int NSUB, bool G8>
static int c22_launch(C22Args& g, hipStream_t st) {
    constexpr int NI = BM / (G8 ? 64 : 256), NPIX = NI * (G8 ? 81 : 17 * 18), LP = ((NPIX + 7) / 8 + 7) / 8;
    const size_t ring = (size_t)4 * BN * 128 + 2 * LP * 8 * 1024, stage = (size_t)BM * (BN * 4 + 16), part = (size_t)(512 / (BN / 4)) * BN * 8;
    size_t lds = ring > stage ? ring : stage;
    if (part > lds) lds = part;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv2x2_kernel<BM, BN, NSUB, G8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int blocks = (g.N / NI) * g.nclass * (g.Cn / BN);
    conv2x2_kernel<BM, BN, NSUB, G8><<<blocks, 512, lds, st>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

static int c22_bn(int Cn) {
    static const int f = getenv("AB_C22_BN") ? atoi(getenv("AB_C22_BN")) : 0;
    if (f == 64 || f == 128) return Cn % f ? 0 : f;
    return Cn % 128 == 0 ? 128 : (Cn % 64 == 0 ? 64 : 0);
}
static bool c22_off() { static const int off = getenv("AB_C22_OFF") ? atoi(getenv("AB_C22_OFF")) : 0; return off != 0; }
static bool c22_g8_off() { static const int off = getenv("AB_C22_G8_OFF") ? atoi(getenv("AB_C22_G8_OFF")) : 0; return off != 0; }

// ConvTranspose2d(4x4, s2, p1) forward as the data gradient of the mirrored convolution: dy planes [N, H/2, W/2, K], wt rows [Cn][4][4][K]
// ("IHWO"), out fp32 [N, H, W, Cn]; H = W = 32 (one image per tile) or 16 (four images per tile, N % 4 == 0).  Rows of BatchNorm partials:
// tiles x 4 parity classes.  0 / AB_ESHAPE: shape not taken.
int conv2x2_tfwd_rows(int N, int H, int W, int Cn, int K) {
    if (c22_off() || K % 32) return 0;
    if (H == 32 && W == 32 && c22_bn(Cn)) return N * 4;
    if (H == 16 && W == 16 && !c22_g8_off() && N % 4 == 0 && Cn % 64 == 0) return N;
    return 0;
}
int conv2x2_tfwd_run(const void* x_hi, const void* x_lo, const void* wt_hi, const void* wt_lo, float* out, int N, int H, int W, int Cn, int K,
                     float* stats, hipStream_t st, const float* ep_scale, const float* ep_shift, int ep_relu, void* out_hi, void* out_lo) {
    if (!conv2x2_tfwd_rows(N, H, W, Cn, K)) return AB_ESHAPE;
    const long delta = (const char*)wt_lo - (const char*)wt_hi;
    if (delta < 0 || delta >= (1L << 31)) return AB_EINVAL;
    C22Args g = {};
    g.X = x_hi; g.X_lo = x_lo; g.Wt = wt_hi; g.wlo_delta = (unsigned)delta; g.Out = out; g.stats = stats;
    g.N = N; g.Hi = H / 2; g.Wi = W / 2; g.C = K; g.Cn = Cn; g.ktot = 16 * K;
    g.in_stride = 1; g.Ho = H; g.Wo = W; g.out_stride = 2; g.nclass = 4;
    g.ep_scale = ep_scale; g.ep_shift = ep_shift; g.ep_relu = ep_relu; g.out_hi = out_hi; g.out_lo = out_lo;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) {
        // class (a, b): kernel rows i with (a + 1 - i) even read input row p + (a + 1 - i) / 2: a = 0: i = 1 -> p, i = 3 -> p - 1; a = 1: i = 0 -> p + 1, i = 2 -> p
        C22Unit& u = g.unit[a * 2 + b];
        u.oy0 = a ? 0 : -1; u.ox0 = b ? 0 : -1; u.sy = u.sx = 0;
        for (int t = 0; t < 4; ++t) {
            const int dh = t >> 1, dw = t & 1;                       // patch row / column of the tap
            const int i = a ? (dh ? 0 : 2) : (dh ? 1 : 3), j = b ? (dw ? 0 : 2) : (dw ? 1 : 3);
            u.koff[t] = (i * 4 + j) * K;
        }
        g.cls_oy[a * 2 + b] = a; g.cls_ox[a * 2 + b] = b;
    }
    if (H == 16) return c22_launch<256, 64, 1, true>(g, st);
    return c22_bn(Cn) == 128 ? c22_launch<256, 128, 1, false>(g, st) : c22_launch<256, 64, 1, false>(g, st);
}

// 4x4 / stride 2 / pad 1 convolution (the data gradient of that ConvTranspose2d): x planes [N, H, W, C], w rows [Cn][4][4][C] (OHWI),
// out fp32 [N, H/2, W/2, Cn]; H = W = 32, or 16 (two images per tile, N even).  BatchNorm partials: one row per tile.
int conv2x2_s2fwd_ok(int N, int H, int W, int C, int Cn) {
    if (c22_off() || C % 32 || Cn % 64) return 0;
    if (H == 32 && W == 32) return N;
    if (H == 16 && W == 16 && !c22_g8_off() && N % 2 == 0) return N / 2;
    return 0;
}
int conv2x2_s2fwd_run(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, float* out, int N, int H, int W, int C, int Cn,
                      float* stats, hipStream_t st) {
    if (!conv2x2_s2fwd_ok(N, H, W, C, Cn)) return AB_ESHAPE;
    const long delta = (const char*)w_lo - (const char*)w_hi;
    if (delta < 0 || delta >= (1L << 31)) return AB_EINVAL;
    C22Args g = {};
    g.X = x_hi; g.X_lo = x_lo; g.Wt = w_hi; g.wlo_delta = (unsigned)delta; g.Out = out; g.stats = stats;
    g.N = N; g.Hi = H; g.Wi = W; g.C = C; g.Cn = Cn; g.ktot = 16 * C;
    g.in_stride = 2; g.Ho = H / 2; g.Wo = W / 2; g.out_stride = 1; g.nclass = 1;
    for (int sy = 0; sy < 2; ++sy) for (int sx = 0; sx < 2; ++sx) {
        // input row 2p + kh - 1: kh = 1, 3 are the even rows p, p + 1 (origin 0); kh = 0, 2 the odd rows p - 1, p (origin -1)
        C22Unit& u = g.unit[sy * 2 + sx];
        u.oy0 = sy ? -1 : 0; u.ox0 = sx ? -1 : 0; u.sy = sy; u.sx = sx;
        for (int t = 0; t < 4; ++t) {
            const int dh = t >> 1, dw = t & 1;
            const int kh = sy ? (dh ? 2 : 0) : (dh ? 3 : 1), kw = sx ? (dw ? 2 : 0) : (dw ? 3 : 1);
            u.koff[t] = (kh * 4 + kw) * C;
        }
    }
    g.cls_oy[0] = g.cls_ox[0] = 0;
    if (H == 16) return c22_launch<128, 64, 4, true>(g, st);
    return c22_launch<256, 64, 4, false>(g, st);
}
