// 3x3 / stride 1 / pad 1 convolution (forward, and data-gradient via flipped taps) with an LDS-resident input halo.
//
// Why: measured on MI355X the tap-by-tap implicit GEMM (conv_gemm*.hip) is bound by L2->LDS load throughput
// (~10 TB/s), not by MFMA: every tap re-fetches the same input pixels, 9x redundant traffic.  Here a workgroup owns a
// TH x TW tile of output pixels; per 64-channel chunk it loads the (TH+2) x (TW+2) input patch ONCE (direct-to-LDS,
// 128 bytes per pixel) and serves all 9 taps from it by shifting the fragment read address; only the weights stream
// per tap (3-deep LDS ring, counted vmcnt, raw s_barrier).  Activation traffic drops ~9x and the bytes loaded per
// FLOP are set by the weight stream alone: BN x 128 B per (BM x BN x 64) MACs.
//
// LDS image (both operands are lane-linear global_load_lds fills, so the swizzle sits on the source address):
//   patch : pixel pp = py*(TW+2) + px at byte pp*128, logical chunk c stored at slot c ^ ((px >> 1) & 7)
//           (TW = 8, where a ds_read_b128 lane group spans four patch rows: c ^ (((px >> 1) & 3) | ((py & 1) << 2)), row pitch TW + 3 --
//           with the px-only mask every fragment read of that tile was a 2-way bank conflict: SQ_LDS_BANK_CONFLICT 0.33 of the LDS cycles)
//   weight: row r (output channel) at byte r*128, logical chunk c at slot c ^ ((r >> 1) & 7)
// Both are conflict-free for the 16-lane groups of ds_read_b128 (for TW = 16 two half-rows complement each other).
#include "conv3x3.h"

static __device__ uint4 c3_zero_page[2];     // zero-initialised: source of every out-of-image chunk

// saddr-form LDS-DMA: per-lane 32-bit byte offset + wave-uniform 64-bit base (no VALU 64-bit address arithmetic)
__device__ __forceinline__ void glds16_s(unsigned voff, const void* sbase, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}

// X3 = 1: split-bf16 operands ("bf16x3").  Every fp32 value v is held as two bf16 planes hi = bf16(v), lo = bf16(v - hi)
// (v = hi + lo to 2^-17 relative) and the product is hi*hi + hi*lo + lo*hi on the bf16 MFMA with fp32 accumulation:
// fp32-grade convolutions (the reference trains in fp32, train_artiboost.py:39-41) at a third of the bf16 matrix peak
// instead of the 1/16 the f32-input MFMA gives.  LDS geometry is unchanged: a 128-byte row now carries a 32-channel
// chunk as [hi: 4 x 16 B][lo: 4 x 16 B], so logical slots 0..3 are the hi k-slices and 4..7 the lo ones; only the fill
// source (plane select per lane) and the MFMA sequence differ.  Output, addend and BN partials are fp32.
//
// X3 = 2: X3 data-gradient launch whose output is the gradient arriving at  relu(bn(bn_y) [+ residual]):  the epilogue masks
// it (dz), stores dz instead of the raw gradient and leaves the BatchNorm-backward partial sums (sum dz, sum dz*xhat) per
// tile in bn_part -- the standalone reduction pass over (dout, y, mask) disappears.  The tile of bn_y (and of the mask
// plane / the addend) this thread will need is requested at kernel entry and sits in registers while the K loop runs,
// so the epilogue adds arithmetic only (exposed reads at the end of a workgroup were what made the bf16 variant lose).
//
// X3 = 3: X3 forward launch in EVAL mode: BatchNorm's (scale, shift) are known before the launch (running statistics), so the
// epilogue applies them, adds the residual, applies the ReLU and writes the next convolution's operand planes directly -- with
// the same expression order as bn_apply_x3_kernel (norm_pool.hip), i.e. bit-identical to conv + separate apply pass.  The fp32
// conv output is never stored (no BatchNorm partials either).
template <int BM, int TW, int BN, int WM, int WN, int FLIP, int X3 = 0>
__global__ __launch_bounds__(64 * WM * WN) void conv3x3_kernel(Conv3Args g) {
    constexpr bool BNR = X3 == 2;
    constexpr int CK = X3 ? 32 : 64;                   // channels per K chunk
    // NW = WM*WN waves (4 or 8).  Measured (tools/probe_fill.hip): a wave pulls ~10 GB/s of L2-resident data into LDS
    // whatever its queue depth, and a CU's fill rate scales with the number of waves issuing loads (4 waves 10 TB/s
    // chip-wide, 8 waves 20, 16 waves 30) -- so the same tile is worked by 8 waves where the fill is the limit.
    constexpr int NW = WM * WN, NT = 64 * NW;
    // patch row pitch: TW + 2, or TW + 3 for TW = 8 so that consecutive patch rows alternate LDS half (pixel parity) --
    // a 16-lane fragment read then spans two image rows with the same px set and would otherwise hit the same banks
    // STACK (BM = 128 on TW = 8): the tile is TWO whole 8 x 8 images, handed over by the launch as one 16-row image (the same bytes); the
    // patch holds each image with its own zero halo rows (2 x 10 rows), so no tap of one image reads the other.  Layer 4: a 128-pixel x
    // 64-channel tile streams half the weight bytes per product of the 64 x 128 one at the same 256 workgroups (the launch waits on its
    // L2 -> LDS weight stream: SQ_WAIT_ANY 0.61 of the wave cycles, tools/pmc_conv.sh).
    constexpr bool STACK = (TW == 8 && BM == 128);
    constexpr int TH = BM / TW, PW = (TW == 8) ? TW + 3 : TW + 2, PH = STACK ? TH + 4 : TH + 2;
    constexpr int NPIX = PH * PW;
    constexpr int PI = (NPIX + 7) / 8;                 // 1-KiB instructions per patch
    constexpr int LP = (PI + NW - 1) / NW;             // per wave
    constexpr int PATCH_BYTES = LP * NW * 1024;        // every wave issues exactly LP fills; the tail past NPIX is slack
    constexpr int IB = BN / 8, LB = IB / NW;           // weight-tile instructions: total / per wave
    static_assert(IB % NW == 0, "weight tile rows must split evenly over the waves");
    constexpr int BBYTES = BN * 128;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int NRING = 3;
    constexpr int PATCH0 = NRING * BBYTES;             // LDS: [weight ring x3][patch 0][patch 1][stats]
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WN, wave_n = wave % WN;
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, within = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const int tiles_n = (g.Cn + BN - 1) / BN;
    const int tile_sp = logical / tiles_n, tile_n = logical - tile_sp * tiles_n;
    const int tpi = g.tiles_x * g.tiles_y;
    const int img = tile_sp / tpi, trem = tile_sp - img * tpi;
    const int ty0 = (trem / g.tiles_x) * TH, tx0 = (trem % g.tiles_x) * TW;
    const int n0 = tile_n * BN;
    const bf16_t* __restrict__ X = (const bf16_t*)g.X;
    const bf16_t* __restrict__ Xlo = (const bf16_t*)g.X_lo;
    const bf16_t* __restrict__ Wt = (const bf16_t*)g.Wt;
    const bf16_t* zp = (const bf16_t*)c3_zero_page;
    const int nchunks = g.C / CK;
    const unsigned lds0 = lds_addr_of(smem);

    // ---- X3 = 2: this thread's share of the BatchNorm operands of the epilogue (rows er0 + k*ERS, channels ec4*4 .. +3)
    constexpr int ECPR = BN / 4, ERS = NT / ECPR, ER = BNR ? BM / ERS : 1;
    // held in registers from here on (the 256-pixel x 128-channel tile has none to spare).  Round 6 probe, closed: on layer 1's 128-pixel
    // tile, prefetching nothing (96 VGPRs) or bn_y only (126) so that TWO workgroups share a CU and one's epilogue runs under the other's
    // K loop: 85 - 88 / 98 us against 92 / 104 us (no residual / residual) -- 1.06 x, below the 1.15 x bar; the launches moved to
    // conv3x3r.hip's persistent kernel instead (64 / 87 us)
    constexpr bool EPRE = BNR && ER <= 8;
    constexpr bool EPRE2 = EPRE;
    const int ec4 = tid % ECPR, er0 = tid / ECPR;
    float4 e_y[EPRE ? ER : 1], e_add[EPRE2 ? ER : 1];
    uint2 e_m[EPRE2 ? ER : 1];
    float e_mean[4], e_istd[4], e_sc[4], e_sh[4];
    if constexpr (BNR) {
        const int col = n0 + ec4 * 4;
        const bool cok = col < g.Cn;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            e_sc[k] = cok ? g.bnp[col + k] : 0.f; e_sh[k] = cok ? g.bnp[g.Cn + col + k] : 0.f;
            e_mean[k] = cok ? g.bnp[2 * g.Cn + col + k] : 0.f; e_istd[k] = cok ? g.bnp[3 * g.Cn + col + k] : 0.f;
        }
        if constexpr (EPRE) {
#pragma unroll
            for (int k = 0; k < ER; ++k) {
                const int row = er0 + k * ERS;
                const int yy = ty0 + row / TW, xx = tx0 + row % TW;
                e_y[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (EPRE2) { e_add[k] = e_y[k]; e_m[k] = make_uint2(0u, 0u); }
                if (yy < g.H && xx < g.W && cok) {
                    const long o = (((long)img * g.H + yy) * g.W + xx) * g.Cn + col;
                    e_y[k] = *(const float4*)((const float*)g.bn_y + o);
                    if constexpr (EPRE2) {
                        if (g.bn_out) e_m[k] = *(const uint2*)((const bf16_t*)g.bn_out + o);
                        if (g.addend) e_add[k] = *(const float4*)((const float*)g.addend + o);
                    }
                }
            }
        }
    }

    // ---- per-lane load assignment
    const bf16_t* p_src[LP];                           // chunk-0 source of this lane's patch fills (zero page when outside)
    bool p_ok[LP];
#pragma unroll
    for (int j = 0; j < LP; ++j) {
        int ii = wave * LP + j;
        int pp = ii * 8 + (lane >> 3);
        int py = pp / PW, px = pp - py * PW;
        int y = ty0 + py - 1, x = tx0 + px - 1;
        if constexpr (STACK) { const int ps = py / 10, pr = py - ps * 10; y = (pr == 0 || pr == 9) ? -1 : ps * 8 + pr - 1; }
        p_ok[j] = (ii < PI) && (pp < NPIX) && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
        int c = (lane & 7) ^ (TW == 8 ? (((px >> 1) & 3) | ((py & 1) << 2)) : ((px >> 1) & 7));
        const bf16_t* plane = (X3 && (c & 4)) ? Xlo : X;
        if (X3) c &= 3;
        p_src[j] = p_ok[j] ? plane + ((((long)img * g.H + y) * g.W + x) * g.C + c * 8) : zp;
    }
    unsigned b_voff[LB];                               // byte offset of this lane's weight-row chunk from the step's base
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        int ii = wave * LB + j;
        int r = ii * 8 + (lane >> 3);
        int col = min(n0 + r, g.Cn - 1);               // rows past Cn are loaded from a valid row and never stored
        int c = (lane & 7) ^ ((r >> 1) & 7);
        const unsigned pl = (X3 && (c & 4)) ? g.wlo_delta : 0u;
        if (X3) c &= 3;
        b_voff[j] = (unsigned)(((long)col * g.ktot + c * 8) * 2) + pl;
    }
    // ---- per-lane fragment addresses (LDS byte offsets)
    const int l32 = lane & 31, fhalf = lane >> 5;
    unsigned b_rel[TN][4], a_rel[TM][3][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int r = (wave_n * TN + j) * 32 + l32;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) b_rel[j][kk] = lds0 + r * 128 + (((kk * 2 + fhalf) ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int row = (wave_m * TM + i) * 32 + l32;        // row-major over the TH x TW output tile
        int oy = row / TW, ox = row - oy * TW;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            int px = ox + d;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                a_rel[i][d][kk] = lds0 + PATCH0 + ((STACK ? oy + 2 * (oy >> 3) : oy) * PW + px) * 128 +
                                  (((kk * 2 + fhalf) ^ (TW == 8 ? (((px >> 1) & 3) | ((oy & 1) << 2)) : ((px >> 1) & 7))) << 4);      // TW = 8: tap row dh flips bit 6 below
        }
    }

    auto issue_patch = [&](int chunk, int pbuf) {
#pragma unroll
        for (int j = 0; j < LP; ++j) {
            const int ii = wave * LP + j;
            glds16(p_ok[j] ? (const void*)(p_src[j] + chunk * CK) : (const void*)zp,
                   __builtin_amdgcn_readfirstlane(lds0 + PATCH0 + pbuf * PATCH_BYTES + ii * 1024));
        }
    };
    auto issue_b = [&](int chunk, int tap, int ring) {  // weights of (chunk, tap) into ring slot `ring`
        const bf16_t* base = Wt + ((long)tap * g.C + chunk * CK);
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            const int ii = wave * LB + j;
            glds16_s(b_voff[j], base, __builtin_amdgcn_readfirstlane(lds0 + ring * BBYTES + ii * 1024));
        }
    };


    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // X3: the two cross products run on a second accumulator chain (more independent MFMA chains per SIMD, and the small
    // terms are summed among themselves before they meet the large one)
    f32x16 accx[X3 ? TM : 1][X3 ? TN : 1];
    if constexpr (X3) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accx[i][j][r] = 0.f;
    }

    // ---- software pipeline.  Issue order: P(0) B(0,0) B(0,1) | per step s after its barrier: B(s+2), and at tap 0
    // of chunk c also P(c+1).  All loads are inline asm (invisible to hipcc's wait counting): the waits below are exact.
    issue_patch(0, 0);
    issue_b(0, 0, 0);
    issue_b(0, 1, 1);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        const unsigned pbase = (chunk & 1) * PATCH_BYTES;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // loads issued after B(step): B(step+1) [+ P(chunk+1) for t == 1, 2]
            const bool last = !more && t == 8;
            if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if ((t == 1 || t == 2) && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LB + LP) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LB) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // WAR on the ring stage restaged below: see conv_gemm2.hip
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + 2 < 9) issue_b(chunk, t + 2, (t + 2) % 3);
            else if (more) issue_b(chunk + 1, t + 2 - 9, (t + 2) % 3);
            if (t == 0 && more) issue_patch(chunk + 1, (chunk + 1) & 1);
            const int t3 = t / 3, tr = t % 3;
            const int dh = FLIP ? 2 - t3 : t3, dw = FLIP ? 2 - tr : tr;     // compile-time per unrolled tap
            const unsigned aoff = pbase + dh * PW * 128;
            const unsigned aflip = (TW == 8 && (dh & 1)) ? 64u : 0u;      // the (py & 1) bit of the TW = 8 swizzle: patch row = oy + dh (bases are 128-byte aligned)
            if constexpr (X3) {
                // slots kk = 0,1: hi k-slices (16 channels each), kk = 2,3: the lo planes of the same channels
                u32x4 fa[4][TM], fb[4][TN];
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) fa[k2 + 2 * h][i] = *(const lds_u32x4*)((a_rel[i][dw][k2 + 2 * h] ^ aflip) + aoff);
#pragma unroll
                        for (int j = 0; j < TN; ++j) fb[k2 + 2 * h][j] = *(const lds_u32x4*)(b_rel[j][k2 + 2 * h] + (t % 3) * BBYTES);
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
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, acc[i][j], 0, 0, 0);
                            accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al, accx[i][j], 0, 0, 0);
                            accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah, accx[i][j], 0, 0, 0);
                        }
                }
            } else {
            u32x4 fa[2][TM], fb[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[0][i] = *(const lds_u32x4*)((a_rel[i][dw][0] ^ aflip) + aoff);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[0][j] = *(const lds_u32x4*)(b_rel[j][0] + (t % 3) * BBYTES);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk < 3) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        fa[(kk + 1) & 1][i] = *(const lds_u32x4*)((a_rel[i][dw][kk + 1] ^ aflip) + aoff);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        fb[(kk + 1) & 1][j] = *(const lds_u32x4*)(b_rel[j][kk + 1] + (t % 3) * BBYTES);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        // weights as the first operand: the accumulator is the TRANSPOSED tile (see the epilogue)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[kk & 1][j]),
                                                                           __builtin_bit_cast(bf16x8, fa[kk & 1][i]), acc[i][j], 0, 0, 0);
            }
            }
        }
    }
    __syncthreads();

    if constexpr (X3) {
        // ---- fp32 epilogue of the split-bf16 launches: a lane owns ONE pixel (lane&31) and per register quad four
        // consecutive channels = one 16-byte LDS store into the pixel-major fp32 staging tile; rows leave as 16-byte vectors.
        float* __restrict__ OutF = (float*)g.Out;
        const float* __restrict__ AddF = (const float*)g.addend;
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
        constexpr int CPRF = BN / 4;                          // 16-byte chunks (4 channels) per tile row
        static_assert(NT % CPRF == 0, "a thread keeps one channel group over all its rows");
        float fs[4] = {0.f, 0.f, 0.f, 0.f}, fq[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (X3 == 3) {
            // a thread owns one 8-channel group over its rows: two 16-byte staging reads -> one 16-byte store per plane
            bf16_t* __restrict__ OutHi = (bf16_t*)g.Out;
            bf16_t* __restrict__ OutLo = (bf16_t*)g.Out_lo;
            const bf16_t* __restrict__ RH = (const bf16_t*)g.res_hi;
            const bf16_t* __restrict__ RL = (const bf16_t*)g.res_lo;
            constexpr int CPR8 = BN / 8;
            static_assert(NT % CPR8 == 0, "a thread keeps one channel group over all its rows");
            const int c8 = tid % CPR8, col = n0 + c8 * 8;
            const bool cok = col < g.Cn;
            float sc[8], sh[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { sc[k] = cok ? g.bnp[col + k] : 0.f; sh[k] = cok ? g.bnp[g.Cn + col + k] : 0.f; }
            for (int row = tid / CPR8; row < BM; row += NT / CPR8) {
                const int yy = ty0 + row / TW, xx = tx0 + row % TW;
                if (yy < g.H && xx < g.W && cok) {
                    const float4 va = *(const float4*)(smem + row * SPF + c8 * 32), vb = *(const float4*)(smem + row * SPF + c8 * 32 + 16);
                    const long o = (((long)img * g.H + yy) * g.W + xx) * g.Cn + col;
                    float v[8] = {va.x * sc[0] + sh[0], va.y * sc[1] + sh[1], va.z * sc[2] + sh[2], va.w * sc[3] + sh[3],
                                  vb.x * sc[4] + sh[4], vb.y * sc[5] + sh[5], vb.z * sc[6] + sh[6], vb.w * sc[7] + sh[7]};
                    if (RH) {
                        const uint4 h4 = *(const uint4*)(RH + o), l4 = *(const uint4*)(RL + o);
                        const uint32_t hw[4] = {h4.x, h4.y, h4.z, h4.w}, lw[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            v[2 * k] += __uint_as_float(hw[k] << 16) + __uint_as_float(lw[k] << 16);
                            v[2 * k + 1] += __uint_as_float(hw[k] & 0xffff0000u) + __uint_as_float(lw[k] & 0xffff0000u);
                        }
                    } else if (AddF) {
                        const float4 a0 = *(const float4*)(AddF + o), a1 = *(const float4*)(AddF + o + 4);
                        v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
                    }
                    if (g.ep_relu) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
                    }
                    uint32_t h[4], l[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        h[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
                        l[k] = pack_bf16x2(v[2 * k] - __uint_as_float(h[k] << 16), v[2 * k + 1] - __uint_as_float(h[k] & 0xffff0000u));
                    }
                    *(uint4*)(OutHi + o) = make_uint4(h[0], h[1], h[2], h[3]);
                    *(uint4*)(OutLo + o) = make_uint4(l[0], l[1], l[2], l[3]);
                    if (g.OutF) {
                        *(float4*)(g.OutF + o) = make_float4(v[0], v[1], v[2], v[3]);
                        *(float4*)(g.OutF + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    }
                }
            }
            return;
        }
        if constexpr (BNR) {
            const float* __restrict__ BnY = (const float*)g.bn_y;
            const bf16_t* __restrict__ BnM = (const bf16_t*)g.bn_out;
            const int col = n0 + ec4 * 4;
#pragma unroll
            for (int k = 0; k < ER; ++k) {
                const int row = er0 + k * ERS;
                const int yy = ty0 + row / TW, xx = tx0 + row % TW;
                if (yy < g.H && xx < g.W && col < g.Cn) {
                    const float4 v4 = *(const float4*)(smem + row * SPF + ec4 * 16);
                    const long o = (((long)img * g.H + yy) * g.W + xx) * g.Cn + col;
                    float4 y4, a4; uint2 m2;
                    if constexpr (EPRE) y4 = e_y[k]; else y4 = *(const float4*)(BnY + o);
                    if constexpr (EPRE2) { a4 = e_add[k]; m2 = e_m[k]; }
                    else {
                        a4 = AddF ? *(const float4*)(AddF + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                        m2 = BnM ? *(const uint2*)(BnM + o) : make_uint2(0u, 0u);
                    }
                    float v[4] = {v4.x + a4.x, v4.y + a4.y, v4.z + a4.z, v4.w + a4.w};
                    const float y[4] = {y4.x, y4.y, y4.z, y4.w};
                    const float m[4] = {__uint_as_float(m2.x << 16), __uint_as_float(m2.x & 0xffff0000u),
                                        __uint_as_float(m2.y << 16), __uint_as_float(m2.y & 0xffff0000u)};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // mask: the stored activation's hi plane, or bn_apply's own expression on y (no residual was added)
                        const bool dead = BnM ? !(m[i] > 0.f) : !(y[i] * e_sc[i] + e_sh[i] > 0.f);
                        v[i] = dead ? 0.f : v[i];
                        fs[i] += v[i]; fq[i] += v[i] * ((y[i] - e_mean[i]) * e_istd[i]);
                    }
                    *(float4*)(OutF + o) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        } else
        for (int id = tid; id < BM * CPRF; id += NT) {
            const int row = id / CPRF, c4 = id - row * CPRF;
            const int yy = ty0 + row / TW, xx = tx0 + row % TW, col = n0 + c4 * 4;
            if (yy < g.H && xx < g.W && col < g.Cn) {
                float4 v = *(const float4*)(smem + row * SPF + c4 * 16);
                const long o = (((long)img * g.H + yy) * g.W + xx) * g.Cn + col;
                if (AddF) { const float4 a = *(const float4*)(AddF + o); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
                *(float4*)(OutF + o) = v;
                fs[0] += v.x; fq[0] += v.x * v.x; fs[1] += v.y; fq[1] += v.y * v.y;
                fs[2] += v.z; fq[2] += v.z * v.z; fs[3] += v.w; fq[3] += v.w * v.w;
            }
        }
        __syncthreads();
        float* part_out = BNR ? g.bn_part : g.stats;
        if (part_out) {
            float* sp = (float*)smem;                          // [NT / CPRF][BN][2], over the consumed staging tile
            const int rg = tid / CPRF, cb = (tid % CPRF) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) { sp[(rg * BN + cb + k) * 2] = fs[k]; sp[(rg * BN + cb + k) * 2 + 1] = fq[k]; }
            __syncthreads();
            for (int c = tid; c < BN; c += NT) {
                const int col = n0 + c;
                if (col < g.Cn) {
                    float s2 = 0.f, q2 = 0.f;
                    for (int r = 0; r < NT / CPRF; ++r) { s2 += sp[(r * BN + c) * 2]; q2 += sp[(r * BN + c) * 2 + 1]; }
                    part_out[((long)tile_sp * g.Cn + col) * 2] = s2;
                    part_out[((long)tile_sp * g.Cn + col) * 2 + 1] = q2;
                }
            }
        }
        return;
    }

    // ---- epilogue.  The MFMAs computed the transposed tile (weights x pixels), so in the C/D layout a lane owns ONE pixel
    // (lane&31) and, per register quad, four consecutive output channels (r&3) + 8*(r>>2) + 4*(lane>>5): a quad is one packed
    // 8-byte LDS store into the pixel-major staging tile (4 ds_write_b64 per 32x32 block instead of 16 ds_write_b16), from
    // which HBM sees whole 16-byte vectors / full lines.
    bf16_t* __restrict__ Out = (bf16_t*)g.Out;
    const bf16_t* __restrict__ Add = (const bf16_t*)g.addend;
    constexpr int SPITCH = BN * 2 + 16;                   // staging row pitch (bytes): +16 spreads rows over banks
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wave_m * TM + i) * 32 + l32;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int cl = (wave_n * TN + j) * 32 + 8 * q4 + 4 * fhalf;
                uint2 w;
                w.x = pack_bf16x2(acc[i][j][q4 * 4], acc[i][j][q4 * 4 + 1]);
                w.y = pack_bf16x2(acc[i][j][q4 * 4 + 2], acc[i][j][q4 * 4 + 3]);
                *(uint2*)(smem + row * SPITCH + cl * 2) = w;
            }
        }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;                            // 16-byte chunks per tile row
    static_assert(NT % CPR == 0, "a thread keeps one channel group over all its rows");
    // Fused BN-backward reduction (data-gradient launches): the tile being stored IS the gradient arriving at
    // relu(bn(bn_y) [+ residual]); each thread owns one 8-channel group over its rows and accumulates sum(dz) and
    // sum(dz * xhat) with dz = masked gradient (mask from the stored activation bn_out, or recomputed from bn_y).
    const bf16_t* __restrict__ BnY = (const bf16_t*)g.bn_y;
    const bf16_t* __restrict__ BnOut = (const bf16_t*)g.bn_out;
    // Forward launches (g.stats): BatchNorm partial sums of the tile AS STORED (bf16-rounded), accumulated by the same
    // thread-owns-a-channel-group scheme and combined through the same LDS partial buffer.
    float bs[8], bq[8], bmean[8], bistd[8], bsc[8], bsh[8];
    if (g.stats) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { bs[k] = 0.f; bq[k] = 0.f; }
    }
    if (BnY) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int c = min(n0 + (tid % CPR) * 8 + k, g.Cn - 1);
            bs[k] = 0.f; bq[k] = 0.f;
            bsc[k] = g.bnp[c]; bsh[k] = g.bnp[g.Cn + c]; bmean[k] = g.bnp[2 * g.Cn + c]; bistd[k] = g.bnp[3 * g.Cn + c];
        }
    }
    {
        for (int id = tid; id < BM * CPR; id += NT) {
            int row = id / CPR, c8 = id - row * CPR;
            int yy = ty0 + row / TW, xx = tx0 + row % TW, col = n0 + c8 * 8;
            if (yy < g.H && xx < g.W && col < g.Cn) {
                uint4 v = *(const uint4*)(smem + row * SPITCH + c8 * 16);
                long o = (((long)img * g.H + yy) * g.W + xx) * g.Cn + col;
                if (Add) {          // residual-branch gradient: added in f32, rounded once more to bf16
                    uint4 a = *(const uint4*)(Add + o);
                    uint32_t vw[4] = {v.x, v.y, v.z, v.w}, aw[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float lo = __uint_as_float(vw[k] << 16) + __uint_as_float(aw[k] << 16);
                        float hi = __uint_as_float(vw[k] & 0xffff0000u) + __uint_as_float(aw[k] & 0xffff0000u);
                        vw[k] = pack_bf16x2(lo, hi);
                    }
                    v = make_uint4(vw[0], vw[1], vw[2], vw[3]);
                }
                *(uint4*)(Out + o) = v;
                if (g.stats) {
                    const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float lo = __uint_as_float(vw[k] << 16), hi = __uint_as_float(vw[k] & 0xffff0000u);
                        bs[2 * k] += lo; bq[2 * k] += lo * lo; bs[2 * k + 1] += hi; bq[2 * k + 1] += hi * hi;
                    }
                }
                if (BnY) {
                    const uint4 yv = *(const uint4*)(BnY + o);
                    uint4 ov = make_uint4(0, 0, 0, 0);
                    if (BnOut) ov = *(const uint4*)(BnOut + o);
                    const uint32_t vw[4] = {v.x, v.y, v.z, v.w}, yw[4] = {yv.x, yv.y, yv.z, yv.w}, ow[4] = {ov.x, ov.y, ov.z, ov.w};
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float gv = __uint_as_float((k & 1) ? (vw[k >> 1] & 0xffff0000u) : (vw[k >> 1] << 16));
                        const float yf = __uint_as_float((k & 1) ? (yw[k >> 1] & 0xffff0000u) : (yw[k >> 1] << 16));
                        const float of = __uint_as_float((k & 1) ? (ow[k >> 1] & 0xffff0000u) : (ow[k >> 1] << 16));
                        const bool dead = BnOut ? !(of > 0.f) : !(yf * bsc[k] + bsh[k] > 0.f);
                        const float dz = dead ? 0.f : gv;
                        bs[k] += dz; bq[k] += dz * ((yf - bmean[k]) * bistd[k]);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (BnY || g.stats) {
        float* part_out = BnY ? g.bn_part : g.stats;
        constexpr int PART_OFF = (BM * SPITCH + 15) / 16 * 16;
        float* sp = (float*)(smem + PART_OFF);             // [NT / CPR][BN][2]
        const int rg = tid / CPR, cb = (tid % CPR) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) { sp[(rg * BN + cb + k) * 2] = bs[k]; sp[(rg * BN + cb + k) * 2 + 1] = bq[k]; }
        __syncthreads();
        for (int c = tid; c < BN; c += NT) {
            int col = n0 + c;
            if (col < g.Cn) {
                float s = 0.f, q = 0.f;
                for (int r = 0; r < NT / CPR; ++r) { s += sp[(r * BN + c) * 2]; q += sp[(r * BN + c) * 2 + 1]; }
                part_out[((long)tile_sp * g.Cn + col) * 2] = s;
                part_out[((long)tile_sp * g.Cn + col) * 2 + 1] = q;
            }
        }
    }
}

template <int BM, int TW, int BN, int WM, int WN, int X3 = 0>
static size_t c3_lds(int nchunks) {
    constexpr int NW = WM * WN;
    constexpr int TH = BM / TW, NPIX = ((TW == 8 && BM == 128) ? TH + 4 : TH + 2) * ((TW == 8) ? TW + 3 : TW + 2), PI = (NPIX + 7) / 8, LP = (PI + NW - 1) / NW;
    size_t need = (size_t)(nchunks > 1 ? 2 : 1) * LP * NW * 1024 + 3 * BN * 128 + WM * BN * 8;
    size_t stage = ((size_t)BM * (BN * 2 + 16) + 15) / 16 * 16;      // epilogue staging tile
    stage += (size_t)(64 * NW / (BN / 8)) * BN * 8;                    // + fused BN-backward partials [NT/CPR][BN][2]
    if (X3) {                                                          // fp32 staging tile; the partials reuse it
        stage = (size_t)BM * (BN * 4 + 16);
        const size_t part = (size_t)(64 * NW / (BN / 4)) * BN * 8;
        if (part > stage) stage = part;
    }
    return need > stage ? need : stage;
}

// config choice: 0 = unsupported, 1 = <128,32,64,2,2>, 2 = <256,32,128,2,2>, 3 = <128,16,128,2,2>, 4 = <128,16,64,2,2>,
// 5 = <64,8,128,2,2>
// x3plain: a split-bf16 launch WITHOUT the fused BatchNorm-backward epilogue (forward, plain data gradient, eval-mode fold).
static int c3_config(int N, int H, int W, int C, int Cn, bool x3plain = false, bool x3 = false) {
    if (C % 64 || Cn % 8 || getenv("AB_CONV3_OFF")) return 0;
    if (const char* f = getenv("AB_C3_FORCE")) return atoi(f);      // tile-shape probes (tools/bench_conv_x3.py, tools/ab_l1_tiles.py)
    if (W >= 24) {
        // Layer 1 in bf16x3 (64 -> 64 channels on 64 x 64 maps), in-process A/B on one box (tools/ab_l1_tiles.py, round 3): the 8 x 16-pixel
        // tile of 128 pixels (4: patch 10 x 18 = 1.41 x its outputs, two workgroups per CU) runs the forward in 63.9 us and the plain data
        // gradient in 61.4 against 71.9 / 70.7 on the 256-pixel tile (7) and 78.4 / 74.7 on 4 x 32 pixels (1) -- but the data gradient WITH the
        // fused BatchNorm-backward epilogue prefers the 256-pixel tile (86.8 vs 95.2 us), so only the plain launches switch.  AB_C3_L1T16=0: off.
        static const int l1t16 = getenv("AB_C3_L1T16") ? atoi(getenv("AB_C3_L1T16")) : 1;
        if (x3plain && l1t16 && Cn <= 64 && W % 16 == 0 && H % 8 == 0) return 4;
        // 64 output channels: a 256-pixel tile (64 x 32 per wave: 1 KB of fragment reads per MFMA instead of 1.33) where that
        // still leaves every CU two workgroups -- layer 1 at B = 64: 10.66 -> 10.54 ms/step in bf16x3.  AB_C3_L1ALT=0: 128 pixels.
        // (=2: whatever the tile count -- tests)
        const int alt1 = getenv("AB_C3_L1ALT") ? atoi(getenv("AB_C3_L1ALT")) : 1;
        if (alt1 && Cn <= 64 && (alt1 == 2 || (long)N * ((H + 7) / 8) * ((W + 31) / 32) >= 512)) return 7;
        return (Cn <= 64) ? 1 : 2;
    }
    if (W >= 12) {
        // whole 16 x 16 image x 64 channels (6) instead of half an image x 128 channels (3): 114 instead of 170 KB of L2 -> LDS fills per
        // 32-channel chunk at the same 256 workgroups; in the step 9.595 -> 9.547 ms over three alternating pairs (round 3).  AB_C3_ALT16=0: off
        static const int alt = getenv("AB_C3_ALT16") ? atoi(getenv("AB_C3_ALT16")) : 1;
        if (alt && W <= 16 && H % 16 == 0 && Cn % 64 == 0) return 6;      // whole 16x16 image x 64 channels per workgroup
        return (Cn <= 64) ? 4 : 3;
    }
    if (W >= 5 && W <= 8 && H <= 8 && Cn > 64) {
        // bf16x3, 8 x 8 maps (layer 4): two images x 64 channels per workgroup (8) instead of one image x 128 channels (5).  AB_C3_STACK=0: off
        static const int stack = getenv("AB_C3_STACK") ? atoi(getenv("AB_C3_STACK")) : 1;
        if (x3 && stack && H == 8 && W == 8 && N % 2 == 0 && Cn % 64 == 0) return 8;
        return 5;      // one whole (<= 8x8) image per workgroup
    }
    return 0;
}
static void c3_geom(int cfg, int* bm, int* tw, int* bn) {
    if (cfg == 1) { *bm = 128; *tw = 32; *bn = 64; }
    else if (cfg == 2) { *bm = 256; *tw = 32; *bn = 128; }
    else if (cfg == 3) { *bm = 128; *tw = 16; *bn = 128; }
    else if (cfg == 5) { *bm = 64; *tw = 8; *bn = 128; }
    else if (cfg == 6) { *bm = 256; *tw = 16; *bn = 64; }
    else if (cfg == 7) { *bm = 256; *tw = 32; *bn = 64; }
    else if (cfg == 8) { *bm = 128; *tw = 8; *bn = 64; }
    else { *bm = 128; *tw = 16; *bn = 64; }
}

// number of BN-partial rows (spatial tiles) this kernel writes; 0 when the shape is not handled here
int conv3x3_tiles(int N, int H, int W, int C, int Cn) {
    int cfg = c3_config(N, H, W, C, Cn);
    if (!cfg) return 0;
    int bm, tw, bn; c3_geom(cfg, &bm, &tw, &bn);
    int th = bm / tw;
    return N * ((H + th - 1) / th) * ((W + tw - 1) / tw);
}

template <int BM, int TW, int BN, int WM, int WN, int FLIP, int X3 = 0>
static int c3_launch(Conv3Args& g, hipStream_t st) {
    constexpr int TH = BM / TW;
    if constexpr (TW == 8 && BM == 128) { g.N /= 2; g.H = 16; }        // two 8 x 8 images as one 16-row image: the same bytes
    g.tiles_x = (g.W + TW - 1) / TW; g.tiles_y = (g.H + TH - 1) / TH;
    int blocks = g.N * g.tiles_x * g.tiles_y * ((g.Cn + BN - 1) / BN);
    size_t lds = c3_lds<BM, TW, BN, WM, WN, X3>(g.C / (X3 ? 32 : 64));
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_kernel<BM, TW, BN, WM, WN, FLIP, X3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)c3_lds<BM, TW, BN, WM, WN, X3>(2));
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    conv3x3_kernel<BM, TW, BN, WM, WN, FLIP, X3><<<blocks, 64 * WM * WN, lds, st>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// x [N,H,W,C] -> out [N,H,W,Cn]; wt rows of length 9*C; flip = 0: forward weights OHWI, taps (kh-1, kw-1);
// flip = 1: data gradient with IHWO weights, taps (1-kh, 1-kw).  Returns AB_ESHAPE when unsupported.
int conv3x3_run(const void* x, const void* wt, void* out, int N, int H, int W, int C, int Cn, int flip,
                const void* addend, float* stats, hipStream_t st, const void* bn_y, const void* bn_out, const float* bnp,
                float* bn_part) {
    int cfg = c3_config(N, H, W, C, Cn);
    if (!cfg) return AB_ESHAPE;
    if (stats && bn_y) return AB_EINVAL;          // the two partial-sum outputs share the epilogue's accumulators
    Conv3Args g = {};
    g.X = x; g.Wt = wt; g.Out = out; g.addend = addend; g.stats = stats;
    g.N = N; g.H = H; g.W = W; g.C = C; g.Cn = Cn; g.ktot = 9 * C;
    g.flip = flip;
    g.bn_y = bn_y; g.bn_out = bn_out; g.bnp = bnp; g.bn_part = bn_part;
    static const int w8 = getenv("AB_C3_W8") ? atoi(getenv("AB_C3_W8")) : 2;     // 0: 4 waves, 1: 8 waves, 2: 8 (16 for the 256-pixel tile)
#define C3_GO(FL) \
    do { \
        if (cfg == 7) return c3_launch<256, 32, 64, 4, 2, FL>(g, st); \
        if (w8 == 2) { \
            if (cfg == 1) return c3_launch<128, 32, 64, 4, 2, FL>(g, st); \
            if (cfg == 2) return c3_launch<256, 32, 128, 4, 4, FL>(g, st); \
            if (cfg == 5) return c3_launch<64, 8, 128, 2, 4, FL>(g, st); \
        } \
        if (w8) { \
            if (cfg == 1) return c3_launch<128, 32, 64, 4, 2, FL>(g, st); \
            if (cfg == 2) return c3_launch<256, 32, 128, 4, 2, FL>(g, st); \
            if (cfg == 3) return c3_launch<128, 16, 128, 4, 2, FL>(g, st); \
            if (cfg == 5) return c3_launch<64, 8, 128, 2, 4, FL>(g, st); \
            if (cfg == 6) return c3_launch<256, 16, 64, 4, 2, FL>(g, st); \
            return c3_launch<128, 16, 64, 4, 2, FL>(g, st); \
        } \
        if (cfg == 1) return c3_launch<128, 32, 64, 2, 2, FL>(g, st); \
        if (cfg == 2) return c3_launch<256, 32, 128, 2, 2, FL>(g, st); \
        if (cfg == 3) return c3_launch<128, 16, 128, 2, 2, FL>(g, st); \
        if (cfg == 5) return c3_launch<64, 8, 128, 2, 2, FL>(g, st); \
        if (cfg == 6) return c3_launch<256, 16, 64, 4, 1, FL>(g, st); \
        return c3_launch<128, 16, 64, 2, 2, FL>(g, st); \
    } while (0)
    if (flip) C3_GO(1);
    C3_GO(0);
#undef C3_GO
}

// ---- split-bf16 ("bf16x3") launches: x / wt given as (hi, lo) bf16 planes, out / addend / stats fp32.
// Tile shapes are those of the bf16 path, always on 8 waves (the accumulators of both chains need the registers).
static int c3_tiles_of(int cfg, int N, int H, int W) {
    if (!cfg) return 0;
    if (cfg == 8) return N / 2;                    // two stacked 8 x 8 images per tile
    int bm, tw, bn; c3_geom(cfg, &bm, &tw, &bn);
    int th = bm / tw;
    return N * ((H + th - 1) / th) * ((W + tw - 1) / tw);
}

// conv3x3r.hip: 64 -> 64 channels with the weights resident in registers (plain forward / data gradient launches of layer 1)
int conv3x3r_rows(int N, int H, int W, int C, int Cn);
int conv3x3rb_rows(int N, int H, int W, int C, int Cn);      // ... and the data gradient with the fused BatchNorm-backward epilogue
int conv3x3r_run(const void* x_hi, const void* x_lo, const void* wt_hi, const void* wt_lo, float* out, int N, int H, int W, int flip,
                 float* stats, hipStream_t st, const float* bn_y = nullptr, const void* mask = nullptr, const float* addend = nullptr,
                 const float* bnp = nullptr);

// BatchNorm partial rows of the split-bf16 launches: of a forward / plain launch, and of a data gradient with the fused reduction
int conv3x3_x3_tiles(int N, int H, int W, int C, int Cn) {
    if (C % 32) return 0;
    if (int r = conv3x3r_rows(N, H, W, C, Cn)) return r;
    return c3_tiles_of(c3_config(N, H, W, (C + 63) / 64 * 64, Cn, true, true), N, H, W);
}
int conv3x3_x3_tiles_bnr(int N, int H, int W, int C, int Cn) {
    if (C % 32) return 0;
    if (int r = conv3x3rb_rows(N, H, W, C, Cn)) return r;
    return c3_tiles_of(c3_config(N, H, W, (C + 63) / 64 * 64, Cn, false, true), N, H, W);
}

int conv3x3_x3_run(const void* x_hi, const void* x_lo, const void* wt_hi, const void* wt_lo, float* out, int N, int H, int W,
                   int C, int Cn, int flip, const float* addend, float* stats, hipStream_t st, const float* bn_y,
                   const void* bn_out_hi, const float* bnp, float* bn_part, const C3EvalBn* ev) {
    if (C % 32) return AB_ESHAPE;
    if (bn_y && (!flip || stats || !bnp || !bn_part)) return AB_EINVAL;
    if (ev && (flip || stats || bn_y || !bnp || !ev->out_hi || !ev->out_lo || (ev->res_hi && (addend || !ev->res_lo)))) return AB_EINVAL;
    // conv3x3r.hip takes the TRAINING launches of layer 1 (forward with BatchNorm partials, plain data gradient).  A forward without
    // statistics is an eval-mode one: it stays on this file's kernel, whose eval-fold form (X3 = 3) must remain bit-identical to
    // conv + ab_bn_apply_x3 (tests/test_gpu_learner.py::test_eval_forward_with_folded_batchnorm_is_bit_identical)
    if (!bn_y && !ev && !addend && (stats || flip) && conv3x3r_rows(N, H, W, C, Cn))
        return conv3x3r_run(x_hi, x_lo, wt_hi, wt_lo, out, N, H, W, flip, stats, st);
    if (bn_y && conv3x3rb_rows(N, H, W, C, Cn))      // (conv3x3_x3_tiles_bnr sized bn_part by the same predicate)
        return conv3x3r_run(x_hi, x_lo, wt_hi, wt_lo, out, N, H, W, flip, bn_part, st, bn_y, bn_out_hi, addend, bnp);
    int cfg = c3_config(N, H, W, (C + 63) / 64 * 64, Cn, /*x3plain=*/!bn_y, /*x3=*/true);
    if (!cfg || Cn % 4) return AB_ESHAPE;
    const long delta = (const char*)wt_lo - (const char*)wt_hi;
    if (delta < 0 || delta >= (1L << 31)) return AB_EINVAL;       // the lo plane is addressed as a 32-bit offset from the hi plane
    Conv3Args g = {};
    g.X = x_hi; g.X_lo = x_lo; g.Wt = wt_hi; g.wlo_delta = (unsigned)delta; g.Out = out; g.addend = addend; g.stats = stats;
    g.N = N; g.H = H; g.W = W; g.C = C; g.Cn = Cn; g.ktot = 9 * C; g.flip = flip;
    g.bn_y = bn_y; g.bn_out = bn_out_hi; g.bnp = bnp; g.bn_part = bn_part;
    if (ev) { g.Out = ev->out_hi; g.Out_lo = ev->out_lo; g.OutF = ev->out_f32; g.res_hi = ev->res_hi; g.res_lo = ev->res_lo; g.ep_relu = ev->relu; }
    if (bn_y) {      // masked gradient + BatchNorm-backward partials from the epilogue
        if (cfg == 1) return c3_launch<128, 32, 64, 4, 2, 1, 2>(g, st);
        if (cfg == 2) return c3_launch<256, 32, 128, 4, 2, 1, 2>(g, st);
        if (cfg == 3) return c3_launch<128, 16, 128, 4, 2, 1, 2>(g, st);
        if (cfg == 5) return c3_launch<64, 8, 128, 2, 4, 1, 2>(g, st);
        if (cfg == 6) return c3_launch<256, 16, 64, 4, 2, 1, 2>(g, st);
        if (cfg == 7) return c3_launch<256, 32, 64, 4, 2, 1, 2>(g, st);
        if (cfg == 8) return c3_launch<128, 8, 64, 4, 2, 1, 2>(g, st);
        return c3_launch<128, 16, 64, 4, 2, 1, 2>(g, st);
    }
    if (g.Out_lo) {      // eval-mode forward with the following BatchNorm folded in (conv3x3_x3_evalbn_run fills these fields)
        if (cfg == 1) return c3_launch<128, 32, 64, 4, 2, 0, 3>(g, st);
        if (cfg == 2) return c3_launch<256, 32, 128, 4, 2, 0, 3>(g, st);
        if (cfg == 3) return c3_launch<128, 16, 128, 4, 2, 0, 3>(g, st);
        if (cfg == 5) return c3_launch<64, 8, 128, 2, 4, 0, 3>(g, st);
        if (cfg == 6) return c3_launch<256, 16, 64, 4, 2, 0, 3>(g, st);
        if (cfg == 7) return c3_launch<256, 32, 64, 4, 2, 0, 3>(g, st);
        if (cfg == 8) return c3_launch<128, 8, 64, 4, 2, 0, 3>(g, st);
        return c3_launch<128, 16, 64, 4, 2, 0, 3>(g, st);
    }
#define C3X_GO(FL) \
    do { \
        if (cfg == 1) return c3_launch<128, 32, 64, 4, 2, FL, 1>(g, st); \
        if (cfg == 2) return c3_launch<256, 32, 128, 4, 2, FL, 1>(g, st); \
        if (cfg == 3) return c3_launch<128, 16, 128, 4, 2, FL, 1>(g, st); \
        if (cfg == 5) return c3_launch<64, 8, 128, 2, 4, FL, 1>(g, st); \
        if (cfg == 6) return c3_launch<256, 16, 64, 4, 2, FL, 1>(g, st); \
        if (cfg == 7) return c3_launch<256, 32, 64, 4, 2, FL, 1>(g, st); \
        if (cfg == 8) return c3_launch<128, 8, 64, 4, 2, FL, 1>(g, st); \
        return c3_launch<128, 16, 64, 4, 2, FL, 1>(g, st); \
    } while (0)
    if (flip) C3X_GO(1);
    C3X_GO(0);
#undef C3X_GO
}
