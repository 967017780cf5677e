// 3x3 / stride 1 / pad 1 convolution, 64 -> 64 channels (layer 1 of the ResNet: anakin/models/resnet.py:85-101 at 64 x 64 maps), split-bf16,
// with the WEIGHTS RESIDENT IN REGISTERS.
//
// Why: at 64 -> 64 channels the K loop of conv3x3.hip is only two 32-channel chunks (18 steps), and every workgroup streams all 147 KB of weight
// planes through its LDS ring whatever its tile -- 2 048 workgroups x 147 KB = 300 MB of L2 -> LDS fills per launch, six times the activation
// patches -- with a barrier every 12 MFMAs per wave.  Here a wave owns SIXTEEN output channels and keeps their complete weights (9 taps x 64 input
// channels x (hi, lo) = 36 B-fragments of v_mfma_f32_16x16x32_bf16 = 144 VGPRs) for the life of a persistent workgroup; the only LDS traffic is
// the input patch (10 x 18 pixels x 64 channels x (hi, lo) = 46 KB per 128-pixel tile, LDS-DMA, double-buffered, ONE barrier per tile = 216 MFMAs
// per wave), every fragment address is a lane constant plus an immediate, and the 16 x 16 accumulator tile (lane = channel, register = pixel)
// stores straight to HBM.  Four waves cover the 64 output channels, two such sets split the tile's eight image rows.
//   FLIP = 0: forward (weights OHWI, tap t reads input (t/3 - 1, t%3 - 1));  FLIP = 1: data gradient (weights IHWO, taps mirrored).
//   stats: BatchNorm partial sums (sum, sum of squares) of the stored values, ONE row per workgroup (its tiles in launch order).
//
// BNR = 1 (round 6): the data gradient WITH the fused BatchNorm-backward epilogue of conv3x3.hip's X3 = 2 launches -- the result is the gradient
// arriving at relu(bn(bn_y) [+ residual]): it is masked (by the stored activation's hi plane `mask`, or by bn_apply's own expression on bn_y),
// optionally after the skip gradient `addend` is added, stored as dz, and (sum dz, sum dz * xhat) per channel leave as one partial row per
// workgroup.  Why here: on the LDS-ring kernel those five layer-1 launches ran 86 - 103 us against 53 - 64 for the plain gradient -- every CU
// alternated between a K loop (HBM idle) and an epilogue that read 8 - 14 bytes per element (MFMA idle), four times in lock step.  In this
// persistent kernel the epilogue operands of image row r + 1 are LDS-DMA'd (no VGPRs: the weights own the register file) into a two-slot ring
// PRIVATE to the wave that will consume them -- (16 pixels x its 16 channels) x {bn_y 1 KiB, addend 1 KiB, mask 512 B} -- while row r's 54 MFMAs
// run, so the epilogue traffic streams under the matrix work at a steady rate and needs no barrier: a wave waits on its own vmcnt only.
#include "conv3x3.h"

typedef float f32x4_t __attribute__((ext_vector_type(4)));
#ifndef C3R_PIN
#define C3R_PIN 1
#endif

struct C3rArgs {
    const void* X; const void* X_lo; const void* Wt; const void* Wt_lo;      // input planes [N, H, W, 64]; weight planes [64][9][64]
    float* Out; float* stats;                                                 // [N, H, W, 64]; [grid][64][2] or NULL
    int N, H, W, tiles_x, tiles_per_img, ntiles;
    int perm;                    // 1: fragment rows permuted for conflict-free ds_read_b128 groups (default); 0: row = pixel (AB_C3R_PERM=0)
    // BNR launches: bn_y fp32 [N, H, W, 64]; mask = hi plane (bf16) of the stored activation or NULL; addend fp32 or NULL;
    // bnp = [scale | shift | mean | istd] x 64; the partial rows go to `stats`
    const float* bn_y; const void* mask; const float* addend; const float* bnp;
};

static __device__ uint4 c3r_zero_page[2];

__device__ __forceinline__ void c3r_store(unsigned voff, float v, const void* sbase) {
    asm volatile("global_store_dword %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase));
}

// saddr-form LDS-DMA (wave-uniform 64-bit base in SGPRs + per-lane 32-bit byte offset): the epilogue operands' addresses cost one constant VGPR
// per access pattern.  16 bytes per lane (1 KiB per wave instruction), and 4 bytes per lane (256 B: the mask plane's bf16 tile); the LDS
// destination is the wave-uniform base + lane * (16 | 4).
__device__ __forceinline__ void c3r_glds16_s(unsigned voff, const void* sbase, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void c3r_glds4_s(unsigned voff, const void* sbase, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ const char* c3r_uniform(const void* p) {      // a wave-uniform pointer, spelled out for an asm SGPR operand
    const unsigned long long u = (unsigned long long)p;
    return (const char*)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(u >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)u));
}

template <int FLIP, int BNR = 0, bool HAS_ADD = false, bool HAS_MASK = false>
__global__ __launch_bounds__(512) void conv3x3r_kernel(C3rArgs g) {
    constexpr int TH = 8, TW = 16, PW = 18, PH = 10, NPIX = PH * PW;          // tile 8 x 16 pixels, patch 10 x 18
    constexpr int PI = (NPIX + 7) / 8, CHUNK_BYTES = 24 * 1024, TILE_BYTES = 2 * CHUNK_BYTES;      // 23 (-> 24) 1-KiB instructions per 32-channel chunk
    constexpr int LPW = 2 * 24 / 8;                                           // fill instructions per wave and tile (both chunks)
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = wave & 3, ph = wave >> 2;                                   // output-channel quarter, half of the tile's rows
    const int l16 = lane & 15, kq = lane >> 4;
    const unsigned lds0 = lds_addr_of(smem);
    const bf16_t* __restrict__ X = (const bf16_t*)g.X;
    const bf16_t* __restrict__ Xlo = (const bf16_t*)g.X_lo;
    const bf16_t* zp = (const bf16_t*)c3r_zero_page;

    // ---- this wave's weights: B fragments (k32 x 16 channels) of every (tap, chunk, plane): lane = (channel l16, k-quarter kq)
    bf16x8 wh[18], wl[18];
    float b_sc = 0.f, b_sh = 0.f, b_mean = 0.f, b_istd = 0.f;      // BNR: this lane's channel of the BatchNorm record (ordinary loads: they must
    if constexpr (BNR) {                                            // retire at the compiler-visible wait below, or hipcc drains vmcnt at their use)
        const int c = q * 16 + l16;
        b_sc = g.bnp[c]; b_sh = g.bnp[64 + c]; b_mean = g.bnp[128 + c]; b_istd = g.bnp[192 + c];
    }
    {
        const bf16_t* ph_ = (const bf16_t*)g.Wt + (size_t)(q * 16 + l16) * 576 + kq * 8;
        const bf16_t* pl_ = (const bf16_t*)g.Wt_lo + (size_t)(q * 16 + l16) * 576 + kq * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                wh[t * 2 + c] = *(const bf16x8*)(ph_ + t * 64 + c * 32);
                wl[t * 2 + c] = *(const bf16x8*)(pl_ + t * 64 + c * 32);
            }
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), visible to hipcc's wait-count pass (gemm_rw.hip)
    }

    // ---- patch fill assignment: instruction ii = wave * LPW + j of 48: chunk = ii / 24 (= wave / 4: wave-uniform), pixels (ii % 24) * 8 .. + 7.
    // Everything about a lane's six fills follows from its first patch pixel f_pp0 (pixel of fill j = f_pp0 + 8 j) and is recomputed per tile
    // (a few dozen VALU operations against 216 MFMAs): held per fill it was ~36 loop-invariant VGPRs beside the 144 of the weights, and the
    // fused-epilogue variants (BNR) spilled weights to scratch.  The asm in issue_tile keeps the compiler from hoisting it back out.
    const int f_pp0 = (wave & 3) * LPW * 8 + (lane >> 3), f_chunk = wave >> 2;
    auto issue_tile = [&](int tile, int slot) {
        const int img = tile / g.tiles_per_img, rem = tile - img * g.tiles_per_img;
        const int ty0 = (rem / g.tiles_x) * TH, tx0 = (rem % g.tiles_x) * TW;
        int pp0 = f_pp0;
        asm volatile("" : "+v"(pp0));
#pragma unroll
        for (int j = 0; j < LPW; ++j) {
            const int pp = pp0 + 8 * j, py = pp / PW, px = pp - py * PW;
            const int c = (lane & 7) ^ ((px >> 1) & 7);
            const int y = ty0 + py - 1, x = tx0 + px - 1;
            const bool ok = pp < NPIX && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
            const bf16_t* src = ok ? ((c & 4) ? Xlo : X) + ((((size_t)img * g.H + y) * g.W + x) * 64 + (unsigned)(f_chunk * 32 + (c & 3) * 8)) : zp;
            glds16(src, __builtin_amdgcn_readfirstlane(lds0 + slot * TILE_BYTES + (wave * LPW + j) * 1024));
        }
    };

    // ---- fragment addresses: output pixel (row r of the tile, column l16), tap column dw: patch pixel (r + dh, l16 + dw), 16-byte slot
    // (plane * 4 + kq) ^ (((l16 + dw) >> 1) & 7); rows and chunks are immediates
    // Which pixel of the row an A-fragment row stands for is free (the MFMA only pairs row i of A with row i of the result): rows {0-3, 12-15}
    // take the EVEN pixels and rows {4-11} the odd ones.  ds_read_b128 is serviced in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...
    // (MI355X_MICROARCH.md, LDS table): with row = pixel such a group read pixels 0-3, 12-15 of k-quarter kq and pixels 4-11 of kq + 1, whose
    // swizzled 16-byte slots collide for the shifted taps (SQ_LDS_BANK_CONFLICT 6.3 M of 15.7 M LDS cycles per launch); now one k-quarter covers
    // one 128-byte half of every pixel pair for every tap shift: conflict-free.
    const int a_pix = g.perm ? 2 * (l16 & 3) + ((l16 >> 2) == 0 ? 0 : (l16 >> 2) == 1 ? 1 : (l16 >> 2) == 2 ? 9 : 8) : l16;
    unsigned a_rel[3][2];
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
        const int px = a_pix + dw;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) a_rel[dw][pl] = lds0 + px * 128 + (((pl * 4 + kq) ^ ((px >> 1) & 7)) << 4);
    }
    // result rows 4 kq + i of this lane = pixels 2 i + {0, 1, 9, 8}[kq] of the image row (the permutation above), this lane's channel
    const unsigned o_voff = (unsigned)(((g.perm ? (kq == 0 ? 0 : kq == 1 ? 1 : kq == 2 ? 9 : 8) : kq * 4) * 64 + q * 16 + l16) * 4);
    const int o_step = g.perm ? 512 : 256;

    // ---- BNR: the wave's private ring of epilogue operands, two image rows deep.  One row = 16 pixels x this wave's 16 channels:
    //   [bn_y fp32: pixel p at p * 64 B][addend fp32, same][mask bf16: pixel p at p * 32 B];  KE LDS-DMA instructions per row
    constexpr int KE = BNR ? 1 + (HAS_ADD ? 1 : 0) + (HAS_MASK ? 2 : 0) : 0;
    constexpr int EO_ADD = 1024, EO_MASK = EO_ADD + (HAS_ADD ? 1024 : 0), ESLOT = EO_MASK + (HAS_MASK ? 512 : 0);
    const unsigned e_lds = lds0 + 2 * TILE_BYTES + wave * (2 * ESLOT);
    // per-lane byte offsets inside a row of 16 pixels: fp32 tensors (16-byte lanes: pixel lane / 4, channels q * 16 + (lane % 4) * 4 .. + 3),
    // mask plane (4-byte lanes: pixel lane / 8, channels q * 16 + (lane % 8) * 2, + 1; the second instruction starts 8 pixels on)
    const unsigned e_v16 = (unsigned)(((lane >> 2) * 64 + q * 16 + (lane & 3) * 4) * 4);
    const unsigned e_v4 = (unsigned)(((lane >> 3) * 64 + q * 16 + (lane & 7) * 2) * 2);
    auto issue_eop = [&](int tile_, int rb_, int eslot) {      // operands of image row ph * 4 + rb_ of tile_ -> ring slot eslot
        if constexpr (BNR) {
            const int img = tile_ / g.tiles_per_img, rem = tile_ - img * g.tiles_per_img;
            const int ty0 = (rem / g.tiles_x) * TH, tx0 = (rem % g.tiles_x) * TW;
            const size_t pix = ((size_t)img * g.H + ty0 + ph * 4 + rb_) * g.W + tx0;
            const unsigned dst = __builtin_amdgcn_readfirstlane(e_lds + eslot * ESLOT);
            c3r_glds16_s(e_v16, c3r_uniform(g.bn_y + pix * 64), dst);
            if constexpr (HAS_ADD) c3r_glds16_s(e_v16, c3r_uniform(g.addend + pix * 64), dst + EO_ADD);
            if constexpr (HAS_MASK) {
                const char* m = c3r_uniform((const bf16_t*)g.mask + pix * 64);
                c3r_glds4_s(e_v4, m, dst + EO_MASK);
                c3r_glds4_s(e_v4, m + 8 * 64 * 2, dst + EO_MASK + 256);
            }
        }
    };
    // this lane's four result pixels of a row (register i): 2 i + {0, 1, 9, 8}[kq] (perm) or 4 kq + i
    const int e_p0 = g.perm ? (kq == 0 ? 0 : kq == 1 ? 1 : kq == 2 ? 9 : 8) : kq * 4, e_pstep = g.perm ? 2 : 1;

    float s_sum = 0.f, s_sq = 0.f;
    int tile = blockIdx.x, slot = 0;
    if (tile < g.ntiles) { issue_tile(tile, 0); issue_eop(tile, 0, 0); }
    bool first = true;
    for (; tile < g.ntiles; tile += gridDim.x) {
        // BNR: this tile's fills are older than every epilogue-operand load the previous tile already waited for (loads return in order);
        // at most the KE loads of this tile's first row are still in flight behind them
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (BNR) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KE) : "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // behind this tile's fills: the 16 stores of the previous tile
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        first = false;
        const int next = tile + gridDim.x;
        if (next < g.ntiles) issue_tile(next, slot ^ 1);
        const int img = tile / g.tiles_per_img, rem = tile - img * g.tiles_per_img;
        const int ty0 = (rem / g.tiles_x) * TH, tx0 = (rem % g.tiles_x) * TW;
        const unsigned sb = slot * TILE_BYTES;
        const char* obase = (const char*)g.Out + ((((size_t)img * g.H + ty0 + ph * 4) * g.W + tx0) * 64) * 4;
        // (BNR: NOT unrolled -- unrolled, the scheduler overlapped neighbouring rows' address arithmetic and epilogues and the register
        //  allocator answered with weight spills to scratch, whose loads then sat inside the K loop behind vmcnt(0) waits)
#pragma unroll(BNR ? 1 : 4)
        for (int rb = 0; rb < 4; ++rb) {                               // four image rows of 16 pixels each
            const int r = ph * 4 + rb;
            if constexpr (BNR) {                                        // the NEXT row's epilogue operands stream in under this row's MFMAs
                if (rb < 3) issue_eop(tile, rb + 1, (rb + 1) & 1);
                else if (next < g.ntiles) issue_eop(next, 0, 0);
            }
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f}, accx = {0.f, 0.f, 0.f, 0.f};
            // 18 (tap, chunk) steps; the fragment reads run one step ahead of the MFMAs that consume them (pinned: gemm_rw.hip's grw_block).
            // (Round 6 knock-out, timing only: reading the fragments on every second / third step only -- what sharing an input row's fragment
            //  between the three vertical taps of neighbouring output rows could save at best -- 65.1 -> 63.2 / 59.4 us: not worth the rebuild.)
            u32x4 fh[2], fl[2];
            auto rd = [&](int s, int b) {
                const int t = s >> 1, c = s & 1, t3 = t / 3, tr = t % 3;
                const int dh = FLIP ? 2 - t3 : t3, dw = FLIP ? 2 - tr : tr;
                const unsigned off = sb + c * CHUNK_BYTES + (r + dh) * PW * 128;
                fh[b] = *(const lds_u32x4*)(a_rel[dw][0] + off);
                fl[b] = *(const lds_u32x4*)(a_rel[dw][1] + off);
            };
            rd(0, 0);
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                if (s + 1 < 18) rd(s + 1, (s + 1) & 1);
                if constexpr (C3R_PIN) __builtin_amdgcn_sched_barrier(0);
                const bf16x8 ah = __builtin_bit_cast(bf16x8, fh[s & 1]), al = __builtin_bit_cast(bf16x8, fl[s & 1]);
                accx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl[s], accx, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh[s], acc, 0, 0, 0);
                accx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh[s], accx, 0, 0, 0);
                if constexpr (C3R_PIN) __builtin_amdgcn_sched_barrier(0);
            }
            const char* orow_v = obase + (size_t)rb * g.W * 256;
            const unsigned long long orow_u = (unsigned long long)orow_v;      // (wave-uniform: spelled out for the asm's SGPR operand)
            const char* orow = (const char*)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(orow_u >> 32)) << 32) |
                                             (unsigned)__builtin_amdgcn_readfirstlane((unsigned)orow_u));
            if constexpr (BNR) {
                // this row's operands: every load issued after them may still fly -- the next row's KE and, on a tile's first row, the
                // next tile's LPW patch fills.  (Stores are not counted in: should they retire out of order with the loads, a smaller count
                // only waits longer.)
                const bool more = next < g.ntiles;
                if (rb == 0) { if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW + KE) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KE) : "memory"); }
                else if (rb < 3 || more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KE) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned eb = e_lds + (rb & 1) * ESLOT;
                float ey[4], ea[4]; unsigned em[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int p = e_p0 + i * e_pstep;
                    ey[i] = *(const __attribute__((address_space(3))) float*)(eb + p * 64 + l16 * 4);
                    if constexpr (HAS_ADD) ea[i] = *(const __attribute__((address_space(3))) float*)(eb + EO_ADD + p * 64 + l16 * 4);
                    if constexpr (HAS_MASK) em[i] = *(const __attribute__((address_space(3))) unsigned short*)(eb + EO_MASK + p * 32 + l16 * 2);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = acc[i] + accx[i];
                    if constexpr (HAS_ADD) v += ea[i];
                    // mask: the stored activation's hi plane, or bn_apply's own expression on y (no residual was added)
                    bool dead;
                    if constexpr (HAS_MASK) dead = !(__uint_as_float(em[i] << 16) > 0.f);
                    else dead = !(ey[i] * b_sc + b_sh > 0.f);
                    v = dead ? 0.f : v;
                    s_sum += v; s_sq += v * ((ey[i] - b_mean) * b_istd);
                    c3r_store(o_voff, v, orow + i * o_step);
                }
            } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {                               // register i: pixel 2 i + (0 | 1 | 9 | 8) of the row
                const float v = acc[i] + accx[i];
                s_sum += v; s_sq += v * v;
                c3r_store(o_voff, v, orow + i * o_step);
            }
            }
        }
        slot ^= 1;
    }
    // ---- BatchNorm partials: a lane holds (sum, sumsq) of its channel over its pixel quarter (kq) and rows (ph); the eight combinations meet in LDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (g.stats) {
        float* sp = (float*)smem;                                        // [2 ph][4 kq][64 ch][2]
        sp[(((ph * 4 + kq) * 64) + q * 16 + l16) * 2] = s_sum;
        sp[(((ph * 4 + kq) * 64) + q * 16 + l16) * 2 + 1] = s_sq;
        __syncthreads();
        if (tid < 64) {
            float s = 0.f, qq = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) { s += sp[(k * 64 + tid) * 2]; qq += sp[(k * 64 + tid) * 2 + 1]; }
            g.stats[((size_t)blockIdx.x * 64 + tid) * 2] = s;
            g.stats[((size_t)blockIdx.x * 64 + tid) * 2 + 1] = qq;
        }
    }
}

static int c3r_ntiles(int N, int H, int W) { return (H % 8 || W % 16) ? 0 : N * (H / 8) * (W / 16); }
static int c3r_grid(int ntiles) {
    static const int wgs = getenv("AB_C3R_WGS") ? atoi(getenv("AB_C3R_WGS")) : 256;
    return ntiles < wgs ? ntiles : wgs;
}
// BatchNorm partial rows (= workgroups) of a launch; 0: shape not taken (64 -> 64 channels, H % 8 == 0, W % 16 == 0, at least 64 tiles)
int conv3x3r_rows(int N, int H, int W, int C, int Cn) {
    // (AB_C3_L1T16=0 asks for the round-2 tiles of conv3x3.hip on layer 1's plain launches: this kernel stands in the 8 x 16 tile's place)
    static const int off = (getenv("AB_C3R_OFF") ? atoi(getenv("AB_C3R_OFF")) : 0) || (getenv("AB_C3_L1T16") && !atoi(getenv("AB_C3_L1T16")));
    if (off || C != 64 || Cn != 64) return 0;
    const int nt = c3r_ntiles(N, H, W);
    static const int min_tiles = getenv("AB_C3R_MIN") ? atoi(getenv("AB_C3R_MIN")) : 64;      // (2 images of 64 x 64: what the parity tests run)
    return nt >= min_tiles ? c3r_grid(nt) : 0;
}
// BatchNorm-backward partial rows of a BNR launch (= workgroups); 0: this kernel does not take the fused gradient of that shape (AB_C3RB=0: never)
int conv3x3rb_rows(int N, int H, int W, int C, int Cn) {
    static const int off = getenv("AB_C3RB") ? !atoi(getenv("AB_C3RB")) : 0;
    return off ? 0 : conv3x3r_rows(N, H, W, C, Cn);
}
int conv3x3r_run(const void* x_hi, const void* x_lo, const void* wt_hi, const void* wt_lo, float* out, int N, int H, int W, int flip,
                 float* stats, hipStream_t st, const float* bn_y, const void* mask, const float* addend, const float* bnp) {
    if (!conv3x3r_rows(N, H, W, 64, 64)) return AB_ESHAPE;
    if (bn_y && (!flip || !stats || !bnp || !conv3x3rb_rows(N, H, W, 64, 64))) return AB_EINVAL;
    if (!bn_y && (mask || addend)) return AB_EINVAL;
    C3rArgs g = {};
    g.X = x_hi; g.X_lo = x_lo; g.Wt = wt_hi; g.Wt_lo = wt_lo; g.Out = out; g.stats = stats;
    g.bn_y = bn_y; g.mask = mask; g.addend = addend; g.bnp = bnp;
    g.N = N; g.H = H; g.W = W; g.tiles_x = W / 16; g.tiles_per_img = (H / 8) * (W / 16); g.ntiles = c3r_ntiles(N, H, W);
    static const int perm = getenv("AB_C3R_PERM") ? atoi(getenv("AB_C3R_PERM")) : 1;
    g.perm = perm;
    const int lds = 2 * 2 * 24 * 1024;
    const int lds_bnr = lds + 8 * 2 * 2560;          // + the eight waves' two-row operand rings (bn_y + addend + mask)
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3r_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3r_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3r_kernel<1, 1, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bnr);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3r_kernel<1, 1, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bnr);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3r_kernel<1, 1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bnr);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3r_kernel<1, 1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bnr);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int grid = c3r_grid(g.ntiles);
    if (bn_y) {
        if (addend && mask) conv3x3r_kernel<1, 1, true, true><<<grid, 512, lds_bnr, st>>>(g);
        else if (addend) conv3x3r_kernel<1, 1, true, false><<<grid, 512, lds_bnr, st>>>(g);
        else if (mask) conv3x3r_kernel<1, 1, false, true><<<grid, 512, lds_bnr, st>>>(g);
        else conv3x3r_kernel<1, 1, false, false><<<grid, 512, lds_bnr, st>>>(g);
    }
    else if (flip) conv3x3r_kernel<1><<<grid, 512, lds, st>>>(g);
    else conv3x3r_kernel<0><<<grid, 512, lds, st>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
