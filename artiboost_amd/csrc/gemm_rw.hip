// Plain NT GEMM on split-bf16 planes with the WEIGHTS RESIDENT IN REGISTERS: the 1x1 / stride 1 convolutions with Cin <= 256
// -- on the benchmark path the final layer of IntegralDeconvHead (anakin/models/simplebaseline.py:95-101,173-175: nn.Conv2d(256,
// NCLASSES * DEPTH, kernel_size=1)), M = 64 * 32 * 32 pixels, N = 22 * 32 logits channels, K = 256.
//
// Why not the implicit GEMM of conv_gemm2.hip: that kernel streams BOTH operands through LDS one 32-channel K step per barrier and
// rebuilds tap addresses every step (7 SALU + 10 VALU instructions per MFMA, round-4 counters); for a 1x1 convolution every workgroup
// re-streams the same 720 KB of weights.  Here
//   * a wave owns 32 output channels and keeps their WHOLE weight rows (K <= 256: 16 k-slices x (hi, lo) x 4 registers = 128 VGPRs) for
//     a whole run of pixel tiles; a workgroup of 8 waves covers a GROUP of 256 channels, G = ceil(N / 256) groups cover N;
//   * workgroups are persistent, one per CU.  The G workgroups that work the same pixel tiles for the G channel groups form a TUPLE on one
//     XCD (block ids equal mod 8) and walk the same tile sequence, so a tile crosses the fabric once and the other G - 1 reads hit that
//     XCD's L2 (measured with contiguous unit runs instead: the A planes crossed G times, 201 MB, and the launch was bound by exactly that);
//   * the only LDS traffic is the activation tile: 64 pixels x K x (hi, lo) = 64 KB per unit, LDS-DMA (global_load_lds_dwordx4, full
//     128-byte lines, swizzle on the source side), ONE barrier per unit (= 96 MFMAs per wave), the next unit's tile in flight meanwhile;
//   * the MFMA runs pixels x channels (A = pixels from LDS, B = weights from registers), so in the C/D layout a LANE owns a CHANNEL and
//     the registers walk the pixels: the epilogue stores straight from the accumulators, each store instruction two full 128-byte lines
//     (no LDS staging tile), and for the soft-argmax head a lane owns ONE DEPTH BIN of the wave's class;
//   * the epilogue of a 32-pixel block is written BEHIND the MFMAs of the next block in program order (the second block's behind the next
//     unit's barrier), so that its VALU work and stores issue in the shadow of those MFMAs instead of leaving the matrix pipe idle;
//   * SAM: the stage-1 statistics of the 3-D soft-argmax (simplebaseline.py:183-189 + 43-71: max, sum exp, first moments per class and
//     64-pixel tile -- the `part` rows ab_softargmax3d's second stage merges) leave from the same epilogue: the 184 MB logits are not read
//     back by a statistics pass.  Wave reductions are DPP (no LDS round trips beside the fragment reads), in a fixed order.
// All VMEM operations inside the unit loop are inline asm (fills, stores), so the counted s_waitcnt vmcnt(N) below are exact.
#include "conv3x3.h"

#ifndef GRW_PIN
#define GRW_PIN 1          // fragment reads pinned one k-slice ahead with sched_barrier(0) (0: hipcc's own placement)
#endif
#ifndef GRW_STIL
#define GRW_STIL 1         // output stores interleaved with the MFMAs of the next block (0: all 16 behind the block)
#endif
#ifndef GRW_ABL
#define GRW_ABL 0          // tools/probe_grw.hip: knock-out builds (1 no stores, 2 no MFMAs, 4 no fragment reads, 8 no fills)
#endif

struct GemmRwArgs {
    const void* A_hi; const void* A_lo;       // [M][K] bf16 planes (K contiguous)
    const void* W_hi; const void* W_lo;       // [N][K]
    const float* bias; float* Out;            // [N] or NULL; [M][N] fp32
    int M, N;
    int tiles, groups;                        // M / 64; ceil(N / 256)
    // SAM (fused soft-argmax stage 1): part[((b * ntile + tile) * C + c) * 8 + {m, s, su, sv, sd}]
    float* sam_part; int C, D, Wimg, npix, ntile; float invW, invH, invD;
    unsigned long long* dbg;                  // tools/probe_grw.hip: per wave {shader cycles, 100 MHz ticks} of its life (NULL: off)
};

__device__ __forceinline__ void grw_fill(unsigned voff, const void* sbase, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
// (no memory clobber: nothing in the kernel reads what it stores; asm volatile keeps its order among the fills / waits / stores)
__device__ __forceinline__ void grw_store(unsigned voff, float v, const void* sbase) {
    asm volatile("global_store_dword %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase));
}

// wave-wide reductions on DPP, result in every lane (read back from lane 63): quad swaps, row mirrors, then the row broadcasts of gfx9
template <bool MAX> __device__ __forceinline__ float grw_wave_reduce(float v) {
    const float ident = MAX ? -3.0e38f : 0.f;
#define GRW_DPP(ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident), __builtin_bit_cast(int, v), ctrl, rmask, 0xF, false))
#define GRW_OP(t) v = MAX ? fmaxf(v, t) : v + t
    { const float t = GRW_DPP(0xB1, 0xF); GRW_OP(t); }      // quad_perm [1,0,3,2]
    { const float t = GRW_DPP(0x4E, 0xF); GRW_OP(t); }      // quad_perm [2,3,0,1]
    { const float t = GRW_DPP(0x141, 0xF); GRW_OP(t); }     // row_half_mirror
    { const float t = GRW_DPP(0x140, 0xF); GRW_OP(t); }     // row_mirror: every lane holds its row's (16 lanes) result
    { const float t = GRW_DPP(0x142, 0xA); GRW_OP(t); }     // row_bcast15 into rows 1, 3
    { const float t = GRW_DPP(0x143, 0xC); GRW_OP(t); }     // row_bcast31 into rows 2, 3: lane 63 holds the wave's result
#undef GRW_DPP
#undef GRW_OP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}


// One 32-pixel block x this wave's 32 channels over the whole K: 3 * KS * 4 MFMAs.  The fragment reads run ONE k-slice ahead of the MFMAs
// that consume them (hipcc on its own issues a slice's ds_reads right in front of its first MFMA and the wave sits out the LDS latency
// sixteen times per block: SQ_WAIT_ANY 0.41 of the wave cycles), pinned with sched_barrier(0); ST: behind the MFMAs of every slice, one
// (16 / slices) of the 16 output-row stores of the block computed BEFORE this one -- VMEM issue in the shadow of the matrix pipe.
template <int KS, bool ST>
__device__ __forceinline__ void grw_block(const bf16x8 (&wh)[KS * 4], const bf16x8 (&wl)[KS * 4], const unsigned (&a_rel)[4], unsigned sboff, float bias_l,
                                          float (&vout)[16], const float (&dv)[16], unsigned o_voff, const char* drow, unsigned rowbytes) {
    constexpr int NK = KS * 4;
    f32x16 acc, accx;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accx[r] = 0.f; }
    u32x4 fh[2], fl[2];
    auto rd = [&](int k, int b) {
        if constexpr (GRW_ABL & 4) { fh[b] = u32x4{a_rel[k & 3], sboff, 0u, 0u}; fl[b] = fh[b]; }
        else {
            fh[b] = *(const lds_u32x4*)(a_rel[k & 3] + sboff + (k >> 2) * 16384);
            fl[b] = *(const lds_u32x4*)(a_rel[k & 3] + sboff + (k >> 2) * 16384 + 8192);
        }
    };
    rd(0, 0);
    int sr = 0;                                        // next row register of dv to store
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        if (k + 1 < NK) rd(k + 1, (k + 1) & 1);
        if constexpr (GRW_PIN) __builtin_amdgcn_sched_barrier(0);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, fh[k & 1]), al = __builtin_bit_cast(bf16x8, fl[k & 1]);
        if constexpr (GRW_ABL & 2) { acc[k & 3] += __uint_as_float(fh[k & 1].x ^ fl[k & 1].y); accx[k & 3] += (float)wh[k][0] + (float)wl[k][1]; }
        else {
            accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl[k], accx, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh[k], acc, 0, 0, 0);
            accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh[k], accx, 0, 0, 0);
        }
        if constexpr (ST) {
            const int upto = GRW_STIL ? (k + 1) * 16 / NK : (k + 1 == NK ? 16 : 0);
#pragma unroll
            for (; sr < upto; ++sr) {
                if constexpr (GRW_ABL & 1) { if (dv[sr] == 12345.678f) grw_store(o_voff, dv[sr], drow); }
                else grw_store(o_voff, dv[sr], drow);
                drow += (sr & 3) == 3 ? 5 * rowbytes : rowbytes;      // rows (r & 3) + 8 (r >> 2): +1, +1, +1, +5
            }
        }
        if constexpr (GRW_PIN) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) vout[r] = (acc[r] + accx[r]) + bias_l;
}

// KS = K / 64 (1..4).  LDS: two tile slots of KS * 16 KB -- [kc][plane][64 rows x 128 B], row r's 16-byte slot c at c ^ ((r >> 1) & 7) --
// then (SAM) three 512-byte tables of a tile's pixel coordinates (three: the second block of unit u - 1 is still read while unit u + 1's is written).
template <int KS, bool SAM>
__global__ __launch_bounds__(512) void gemm_rw_kernel(GemmRwArgs g) {
    constexpr int K = KS * 64, NK16 = KS * 4;
    constexpr int TILE_BYTES = KS * 16384;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, half = lane >> 5;
    const unsigned lds0 = lds_addr_of(smem);
    const int N = g.N, nblk = N >> 5;                   // 32-channel blocks
    const unsigned rowbytes = (unsigned)N * 4u;

    // ---- which units: XCD x (block id mod 8) owns the tiles [x T / 8, (x + 1) T / 8); its P workgroups form Q = P / G tuples of G (one per
    // channel group) that take U = ceil(Tx / Q) tiles each, in step
    int run_g0, t0, t1;
    {
        const int G = g.groups, P = (int)gridDim.x >> 3, x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
        const int tx0 = (int)((long)x * g.tiles / 8), Tx = (int)((long)(x + 1) * g.tiles / 8) - tx0;
        // (the P % G workgroups left over stay idle: as "spares" walking the remaining tiles through all G groups they paid a drain, a weight
        //  reload and an exposed first fill per group -- 141 k cycles for their 12 units where a tuple member takes 108 k -- and set the launch time)
        const int Q = P / G;
        const int U = (Tx + Q - 1) / Q;
        const int q = j / G;
        run_g0 = j - q * G; t0 = q < Q ? min(q * U, Tx) : Tx; t1 = min(t0 + U, Tx);
        t0 += tx0; t1 += tx0;
    }
    if (t1 <= t0) return;
    const unsigned long long dbg_c0 = g.dbg ? __builtin_amdgcn_s_memtime() : 0, dbg_r0 = g.dbg ? __builtin_amdgcn_s_memrealtime() : 0;

    // ---- fill assignment: per (kc, plane) one 1-KiB instruction per wave = rows 8 * wave .. + 7 of the tile
    const int frow = 8 * wave + (lane >> 3);
    const unsigned f_voff = (unsigned)(frow * (K * 2) + (((lane & 7) ^ ((frow >> 1) & 7)) << 4));
    auto issue_tile = [&](int tile, int slot) {
        const size_t row0 = (size_t)tile * 64 * (K * 2);
        const char* ah = (const char*)g.A_hi + row0;
        const char* al = (const char*)g.A_lo + row0;
        const unsigned dst = lds0 + slot * TILE_BYTES + wave * 1024;
#pragma unroll
        for (int kc = 0; kc < KS; ++kc) {
            if constexpr (GRW_ABL & 8) break;
            grw_fill(f_voff, ah + kc * 128, __builtin_amdgcn_readfirstlane(dst + kc * 16384));
            grw_fill(f_voff, al + kc * 128, __builtin_amdgcn_readfirstlane(dst + kc * 16384 + 8192));
        }
    };

    // ---- fragment read addresses: row = pb * 32 + l32, k-slice kk of stage kc -> 16-byte slot (2 kk + half) ^ ((row >> 1) & 7)
    unsigned a_rel[4];
    {
        const unsigned base = (unsigned)(l32 * 128 + ((half ^ ((l32 >> 1) & 7)) << 4));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a_rel[kk] = lds0 + (base ^ (unsigned)(kk << 5));
    }

    // ---- weights of this wave's 32 channels, both planes, whole K: registers
    bf16x8 wh[NK16], wl[NK16];
    int n0 = 0;
    bool active = false;
    float bias_l = 0.f;
    unsigned o_voff = 0;
    auto load_weights = [&](int grp) {
        const int cb = grp * 8 + wave;
        active = cb < nblk;
        n0 = (active ? cb : nblk - 1) * 32;
        const bf16_t* ph = (const bf16_t*)g.W_hi + (size_t)(n0 + l32) * K + half * 8;
        const bf16_t* pl = (const bf16_t*)g.W_lo + (size_t)(n0 + l32) * K + half * 8;
#pragma unroll
        for (int k = 0; k < NK16; ++k) {
            wh[k] = *(const bf16x8*)(ph + k * 16);
            wl[k] = *(const bf16x8*)(pl + k * 16);
        }
        bias_l = g.bias ? g.bias[n0 + l32] : 0.f;
        o_voff = (unsigned)(((size_t)(4 * half) * N + n0 + l32) * 4);
        // the BUILTIN wait (not asm): hipcc's wait-count pass sees it and knows these loads have landed -- otherwise it guards the first
        // use of every weight register inside the loop with its own vmcnt(15..0), which drains the fills issued there
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    };

    float* s_cu = (float*)(smem + 2 * TILE_BYTES);      // [3 slots][cu 64 | cv 64]
    const bool dv = l32 < g.D;                            // SAM: depth bins D .. 31 of a class are padding (zero weights): not part of the softmax

    // one 32-pixel block's 16 output rows: lane = channel n0 + l32, register r = pixel (r & 3) + 8 (r >> 2) [+ 4 half in o_voff]
    auto emit = [&](const float* v, const char* orow) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (GRW_ABL & 1) { if (v[r] == 12345.678f) grw_store(o_voff, v[r], orow); }
            else grw_store(o_voff, v[r], orow);
            orow += (r & 3) == 3 ? 5 * rowbytes : rowbytes;      // +1, +1, +1, +5 rows
        }
    };
    // SAM: the block's softmax statistics for this wave's class: max over the block (uniform), per-lane sums of exp and of exp * (u, v)
    auto sam_block = [&](const float* v, const float* tcu, int pb, float& mo, float& so, float& uo, float& vo) {
        float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
#pragma unroll
        for (int r = 4; r < 16; r += 4) mx = fmaxf(mx, fmaxf(fmaxf(v[r], v[r + 1]), fmaxf(v[r + 2], v[r + 3])));
        mx = grw_wave_reduce<true>(dv ? mx : -3.0e38f);
        const float mxl = dv ? mx : 3.0e38f;              // padding lanes: exp(v - 3e38) = 0
        float s = 0.f, su = 0.f, sv = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 cu4 = *(const float4*)(tcu + pb * 32 + 8 * q + 4 * half);
            const float4 cv4 = *(const float4*)(tcu + 64 + pb * 32 + 8 * q + 4 * half);
            const float cu[4] = {cu4.x, cu4.y, cu4.z, cu4.w}, cv[4] = {cv4.x, cv4.y, cv4.z, cv4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float e = __expf(v[q * 4 + i] - mxl);
                s += e; su = __builtin_fmaf(e, cu[i], su); sv = __builtin_fmaf(e, cv[i], sv);
            }
        }
        mo = mx; so = s; uo = su; vo = sv;
    };
    // ... the two blocks merged (uniform maxima), the 64 lanes (depth bins x pixel halves) summed, the row of `part` written
    auto sam_finish = [&](const float* m2, const float* s2, const float* u2, const float* v2, float* po) {
        const float m = fmaxf(m2[0], m2[1]);
        const float f0 = __expf(m2[0] - m), f1 = __expf(m2[1] - m);
        float s = __builtin_fmaf(s2[0], f0, s2[1] * f1), su = __builtin_fmaf(u2[0], f0, u2[1] * f1), sv = __builtin_fmaf(v2[0], f0, v2[1] * f1);
        float sd = s * (l32 * g.invD);
        s = grw_wave_reduce<false>(s); su = grw_wave_reduce<false>(su); sv = grw_wave_reduce<false>(sv); sd = grw_wave_reduce<false>(sd);
        const float val = lane == 0 ? m : lane == 1 ? s : lane == 2 ? su : lane == 3 ? sv : lane == 4 ? sd : 0.f;
        if (lane < 8) grw_store((unsigned)(lane * 4), val, po);
    };
    // (tile indices run over all images: row (b * ntile + tl) of `part` IS row `tile`)
    auto part_row = [&](int tile) -> float* { return g.sam_part + ((size_t)tile * g.C + (n0 >> 5)) * 8; };
    {
        // ---- this wave's channel block over the tiles t0 .. t1 - 1
        load_weights(run_g0);
        issue_tile(t0, 0);
        float vd[16];                                  // second block of the previous unit: its epilogue rides behind this unit's first MFMAs
        float sm_m[2], sm_s[2], sm_u[2], sm_v[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) vd[r] = 0.f;
        sm_m[0] = sm_m[1] = 0.f; sm_s[0] = sm_s[1] = 0.f; sm_u[0] = sm_u[1] = 0.f; sm_v[0] = sm_v[1] = 0.f;
        int img_t = SAM ? t0 % g.ntile : 0, tslot = 0, tprev = 0;
        const char* drow = (const char*)g.Out;         // output rows of the deferred block
        for (int tile = t0; tile < t1; ++tile) {
            const int slot = (tile - t0) & 1;
            const bool first = tile == t0;
            if constexpr (SAM) {
                if (tid < 64) {                        // pixel coordinates of this tile (read behind the barrier below)
                    const int pix = img_t * 64 + tid;
                    const int h = pix / g.Wimg, w = pix - h * g.Wimg;
                    s_cu[tslot * 128 + tid] = w * g.invW;
                    s_cu[tslot * 128 + 64 + tid] = h * g.invH;
                }
                if (++img_t == g.ntile) img_t = 0;
            }
            // The fills of this tile are the oldest outstanding operations of the wave; behind them at least the 16 stores of the previous
            // unit's first block (issued last): everything older than those 16 has landed when at most 16 remain outstanding.
            if (first || !active) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (tile + 1 < t1) issue_tile(tile + 1, slot ^ 1);      // its slot was last read in the previous unit: every wave is past those reads
            if (active) {
                const unsigned sbase = slot * TILE_BYTES;
                const char* orow = (const char*)g.Out + (size_t)tile * 64 * rowbytes;
                float v0[16];
                // first block; behind its MFMAs the stores of the previous unit's second block (none in the first unit of a run)
                if (first) grw_block<KS, false>(wh, wl, a_rel, sbase, bias_l, v0, vd, o_voff, drow, rowbytes);
                else {
                    grw_block<KS, true>(wh, wl, a_rel, sbase, bias_l, v0, vd, o_voff, drow, rowbytes);
                    if constexpr (SAM) {
                        sam_block(vd, s_cu + tprev * 128, 1, sm_m[1], sm_s[1], sm_u[1], sm_v[1]);
                        sam_finish(sm_m, sm_s, sm_u, sm_v, part_row(tile - 1));
                    }
                }
                // second block; behind its MFMAs the stores of the first
                grw_block<KS, true>(wh, wl, a_rel, sbase + 4096, bias_l, vd, v0, o_voff, orow, rowbytes);      // (vd's previous contents were consumed above)
                drow = orow + 32 * (size_t)rowbytes;
                if constexpr (SAM) sam_block(v0, s_cu + tslot * 128, 0, sm_m[0], sm_s[0], sm_u[0], sm_v[0]);
            }
            tprev = tslot;
            tslot = tslot == 2 ? 0 : tslot + 1;
        }
        if (active) {                                  // the last unit's second block
            emit(vd, drow);
            if constexpr (SAM) {
                sam_block(vd, s_cu + tprev * 128, 1, sm_m[1], sm_s[1], sm_u[1], sm_v[1]);
                sam_finish(sm_m, sm_s, sm_u, sm_v, part_row(t1 - 1));
            }
        }
        if (g.dbg && lane == 0) {
            g.dbg[((size_t)blockIdx.x * 8 + wave) * 2] = __builtin_amdgcn_s_memtime() - dbg_c0;
            g.dbg[((size_t)blockIdx.x * 8 + wave) * 2 + 1] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
        }
    }
}

template <int KS, bool SAM>
static int grw_launch(GemmRwArgs& g, hipStream_t st) {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return AB_EINVAL;
        ncu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    const size_t lds = (size_t)2 * KS * 16384 + (SAM ? 1536 : 0);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_rw_kernel<KS, SAM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    // one workgroup per CU, a multiple of 8 (the schedule is per XCD) and at least one tuple per XCD
    static const int wgs_env = getenv("AB_GRW_WGS") ? atoi(getenv("AB_GRW_WGS")) : 0;
    int grid = ((wgs_env > 0 ? wgs_env : ncu) / 8) * 8;
    if (grid < 8 * g.groups) grid = 8 * g.groups;
    gemm_rw_kernel<KS, SAM><<<grid, 512, lds, st>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// Shapes this kernel takes: K in {64, 128, 192, 256}, N a multiple of 32, M a multiple of 64.  AB_ESHAPE otherwise (the caller falls back
// to the implicit GEMM).  sam: NULL, or the fused soft-argmax statistics (DEPTH_PITCH must be 32 = one class per wave; H * W % 64 == 0).
struct GemmRwSam { float* part; int C, D, H, W; };
int gemm_rw_ok(long M, int N, int K) {
    static const int off = getenv("AB_GRW_OFF") ? atoi(getenv("AB_GRW_OFF")) : 0;
    return !off && M > 0 && M % 64 == 0 && M / 64 < (1 << 24) && N % 32 == 0 && N >= 32 && N <= 8192 && K % 64 == 0 && K >= 64 && K <= 256;
}
unsigned long long* g_grw_dbg = nullptr;       // tools/probe_grw.hip sets it
int gemm_rw_run(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias, float* out, long M, int N,
                int K, const GemmRwSam* sam, hipStream_t st) {
    if (!gemm_rw_ok(M, N, K)) return AB_ESHAPE;
    GemmRwArgs g = {};
    g.A_hi = a_hi; g.A_lo = a_lo; g.W_hi = w_hi; g.W_lo = w_lo; g.bias = bias; g.Out = out;
    g.M = (int)M; g.N = N; g.tiles = (int)(M / 64);
    g.groups = (N / 32 + 7) / 8;
    g.dbg = g_grw_dbg;
    if (sam) {
        const long npix = (long)sam->H * sam->W;
        if (!sam->part || sam->C * 32 != N || sam->D <= 0 || sam->D > 32 || npix % 64 || M % npix) return AB_ESHAPE;
        g.sam_part = sam->part; g.C = sam->C; g.D = sam->D; g.Wimg = sam->W; g.npix = (int)npix; g.ntile = (int)(npix / 64);
        g.invW = 1.f / sam->W; g.invH = 1.f / sam->H; g.invD = 1.f / sam->D;
    }
#define GRW_GO(KS_) return sam ? grw_launch<KS_, true>(g, st) : grw_launch<KS_, false>(g, st)
    switch (K / 64) {
        case 1: GRW_GO(1);
        case 2: GRW_GO(2);
        case 3: GRW_GO(3);
        default: GRW_GO(4);
    }
#undef GRW_GO
}
