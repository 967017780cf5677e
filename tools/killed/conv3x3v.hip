// 3x3 / stride 1 / pad 1 split-bf16 convolution (forward, and data gradient via flipped taps), second generation:
// the input halo patch lives in LDS (as in conv3x3.hip), the WEIGHTS never touch LDS -- every wave streams the MFMA fragments of its
// own 32 output channels straight from L2 into VGPRs, from a copy of the weights stored in fragment order (c3v_pack_kernel: one
// contiguous KiB per wave instruction).  What that removes from the K loop of conv3x3.hip: the 3-deep LDS weight ring, its fragment
// reads (half of all ds_read_b128 there) and the nine barrier + vmcnt rendezvous per 32-channel chunk -- ONE barrier per chunk is left,
// for the patch hand-over.  resnet.py:41-44,85-101 (conv3x3 of BasicBlock, forward and backward).
//
// Workgroup: 256 output pixels (TH x TW) x BN = 32 * WN output channels on 4 waves, one per SIMD (up to 512 registers each):
//   wave (wn, gk): output channels [32 wn, 32 wn + 32) of the tile, ALL 256 pixels (8 accumulator blocks of 32 x 32, two chains),
//   and, when WK = 2, the k16 slice gk of every 32-channel chunk (the two slices are summed through LDS in the epilogue) --
//   so no two waves of a workgroup ever load the same weight bytes: per launch the weight stream is Cn * 9C * 4 B * (pixels / 256).
// Per step (one tap, one 16-channel slice): 2 weight fragments (hi, lo; prefetched two steps ahead into a 3-deep register ring),
// 16 patch fragments (ds_read_b128, half a step ahead), 24 MFMAs.
//
// LDS: [patch 0][patch 1], pixel-major 128-byte rows = 32 channels as [hi: 4 x 16 B][lo: 4 x 16 B], 16-byte slot c of patch pixel
// (py, px) stored at c ^ ((px >> 1) & 7) (conflict-free ds_read_b128 for the 16-lane groups; the swizzle sits on the DMA source).
// All loads of the K loop are inline asm with exact hand-counted vmcnt waits (cdna_hip_programming.md 5.7).
#include <type_traits>
#include "conv3x3.h"

static __device__ uint4 c3v_zero_page[2];

template <int I, int N, class F> __device__ __forceinline__ void c3v_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); c3v_for<I + 1, N>(f); }
}

__device__ __forceinline__ void c3v_gload(u32x4& d, unsigned voff, const void* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void c3v_gload_1k(u32x4& d, unsigned voff, const void* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(d) : "v"(voff), "s"(sbase) : "memory");
}
// the wait that retires a weight-fragment pair: tied to the registers so that no consumer can be scheduled above it
template <int N> __device__ __forceinline__ void c3v_wait(u32x4& h, u32x4& l) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(h), "+v"(l) : "n"(N) : "memory");
}

// patch-fragment read and the wait that retires a half step's eight of them (same register-tied form)
template <int IMM> __device__ __forceinline__ void c3v_dsread(u32x4& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM) : "memory");
}
__device__ __forceinline__ void c3v_wait_frags(u32x4 (&f)[4][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]),
                 "+v"(f[2][0]), "+v"(f[2][1]), "+v"(f[3][0]), "+v"(f[3][1]) :: "memory");
}

// weights [Cn][9][C] (hi, lo planes) -> fragment order [Cn/32][C/32][9 taps][2 k16][2 planes][64 lanes] x 16 B:
// lane l of fragment (cb, c, t, kk, plane) holds channels c*32 + kk*16 + (l>>5)*8 .. +8 of tap t of output channel cb*32 + (l&31)
__global__ __launch_bounds__(256) void c3v_pack_kernel(const bf16_t* __restrict__ w_hi, const bf16_t* __restrict__ w_lo, int Cn, int C,
                                                       uint4* __restrict__ out, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        long r = i >> 6;
        const int plane = (int)(r & 1); r >>= 1;
        const int kk = (int)(r & 1); r >>= 1;
        const int t = (int)(r % 9); r /= 9;
        const int nch = C / 32;
        const int c = (int)(r % nch), cb = (int)(r / nch);
        const bf16_t* src = (plane ? w_lo : w_hi) + ((long)(cb * 32 + (lane & 31)) * 9 + t) * C + c * 32 + kk * 16 + (lane >> 5) * 8;
        out[i] = *(const uint4*)src;
    }
}

long c3v_frag_bytes(int C, int Cn) { return (long)(Cn / 32) * (C / 32) * 9 * 4096; }

int c3v_pack(const void* w_hi, const void* w_lo, int C, int Cn, void* out, hipStream_t st) {
    if (C % 32 || Cn % 32) return AB_ESHAPE;
    const long total = c3v_frag_bytes(C, Cn) / 16;
    long b = (total + 255) / 256; if (b > 4096) b = 4096;
    c3v_pack_kernel<<<(int)b, 256, 0, st>>>((const bf16_t*)w_hi, (const bf16_t*)w_lo, Cn, C, (uint4*)out, total);
    AB_LAUNCH_CHECK(); return 0;
}

// patch pieces (of LP per wave) issued in step u of a chunk: spread evenly over the first `psteps` steps, the remainder first
__host__ __device__ constexpr int c3v_np(int u, int lp, int psteps) { return (u < 0 || u >= psteps) ? 0 : lp / psteps + (u < lp % psteps ? 1 : 0); }
__host__ __device__ constexpr int c3v_ps(int u, int lp, int psteps) { int s = 0; for (int k = 0; k < u; ++k) s += c3v_np(k, lp, psteps); return s; }

template <int TW, int WN, int WK, int FLIP, int X3, int ABL = 0>      // ABL: timing-only ablations (bit 0: no patch pieces, bit 1: no weight loads in the loop)
__global__ __launch_bounds__(256, 1) void conv3x3v_kernel(Conv3Args g) {
    static_assert(WN * WK == 4, "four waves, one per SIMD");
    constexpr bool BNR = X3 == 2;
    constexpr int BM = 256, TM = 8, BN = 32 * WN, NW = 4, NT = 256;
    constexpr int KK = 2 / WK;                         // k16 slices of a 32-channel chunk this wave multiplies
    constexpr int U = 9 * KK;                          // steps per chunk
    constexpr int TH = BM / TW, PW = TW + 2, PH = TH + 2, NPIX = PH * PW;
    constexpr int PI = (NPIX + 7) / 8, LP = (PI + NW - 1) / NW, PATCH_BYTES = LP * NW * 1024;
    static_assert(U - 2 >= 1 && LP <= 2 * (U - 2), "patch pieces are issued in steps 0 .. U-3, at most two per step");
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_n = wave % WN, gk = wave / WN;
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, within = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const int tiles_n = g.Cn / BN;
    const int tile_sp = logical / tiles_n, tile_n = logical - tile_sp * tiles_n;
    const int tpi = g.tiles_x * g.tiles_y;
    const int img = tile_sp / tpi, trem = tile_sp - img * tpi;
    const int ty0 = (trem / g.tiles_x) * TH, tx0 = (trem % g.tiles_x) * TW;
    const int n0 = tile_n * BN;
    const bf16_t* __restrict__ X = (const bf16_t*)g.X;
    const bf16_t* __restrict__ Xlo = (const bf16_t*)g.X_lo;
    const bf16_t* zp = (const bf16_t*)c3v_zero_page;
    const int nchunks = g.C / 32;
    const unsigned lds0 = lds_addr_of(smem);

    // ---- patch fill: this lane's 16 bytes of each of the wave's LP one-KiB pieces (8 pixels x 8 slots per piece)
    const bf16_t* p_src[LP];
    bool p_ok[LP];
#pragma unroll
    for (int j = 0; j < LP; ++j) {
        const int ii = wave * LP + j;
        const int pp = ii * 8 + (lane >> 3);
        const int py = pp / PW, px = pp - py * PW;
        const int y = ty0 + py - 1, x = tx0 + px - 1;
        p_ok[j] = (ii < PI) && (pp < NPIX) && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
        int c = (lane & 7) ^ ((px >> 1) & 7);
        const bf16_t* plane = (c & 4) ? Xlo : X;
        c &= 3;
        p_src[j] = p_ok[j] ? plane + ((((long)img * g.H + y) * g.W + x) * g.C + c * 8) : zp;
    }
    // ---- patch fragment addresses: block i = tile rows (32 / TW) i .., lane = pixel; slot of (plane 0, k16 slice 0 or gk, half fh)
    const int l32 = lane & 31, fh = lane >> 5;
    unsigned a0[TM][3];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = i * 32 + l32;
        const int oy = row / TW, ox = row - oy * TW;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int px = ox + d;
            a0[i][d] = lds0 + (oy * PW + px) * 128 + (((fh | (WK == 2 ? gk << 1 : 0)) ^ ((px >> 1) & 7)) << 4);
        }
    }
    // ---- weight fragments: [cb][chunk][tap][kk][plane][lane] x 16 B
    const unsigned wvoff = lane * 16;
    const char* wq = (const char*)g.Wf + ((long)(tile_n * WN + wave_n) * nchunks) * (9 * 4096) + (WK == 2 ? gk * 2048 : 0);

    f32x16 acc[TM], accx[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; accx[i][r] = 0.f; }
    u32x4 bh[3], bl[3];            // weight-fragment ring: step u uses slot u % 3
    u32x4 fa[2][4][2];             // patch fragments of one half step (4 blocks x (hi, lo)), double-buffered

    auto issue_b = [&](int chunk, auto vc, auto rc) {       // weights of step v of `chunk` into ring slot r
        constexpr int v = decltype(vc)::value, r = decltype(rc)::value;
        const char* s = wq + ((long)chunk * 9 + v / KK) * 4096 + (v % KK) * 2048;
        c3v_gload(bh[r], wvoff, s);
        c3v_gload_1k(bl[r], wvoff, s);
    };
    auto issue_piece = [&](auto jc, int chunk, unsigned pbuf) {
        constexpr int j = decltype(jc)::value;
        const int ii = wave * LP + j;
        glds16(p_ok[j] ? (const void*)(p_src[j] + chunk * 32) : (const void*)zp, __builtin_amdgcn_readfirstlane(lds0 + pbuf + ii * 1024));
    };
    constexpr int PSTEPS = U - 2;          // pieces of the NEXT chunk's patch are issued in steps 0 .. U-3 (c3v_np of them in step u)

    auto stamp = [&](int k) { if (g.dbg && lane == 0) g.dbg[((long)blockIdx.x * 4 + wave) * 8 + k] = __builtin_readcyclecounter(); };
    stamp(0);
    if (g.dbg && lane == 0) g.dbg[((long)blockIdx.x * 4 + wave) * 8 + 7] = __builtin_amdgcn_s_memrealtime();
    // ---- prologue: patch of chunk 0, weights of steps 0 and 1
    c3v_for<0, LP>([&](auto jc) { issue_piece(jc, 0, 0u); });
    issue_b(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    issue_b(0, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});

    // One body for every chunk, the last included (a second copy for the last chunk made hipcc permute all 256 accumulator registers
    // between the two: ~570 v_accvgpr moves per workgroup): the last chunk "prefetches" a patch and two weight steps it never uses
    // (addresses clamped to the last chunk: valid memory), retired by the vmcnt(0) in front of the epilogue.
    stamp(1);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (chunk == 1) stamp(2);
        const int cnext = min(chunk + 1, nchunks - 1);
        const unsigned pnext = ((chunk + 1) & 1) * PATCH_BYTES;
        c3v_for<0, U>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int r = u % 3;
            // loads younger than this step's weights: pieces of step u-2, weights + pieces of step u-1
            constexpr int NWAIT = ((ABL & 1) ? 0 : c3v_np(u - 2, LP, PSTEPS) + c3v_np(u - 1, LP, PSTEPS)) + 2;
            if constexpr (!(ABL & 2)) c3v_wait<NWAIT>(bh[r], bl[r]);
            if constexpr (u == 0) {
                // the patch of this chunk: every wave's pieces are older than the weights just waited for; the barrier also
                // proves that every wave is done reading the other buffer (restaged below)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            // ---- the step itself, fully pinned (sched_barrier(0) between slots): hipcc sinks compiler-visible ds_reads to just before
            // their consumer under this register pressure, which exposes the LDS latency on a one-wave-per-SIMD kernel.  The fragment
            // reads are asm with hand-placed waits: those of half step hh + 1 ride in the shadow of the first MFMAs of half step hh.
            auto read_frag = [&](auto hc, auto ic) {      // fragment pair (hi, lo) of block i of half step hh
                constexpr int hh = decltype(hc)::value, i = decltype(ic)::value;
                constexpr int uu = hh / 2, h = hh % 2, tt = uu / KK, kl = uu % KK;
                constexpr int ddh = FLIP ? 2 - tt / 3 : tt / 3, ddw = FLIP ? 2 - tt % 3 : tt % 3;
                const unsigned ah = a0[h * 4 + i][ddw] ^ (unsigned)(kl << 5), al = ah ^ 64u;
                c3v_dsread<ddh * PW * 128>(fa[hh & 1][i][0], ah);
                c3v_dsread<ddh * PW * 128>(fa[hh & 1][i][1], al);
            };
            if constexpr (u == 0) {
                c3v_for<0, 4>([&](auto ic) { read_frag(std::integral_constant<int, 0>{}, ic); });
            }
            c3v_for<0, 2>([&](auto hcc) {
                constexpr int h = decltype(hcc)::value, hh = 2 * u + h, fb = hh & 1;
                c3v_wait_frags(fa[fb]);
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 wh = __builtin_bit_cast(bf16x8, bh[r]), wl = __builtin_bit_cast(bf16x8, bl[r]);
                c3v_for<0, 12>([&](auto mc) {
                    constexpr int m = decltype(mc)::value, i = m % 4, pass = m / 4;
                    if constexpr (pass == 0)
                        acc[h * 4 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, __builtin_bit_cast(bf16x8, fa[fb][i][0]), acc[h * 4 + i], 0, 0, 0);
                    else if constexpr (pass == 1)
                        accx[h * 4 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, __builtin_bit_cast(bf16x8, fa[fb][i][1]), accx[h * 4 + i], 0, 0, 0);
                    else
                        accx[h * 4 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, __builtin_bit_cast(bf16x8, fa[fb][i][0]), accx[h * 4 + i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // in the shadow of MFMA m: the next half step's fragments (m = 0..3), this step's VMEM (half 0: weights of step
                    // u + 2 behind m = 5, half 1: the patch pieces behind m = 5, 7)
                    if constexpr (m < 4 && hh + 1 < 2 * U) read_frag(std::integral_constant<int, hh + 1>{}, std::integral_constant<int, m>{});
                    if constexpr (h == 0 && m == 5) {
                        if constexpr (ABL & 2) { if (u == 0 && chunk == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                        else if constexpr (u + 2 < U) issue_b(chunk, std::integral_constant<int, u + 2>{}, std::integral_constant<int, (u + 2) % 3>{});
                        else issue_b(cnext, std::integral_constant<int, u + 2 - U>{}, std::integral_constant<int, (u + 2) % 3>{});
                    }
                    if constexpr (h == 1 && (m == 5 || m == 7) && !(ABL & 1)) {
                        constexpr int p0 = c3v_ps(u, LP, PSTEPS), np = c3v_np(u, LP, PSTEPS);
                        if constexpr (m == 5 && np >= 1) issue_piece(std::integral_constant<int, p0>{}, cnext, pnext);
                        if constexpr (m == 7 && np >= 2) issue_piece(std::integral_constant<int, p0 + 1>{}, cnext, pnext);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        });
        // the next chunk reads the other patch buffer
        const unsigned flipbuf = (chunk & 1) ? (unsigned)-PATCH_BYTES : (unsigned)PATCH_BYTES;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int d = 0; d < 3; ++d) a0[i][d] += flipbuf;
    }
    stamp(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last chunk's unused prefetches: nothing may land in LDS after this point
    __syncthreads();

    // ---- epilogue: (acc + accx) -> pixel-major fp32 staging tile in LDS (a lane owns ONE pixel and per register quad four consecutive
    // channels: one 16-byte store); with WK = 2 the two k16 slices meet there.  From the staging tile on: as conv3x3.hip's split-bf16 path.
    constexpr int SPF = BN * 4 + 16;
    auto stage = [&](bool add) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = i * 32 + l32;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int cl = wave_n * 32 + 8 * q4 + 4 * fh;
                float4 w;
                w.x = acc[i][q4 * 4] + accx[i][q4 * 4]; w.y = acc[i][q4 * 4 + 1] + accx[i][q4 * 4 + 1];
                w.z = acc[i][q4 * 4 + 2] + accx[i][q4 * 4 + 2]; w.w = acc[i][q4 * 4 + 3] + accx[i][q4 * 4 + 3];
                float4* p = (float4*)(smem + row * SPF + cl * 4);
                if (add) { const float4 o = *p; w.x += o.x; w.y += o.y; w.z += o.z; w.w += o.w; }
                *p = w;
            }
        }
    };
    if constexpr (WK == 2) {
        if (gk == 1) stage(false);
        __syncthreads();
        if (gk == 0) stage(true);
    } else stage(false);
    __syncthreads();
    stamp(4);

    float* __restrict__ OutF = (float*)g.Out;
    const float* __restrict__ AddF = (const float*)g.addend;
    constexpr int CPRF = BN / 4;                          // 16-byte chunks (4 channels) per tile row
    float fs[4] = {0.f, 0.f, 0.f, 0.f}, fq[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (X3 == 3) {
        // eval-mode forward: BatchNorm (scale, shift) + residual + ReLU, written as the next convolution's operand planes --
        // the expression order of bn_apply_x3_kernel (norm_pool.hip), bit-identical to conv + separate apply pass
        bf16_t* __restrict__ OutHi = (bf16_t*)g.Out;
        bf16_t* __restrict__ OutLo = (bf16_t*)g.Out_lo;
        const bf16_t* __restrict__ RH = (const bf16_t*)g.res_hi;
        const bf16_t* __restrict__ RL = (const bf16_t*)g.res_lo;
        constexpr int CPR8 = BN / 8;
        const int c8 = tid % CPR8, col = n0 + c8 * 8;
        float sc[8], sh[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { sc[k] = g.bnp[col + k]; sh[k] = g.bnp[g.Cn + col + k]; }
        for (int row = tid / CPR8; row < BM; row += NT / CPR8) {
            const int yy = ty0 + row / TW, xx = tx0 + row % TW;
            if (yy < g.H && xx < g.W) {
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
                    const float4 a0f = *(const float4*)(AddF + o), a1f = *(const float4*)(AddF + o + 4);
                    v[0] += a0f.x; v[1] += a0f.y; v[2] += a0f.z; v[3] += a0f.w; v[4] += a1f.x; v[5] += a1f.y; v[6] += a1f.z; v[7] += a1f.w;
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
    // a thread keeps one 4-channel group over all its rows (NT % CPRF == 0)
    static_assert(NT % CPRF == 0, "a thread keeps one channel group over all its rows");
    const int ec4 = tid % CPRF, er0 = tid / CPRF;
    constexpr int ERS = NT / CPRF;
    const int col = n0 + ec4 * 4;
    if constexpr (BNR) {
        // data gradient arriving at relu(bn(bn_y) [+ residual]): mask it (dz), store dz, leave the BatchNorm-backward partial sums
        const float* __restrict__ BnY = (const float*)g.bn_y;
        const bf16_t* __restrict__ BnM = (const bf16_t*)g.bn_out;
        float e_mean[4], e_istd[4], e_sc[4], e_sh[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            e_sc[k] = g.bnp[col + k]; e_sh[k] = g.bnp[g.Cn + col + k];
            e_mean[k] = g.bnp[2 * g.Cn + col + k]; e_istd[k] = g.bnp[3 * g.Cn + col + k];
        }
        for (int row = er0; row < BM; row += ERS) {
            const int yy = ty0 + row / TW, xx = tx0 + row % TW;
            if (yy < g.H && xx < g.W) {
                const float4 v4 = *(const float4*)(smem + row * SPF + ec4 * 16);
                const long o = (((long)img * g.H + yy) * g.W + xx) * g.Cn + col;
                const float4 y4 = *(const float4*)(BnY + o);
                const float4 a4 = AddF ? *(const float4*)(AddF + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                const uint2 m2 = BnM ? *(const uint2*)(BnM + o) : make_uint2(0u, 0u);
                float v[4] = {v4.x + a4.x, v4.y + a4.y, v4.z + a4.z, v4.w + a4.w};
                const float y[4] = {y4.x, y4.y, y4.z, y4.w};
                const float m[4] = {__uint_as_float(m2.x << 16), __uint_as_float(m2.x & 0xffff0000u),
                                    __uint_as_float(m2.y << 16), __uint_as_float(m2.y & 0xffff0000u)};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool dead = BnM ? !(m[i] > 0.f) : !(y[i] * e_sc[i] + e_sh[i] > 0.f);
                    v[i] = dead ? 0.f : v[i];
                    fs[i] += v[i]; fq[i] += v[i] * ((y[i] - e_mean[i]) * e_istd[i]);
                }
                *(float4*)(OutF + o) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    } else {
        for (int row = er0; row < BM; row += ERS) {
            const int yy = ty0 + row / TW, xx = tx0 + row % TW;
            if (yy < g.H && xx < g.W) {
                float4 v = *(const float4*)(smem + row * SPF + ec4 * 16);
                const long o = (((long)img * g.H + yy) * g.W + xx) * g.Cn + col;
                if (AddF) { const float4 a = *(const float4*)(AddF + o); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
                *(float4*)(OutF + o) = v;
                fs[0] += v.x; fq[0] += v.x * v.x; fs[1] += v.y; fq[1] += v.y * v.y;
                fs[2] += v.z; fq[2] += v.z * v.z; fs[3] += v.w; fq[3] += v.w * v.w;
            }
        }
    }
    stamp(5);
    float* part_out = BNR ? g.bn_part : g.stats;
    if (part_out) {
        __syncthreads();
        float* sp = (float*)smem;                          // [ERS][BN][2], over the consumed staging tile
        const int rg = tid / CPRF, cb = (tid % CPRF) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) { sp[(rg * BN + cb + k) * 2] = fs[k]; sp[(rg * BN + cb + k) * 2 + 1] = fq[k]; }
        __syncthreads();
        for (int c = tid; c < BN; c += NT) {
            float s2 = 0.f, q2 = 0.f;
            for (int r = 0; r < ERS; ++r) { s2 += sp[(r * BN + c) * 2]; q2 += sp[(r * BN + c) * 2 + 1]; }
            part_out[((long)tile_sp * g.Cn + n0 + c) * 2] = s2;
            part_out[((long)tile_sp * g.Cn + n0 + c) * 2 + 1] = q2;
        }
    }
    stamp(6);
    if (g.dbg && lane == 0) g.dbg[((long)blockIdx.x * 4 + wave) * 8 + 7] = __builtin_amdgcn_s_memrealtime() - g.dbg[((long)blockIdx.x * 4 + wave) * 8 + 7];
}

// ---------------------------------------------------------------- host side
// which (TW, WN, WK) takes the shape; 0: none.  code = TW * 100 + WN * 10 + WK
int c3v_config(int N, int H, int W, int C, int Cn) {
    static const int on = getenv("AB_C3V") ? atoi(getenv("AB_C3V")) : 0;
    if (!on || C % 32 || Cn % 64 || C < 64) return 0;
    int tw;
    if (W % 32 == 0 && H % 8 == 0) tw = 32;
    else if (W % 16 == 0 && H % 16 == 0) tw = 16;
    else return 0;
    const long tiles = (long)N * (H / (256 / tw)) * (W / tw);
    if (Cn % 128 == 0 && tiles * (Cn / 128) >= 256) return tw * 100 + 41;
    return tw * 100 + 22;
}

int c3v_tiles(int N, int H, int W, int C, int Cn) {
    const int cfg = c3v_config(N, H, W, C, Cn);
    if (!cfg) return 0;
    const int tw = cfg / 100;
    return N * (H / (256 / tw)) * (W / tw);
}

template <int TW, int WN, int WK, int FLIP, int X3, int ABL = 0>
static int c3v_launch(Conv3Args& g, hipStream_t st) {
    constexpr int TH = 256 / TW, BN = 32 * WN;
    constexpr int NPIX = (TH + 2) * (TW + 2), PI = (NPIX + 7) / 8, LP = (PI + 3) / 4;
    g.tiles_x = g.W / TW; g.tiles_y = g.H / TH;
    const int blocks = g.N * g.tiles_x * g.tiles_y * (g.Cn / BN);
    size_t lds = (size_t)2 * LP * 4 * 1024;
    const size_t stage = (size_t)256 * (BN * 4 + 16);
    if (stage > lds) lds = stage;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3v_kernel<TW, WN, WK, FLIP, X3, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    conv3x3v_kernel<TW, WN, WK, FLIP, X3, ABL><<<blocks, 256, lds, st>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

static unsigned long long* c3v_dbg = nullptr;
extern "C" void ab_c3v_debug_buffer(void* p) { c3v_dbg = (unsigned long long*)p; }

// g: as conv3x3_x3_run fills it, plus Wf; X3: 1 plain (forward with stats / data gradient), 2 fused BatchNorm backward, 3 eval fold
int c3v_run(Conv3Args& g, int x3, hipStream_t st) {
    const int cfg = c3v_config(g.N, g.H, g.W, g.C, g.Cn);
    if (!cfg || !g.Wf) return AB_ESHAPE;
    g.dbg = c3v_dbg;
    static const int abl = getenv("AB_C3V_ABL") ? atoi(getenv("AB_C3V_ABL")) : 0;
    if (abl && x3 == 1 && !g.flip) {      // timing-only ablations of the forward launch (results are garbage)
        if (cfg == 1622) { if (abl == 1) return c3v_launch<16, 2, 2, 0, 1, 1>(g, st); if (abl == 2) return c3v_launch<16, 2, 2, 0, 1, 2>(g, st); return c3v_launch<16, 2, 2, 0, 1, 3>(g, st); }
        if (cfg == 3241) { if (abl == 1) return c3v_launch<32, 4, 1, 0, 1, 1>(g, st); if (abl == 2) return c3v_launch<32, 4, 1, 0, 1, 2>(g, st); return c3v_launch<32, 4, 1, 0, 1, 3>(g, st); }
    }
#define C3V_CFG(TW, WN, WK) \
    if (cfg == TW * 100 + WN * 10 + WK) { \
        if (x3 == 2) return c3v_launch<TW, WN, WK, 1, 2>(g, st); \
        if (x3 == 3) return c3v_launch<TW, WN, WK, 0, 3>(g, st); \
        if (g.flip) return c3v_launch<TW, WN, WK, 1, 1>(g, st); \
        return c3v_launch<TW, WN, WK, 0, 1>(g, st); \
    }
    C3V_CFG(32, 4, 1)
    C3V_CFG(32, 2, 2)
    C3V_CFG(16, 4, 1)
    C3V_CFG(16, 2, 2)
#undef C3V_CFG
    return AB_ESHAPE;
}
