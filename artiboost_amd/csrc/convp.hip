// Stride-2 3x3 convolutions of the ResNet-34 stage entries (torchvision resnet.py:59-101 as anakin/models/resnet.py uses it: layerN.0.conv1 +
// layerN.0.downsample.0, N = 2..4) on LDS-resident patches, split-bf16:
//
//   forward  : out[n, p, q, :] = sum over 3 x 3 taps of in[n, 2p + kh - 1, 2q + kw - 1, :] . W[kh, kw].  The nine taps fall on the four parity
//              sub-grids of the input as 1 + 2 + 2 + 4: per 32-channel chunk a workgroup walks four UNITS (a sub-grid patch + its taps), all
//              summed into one accumulator tile.  Every input pixel is fetched once per output-channel tile.
//   backward : the data gradient, dx[n, 2p + a, 2q + b, :], is four output-parity classes of 1 / 2 / 2 / 4 taps over ONE patch of dy; a
//              workgroup owns one class of a 16 x 16 (or 2 images x 8 x 8) base tile.  The 1x1 / stride-2 downsample branch that shares the
//              block's input arrives at class (even, even) only: it is that class's second unit (own dy tensor, own weight matrix), so both
//              branches are one launch and class (0, 0) costs the same two taps as (0, 1) and (1, 0).
//
// conv_gemm2.hip ran these six launches tap by tap (every tap re-fetching its rows from L2, one barrier and one address rebuild per 32-channel
// step): 50 us forward, 68 - 89 us backward, 13 - 23 % of the split-bf16 roof.  This is conv2x2.hip's loop (patch DMA'd once per unit, taps by
// fragment-address shifts, weight ring two steps ahead, counted vmcnt + raw s_barrier) with the tap count a property of the unit: the step
// sequence is driven by three scalar cursors (executing step, weight stage two steps ahead, patch two units ahead) over a per-class table in
// the kernel arguments, read with scalar loads; THREE patch buffers so that a one-tap unit still has its patch requested two steps early.
#include "conv3x3.h"
#include <cstddef>
#include <type_traits>
#ifndef CP_BLEAD
#define CP_BLEAD 2     // weight requests run this many steps ahead (2: conv2x2.hip's distance; 3 fits the four-stage ring and measured the same:
                       // tools/probe_cp.hip -DCP_BLEAD=3, forward 34.7 vs 34.9 us, data gradients 53.8 vs 52.8 us on layer 3)
#endif
#ifndef CP_ABL
#define CP_ABL 0      // tools/probe_cp.hip: knock-outs (1 no MFMAs, 2 no patch requests in the loop, 4 no weight requests in the loop, 8 no fragment reads)
#endif

static __device__ uint4 cp_zero_page[2];

// fields of a unit in the table (ints): 0 src, 1 wsrc, 2 oy0, 3 ox0, 4 sy, 5 sx, 6 ntap, 7 program code (CPProg), 8..11 koff[tap], 12..15 (dh << 1 | dw)[tap]
// (the kernel reads the code of a class's first unit and the koff entries; the rest documents what the compile-time programs assume and is
// checked against them on the host)
struct CPUnit { int src, wsrc, oy0, ox0, sy, sx, ntap, code; int koff[4]; int dhw[4]; };
struct CPArgs {
    CPUnit unit[4][4];                // [class][unit]  (first member: the kernel reads it through the kernarg segment pointer)
    const void* X[2]; const void* Xlo[2];      // input planes [N, Hi, Wi, C] of source 0 / 1
    const void* Wt[2]; unsigned wlo_delta[2]; int ktot[2];      // weight rows [Cn][ktot] of weight source 0 / 1, lo plane at + delta bytes
    float* Out; float* stats;         // [N, Ho, Wo, Cn]; BatchNorm partials [rows][Cn][2] or NULL
    int N, Hi, Wi, C, Cn;
    int in_stride;                    // 1: input pixel = base + origin + tap;  2: = 2 * (base + origin + tap) + (sy, sx)
    int Ho, Wo, out_stride;           // output pixel = base * out_stride + (cls_oy, cls_ox)
    int nclass, tiles_y, tiles_x;     // output classes in the grid; tiles of the base grid per image (G8: 1 x 1)
    int nunit[4]; int cls_oy[4], cls_ox[4];
    // optional (data gradient): the result arrives at relu(bn(bn_y) [+ residual]) -- Out receives the MASKED gradient (mask: hi plane bn_out of the stored
    // activation, or recomputed from bn_y) and bn_part [rows][Cn][2] the tile's (sum dz, sum dz * xhat): conv3x3.hip's X3 = 2 epilogue
    const float* bn_y; const void* bn_out; const float* bnp; float* bn_part;
    // optional (forward, eval mode): the BatchNorm (+ ReLU) that follows as a per-channel affine, out = relu?(acc * ep_scale[c] + ep_shift[c]) -- conv_gemm2.hip's
    // expression, bit-identical to this kernel + ab_bn_apply_x3 -- written as fp32 `Out` or, when out_hi != NULL, as the (hi, lo) planes the next convolution reads
    const float* ep_scale; const float* ep_shift; int ep_relu; void* out_hi; void* out_lo;
};

typedef const __attribute__((address_space(4))) int* cp_cint;


// A chunk's program: units with NT... taps each.  Step k of a chunk -> (unit, tap); allowed(k) = the requests that may still be in flight
// when step k starts: the weights of step k + 1 and the patches requested one step ago and -- unless step k is a unit's first tap whose patch
// is that very request (two one-tap units in a row) -- two steps ago.  FIRST: steps -1 / -2 are the prologue (B(1), P(1) | P(0), B(0)).
template <int... CODE>      // per unit: 4 = taps (dh, dw) in {0,1} x {0,1}; 2 = two taps, vertical (dh = 0, 1); 3 = two taps, horizontal (dw = 0, 1); 1 = one tap
struct CPProg {
    static constexpr int NU = sizeof...(CODE);
    static constexpr int code(int u) { constexpr int c[NU] = {CODE...}; return c[u]; }
    static constexpr int ntap(int u) { return code(u) == 4 ? 4 : (code(u) == 1 ? 1 : 2); }
    static constexpr int steps() { int s = 0; for (int u = 0; u < NU; ++u) s += ntap(u); return s; }
    static constexpr int unit_of(int k) { int u = 0; while (k >= ntap(u)) { k -= ntap(u); ++u; } return u; }
    static constexpr int tap_of(int k) { int u = 0; while (k >= ntap(u)) { k -= ntap(u); ++u; } return k; }
    static constexpr int dh(int u, int t) { return code(u) == 4 ? t >> 1 : (code(u) == 2 ? t : 0); }
    static constexpr int dw(int u, int t) { return code(u) == 4 ? t & 1 : (code(u) == 3 ? t : 0); }
    // forward (stride-2 input): the unit's parity sub-grid -- taps along an axis <=> the odd rows / columns, origin -1
    static constexpr int sy(int u) { return code(u) == 4 || code(u) == 2; }
    static constexpr int sx(int u) { return code(u) == 4 || code(u) == 3; }
    // REM = chunks after this one (2 = two or more): a request reaching r chunks ahead is made iff r <= REM.  PL = patch lead in units (1 | 2).
    static constexpr bool b_made(int rem, int k, int BL) { return (k + BL) / steps() <= rem; }        // the weights of step k + BL, at step k
    static constexpr bool p_made(int rem, int u, int PL) { return (u + PL) / NU <= rem; }             // the patch of unit u + PL, at unit u's first tap
    // Requests in issue order are pairs (step, kind): kind 0 = the weights a step requests, kind 1 = the patch it requests after them.  The prologue
    // stands for the steps before 0: P(0) at (-BL - 1, 1), B(i) at (i - BL, 0) for i < BL, and with PL = 2 P(1) at (-1, 1).  BL = weight lead in steps (2 | 3).
    static constexpr bool made(bool first, int rem, int j, int kind, int PL, int BL) {
        if (j < 0) {
            if (first) return kind == 0 ? (j >= -BL && b_made(rem, j, BL)) : (j == -BL - 1 || (j == -1 && PL == 2));
            const int r1 = rem + 1 > 2 ? 2 : rem + 1, jl = j + steps();          // (steps() >= 2 for every two-step look-back that matters: see below)
            if (jl < 0) return kind == 0 ? true : tap_of(jl + steps()) == 0;        // two chunks back (one-step programs): everything was requested
            return kind == 0 ? b_made(r1, jl, BL) : (tap_of(jl) == 0 && p_made(r1, unit_of(jl), PL));
        }
        return kind == 0 ? b_made(rem, j, BL) : (tap_of(j) == 0 && p_made(rem, unit_of(j), PL));
    }
    static constexpr int allowed(bool first, int rem, int k, int LB, int LP, int PL, int BL) {
        // what step k needs: its weights, requested at (k - BL, 0); at a unit's first tap also its patch, requested at the first tap of unit u - PL
        int ns = k - BL, nk = 0;
        if (tap_of(k) == 0) {
            const int u = unit_of(k);
            int ago = 0;
            for (int i = 1; i <= PL; ++i) ago += ntap(((u - i) % NU + NU) % NU);
            int ps = k - ago;
            if (first && u < PL) ps = u == 0 ? -BL - 1 : -1;                         // the prologue's patches
            if (ps > ns || (ps == ns)) { ns = ps; nk = 1; }
        }
        // everything issued after (ns, nk) up to step k - 1 may still be in flight
        int n = 0;
        for (int j = ns; j <= k - 1; ++j)
            for (int kind = 0; kind < 2; ++kind)
                if ((j > ns || kind > nk) && made(first, rem, j, kind, PL, BL)) n += kind ? LP : LB;
        return n;
    }
};

__device__ __forceinline__ void cp_wait_dyn(int n) {      // n is a constant after unrolling: one s_waitcnt remains
    switch (n) {
#define CP_W_(v) case v: asm volatile("s_waitcnt vmcnt(" #v ")" ::: "memory"); break;
        CP_W_(0) CP_W_(1) CP_W_(2) CP_W_(3) CP_W_(4) CP_W_(5) CP_W_(6) CP_W_(7) CP_W_(8) CP_W_(9) CP_W_(10) CP_W_(11) CP_W_(12) CP_W_(13) CP_W_(14)
        CP_W_(15) CP_W_(16) CP_W_(17) CP_W_(18) CP_W_(19) CP_W_(20)
#undef CP_W_
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// G8 = false: base tile 16 x 16 of one image (BM = 256), patch 17 rows x 18 (17 used), swizzle key (px >> 1) & 7.
// G8 = true : base grid 8 x 8, a tile is NI = BM / 64 consecutive images with a 9 x 9 patch each (conv2x2.hip's layout).
// NPB = 3 patch buffers + ring of four weight stages: one workgroup per CU.  NPB = 2 (+ ring of three, BM = 128): 72 KB of LDS and <= 128 VGPRs, TWO
// workgroups per CU -- one's epilogue (fp32 store, BatchNorm operands of the fused reduction) runs under the other's K loop.
template <int BM, int BN, bool G8, int MODE, int NPB>
__global__ __launch_bounds__(512, NPB == 2 ? 4 : 2) void convp_kernel(CPArgs g) {
    constexpr int TW = G8 ? 8 : 16, TH = G8 ? 8 : BM / 16, NI = G8 ? BM / 64 : 1, PW = G8 ? 9 : 18, PH = TH + 1, IPIX = PH * PW, NPIX = NI * IPIX;
    constexpr int PL = NPB - 1, NRING = NPB == 2 ? 3 : 4;
    static_assert(BM == NI * TW * TH && (BM == 128 || BM == 256), "tile = whole base tiles");
    constexpr int WM = 4, WN = 2, NW = 8, NT = 512;
    constexpr int PI = (NPIX + 7) / 8, LP = (PI + NW - 1) / NW, PATCH_BYTES = LP * NW * 1024;
    constexpr int IB = BN / 8, LB = IB / NW, BBYTES = BN * 128;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int PATCH0 = NRING * BBYTES;                    // LDS: [weight ring][patch 0][patch 1]([patch 2])
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WN, wave_n = wave % WN;
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, within = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const int tiles_n = g.Cn / BN;
    const int row_id = logical / tiles_n, tile_n = logical - row_id * tiles_n;        // row_id = base tile * nclass + class: the BatchNorm partial row
    const int tile_sp = row_id / g.nclass, cls = row_id - tile_sp * g.nclass;
    const int ntile = g.tiles_y * g.tiles_x, grp = tile_sp / ntile, tt = tile_sp - grp * ntile;
    const int by = (tt / g.tiles_x) * TH, bx = (tt % g.tiles_x) * TW, img = grp * NI;
    const int n0 = tile_n * BN;
    const bf16_t* zp = (const bf16_t*)cp_zero_page;
    const int nch = g.C / 32;
    const unsigned lds0 = lds_addr_of(smem);
    const int nunit = g.nunit[cls];
    const cp_cint tab = (cp_cint)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(CPArgs, unit)) + cls * 64;
#define CPU_(u, f) (tab[(u) * 16 + (f)])

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
    unsigned b_voff[2][LB];
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        const int ii = wave * LB + j, r = ii * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
#pragma unroll
        for (int w = 0; w < 2; ++w)
            b_voff[w][j] = (unsigned)(((long)(n0 + r) * g.ktot[w] + (c & 3) * 8) * 2) + ((c & 4) ? g.wlo_delta[w] : 0u);
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
            for (int kk = 0; kk < 4; ++kk)
                a_rel[i][d][kk] = lds0 + PATCH0 + (il * IPIX + oy * PW + px) * 128 +
                                  (((kk * 2 + fhalf) ^ (G8 ? (((px >> 1) & 3) | ((oy & 1) << 2)) : ((px >> 1) & 7))) << 4);
        }
    }

    // (the parameters of the next patch / weight stage are fetched when their cursor advances: the scalar loads return a step before use)
    // address of a patch element = (plane of the unit's source + the unit's scalar offset) + a lane constant: the per-request vector work is
    // two range checks, one 64-bit add and the selects (the full per-lane index arithmetic cost 18 us of a 59 us launch: tools/probe_cp.hip)
    int r_l[LP], c_l[LP]; unsigned l_off[LP];
#pragma unroll
    for (int j = 0; j < LP; ++j) {
        r_l[j] = p_py[j] * g.in_stride; c_l[j] = p_px[j] * g.in_stride;
        l_off[j] = (unsigned)(((((long)p_il[j] * g.Hi + r_l[j]) * g.Wi + c_l[j]) * g.C + p_coff[j]) * 2);
    }
    auto issue_patch = [&](int src, int oy0, int ox0, int sy, int sx, int chunk, int pbuf) {
        const int r_u = (oy0 + by) * g.in_stride + sy, c_u = (ox0 + bx) * g.in_stride + sx;
        const long uoff = (((long)img * g.Hi + r_u) * g.Wi + c_u) * g.C + chunk * 32;
        const char* Xu = (const char*)((const bf16_t*)(src ? g.X[1] : g.X[0]) + uoff);
        const char* Xlu = (const char*)((const bf16_t*)(src ? g.Xlo[1] : g.Xlo[0]) + uoff);
#pragma unroll
        for (int j = 0; j < LP; ++j) {
            const int ii = wave * LP + j;
            const bool ok = p_in[j] && (unsigned)(r_l[j] + r_u) < (unsigned)g.Hi && (unsigned)(c_l[j] + c_u) < (unsigned)g.Wi;
            const char* sp = ok ? (p_lo[j] ? Xlu : Xu) + l_off[j] : (const char*)zp;
            glds16(sp, __builtin_amdgcn_readfirstlane(lds0 + PATCH0 + pbuf * PATCH_BYTES + ii * 1024));
        }
    };
    auto issue_b = [&](int chunk, int koff, int wsrc, int slot) {
        const bf16_t* base = (const bf16_t*)(wsrc ? g.Wt[1] : g.Wt[0]) + (koff + chunk * 32);      // (wsrc is a constant after unrolling)
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            const int ii = wave * LB + j;
            const unsigned vo = wsrc ? b_voff[1][j] : b_voff[0][j];
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vo), "s"(base), "s"(__builtin_amdgcn_readfirstlane(lds0 + slot * BBYTES + ii * 1024)) : "memory");
        }
    };

    f32x16 acc[TM][TN], accx[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accx[i][j][r] = 0.f; }

    // ---- the step sequence of a chunk is a compile-time PROGRAM (units x taps), unrolled; the unit parameters are scalars fetched once.
    // Requests run ahead of the executing step: the weight stage of step k + 2 (ring of four), the patch of unit instance sc + 2 at the first tap of
    // instance sc (three buffers: a one-tap unit still has its patch requested two steps early).  Issue order inside a step: weights, then patch;
    // every vmcnt below is the number of requests issued after the ones the step needs (cp_allowed).  Stage and buffer indices are running scalars.
    auto run = [&](auto prog) {
        using P = decltype(prog);
        constexpr int NU = P::NU, S = P::steps();
        constexpr int BL = (NRING == 4 && S >= 2 && CP_BLEAD == 3) ? 3 : 2;      // weights three steps ahead where the ring and the chunk allow
        // per unit: source (the second unit of the data gradient's class (0, 0) is the downsample branch), sub-grid parity and origin (forward
        // only) are properties of the program; the weight-row offsets of its taps come from the host's table
        int u_koff[S];
#pragma unroll
        for (int k = 0; k < S; ++k) u_koff[k] = __builtin_amdgcn_readfirstlane(CPU_(P::unit_of(k), 8 + P::tap_of(k)));
        auto patch_of = [&](int u, int chunk, int pbuf) {
            const int src = MODE == 1 && u == 1, sy = MODE == 0 ? P::sy(u) : 0, sx = MODE == 0 ? P::sx(u) : 0;
            issue_patch(src, -sy, -sx, sy, sx, chunk, pbuf);
        };
        auto weights_of = [&](int k, int chunk, int slot) { issue_b(chunk, u_koff[k], MODE == 1 && P::unit_of(k) == 1, slot); };
        // prologue: P(0), B(0) .. B(BL - 1) [, P(1)]
        patch_of(0, 0, 0);
#pragma unroll
        for (int i = 0; i < BL; ++i) if (i / S < nch) weights_of(i % S, i / S, i);
        if (PL == 2) patch_of(1 % NU, 1 / NU, 1);
        int st = 0, pb_e = 0, pb_i = PL == 2 ? 2 : 1;      // ring stage of the executing step, patch buffer of its unit, buffer of the next patch request
        auto body = [&](auto first_c, auto rem_c, int chunk) {
            constexpr bool FIRST = decltype(first_c)::value; constexpr int REM = decltype(rem_c)::value;
#pragma unroll
            for (int k = 0; k < S; ++k) {
                // (k is a constant after unrolling; hipcc folds the constexpr lookups below)
                const int u = P::unit_of(k), t = P::tap_of(k);
                cp_wait_dyn(P::allowed(FIRST, REM, k, LB, LP, PL, BL));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // WAR on the ring stage / patch buffer restaged below
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (P::b_made(REM, k, BL) && !(CP_ABL & 4)) {
                    weights_of((k + BL) % S, chunk + (k + BL) / S, (st + BL) % NRING);
                }
                if (t == 0 && P::p_made(REM, u, PL) && !(CP_ABL & 2)) {
                    patch_of((u + PL) % NU, chunk + (u + PL) / NU, pb_i);
                    pb_i = pb_i == NPB - 1 ? 0 : pb_i + 1;
                }
                const int dh = P::dh(u, t), dw = P::dw(u, t);
                const unsigned aoff = pb_e * PATCH_BYTES + dh * PW * 128;
                const unsigned aflip = (G8 && dh) ? 64u : 0u;
                const unsigned boff = st * BBYTES;
                u32x4 fa[4][TM] = {}, fb[4][TN] = {};
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    if (CP_ABL & 8) break;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int ii = 0; ii < TM; ++ii)
                            fa[k2 + 2 * h][ii] = *(const lds_u32x4*)(((dw ? a_rel[ii][1][k2 + 2 * h] : a_rel[ii][0][k2 + 2 * h]) ^ aflip) + aoff);
#pragma unroll
                        for (int jj = 0; jj < TN; ++jj) fb[k2 + 2 * h][jj] = *(const lds_u32x4*)(b_rel[jj][k2 + 2 * h] + boff);
                    }
                }
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    if (CP_ABL & 1) break;
#pragma unroll
                    for (int ii = 0; ii < TM; ++ii)
#pragma unroll
                        for (int jj = 0; jj < TN; ++jj) {
                            const bf16x8 bh = __builtin_bit_cast(bf16x8, fb[k2][jj]), bl = __builtin_bit_cast(bf16x8, fb[k2 + 2][jj]);
                            const bf16x8 ah = __builtin_bit_cast(bf16x8, fa[k2][ii]), al = __builtin_bit_cast(bf16x8, fa[k2 + 2][ii]);
                            // weights as the first operand: the accumulator is the TRANSPOSED tile (a lane owns one pixel, see the epilogue)
                            acc[ii][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, acc[ii][jj], 0, 0, 0);
                            accx[ii][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al, accx[ii][jj], 0, 0, 0);
                            accx[ii][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah, accx[ii][jj], 0, 0, 0);
                        }
                }
                st = st == NRING - 1 ? 0 : st + 1;
                if (t + 1 == P::ntap(u)) pb_e = pb_e == NPB - 1 ? 0 : pb_e + 1;
            }
        };
        auto chunk_body = [&](auto first_c, int chunk) {           // nch >= 2 (host)
            const int rem = nch - 1 - chunk;
            if (rem >= 2) body(first_c, std::integral_constant<int, 2>{}, chunk);
            else if (rem == 1) body(first_c, std::integral_constant<int, 1>{}, chunk);
            else body(first_c, std::integral_constant<int, 0>{}, chunk);
        };
        chunk_body(std::true_type{}, 0);
        for (int chunk = 1; chunk < nch; ++chunk) chunk_body(std::false_type{}, chunk);
    };
    if (MODE == 0) run(CPProg<4, 2, 3, 1>{});
    else if (nunit == 2) run(CPProg<1, 1>{});
    else {
        const int code0 = CPU_(0, 7);
        if (code0 == 4) run(CPProg<4>{});
        else if (code0 == 2) run(CPProg<2>{});
        else if (code0 == 3) run(CPProg<3>{});
        else run(CPProg<1>{});
    }
#undef CPU_
    __syncthreads();

    // ---- fp32 epilogue (conv2x2.hip's): a lane owns ONE pixel and per register quad four consecutive channels = one 16-byte LDS store into
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
    const bool bnr = MODE == 1 && g.bn_y != nullptr;
    float e_sc[4] = {}, e_sh[4] = {}, e_mean[4] = {}, e_istd[4] = {};
    if (bnr) {           // (a thread keeps one channel group: NT % CPRF == 0)
        const int col = n0 + (tid % CPRF) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) { e_sc[k] = g.bnp[col + k]; e_sh[k] = g.bnp[g.Cn + col + k]; e_mean[k] = g.bnp[2 * g.Cn + col + k]; e_istd[k] = g.bnp[3 * g.Cn + col + k]; }
    }
    const bf16_t* __restrict__ BnM = (const bf16_t*)g.bn_out;
    for (int id = tid; id < BM * CPRF; id += NT) {
        const int row = id / CPRF, c4 = id - row * CPRF;
        const int il = row / (TW * TH), rr = row - il * (TW * TH);
        const int yy = (by + rr / TW) * g.out_stride + coy, xx = (bx + rr % TW) * g.out_stride + cox, col = n0 + c4 * 4;
        const float4 v4 = *(const float4*)(smem + row * SPF + c4 * 16);
        const long o = (((long)(img + il) * g.Ho + yy) * g.Wo + xx) * g.Cn + col;
        if (bnr) {
            const float4 y4 = *(const float4*)(g.bn_y + o);
            const uint2 m2 = BnM ? *(const uint2*)(BnM + o) : make_uint2(0u, 0u);
            float v[4] = {v4.x, v4.y, v4.z, v4.w};
            const float y[4] = {y4.x, y4.y, y4.z, y4.w};
            const float m[4] = {__uint_as_float(m2.x << 16), __uint_as_float(m2.x & 0xffff0000u), __uint_as_float(m2.y << 16), __uint_as_float(m2.y & 0xffff0000u)};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool dead = BnM ? !(m[k] > 0.f) : !(y[k] * e_sc[k] + e_sh[k] > 0.f);
                v[k] = dead ? 0.f : v[k];
                fs[k] += v[k]; fq[k] += v[k] * ((y[k] - e_mean[k]) * e_istd[k]);
            }
            *(float4*)(g.Out + o) = make_float4(v[0], v[1], v[2], v[3]);
            continue;
        }
        float4 v = v4;
        if (MODE == 0 && g.ep_scale) {
            const float4 sc = *(const float4*)(g.ep_scale + col), sh = *(const float4*)(g.ep_shift + col);
            v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
            if (g.ep_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        }
        if (MODE == 0 && g.out_hi) {
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
    float* part_out = bnr ? g.bn_part : g.stats;
    if (part_out) {
        float* sp = (float*)smem;                          // [NT / CPRF][BN][2], over the consumed staging tile
        const int rg = tid / CPRF, cb = (tid % CPRF) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) { sp[(rg * BN + cb + k) * 2] = fs[k]; sp[(rg * BN + cb + k) * 2 + 1] = fq[k]; }
        __syncthreads();
        for (int c = tid; c < BN; c += NT) {
            float s2 = 0.f, q2 = 0.f;
            for (int r = 0; r < NT / CPRF; ++r) { s2 += sp[(r * BN + c) * 2]; q2 += sp[(r * BN + c) * 2 + 1]; }
            part_out[((long)row_id * g.Cn + n0 + c) * 2] = s2;
            part_out[((long)row_id * g.Cn + n0 + c) * 2 + 1] = q2;
        }
    }
}

template <int BM, int BN, bool G8, int MODE, int NPB = 3>
static int cp_launch(CPArgs& g, hipStream_t st) {
    constexpr int NI = G8 ? BM / 64 : 1, NPIX = NI * (G8 ? 81 : (BM / 16 + 1) * 18), LP = ((NPIX + 7) / 8 + 7) / 8;
    const size_t ring = (size_t)(NPB == 2 ? 3 : 4) * BN * 128 + NPB * LP * 8 * 1024, stage = (size_t)BM * (BN * 4 + 16), part = (size_t)(512 / (BN / 4)) * BN * 8;
    size_t lds = ring > stage ? ring : stage;
    if (part > lds) lds = part;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)convp_kernel<BM, BN, G8, MODE, NPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int blocks = (g.N / NI) * g.tiles_y * g.tiles_x * g.nclass * (g.Cn / BN);
    convp_kernel<BM, BN, G8, MODE, NPB><<<blocks, 512, lds, st>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

static bool cp_off() { static const int off = getenv("AB_CP_OFF") ? atoi(getenv("AB_CP_OFF")) : 0; return off != 0; }
// base grid (the output of the forward, dy of the backward): multiples of 16 x 16, or 8 x 8 with an even batch.  Returns workgroup rows per
// class and n-tile (= BatchNorm partial rows of the forward), 0: not taken.
static int cp_rows(int N, int P, int Q) {
    if (P % 16 == 0 && Q % 16 == 0) return N * (P / 16) * (Q / 16);
    if (P == 8 && Q == 8 && N % 2 == 0) return N / 2;
    return 0;
}

// 3x3 / stride 2 / pad 1 forward: x planes [N, H, W, C], w rows [Cn][3][3][C] (OHWI), out fp32 [N, H/2, W/2, Cn].
int convp_s2fwd_rows(int N, int H, int W, int C, int Cn) {
    // (C = 64, layer 2's entry: two chunks of nine steps per workgroup -- 46.7 us against the tap-by-tap kernel's 47.8: not taken)
    if (cp_off() || C % 32 || C < 128 || Cn % 64 || (H & 1) || (W & 1)) return 0;
    return cp_rows(N, H / 2, W / 2);
}
int convp_s2fwd_run(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, float* out, int N, int H, int W, int C, int Cn,
                    float* stats, hipStream_t st, const float* ep_scale, const float* ep_shift, int ep_relu, void* out_hi, void* out_lo) {
    if (!convp_s2fwd_rows(N, H, W, C, Cn)) return AB_ESHAPE;
    const long delta = (const char*)w_lo - (const char*)w_hi;
    if (delta < 0 || delta >= (1L << 31)) return AB_EINVAL;
    CPArgs g = {};
    g.X[0] = x_hi; g.Xlo[0] = x_lo; g.Wt[0] = w_hi; g.wlo_delta[0] = (unsigned)delta; g.ktot[0] = 9 * C;
    g.X[1] = x_hi; g.Xlo[1] = x_lo; g.Wt[1] = w_hi; g.wlo_delta[1] = (unsigned)delta; g.ktot[1] = 9 * C;
    g.Out = out; g.stats = stats;
    g.ep_scale = ep_scale; g.ep_shift = ep_shift; g.ep_relu = ep_relu; g.out_hi = out_hi; g.out_lo = out_lo;
    g.N = N; g.Hi = H; g.Wi = W; g.C = C; g.Cn = Cn;
    g.in_stride = 2; g.Ho = H / 2; g.Wo = W / 2; g.out_stride = 1; g.nclass = 1;
    const bool g8 = g.Ho == 8;
    g.tiles_y = g8 ? 1 : g.Ho / 16; g.tiles_x = g8 ? 1 : g.Wo / 16;
    // input row 2p + kh - 1: kh = 1 is the even row p (origin 0, patch row 0); kh = 0 / 2 the odd rows p - 1 / p (origin -1, patch rows 0 / 1).
    // The four-tap unit first: the patch of a unit is requested two units ahead, the longest unit hides the most.
    const int order[4][2] = {{1, 1}, {1, 0}, {0, 1}, {0, 0}};
    for (int u = 0; u < 4; ++u) {
        const int sy = order[u][0], sx = order[u][1];
        CPUnit& U = g.unit[0][u];
        U.src = 0; U.wsrc = 0; U.oy0 = sy ? -1 : 0; U.ox0 = sx ? -1 : 0; U.sy = sy; U.sx = sx; U.ntap = 0;
        for (int dh = 0; dh <= sy; ++dh) for (int dw = 0; dw <= sx; ++dw) {
            const int kh = sy ? 2 * dh : 1, kw = sx ? 2 * dw : 1;
            U.koff[U.ntap] = (kh * 3 + kw) * C; U.dhw[U.ntap] = dh * 2 + dw; ++U.ntap;
        }
        U.code = sy && sx ? 4 : (sy ? 2 : (sx ? 3 : 1));
    }
    g.nunit[0] = 4; g.cls_oy[0] = g.cls_ox[0] = 0;
    return g8 ? cp_launch<128, 64, true, 0>(g, st) : cp_launch<256, 64, false, 0>(g, st);
}

// Data gradient of that convolution, optionally with the 1x1 / stride 2 / pad 0 branch of the same input (resnet.py:59-101 backwards: conv1 and
// downsample.0 of a stage's first block): dy planes [N, H/2, W/2, K], wt rows [Cn][3][3][K] ("IHWO"), dy2 / wt2 rows [Cn][K] or NULL,
// dx fp32 [N, H, W, Cn] (every element written).
// (AB_CP_HALF=1: 16 x 16 base grids and larger run 8 x 16 tiles with two patch buffers, two workgroups per CU -- meant to put one workgroup's
// epilogue under the other's K loop; measured 9.124 vs 9.115 ms per step over two alternating pairs, i.e. nothing: off by default)
static bool cp_half() { static const int on = getenv("AB_CP_HALF") ? atoi(getenv("AB_CP_HALF")) : 0; return on != 0; }
int convp_s2dgrad_ok(int N, int H, int W, int Cn, int K) {
    if (cp_off() || K % 32 || K < 64 || Cn % 64 || (H & 1) || (W & 1)) return 0;
    const int r = cp_rows(N, H / 2, W / 2);
    return (r && H / 2 != 8 && cp_half()) ? 2 * r : r;
}
// rows of bn_part a launch with the BatchNorm-backward epilogue writes (one per base tile and parity class); 0: not taken
int convp_s2dgrad_bn_rows(int N, int H, int W, int Cn, int K) {
    static const int off = getenv("AB_CP_BN_OFF") ? atoi(getenv("AB_CP_BN_OFF")) : 0;
    return off ? 0 : 4 * convp_s2dgrad_ok(N, H, W, Cn, K);
}
int convp_s2dgrad_run(const void* dy_hi, const void* dy_lo, const void* wt_hi, const void* wt_lo, const void* dy2_hi, const void* dy2_lo,
                      const void* wt2_hi, const void* wt2_lo, float* dx, int N, int H, int W, int Cn, int K, hipStream_t st,
                      const float* bn_y, const void* bn_out, const float* bnp, float* bn_part) {
    if (!convp_s2dgrad_ok(N, H, W, Cn, K)) return AB_ESHAPE;
    if (bn_y && (!bnp || !bn_part)) return AB_EINVAL;
    const long delta = (const char*)wt_lo - (const char*)wt_hi, delta2 = dy2_hi ? (const char*)wt2_lo - (const char*)wt2_hi : 0;
    if (delta < 0 || delta >= (1L << 31) || delta2 < 0 || delta2 >= (1L << 31)) return AB_EINVAL;
    CPArgs g = {};
    g.X[0] = dy_hi; g.Xlo[0] = dy_lo; g.Wt[0] = wt_hi; g.wlo_delta[0] = (unsigned)delta; g.ktot[0] = 9 * K;
    g.X[1] = dy2_hi ? dy2_hi : dy_hi; g.Xlo[1] = dy2_hi ? dy2_lo : dy_lo;
    g.Wt[1] = dy2_hi ? wt2_hi : wt_hi; g.wlo_delta[1] = dy2_hi ? (unsigned)delta2 : (unsigned)delta; g.ktot[1] = dy2_hi ? K : 9 * K;
    g.Out = dx; g.stats = nullptr;
    g.bn_y = bn_y; g.bn_out = bn_out; g.bnp = bnp; g.bn_part = bn_part;
    g.N = N; g.Hi = H / 2; g.Wi = W / 2; g.C = K; g.Cn = Cn;
    g.in_stride = 1; g.Ho = H; g.Wo = W; g.out_stride = 2; g.nclass = 4;
    const bool g8 = g.Hi == 8, half = !g8 && cp_half();
    g.tiles_y = g8 ? 1 : g.Hi / (half ? 8 : 16); g.tiles_x = g8 ? 1 : g.Wi / 16;
    // dx row 2p + a <- dy row (2p + a + 1 - kh) / 2: a = 0: kh = 1 (row p); a = 1: kh = 2 (row p, patch row 0), kh = 0 (row p + 1, patch row 1).
    // Class order in the grid: (1, 1) first -- its workgroups run four taps per chunk, the others two.
    const int order[4][2] = {{1, 1}, {1, 0}, {0, 1}, {0, 0}};
    for (int c = 0; c < 4; ++c) {
        const int a = order[c][0], b = order[c][1];
        CPUnit& U = g.unit[c][0];
        U.src = 0; U.wsrc = 0; U.oy0 = U.ox0 = 0; U.sy = U.sx = 0; U.ntap = 0;
        for (int dh = 0; dh <= a; ++dh) for (int dw = 0; dw <= b; ++dw) {
            const int kh = a ? 2 - 2 * dh : 1, kw = b ? 2 - 2 * dw : 1;
            U.koff[U.ntap] = (kh * 3 + kw) * K; U.dhw[U.ntap] = dh * 2 + dw; ++U.ntap;
        }
        U.code = a && b ? 4 : (a ? 2 : (b ? 3 : 1));
        g.nunit[c] = 1; g.cls_oy[c] = a; g.cls_ox[c] = b;
        if (!a && !b && dy2_hi) {        // the downsample branch: dx[2p, 2q] += dy2[p, q] . wt2
            CPUnit& V = g.unit[c][1];
            V.src = 1; V.wsrc = 1; V.oy0 = V.ox0 = 0; V.sy = V.sx = 0; V.ntap = 1; V.code = 1; V.koff[0] = 0; V.dhw[0] = 0;
            g.nunit[c] = 2;
        }
    }
    if (half) return cp_launch<128, 64, false, 1, 2>(g, st);
    return g8 ? cp_launch<128, 64, true, 1>(g, st) : cp_launch<256, 64, false, 1>(g, st);
}
