// Generic weight gradient (any tap set / stride), bf16, direct-to-LDS operands:
//
//   dW[co][tap][ci] = sum over output pixels m of dy[m][co] * x[pix(m, tap)][ci]
//
// Same GEMM as wgrad_kernel (conv_wgrad.hip) -- which stages through registers with one __syncthreads per 32 rows and is
// latency-bound -- rebuilt on the scheme of the other fast kernels: both operand tiles (64 reduction rows per step) are
// DMA'd straight into a 3-deep LDS ring (global_load_lds_dwordx4 from inline asm, counted s_waitcnt vmcnt, raw s_barrier),
// and, being reduction-major in HBM, are read as MFMA fragments through the transpose read ds_read_b64_tr_b16.
// A workgroup owns BI output channels x BJ columns of ONE tap (BJ | Cin) over a slice of the pixels; slices write
// separate fp32 slabs that wgrad_reduce sums in fixed order.
//
// LDS image of a tile: row r (a pixel) at byte r*P, P = 2*B (128 or 256 bytes, lane-linear DMA so no padding); the
// 16-byte chunk c of a row is stored at slot c ^ (f(r) << 2) with f(r) = (r >> 1) & 1 for P = 128 and r & 3 for P = 256:
// the four consecutive rows x 64 bytes a transpose read touches then sit on four distinct 64-byte bank groups.
// Pixel -> (image, row, col) uses host-made multiply-shift reciprocals (exact and overflow-free for m < 2^21).
#include "conv_common.h"

typedef short short4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4_t lds_short4;

static __device__ uint4 wg2_zero_page[2];

struct Wg2Args {
    const void* X; const void* DY; float* slabs;
    const void* X_lo; const void* DY_lo;         // split-bf16 (X3) launches: low-order planes
    int N, Ha, Wa, Ca;          // x tensor
    int P, Q, Cout;             // dy tensor [N,P,Q,Cout]
    int stride;
    int ntaps, Cin, jtot;
    int M, rows_per_slice;
    int xcd_map;                // workgroups renumbered slice-major per XCD (conv_common.h)
    unsigned long long magic_pq, magic_q;     // floor(2^42 / d) + 1
    int8_t dh[16], dw[16];
};

__device__ __forceinline__ int fastdiv(int n, unsigned long long magic) {
    return (int)(((unsigned long long)(unsigned)n * magic) >> 42);
}

template <int B> __device__ __forceinline__ int swz(int row) { return (B == 64) ? ((row >> 1) & 1) : (row & 3); }

__device__ __forceinline__ uint4 wg2_tr_pair(unsigned lo_addr, unsigned hi_addr) {
    short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(size_t)lo_addr);
    short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(size_t)hi_addr);
    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

// STEM: x is the zero-bordered NHWC4 image and a dW row is [8 kernel rows (7 + 1 pad)][8 px][4 ch] = 256 columns (see
// ab_conv2d_stem_fwd); BJ = 256 covers all of them, the 64-byte segment of kernel row t comes from image row 2p + t.
// X3 = 1: split-bf16 operands (conv3x3.hip): 32 reduction rows per step, each operand tile staged once per plane
// ([dy hi][dy lo][x hi][x lo]); the DMA instruction index runs over (plane, row group).
// X3 = 2 (stem only): dy as (hi, lo) planes, the image as ONE plane of odd integers n = 2 v - 255 (exact in bf16: AB_DT_U8N, stem_halo.hip) --
// two MFMAs per fragment pair (dy_hi . n, dy_lo . n) instead of three, no lo plane of the image staged, the factor 1 / 510 in the slab store.
// NBUF = 2 (the 256 x 256 split-bf16 tile: two 64 KB stages): one step in flight instead of two.
template <int BI, int BJ, bool STEM = false, int WI = 2, int WJ = 2, int X3 = 0, int NBUF = 3>
__global__ __launch_bounds__(64 * WI * WJ) void wgrad_gemm2_kernel(Wg2Args g) {
    constexpr int NW = WI * WJ;                             // 4 or 8 waves (the LDS fill rate scales with the waves issuing loads)
    constexpr int BR = X3 ? 32 : 64;                        // reduction rows per step
    constexpr int NPL = X3 ? 2 : 1;                         // operand planes
    constexpr int NPLB = X3 == 1 ? 2 : 1;                   // ... of the x operand
    constexpr int PA = BI * 2, PB = BJ * 2;                 // row pitches (bytes)
    constexpr int RA = 1024 / PA, RB = 1024 / PB;           // rows per 1-KiB DMA instruction
    constexpr int IA1 = BR / RA, IB1 = BR / RB;             // instructions per tile plane
    constexpr int IA = NPL * IA1, IB = NPLB * IB1;          // ... over the planes
    constexpr int LA = IA / NW, LB = IB / NW;               // per wave
    static_assert(IA % NW == 0 && IB % NW == 0, "tile rows must split evenly over the waves");
    constexpr int APL = BR * PA, BPL = BR * PB;             // bytes of one plane of a tile
    constexpr int ABYTES = NPL * APL, STAGE = NPL * APL + NPLB * BPL;
    constexpr int PF = NBUF - 1;                            // steps in flight ahead of the one being multiplied
    constexpr bool PARTIAL_I = BI == 256 && BJ == 256;      // only these launches may have a partial last channel tile (Cout = 704)
    constexpr int TI = BI / WI / 32, TJ = BJ / WJ / 32;     // 32x32 tiles per wave (waves WI x WJ)
    __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * STAGE];
    __shared__ int s_xoff[2][BR];                           // element offset of each row's input pixel (-1: padding / past the end)
    const unsigned lds0 = lds_addr_of(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_i = wave / WJ, wave_j = wave % WJ;
    const int tiles_j = g.jtot / BJ;
    int bx, by; xcd_slice_major(g.xcd_map, bx, by);
    const int tile_i = bx / tiles_j, tile_j = bx - tile_i * tiles_j;
    const int i0 = tile_i * BI, j0 = tile_j * BJ;
    const int tap = STEM ? 0 : j0 / g.Cin, ci0 = STEM ? 0 : j0 - tap * g.Cin;
    const int dh = STEM ? 0 : g.dh[tap], dw = STEM ? 0 : g.dw[tap];   // read once, before any DMA is in flight
    const int r_begin = by * g.rows_per_slice;
    const int r_end = min(g.M, r_begin + g.rows_per_slice);
    const int PQ = g.P * g.Q;
    const bf16_t* __restrict__ X = (const bf16_t*)g.X;
    const bf16_t* __restrict__ DY = (const bf16_t*)g.DY;
    const bf16_t* __restrict__ Xl = (const bf16_t*)g.X_lo;
    const bf16_t* __restrict__ DYl = (const bf16_t*)g.DY_lo;
    const bf16_t* zp = (const bf16_t*)wg2_zero_page;

    // per-lane DMA assignment: instruction ii covers rows ii*R .. ii*R+R-1; lane -> (row in instr, slot); the lane fetches
    // the logical chunk that belongs in its slot
    int a_row[LA], a_col[LA], b_row[LB], b_col[LB];
    bool a_pl[LA], b_pl[LB];                                // X3: the instruction fills the lo plane
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        constexpr int LPR = PA / 16;                        // lanes (chunks) per row
        const int ii = wave * LA + j;
        a_pl[j] = ii >= IA1;
        int r = (ii % IA1) * RA + lane / LPR, slot = lane % LPR;
        a_row[j] = r; a_col[j] = (slot ^ (swz<BI>(r) << 2)) * 8;
    }
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        constexpr int LPR = PB / 16;
        const int ii = wave * LB + j;
        b_pl[j] = ii >= IB1;
        int r = (ii % IB1) * RB + lane / LPR, slot = lane % LPR;
        int c = slot ^ (swz<BJ>(r) << 2);
        b_row[j] = r; b_col[j] = STEM ? (c >> 2) * g.Wa * 4 + (c & 3) * 8 : c * 8;
    }
    // The pixel -> input-offset arithmetic (two divisions per row) is done ONCE per row by the first wave and handed to the
    // loaders through a small LDS table, two steps ahead: as per-load VALU work it outweighed the MFMAs of a step.
    auto make_table = [&](int rbase, int slot) {
        if (tid < BR) {
            int m = rbase + tid, off = -1;
            if (m < r_end) {
                int n = fastdiv(m, g.magic_pq), rem = m - n * PQ;
                int p = fastdiv(rem, g.magic_q), q = rem - p * g.Q;
                int hi = p * g.stride + dh, wi = q * g.stride + dw;
                if (STEM || ((unsigned)hi < (unsigned)g.Ha && (unsigned)wi < (unsigned)g.Wa)) off = ((n * g.Ha + hi) * g.Wa + wi) * g.Ca + ci0;
            }
            s_xoff[slot][tid] = off;
        }
    };
    auto issue = [&](int rbase, int buf, int slot) {
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            int m = rbase + a_row[j];
            const bool in = m < r_end && (!PARTIAL_I || i0 + a_col[j] < g.Cout);
            const bf16_t* src = in ? ((X3 && a_pl[j]) ? DYl : DY) + ((long)m * g.Cout + i0 + a_col[j]) : zp;
            glds16(src, __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE + (wave * LA + j) * 1024));
        }
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            int off = s_xoff[slot][b_row[j]];
            const bf16_t* src = off >= 0 ? ((X3 == 1 && b_pl[j]) ? Xl : X) + (off + b_col[j]) : zp;
            glds16(src, __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE + ABYTES + (wave * LB + j) * 1024));
        }
    };

    // fragment bases (transpose-read geometry, see conv_wgrad.hip): 16-lane group grp, k row = (grp>>1)*8 + (l16>>2) [+4],
    // columns (grp&1)*16 + (l16&3)*4 .. +3 of the wave's 32-wide tile
    const int grp = lane >> 4, l16 = lane & 15;
    const int krow = (grp >> 1) * 8 + (l16 >> 2), csub = (grp & 1) * 16 + (l16 & 3) * 4;
    unsigned a_base[TI][2], b_base[TJ][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = krow + 4 * h;
#pragma unroll
        for (int a = 0; a < TI; ++a) {
            int col = (wave_i * TI + a) * 32 + csub;
            a_base[a][h] = lds0 + r * PA + (((col >> 3) ^ (swz<BI>(r) << 2)) << 4) + (col & 7) * 2;
        }
#pragma unroll
        for (int b = 0; b < TJ; ++b) {
            int col = (wave_j * TJ + b) * 32 + csub;
            b_base[b][h] = lds0 + ABYTES + r * PB + (((col >> 3) ^ (swz<BJ>(r) << 2)) << 4) + (col & 7) * 2;
        }
    }

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    constexpr int L = LA + LB;
    const int nsteps = (r_end - r_begin + BR - 1) / BR;
    // tables: step s uses slot s & 1; with PF steps in flight the table of step s+PF+1 is written during step s (its slot was
    // last read at step s-1's issue, i.e. before this step's barrier) and read at step s+1's issue (after the next barrier)
    make_table(r_begin, 0);
    make_table(r_begin + BR, 1);
    __syncthreads();
    if (nsteps > 0) issue(r_begin, 0, 0);
    if (PF > 1 && nsteps > 1) issue(r_begin + BR, 1, 1);
    __syncthreads();                                         // the issued tables are consumed (s_waitcnt lgkmcnt is implied by use)
    if (PF > 1) make_table(r_begin + 2 * BR, 0);
    int cur = 0;
    for (int step = 0; step < nsteps; ++step) {
        if (PF > 1 && step + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * L) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (step + PF < nsteps) { int nb = cur + PF; if (nb >= NBUF) nb -= NBUF; issue(r_begin + (step + PF) * BR, nb, (step + PF) & 1); }
        make_table(r_begin + (step + PF + 1) * BR, (step + PF + 1) & 1);
        const unsigned so = cur * STAGE;
#pragma unroll
        for (int s = 0; s < BR / 16; ++s) {
            uint4 fa[TI], fb[TJ];
#pragma unroll
            for (int a = 0; a < TI; ++a) fa[a] = wg2_tr_pair(a_base[a][0] + so + s * 16 * PA, a_base[a][1] + so + s * 16 * PA);
#pragma unroll
            for (int b = 0; b < TJ; ++b) fb[b] = wg2_tr_pair(b_base[b][0] + so + s * 16 * PB, b_base[b][1] + so + s * 16 * PB);
#pragma unroll
            for (int a = 0; a < TI; ++a)
#pragma unroll
                for (int b = 0; b < TJ; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[a]),
                                                                       __builtin_bit_cast(bf16x8, fb[b]), acc[a][b], 0, 0, 0);
            if constexpr (X3 == 2) {
                uint4 fal[TI];
#pragma unroll
                for (int a = 0; a < TI; ++a) fal[a] = wg2_tr_pair(a_base[a][0] + so + APL + s * 16 * PA, a_base[a][1] + so + APL + s * 16 * PA);
#pragma unroll
                for (int a = 0; a < TI; ++a)
#pragma unroll
                    for (int b = 0; b < TJ; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fal[a]),
                                                                           __builtin_bit_cast(bf16x8, fb[b]), acc[a][b], 0, 0, 0);
            } else if constexpr (X3) {
                uint4 fal[TI], fbl[TJ];
#pragma unroll
                for (int a = 0; a < TI; ++a) fal[a] = wg2_tr_pair(a_base[a][0] + so + APL + s * 16 * PA, a_base[a][1] + so + APL + s * 16 * PA);
#pragma unroll
                for (int b = 0; b < TJ; ++b) fbl[b] = wg2_tr_pair(b_base[b][0] + so + BPL + s * 16 * PB, b_base[b][1] + so + BPL + s * 16 * PB);
#pragma unroll
                for (int a = 0; a < TI; ++a)
#pragma unroll
                    for (int b = 0; b < TJ; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[a]),
                                                                           __builtin_bit_cast(bf16x8, fbl[b]), acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fal[a]),
                                                                           __builtin_bit_cast(bf16x8, fb[b]), acc[a][b], 0, 0, 0);
                    }
            }
        }
        if (++cur == NBUF) cur = 0;
    }
    float* out = g.slabs + (long)by * g.Cout * g.jtot;
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = i0 + (wave_i * TI + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                int col = j0 + (wave_j * TJ + b) * 32 + (lane & 31);
                if (!PARTIAL_I || row < g.Cout) out[(long)row * g.jtot + col] = X3 == 2 ? acc[a][b][r] * (1.0f / 510.0f) : acc[a][b][r];
            }
}

static int wg2_xcd_map() { static const int v = getenv("AB_WG_XCD") ? atoi(getenv("AB_WG_XCD")) : 1; return v; }

static void wg2_pick(int M, int Cout, int Cin, int jtot, int* bi, int* bj, int* ns, int* rows) {
    *bi = (Cout % 128 == 0) ? 128 : 64;
    *bj = (Cin % 256 == 0 && *bi == 128) ? 256 : (Cin % 128 == 0) ? 128 : 64;
    long tiles = (long)(Cout / *bi) * (jtot / *bj);
    // workgroups that fit on a CU side by side (3 x 64 x (BI+BJ) x 2 bytes of LDS ring each): aim at one full wave of them
    const int per_cu = (160 * 1024) / (3 * 64 * (*bi + *bj) * 2 + 1024);
    static int target_env = getenv("AB_WG2_TARGET") ? atoi(getenv("AB_WG2_TARGET")) : 0;
    const int target = target_env ? target_env : 256 * (per_cu < 1 ? 1 : per_cu > 2 ? 2 : per_cu);
    int want = (int)((target + tiles - 1) / tiles);
    int maxs = (M + 511) / 512;                              // at least 512 pixels (8 steps) per slice
    int n = want < 1 ? 1 : want; if (n > maxs) n = maxs; if (n < 1) n = 1; if (n > 512) n = 512;
    int r = (M + n - 1) / n; r = (r + 63) / 64 * 64;
    *ns = (M + r - 1) / r; *rows = r;
}

// slabs needed (0: shape not handled by this kernel)
int wgrad_gemm2_slices(int M, int Cout, int Cin, int ntaps) {
    if (Cout % 64 || Cin % 64 || ntaps > 16 || M >= (1 << 21) || getenv("AB_WGRAD2_OFF")) return 0;
    int bi, bj, ns, rows; wg2_pick(M, Cout, Cin, ntaps * Cin, &bi, &bj, &ns, &rows);
    return ns;
}

int wgrad_gemm2_run(const void* x, const void* dy, float* slabs, int N, int H, int W, int Cin, int Cout, int kh, int kw,
                    int stride, int pad, hipStream_t st) {
    Wg2Args g = {};
    g.X = x; g.DY = dy; g.slabs = slabs;
    g.N = N; g.Ha = H; g.Wa = W; g.Ca = Cin;
    g.P = (H + 2 * pad - kh) / stride + 1; g.Q = (W + 2 * pad - kw) / stride + 1; g.Cout = Cout;
    g.stride = stride; g.ntaps = kh * kw; g.Cin = Cin; g.jtot = kh * kw * Cin; g.M = N * g.P * g.Q;
    if (!wgrad_gemm2_slices(g.M, Cout, Cin, g.ntaps)) return AB_ESHAPE;
    for (int i = 0; i < kh; ++i) for (int j = 0; j < kw; ++j) { g.dh[i * kw + j] = (int8_t)(i - pad); g.dw[i * kw + j] = (int8_t)(j - pad); }
    g.magic_pq = (1ull << 42) / (unsigned long long)(g.P * g.Q) + 1;
    g.magic_q = (1ull << 42) / (unsigned long long)g.Q + 1;
    int bi, bj, ns, rows; wg2_pick(g.M, Cout, Cin, g.jtot, &bi, &bj, &ns, &rows);
    g.rows_per_slice = rows;
    g.xcd_map = wg2_xcd_map();
    dim3 grid((Cout / bi) * (g.jtot / bj), ns);
    static const int w8 = getenv("AB_WG2_W8") ? atoi(getenv("AB_WG2_W8")) : 1;
    if (bi == 128 && bj == 256) { if (w8) wgrad_gemm2_kernel<128, 256, false, 2, 4><<<grid, 512, 0, st>>>(g); else wgrad_gemm2_kernel<128, 256><<<grid, 256, 0, st>>>(g); }
    else if (bi == 64 && bj == 256) wgrad_gemm2_kernel<64, 256><<<grid, 256, 0, st>>>(g);
    else if (bi == 128 && bj == 128) { if (w8) wgrad_gemm2_kernel<128, 128, false, 2, 4><<<grid, 512, 0, st>>>(g); else wgrad_gemm2_kernel<128, 128><<<grid, 256, 0, st>>>(g); }
    else if (bi == 128 && bj == 64) { if (w8) wgrad_gemm2_kernel<128, 64, false, 4, 2><<<grid, 512, 0, st>>>(g); else wgrad_gemm2_kernel<128, 64><<<grid, 256, 0, st>>>(g); }
    else if (bi == 64 && bj == 128) { if (w8) wgrad_gemm2_kernel<64, 128, false, 2, 4><<<grid, 512, 0, st>>>(g); else wgrad_gemm2_kernel<64, 128><<<grid, 256, 0, st>>>(g); }
    else wgrad_gemm2_kernel<64, 64><<<grid, 256, 0, st>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// upper bound of the slab count for a (M, Cout, jtot) problem whatever its tap split (workspace sizing)
int wgrad_gemm2_max_slices(int M, int Cout, int jtot) {
    if (Cout % 64 || jtot % 64 || M >= (1 << 21)) return 0;
    long ti = Cout / 128 > 0 ? Cout / 128 : 1, tj = jtot / 128 > 0 ? jtot / 128 : 1;
    const int target = 512;
    long want = (target + ti * tj - 1) / (ti * tj), maxs = (M + 511) / 512;
    long n = want < maxs ? want : maxs; if (n < 1) n = 1; if (n > 512) n = 512;
    return (int)n + 1;
}

// Stem weight gradient (xpad: zero-bordered NHWC4 image [N, H+6, W+8, 4]; dy [N, H/2, W/2, Cout]); slabs hold
// [Cout][8][8][4] rows.  Returns the slab count, or AB_ESHAPE (< 0) when not handled.
int wgrad_gemm2_stem_slices(int N, int H, int W, int Cout) {
    long M = (long)N * (H / 2) * (W / 2);
    if (Cout % 64 || M >= (1 << 21) || getenv("AB_WGRAD2_OFF")) return 0;
    long tiles = Cout / 64;
    int n = (int)((256 + tiles - 1) / tiles);
    int maxs = (int)((M + 1023) / 1024);
    if (n > maxs) n = maxs; if (n < 1) n = 1;
    int r = (int)((M + n - 1) / n); r = (r + 63) / 64 * 64;
    return (int)((M + r - 1) / r);
}

int wgrad_gemm2_stem_run(const void* xpad, const void* dy, float* slabs, int N, int H, int W, int Cout, hipStream_t st) {
    int ns = wgrad_gemm2_stem_slices(N, H, W, Cout);
    if (!ns) return AB_ESHAPE;
    Wg2Args g = {};
    g.X = xpad; g.DY = dy; g.slabs = slabs;
    g.N = N; g.Ha = H + 6; g.Wa = W + 8; g.Ca = 4;
    g.P = H / 2; g.Q = W / 2; g.Cout = Cout; g.stride = 2; g.ntaps = 8; g.Cin = 256; g.jtot = 256; g.M = N * g.P * g.Q;
    g.magic_pq = (1ull << 42) / (unsigned long long)(g.P * g.Q) + 1;
    g.magic_q = (1ull << 42) / (unsigned long long)g.Q + 1;
    int r = (g.M + ns - 1) / ns; r = (r + 63) / 64 * 64;
    g.rows_per_slice = r;
    g.xcd_map = wg2_xcd_map();
    dim3 grid(Cout / 64, ns);
    static const int w8 = getenv("AB_WG2_W8") ? atoi(getenv("AB_WG2_W8")) : 1;
    if (w8) wgrad_gemm2_kernel<64, 256, true, 2, 4><<<grid, 512, 0, st>>>(g);
    else wgrad_gemm2_kernel<64, 256, true><<<grid, 256, 0, st>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// ---- split-bf16 ("bf16x3") launches (x / dy as bf16 plane pairs).  Tiles up to 128 x 128: three ring stages of both
// planes of a 32-row step are 96 KB.
static int wg2x_mode() { static const int v = getenv("AB_WG2X_MODE") ? atoi(getenv("AB_WG2X_MODE")) : 3; return v; }

static void wg2x_pick(int M, int Cout, int Cin, int jtot, int* bi, int* bj, int* ns, int* rows) {
    *bi = (Cout % 128 == 0) ? 128 : 64;
    *bj = (Cin % 128 == 0) ? 128 : 64;
    if (wg2x_mode() >= 2 && Cout % 256 == 0 && *bj == 128) *bi = 256;
    // 256 x 256 (two stages of 64 KB, one step in flight; the last channel tile may be partial) pays from 16k pixels up: the
    // transposed 256->256 layer 114 -> 102 us, the final 1x1 layer 104 -> 93 us, but 4 096-pixel launches 44 -> 50 us
    if (wg2x_mode() == 3 && Cout >= 256 && Cin % 256 == 0 && M >= 16384) { *bi = 256; *bj = 256; }
    long tiles = (long)((Cout + *bi - 1) / *bi) * (jtot / *bj);
    int maxs = (M + 255) / 256;                              // at least 256 pixels (8 steps) per slice
    if (maxs > 512) maxs = 512;
    // The chip retires workgroups 256 at a time (one per CU, or two side by side at half the rate each): tiles x slices just over
    // a multiple of 256 costs a whole extra round (18 tiles x 15 slices = 270 workgroups ran as two rounds of 1 120 pixels where
    // 14 slices run as one round of 1 184).  Pick the slice count with the fewest pixel rows on the busiest CU; WG_ROWS prices a
    // workgroup's fixed part (ring fill, 64 KB slab store) in pixel rows.
    static const int legacy = getenv("AB_WG2X_LEGACY") ? atoi(getenv("AB_WG2X_LEGACY")) : 0;
    const long WG_ROWS = 96;
    int n = 1; long best = -1;
    if (legacy) { n = (int)((256 + tiles - 1) / tiles); if (n > maxs) n = maxs; if (n < 1) n = 1; }
    else for (int c = 1; c <= maxs && tiles * c <= 2048; ++c) {
        long r = ((M + c - 1) / c + 31) / 32 * 32, rounds = (tiles * c + 255) / 256;
        long cost = rounds * (r + WG_ROWS);
        if (best < 0 || cost * 100 < best * 98) { best = cost; n = c; }     // more slices only for 2 % or more
    }
    int r = (M + n - 1) / n; r = (r + 31) / 32 * 32;
    *ns = (M + r - 1) / r; *rows = r;
}

int wgrad_gemm2_x3_slices(int M, int Cout, int Cin, int ntaps) {
    if (Cout % 64 || Cin % 64 || ntaps > 16 || M >= (1 << 21)) return 0;
    int bi, bj, ns, rows; wg2x_pick(M, Cout, Cin, ntaps * Cin, &bi, &bj, &ns, &rows);
    return ns;
}

int wgrad_gemm2_x3_run(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, float* slabs, int N, int H, int W,
                       int Cin, int Cout, int kh, int kw, int stride, int pad, hipStream_t st) {
    Wg2Args g = {};
    g.X = x_hi; g.X_lo = x_lo; g.DY = dy_hi; g.DY_lo = dy_lo; g.slabs = slabs;
    g.N = N; g.Ha = H; g.Wa = W; g.Ca = Cin;
    g.P = (H + 2 * pad - kh) / stride + 1; g.Q = (W + 2 * pad - kw) / stride + 1; g.Cout = Cout;
    g.stride = stride; g.ntaps = kh * kw; g.Cin = Cin; g.jtot = kh * kw * Cin; g.M = N * g.P * g.Q;
    if (!wgrad_gemm2_x3_slices(g.M, Cout, Cin, g.ntaps)) return AB_ESHAPE;
    for (int i = 0; i < kh; ++i) for (int j = 0; j < kw; ++j) { g.dh[i * kw + j] = (int8_t)(i - pad); g.dw[i * kw + j] = (int8_t)(j - pad); }
    g.magic_pq = (1ull << 42) / (unsigned long long)(g.P * g.Q) + 1;
    g.magic_q = (1ull << 42) / (unsigned long long)g.Q + 1;
    int bi, bj, ns, rows; wg2x_pick(g.M, Cout, Cin, g.jtot, &bi, &bj, &ns, &rows);
    g.rows_per_slice = rows;
    g.xcd_map = wg2_xcd_map();
    dim3 grid(((Cout + bi - 1) / bi) * (g.jtot / bj), ns);
    if (bi == 256 && bj == 256) wgrad_gemm2_kernel<256, 256, false, 4, 2, 1, 2><<<grid, 512, 0, st>>>(g);
    else if (bi == 256 && bj == 128) wgrad_gemm2_kernel<256, 128, false, 4, 2, 1><<<grid, 512, 0, st>>>(g);
    else if (bi == 128 && bj == 128 && wg2x_mode() == 1) wgrad_gemm2_kernel<128, 128, false, 2, 2, 1><<<grid, 256, 0, st>>>(g);
    else if (bi == 128 && bj == 128) wgrad_gemm2_kernel<128, 128, false, 2, 4, 1><<<grid, 512, 0, st>>>(g);
    else if (bi == 128 && bj == 64) wgrad_gemm2_kernel<128, 64, false, 4, 2, 1><<<grid, 512, 0, st>>>(g);
    else if (bi == 64 && bj == 128) wgrad_gemm2_kernel<64, 128, false, 2, 4, 1><<<grid, 512, 0, st>>>(g);
    else wgrad_gemm2_kernel<64, 64, false, 2, 2, 1><<<grid, 256, 0, st>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// split-bf16 stem weight gradient (image and dy as plane pairs); slabs as wgrad_gemm2_stem_run
int wgrad_gemm2_x3_stem_run(const void* xpad_hi, const void* xpad_lo, const void* dy_hi, const void* dy_lo, float* slabs, int N,
                            int H, int W, int Cout, hipStream_t st) {
    int ns = wgrad_gemm2_stem_slices(N, H, W, Cout);
    if (!ns) return AB_ESHAPE;
    Wg2Args g = {};
    g.X = xpad_hi; g.X_lo = xpad_lo; g.DY = dy_hi; g.DY_lo = dy_lo; g.slabs = slabs;
    g.N = N; g.Ha = H + 6; g.Wa = W + 8; g.Ca = 4;
    g.P = H / 2; g.Q = W / 2; g.Cout = Cout; g.stride = 2; g.ntaps = 8; g.Cin = 256; g.jtot = 256; g.M = N * g.P * g.Q;
    g.magic_pq = (1ull << 42) / (unsigned long long)(g.P * g.Q) + 1;
    g.magic_q = (1ull << 42) / (unsigned long long)g.Q + 1;
    int r = (g.M + ns - 1) / ns; r = (r + 63) / 64 * 64;
    g.rows_per_slice = r;
    g.xcd_map = wg2_xcd_map();
    dim3 grid(Cout / 64, ns);
    if (!xpad_lo) wgrad_gemm2_kernel<64, 256, true, 2, 4, 2><<<grid, 512, 0, st>>>(g);      // the integer image plane (AB_DT_U8N)
    else wgrad_gemm2_kernel<64, 256, true, 2, 4, 1><<<grid, 512, 0, st>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
