// Weight gradient of a 3x3 / stride 1 / pad 1 convolution, all nine taps per workgroup (bf16, gfx950).
//
//   dW[co][kh][kw][ci] = sum over output pixels (n,y,x) of dy[n,y,x,co] * x[n,y+kh-1,x+kw-1,ci]
//
// The tap-by-tap kernel (conv_wgrad.hip) re-reads dy and the input once per tap and is bound by L2->LDS load
// throughput.  Here a workgroup owns a 64(co) x 64(ci) x 9(tap) block of dW (144 f32 accumulators per lane) and walks a
// range of "bands" (TH full image rows, TH*W = 128 or 64 pixels): per band it loads the dy band and the input patch
// with its halo ONCE (direct-to-LDS, double buffered, counted vmcnt) and feeds all nine taps from the patch by
// shifting the fragment address.  Both MFMA operands are reduction-major in HBM, so fragments come through the
// gfx950 transpose read ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group), as in conv_wgrad.hip.
//
// LDS images (lane-linear LDS-DMA fills; swizzle on the source side): pixel q at byte q*128, logical 16-byte chunk c at
// slot c ^ (((q >> 1) & 1) << 2)  -- swapping the 64-byte halves of every other pixel pair puts the four consecutive
// pixels a transpose read touches on four distinct 64-byte bank groups.  The patch row pitch is W+4 pixels (a multiple
// of 4) so that tap / k-slice displacements never change bit 1 of the pixel index: every fragment read is then
// "lane base register + compile-time immediate".
//
// Partial results of the pixel slices go to separate slabs and are summed in fixed order by wgrad_reduce (no atomics).
#include "conv_common.h"
#include <type_traits>

typedef short short4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4_t lds_short4;

static __device__ uint4 wg3_zero_page[2];

struct Wg3Args {
    const void* X; const void* DY; float* slabs;
    const void* X_lo; const void* DY_lo;         // split-bf16 (X3) launches: low-order planes
    int N, H, Cin, Cout;
    int bands_per_slice, nbands;
    int xcd_map;                // workgroups renumbered slice-major per XCD (conv_common.h)
    // Grouped launch (round 6): up to eight SAME-SHAPE layers in one grid.  Slice by of the grid belongs to problem by / ns_per (0: the fields
    // above, 1..7: entry - 1 of the arrays below) and is that problem's pixel slice by % ns_per.  With the launch held at one workgroup per
    // CU, G problems get 1 / G of the slices each: every workgroup works through G times the pixels before it writes its 147 KB partial
    // tile, so the slab bytes written here and read back by the reduction -- 37.7 MB per LAYER at 256 workgroups -- are paid per GROUP.
    int ns_per;                 // 0: one problem
    const void* Xg[7]; const void* Xg_lo[7]; const void* DYg[7]; const void* DYg_lo[7]; float* slabsg[7];
};

__device__ __forceinline__ uint4 tr_pair(unsigned addr_lo, unsigned addr_hi) {
    short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(size_t)addr_lo);
    short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(size_t)addr_hi);
    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

// NW = 4: every wave (2 co-halves x 2 ci-halves) accumulates all nine taps.  NW = 8: the taps are split 5 + 4 over two
// groups of four waves -- twice the waves issue the band / patch DMA (the L2->LDS fill rate scales with the number of
// issuing waves, tools/probe_fill.hip) and each holds 80 instead of 144 accumulators.
// X3 = 1: split-bf16 operands (conv3x3.hip): the band and the patch are staged once per plane ([dy hi][dy lo][x hi][x lo],
// bands of 64 pixels so that two stages still fit) and every (dy, x) fragment pair feeds hi*hi + hi*lo + lo*hi.
// PIN (X3 only, round 5): the transposed fragment reads of tap t + 1 (and of the next k-slice's dy fragments) are issued in front of the MFMAs of
// tap t and held there with sched_barrier(0) -- hipcc on its own issues a tap's reads right in front of its first MFMA (gemm_rw.hip's finding).
template <int W, int TH, int NW = 4, int X3 = 0, bool PIN = false>
__global__ __launch_bounds__(64 * NW) void wgrad3x3_kernel(Wg3Args g) {
    constexpr int BP = TH * W;                       // band pixels (multiple of 16)
    constexpr int KS = BP / 16;                      // k-slices per band
    constexpr int PW = W + 4, PH = TH + 2;           // patch: 1-pixel halo (+2 slack columns so PW % 4 == 0)
    constexpr int NPIX = PH * PW;
    constexpr int IA = BP / 8, LA = (IA + NW - 1) / NW;    // dy-band fills: total / per wave
    constexpr int IX = (NPIX + 7) / 8, LX = (IX + NW - 1) / NW;
    constexpr int ABYTES = LA * NW * 1024, XBYTES = LX * NW * 1024;
    constexpr int NACC = NW == 8 ? 5 : 9;
    constexpr int NPL = X3 ? 2 : 1;                  // operand planes
    constexpr int BUF = NPL * (ABYTES + XBYTES);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = lds_addr_of(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_ci = g.Cin / 64;
    int bx, by; xcd_slice_major(g.xcd_map, bx, by);
    const int tile_co = bx / tiles_ci, tile_ci = bx - tile_co * tiles_ci;
    const int co0 = tile_co * 64, ci0 = tile_ci * 64;
    const int wco = (wave & 1) * 32, wci = ((wave >> 1) & 1) * 32;
    const int tg = wave >> 2;                                // tap group (NW = 8): 0 -> taps 0..4, 1 -> taps 5..8
    const int t0 = (NW == 8 && tg) ? 5 : 0, ntap = NW == 8 ? (tg ? 4 : 5) : 9;
    int prob = 0;
    if (g.ns_per) { prob = by / g.ns_per; by -= prob * g.ns_per; }
    const int band_begin = by * g.bands_per_slice;
    const int band_end = min(g.nbands, band_begin + g.bands_per_slice);
    const int bands_per_img = g.H / TH;
    // (uniform selects, not a dynamically indexed kernarg array: that would become a vector load)
    const void* pX = g.X; const void* pXl = g.X_lo; const void* pDY = g.DY; const void* pDYl = g.DY_lo; float* pslabs = g.slabs;
#pragma unroll
    for (int k = 0; k < 7; ++k)
        if (prob == k + 1) { pX = g.Xg[k]; pXl = g.Xg_lo[k]; pDY = g.DYg[k]; pDYl = g.DYg_lo[k]; pslabs = g.slabsg[k]; }
    const bf16_t* __restrict__ X = (const bf16_t*)pX;
    const bf16_t* __restrict__ DY = (const bf16_t*)pDY;
    const bf16_t* __restrict__ Xl = (const bf16_t*)pXl;
    const bf16_t* __restrict__ DYl = (const bf16_t*)pDYl;
    const bf16_t* zp = (const bf16_t*)wg3_zero_page;

    // ---- per-lane fill assignment
    long a_off[LA]; bool a_ok[LA];
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        int ii = wave * LA + j;
        int p = ii * 8 + (lane >> 3);
        a_ok[j] = ii < IA;
        int c = (lane & 7) ^ (((p >> 1) & 1) << 2);
        a_off[j] = (long)p * g.Cout + co0 + c * 8;                 // + band pixel base * Cout
    }
    int x_pr[LX], x_pc[LX], x_c[LX]; bool x_in[LX];
#pragma unroll
    for (int j = 0; j < LX; ++j) {
        int ii = wave * LX + j;
        int q = ii * 8 + (lane >> 3);
        x_pr[j] = q / PW; x_pc[j] = q - x_pr[j] * PW;
        x_in[j] = (ii < IX) && (q < NPIX) && x_pc[j] >= 1 && x_pc[j] <= W;
        x_c[j] = ((lane & 7) ^ (((q >> 1) & 1) << 2)) * 8;
    }
    auto issue_band = [&](int band, int buf) {
        const int img = band / bands_per_img, y0 = (band - img * bands_per_img) * TH;
        const long pix0 = ((long)img * g.H + y0) * W;
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            const int ii = wave * LA + j;
            glds16(a_ok[j] ? (const void*)(DY + pix0 * g.Cout + a_off[j]) : (const void*)zp,
                   __builtin_amdgcn_readfirstlane(lds0 + buf * BUF + ii * 1024));
            if constexpr (X3)
                glds16(a_ok[j] ? (const void*)(DYl + pix0 * g.Cout + a_off[j]) : (const void*)zp,
                       __builtin_amdgcn_readfirstlane(lds0 + buf * BUF + ABYTES + ii * 1024));
        }
#pragma unroll
        for (int j = 0; j < LX; ++j) {
            const int ii = wave * LX + j;
            int y = y0 + x_pr[j] - 1;
            bool ok = x_in[j] && (unsigned)y < (unsigned)g.H;
            const long xo = (((long)img * g.H + y) * W + (x_pc[j] - 1)) * g.Cin + ci0 + x_c[j];
            glds16(ok ? X + xo : zp, __builtin_amdgcn_readfirstlane(lds0 + buf * BUF + NPL * ABYTES + ii * 1024));
            if constexpr (X3)
                glds16(ok ? Xl + xo : zp, __builtin_amdgcn_readfirstlane(lds0 + buf * BUF + NPL * ABYTES + XBYTES + ii * 1024));
        }
    };

    // ---- per-lane fragment bases (transpose-read geometry: 16-lane group grp, row = k index, 4 columns per lane)
    const int grp = lane >> 4, l16 = lane & 15;
    const int krow = (grp >> 1) * 8 + (l16 >> 2);            // k (pixel) index inside a 16-slice; +4 for the second read
    const int csub = (grp & 1) * 16 + (l16 & 3) * 4;         // column inside the wave's 32
    // dy band: pixel p = 16*s + krow (+4); chunk / byte inside the pixel
    unsigned a_base[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int p = krow + h * 4, col = wco + csub;
        a_base[h] = p * 128 + (((col >> 3) ^ (((p >> 1) & 1) << 2)) << 4) + (col & 7) * 2;
    }
    // patch: pixel q = (ty + kh)*PW + tx + kw + 1  with (ty, tx) of band pixel p; the lane part is q_lane = krow (+4) mapped
    // through the row layout below; bit 1 of q depends only on (tx + kw + 1): three variants per read half
    unsigned x_base[3][2];
    int tx_l[2], ty_l[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) { int p = krow + h * 4; ty_l[h] = p / W; tx_l[h] = p - ty_l[h] * W; }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int q = ty_l[h] * PW + tx_l[h] + kw;                    // patch column = (tx + 1) + (kw - 1)
            int col = wci + csub;
            x_base[kw][h] = q * 128 + (((col >> 3) ^ (((q >> 1) & 1) << 2)) << 4) + (col & 7) * 2;
        }

    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // all MFMAs of one band for the taps T0 .. T0+NTP-1 (compile-time: every fragment address is base + immediate)
    auto band_compute = [&](auto t0c, auto ntc, unsigned ab, unsigned xb) {
        constexpr int T0 = decltype(t0c)::value, NTP = decltype(ntc)::value;
        unsigned a_cur[2] = {ab + a_base[0], ab + a_base[1]};
        unsigned x_cur[3][2];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) { x_cur[kw][0] = xb + x_base[kw][0]; x_cur[kw][1] = xb + x_base[kw][1]; }
        if constexpr (X3 && PIN) {
            // linearised (k-slice, tap) steps with the next step's fragments in flight
            auto rd_a = [&](int sl, uint4& a, uint4& al) {
                a = tr_pair(a_cur[0] + sl * 16 * 128, a_cur[1] + sl * 16 * 128);
                al = tr_pair(a_cur[0] + ABYTES + sl * 16 * 128, a_cur[1] + ABYTES + sl * 16 * 128);
            };
            auto rd_b = [&](int sl, int tt, uint4& b, uint4& bl) {
                const int kh = (T0 + tt) / 3, kw = (T0 + tt) % 3, p0 = 16 * sl;
                const int disp = ((p0 / W + kh) * PW + p0 % W) * 128;
                b = tr_pair(x_cur[kw][0] + disp, x_cur[kw][1] + disp);
                bl = tr_pair(x_cur[kw][0] + XBYTES + disp, x_cur[kw][1] + XBYTES + disp);
            };
            uint4 fa[2], fal[2], fb[2], fbl[2];
            rd_a(0, fa[0], fal[0]);
            rd_b(0, 0, fb[0], fbl[0]);
#pragma unroll
            for (int st = 0; st < KS * NTP; ++st) {
                const int sl = st / NTP, tt = st - sl * NTP;
                if (st + 1 < KS * NTP) {
                    const int sl2 = (st + 1) / NTP, tt2 = (st + 1) - sl2 * NTP;
                    if (tt2 == 0) rd_a(sl2, fa[sl2 & 1], fal[sl2 & 1]);
                    rd_b(sl2, tt2, fb[(st + 1) & 1], fbl[(st + 1) & 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 a = __builtin_bit_cast(bf16x8, fa[sl & 1]), al = __builtin_bit_cast(bf16x8, fal[sl & 1]);
                const bf16x8 b = __builtin_bit_cast(bf16x8, fb[st & 1]), bl = __builtin_bit_cast(bf16x8, fbl[st & 1]);
                acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[tt], 0, 0, 0);
                acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, acc[tt], 0, 0, 0);
                acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b, acc[tt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // band pixel 16*s + k: same image row for all 16 k when W >= 16; for W == 8 two rows (k >= 8 -> next row),
            // handled by ty_l/tx_l in x_base, so only the slice origin (row/col of pixel p0) is added here
            const int p0 = 16 * s;
            uint4 fa = tr_pair(a_cur[0] + p0 * 128, a_cur[1] + p0 * 128);
            uint4 fal;
            if constexpr (X3) fal = tr_pair(a_cur[0] + ABYTES + p0 * 128, a_cur[1] + ABYTES + p0 * 128);
#pragma unroll
            for (int tt = 0; tt < NTP; ++tt) {
                const int kh = (T0 + tt) / 3, kw = (T0 + tt) % 3;
                const int oy = p0 / W, ox = p0 % W;
                const int disp = ((oy + kh) * PW + ox) * 128;
                uint4 fb = tr_pair(x_cur[kw][0] + disp, x_cur[kw][1] + disp);
                acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb),
                                                                  acc[tt], 0, 0, 0);
                if constexpr (X3) {
                    uint4 fbl = tr_pair(x_cur[kw][0] + XBYTES + disp, x_cur[kw][1] + XBYTES + disp);
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fbl),
                                                                      acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fal), __builtin_bit_cast(bf16x8, fb),
                                                                      acc[tt], 0, 0, 0);
                }
            }
        }
    };

    if (band_begin < band_end) issue_band(band_begin, 0);
    int buf = 0;
    for (int band = band_begin; band < band_end; ++band) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // WAR on the band buffer restaged below: see conv_gemm2.hip
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (band + 1 < band_end) issue_band(band + 1, buf ^ 1);
        const unsigned ab = lds0 + buf * BUF, xb = ab + NPL * ABYTES;
        if constexpr (NW == 8) {
            if (tg == 0) band_compute(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{}, ab, xb);
            else band_compute(std::integral_constant<int, 5>{}, std::integral_constant<int, 4>{}, ab, xb);
        } else {
            band_compute(std::integral_constant<int, 0>{}, std::integral_constant<int, 9>{}, ab, xb);
        }
        buf ^= 1;
    }
    // ---- slab write: dW[co][tap][ci] (row length 9*Cin)
    float* out = pslabs + (long)by * g.Cout * 9 * g.Cin;
    const int jt = 9 * g.Cin;
#pragma unroll
    for (int tt = 0; tt < NACC; ++tt) {
        if (tt >= ntap) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = co0 + wco + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            int col = (t0 + tt) * g.Cin + ci0 + wci + (lane & 31);
            out[(long)row * jt + col] = acc[tt][r];
        }
    }
}

template <int W, int TH, int NW, int X3 = 0>
static size_t wg3_lds() {
    constexpr int BP = TH * W, NPIX = (TH + 2) * (W + 4);
    constexpr int LA = (BP / 8 + NW - 1) / NW, LX = ((NPIX + 7) / 8 + NW - 1) / NW;
    return (size_t)2 * (X3 ? 2 : 1) * (LA + LX) * NW * 1024;
}

template <int W, int TH, int NW, int X3 = 0>
static int wg3_launch(Wg3Args& g, int tiles, int nslices, hipStream_t st) {
    size_t lds = wg3_lds<W, TH, NW, X3>();
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad3x3_kernel<W, TH, NW, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if constexpr (X3 != 0) { if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wgrad3x3_kernel<W, TH, NW, X3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    static const int xcd = getenv("AB_WG_XCD") ? atoi(getenv("AB_WG_XCD")) : 1;
    g.xcd_map = xcd;
    static const int pin = getenv("AB_WG3_PIN") ? atoi(getenv("AB_WG3_PIN")) : 1;      // 9.08 -> 9.02 ms per step over two alternating pairs (round 5)
    bool launched = false;
    if constexpr (X3 != 0) { if (pin) { wgrad3x3_kernel<W, TH, NW, X3, true><<<dim3(tiles, nslices), 64 * NW, lds, st>>>(g); launched = true; } }
    if (!launched) wgrad3x3_kernel<W, TH, NW, X3><<<dim3(tiles, nslices), 64 * NW, lds, st>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

static int wg3_th(int H, int W) {
    if (W == 64 && H % 2 == 0) return 2;
    if (W == 32 && H % 4 == 0) return 4;
    if (W == 16 && H % 8 == 0) return 8;
    if (W == 8 && H % 8 == 0) return 8;
    return 0;
}

// returns the number of slabs (pixel slices) this kernel will write, 0 when the shape is not handled here
int wgrad3x3_slices(int N, int H, int W, int Cin, int Cout) {
    int th = wg3_th(H, W);
    if (!th || Cin % 64 || Cout % 64 || getenv("AB_WGRAD3_OFF")) return 0;
    int nbands = N * (H / th);
    int tiles = (Cin / 64) * (Cout / 64);
    static int target = getenv("AB_WG3_TARGET") ? atoi(getenv("AB_WG3_TARGET")) : 256;
    int want = (target + tiles - 1) / tiles;
    int ns = want < 1 ? 1 : want;
    if (ns > nbands / 2) ns = nbands / 2 > 0 ? nbands / 2 : 1;      // at least two bands per slice (double buffering)
    if (ns > 256) ns = 256;
    int bps = (nbands + ns - 1) / ns;
    return (nbands + bps - 1) / bps;
}

int wgrad3x3_run(const void* x, const void* dy, float* slabs, int N, int H, int W, int Cin, int Cout, hipStream_t st) {
    int ns = wgrad3x3_slices(N, H, W, Cin, Cout);
    if (!ns) return AB_ESHAPE;
    int th = wg3_th(H, W);
    Wg3Args g = {};
    g.X = x; g.DY = dy; g.slabs = slabs; g.N = N; g.H = H; g.Cin = Cin; g.Cout = Cout;
    g.nbands = N * (H / th);
    g.bands_per_slice = (g.nbands + ns - 1) / ns;
    int tiles = (Cin / 64) * (Cout / 64);
    static const int w8 = getenv("AB_WG3_W8") ? atoi(getenv("AB_WG3_W8")) : 1;
    if (w8) {
        if (W == 64) return wg3_launch<64, 2, 8>(g, tiles, ns, st);
        if (W == 32) return wg3_launch<32, 4, 8>(g, tiles, ns, st);
        if (W == 16) return wg3_launch<16, 8, 8>(g, tiles, ns, st);
        return wg3_launch<8, 8, 8>(g, tiles, ns, st);
    }
    if (W == 64) return wg3_launch<64, 2, 4>(g, tiles, ns, st);
    if (W == 32) return wg3_launch<32, 4, 4>(g, tiles, ns, st);
    if (W == 16) return wg3_launch<16, 8, 4>(g, tiles, ns, st);
    return wg3_launch<8, 8, 4>(g, tiles, ns, st);
}

// ---- split-bf16 ("bf16x3") launches: bands of 64 pixels (two stages of both planes fit in LDS)
static int wg3x_th(int H, int W) {
    if (W == 64) return 1;
    if (W == 32 && H % 2 == 0) return 2;
    if (W == 16 && H % 4 == 0) return 4;
    if (W == 8 && H % 8 == 0) return 8;
    return 0;
}

int wgrad3x3_x3_slices(int N, int H, int W, int Cin, int Cout) {
    int th = wg3x_th(H, W);
    if (!th || Cin % 64 || Cout % 64) return 0;
    int nbands = N * (H / th);
    int tiles = (Cin / 64) * (Cout / 64);
    // (AB_WG3X_TARGET: workgroups per launch the slice count aims at; 256 = one per CU.  512 was tried for review item 7 -- launches that lose
    // a CU to a collective's kernel then lose 1/512 of their work instead of running a second round: see DESIGN 14.3)
    static const int target = getenv("AB_WG3X_TARGET") ? atoi(getenv("AB_WG3X_TARGET")) : 256;
    int want = (target + tiles - 1) / tiles;
    int ns = want < 1 ? 1 : want;
    if (ns > nbands / 2) ns = nbands / 2 > 0 ? nbands / 2 : 1;
    if (ns > 256) ns = 256;
    int bps = (nbands + ns - 1) / ns;
    return (nbands + bps - 1) / bps;
}

// slices PER PROBLEM of a grouped launch over G same-shape layers (0: shape not handled): the single-launch rule with the workgroup target
// divided by G -- and at least two bands per slice, as there
int wgrad3x3_x3_group_slices(int G, int N, int H, int W, int Cin, int Cout) {
    int th = wg3x_th(H, W);
    if (!th || Cin % 64 || Cout % 64 || G < 1 || G > 8) return 0;
    int nbands = N * (H / th);
    int tiles = (Cin / 64) * (Cout / 64);
    static const int target = getenv("AB_WG3X_TARGET") ? atoi(getenv("AB_WG3X_TARGET")) : 256;
    // floor, not the single launch's ceil: 16 tiles x 3 problems x ceil(256 / 48) = 288 workgroups ran as a second round on 32 CUs
    // (measured: the step 0.45 ms SLOWER with groups of three than ungrouped); x 5 = 240 fill 94 % of the chip in one round
    if (G > 1 && tiles * G > target) return 0;          // (more problems than one round of workgroups holds: the caller groups fewer)
    int want = G > 1 ? target / (tiles * G) : (target + tiles - 1) / tiles;
    int ns = want < 1 ? 1 : want;
    if (ns > nbands / 2) ns = nbands / 2 > 0 ? nbands / 2 : 1;
    if (ns > 256) ns = 256;
    int bps = (nbands + ns - 1) / ns;
    return (nbands + bps - 1) / bps;
}

// G <= 8 same-shape problems in one grid; slabs[p]: problem p's [ns][Cout][9][Cin] partials
int wgrad3x3_x3_group_run(int G, const void* const* x_hi, const void* const* x_lo, const void* const* dy_hi, const void* const* dy_lo,
                          float* const* slabs, int N, int H, int W, int Cin, int Cout, hipStream_t st) {
    int ns = wgrad3x3_x3_group_slices(G, N, H, W, Cin, Cout);
    if (!ns) return AB_ESHAPE;
    int th = wg3x_th(H, W);
    Wg3Args g = {};
    g.X = x_hi[0]; g.X_lo = x_lo[0]; g.DY = dy_hi[0]; g.DY_lo = dy_lo[0]; g.slabs = slabs[0];
    for (int p = 1; p < G; ++p) { g.Xg[p - 1] = x_hi[p]; g.Xg_lo[p - 1] = x_lo[p]; g.DYg[p - 1] = dy_hi[p]; g.DYg_lo[p - 1] = dy_lo[p]; g.slabsg[p - 1] = slabs[p]; }
    g.ns_per = G > 1 ? ns : 0;
    g.N = N; g.H = H; g.Cin = Cin; g.Cout = Cout;
    g.nbands = N * (H / th);
    g.bands_per_slice = (g.nbands + ns - 1) / ns;
    int tiles = (Cin / 64) * (Cout / 64);
    if (W == 64) return wg3_launch<64, 1, 8, 1>(g, tiles, ns * G, st);
    if (W == 32) return wg3_launch<32, 2, 8, 1>(g, tiles, ns * G, st);
    if (W == 16) return wg3_launch<16, 4, 8, 1>(g, tiles, ns * G, st);
    return wg3_launch<8, 8, 8, 1>(g, tiles, ns * G, st);
}

int wgrad3x3_x3_run(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, float* slabs, int N, int H, int W,
                    int Cin, int Cout, hipStream_t st) {
    int ns = wgrad3x3_x3_slices(N, H, W, Cin, Cout);
    if (!ns) return AB_ESHAPE;
    int th = wg3x_th(H, W);
    Wg3Args g = {};
    g.X = x_hi; g.X_lo = x_lo; g.DY = dy_hi; g.DY_lo = dy_lo; g.slabs = slabs; g.N = N; g.H = H; g.Cin = Cin; g.Cout = Cout;
    g.nbands = N * (H / th);
    g.bands_per_slice = (g.nbands + ns - 1) / ns;
    int tiles = (Cin / 64) * (Cout / 64);
    if (W == 64) return wg3_launch<64, 1, 8, 1>(g, tiles, ns, st);
    if (W == 32) return wg3_launch<32, 2, 8, 1>(g, tiles, ns, st);
    if (W == 16) return wg3_launch<16, 4, 8, 1>(g, tiles, ns, st);
    return wg3_launch<8, 8, 8, 1>(g, tiles, ns, st);
}
