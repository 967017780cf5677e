// Implicit-GEMM convolution, bf16 fast path: direct-to-LDS loads (global_load_lds_dwordx4), no VGPR staging.
//
// Same contract as conv_gemm_kernel (conv_common.h), specialised for the benchmark path:
//   * every K step moves 128 bytes (64 bf16 channels of one tap) per tile row straight from L2/HBM into LDS with
//     global_load_lds; a wave instruction fills 1 KiB = 8 rows x 8 chunks, lane-linear, so the bank-conflict
//     swizzle is applied on the SOURCE side: LDS slot (row r, slot c') holds logical chunk c = c' ^ ((r >> 1) & 7),
//     and the MFMA fragment reads apply the same involution (conflict-free ds_read_b128 for any 16 distinct rows
//     mod 16);
//   * rows that fall outside the image (padding taps, ragged M / N tails) read a 16-byte zero page instead of
//     branching, so the load stream is uniform;
//   * workgroups are renumbered so that each XCD (block id mod 8) owns a contiguous range of M tiles: neighbouring
//     tiles re-read each other's halo pixels and the 9 taps re-read the same rows from that XCD's L2;
//   * a ring of NBUF (3) LDS stages; the loads run two K-steps ahead of the MFMAs, issued from inline asm and retired
//     with counted s_waitcnt vmcnt(N) + a raw s_barrier per step (see the loop);
//   * 4 or 8 waves per workgroup (WM x WN): the L2->LDS fill rate scales with the number of waves issuing loads;
//   * nclass > 1: the four output-parity classes of a stride-2 data gradient share one grid.
#include "conv_common.h"

__device__ uint4 ab_zero_page[2];    // zero-initialised device memory: source of every out-of-range chunk

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// STEM: A is the zero-bordered NHWC4 image; a K step covers TWO kernel rows (2 x 32 elements = 128 bytes per output pixel:
// chunks 0..3 from image row 2p + 2s, chunks 4..7 from row 2p + 2s + 1, both starting at column 2q), four steps for the 7
// rows + 1 row of padding; weight rows are [7][8][4] = 224 elements, the missing 32 are read from the zero page.
// X3 = 1: split-bf16 operands (see conv3x3.hip): a 128-byte LDS row carries a 32-channel chunk as [hi 64 B][lo 64 B];
// g.cpt counts 32-channel chunks; Out / addend are fp32.
// ALT (X3 only): one tap reads the second operand pair (ConvGemmArgs::alt_tap1).  A template parameter, not a run-time test: the
// extra pointer selects in the load issue cost the launches that do not use it 5-20 %.
template <int BM, int BN, int WM, int WN, int NBUF = 3, bool STEM = false, int X3 = 0, bool ALT = false>
__global__ __launch_bounds__(64 * WM * WN) void conv_gemm2_kernel(ConvGemmArgs g) {
    constexpr int CK = X3 ? 32 : 64;                          // channels per K step
    constexpr int NW = WM * WN, NT = 64 * NW;                // 4 or 8 waves: the LDS fill rate scales with the waves issuing loads
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;     // 32x32 MFMA tiles per wave
    constexpr int IA = BM / 8, IB = BN / 8;                 // 1-KiB load instructions (8 rows each) for the A / B tile
    constexpr int BUFSZ = (BM + BN) * 128;
    // LDS ring of NBUF stages: loads run NBUF-1 K-steps ahead of the MFMAs.  The fill is latency-bound (bytes in flight
    // per CU / L2 latency), so grids that put a single workgroup on a CU use a deeper ring (5) than those that co-run 2-3.
    constexpr int PD = NBUF - 1;
    // (X3: the fp32 epilogue staging tile may exceed a two-stage ring)
    constexpr int STAGE_X3 = X3 ? BM * (BN * 4 + 16) : 0;
    constexpr int RINGSZ = NBUF * BUFSZ > STAGE_X3 ? NBUF * BUFSZ : STAGE_X3;
    __shared__ __attribute__((aligned(16))) unsigned char smem[RINGSZ + BM * 4 + WM * BN * 8 + 3 * CG_MAXTAPS * 4];
    int* s_outpix = (int*)(smem + RINGSZ);
    float* s_stat = (float*)(s_outpix + BM);                // [WM][BN][2]
    // tap tables copied to LDS: indexing the kernarg arrays with the runtime tap id compiles to VMEM loads inside the K
    // loop, and the vmcnt wait for those would drain the in-flight LDS-DMA prefetch
    int* s_tap = (int*)(s_stat + WM * BN * 2);              // [3][CG_MAXTAPS] = dh, dw, koff

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WN, wave_n = wave % WN;
    // XCD-aware renumbering (bijective for any grid size): XCD x = bid % 8 gets logical ids [start_x, start_x + cnt_x)
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, within = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const int tiles_n = (g.Cn + BN - 1) / BN, tiles_mc = (g.M + BM - 1) / BM;      // M tiles per parity class
    const int tiles_m = tiles_mc * (g.nclass > 1 ? g.nclass : 1);
    int tile_m, tile_n;
    if (g.nmajor) { tile_n = logical / tiles_m; tile_m = logical - tile_n * tiles_m; }
    else { tile_m = logical / tiles_n; tile_n = logical - tile_m * tiles_n; }
    const int stat_row = tile_m;                             // BN-partial row: unique over (parity class, M tile)
    int ntaps = g.ntaps, out_oh = g.out_oh, out_ow = g.out_ow, tap0 = 0;
    if (g.nclass > 1) {
        const int cls = tile_m / tiles_mc;
        tile_m -= cls * tiles_mc;
        ntaps = g.cls_ntaps[cls]; out_oh = g.cls_oh[cls]; out_ow = g.cls_ow[cls]; tap0 = cls * 4;
    }
    if (threadIdx.x < CG_MAXTAPS - tap0) {
        s_tap[threadIdx.x] = g.dh[tap0 + threadIdx.x];
        s_tap[CG_MAXTAPS + threadIdx.x] = g.dw[tap0 + threadIdx.x];
        s_tap[2 * CG_MAXTAPS + threadIdx.x] = g.koff[tap0 + threadIdx.x];
    }
    __syncthreads();
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const bf16_t* __restrict__ A = (const bf16_t*)g.A;
    const bf16_t* __restrict__ Bw = (const bf16_t*)g.Bw;
    const bf16_t* __restrict__ Alo = (const bf16_t*)g.A_lo;
    const bf16_t* __restrict__ Bwlo = (const bf16_t*)g.Bw_lo;
    const bf16_t* __restrict__ A2 = (const bf16_t*)g.A2;
    const bf16_t* __restrict__ A2lo = (const bf16_t*)g.A2_lo;
    const bf16_t* __restrict__ Bw2 = (const bf16_t*)g.Bw2;
    const bf16_t* __restrict__ Bw2lo = (const bf16_t*)g.Bw2_lo;
    const int alt_t = ALT ? g.alt_tap1 - 1 - tap0 : -1;       // class-local index of the tap that reads the second operand pair
    const int PQ = g.P * g.Q;
    const bf16_t* zp = (const bf16_t*)ab_zero_page;

    // ---- per-lane load assignment: instruction ii (0..IA-1 over the 4 waves) covers rows ii*8 .. ii*8+7
    const int lrow = lane >> 3, lslot = lane & 7;
    constexpr int NA = IA / NW, NB = IB / NW;               // per wave
    static_assert(IA % NW == 0 && IB % NW == 0, "tile rows must split evenly over the waves");
    int a_h[NA], a_w[NA], a_chunk[NA]; long a_base[NA]; bool a_ok[NA]; bool a_lo[NA], b_lo[NB];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        int ii = wave * NA + j;
        int r = ii * 8 + lrow;
        int m = m0 + r;
        a_ok[j] = (ii < IA) && (m < g.M);
        int mm = a_ok[j] ? m : 0;
        int n = mm / PQ, rem = mm - n * PQ;
        int p = rem / g.Q, q = rem - p * g.Q;
        a_h[j] = p * g.a_sh; a_w[j] = q * g.a_sw;
        a_base[j] = (long)n * g.Ha * g.Wa;
        a_chunk[j] = (lslot ^ ((r >> 1) & 7)) * 8;      // element offset of the logical chunk this lane fetches
        a_lo[j] = false;
        if (X3) { const int c = lslot ^ ((r >> 1) & 7); a_lo[j] = (c & 4) != 0; a_chunk[j] = (c & 3) * 8; }
        if (STEM && !X3) { const int c = lslot ^ ((r >> 1) & 7); a_chunk[j] = ((c >> 2) << 16) | ((c & 3) * 8); }   // (row of the pair, offset)
        if (lslot == 0 && ii < IA) {
            int op = (n * g.Ho + p * g.out_sh + out_oh) * g.Wo + q * g.out_sw + out_ow;
            s_outpix[r] = a_ok[j] ? op : -1;
        }
    }
    long b_off[NB]; long b_off2[ALT ? NB : 1]; int b_chunk[NB]; bool b_ok[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        int ii = wave * NB + j;
        int r = ii * 8 + lrow;
        int col = n0 + r;
        b_ok[j] = (ii < IB) && (col < g.Cn);
        b_off[j] = (long)(b_ok[j] ? col : 0) * g.ktot;
        if constexpr (ALT) b_off2[j] = (long)(b_ok[j] ? col : 0) * g.ktot2;
        b_chunk[j] = (lslot ^ ((r >> 1) & 7)) * 8;
        b_lo[j] = false;
        if (X3) { const int c = lslot ^ ((r >> 1) & 7); b_lo[j] = (c & 4) != 0; b_chunk[j] = (c & 3) * 8; }
    }
    const int nsteps = STEM ? (X3 ? 7 : 4) : ntaps * g.cpt;      // STEM + X3: one kernel row (32 elements, both planes) per step

    auto issue = [&](int step, int buf) {
        const int t = STEM ? 0 : step / g.cpt, c0 = STEM ? step * CK : (step - t * g.cpt) * CK;
        const int dh = STEM ? 0 : s_tap[t], dw = STEM ? 0 : s_tap[CG_MAXTAPS + t], ko = STEM ? 0 : s_tap[2 * CG_MAXTAPS + t];
        const bool alt = ALT && t == alt_t;                   // wave-uniform: this step's rows come from (A2, Bw2)
        const bf16_t* Ah = alt ? A2 : A; const bf16_t* Al = alt ? A2lo : Alo;
        const bf16_t* Bh = alt ? Bw2 : Bw; const bf16_t* Bl = alt ? Bw2lo : Bwlo;
        unsigned char* base = smem + buf * BUFSZ;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int ii = wave * NA + j;
            if (ii < IA) {
                const bf16_t* src;
                if (STEM) {
                    const int krow = X3 ? step : 2 * step + (a_chunk[j] >> 16);     // kernel row 0..7 (7 = padding)
                    src = (a_ok[j] && krow < 7) ? ((X3 && a_lo[j]) ? Alo : A) + ((a_base[j] + (long)(a_h[j] + krow) * g.Wa + a_w[j]) * 4 + (a_chunk[j] & 0xffff)) : zp;
                } else {
                    int hi = a_h[j] + dh, wi = a_w[j] + dw;
                    bool ok = a_ok[j] && (unsigned)hi < (unsigned)g.Ha && (unsigned)wi < (unsigned)g.Wa;
                    src = ok ? (((X3 && a_lo[j]) ? Al : Ah) + ((a_base[j] + (long)hi * g.Wa + wi) * g.Ca + c0 + a_chunk[j])) : zp;
                }
                glds16(src, __builtin_amdgcn_readfirstlane(lds_addr_of(base + ii * 1024)));
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int ii = wave * NB + j;
            if (ii < IB) {
                const bool ok = b_ok[j] && (!STEM || c0 + b_chunk[j] < g.ktot);
                long bo = b_off[j] + ko;
                if constexpr (ALT) { if (alt) bo = b_off2[j]; }
                const bf16_t* src = ok ? (((X3 && b_lo[j]) ? Bl : Bh) + (bo + c0 + b_chunk[j])) : zp;
                glds16(src, __builtin_amdgcn_readfirstlane(lds_addr_of(base + BM * 128 + ii * 1024)));
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x16 accx[X3 ? TM : 1][X3 ? TN : 1];                  // X3: the two cross products hi*lo + lo*hi (second chain)
    if constexpr (X3) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accx[i][j][r] = 0.f;
    }

    // ring with counted waits (raw s_barrier: __syncthreads() would drain vmcnt to 0 and kill the overlap):
    //   step s:  wait until only the loads of steps s+1 .. s+PD-1 are in flight -> barrier (step-s tile visible to all
    //            waves, and all waves are done reading the buffer of step s-1) -> issue step s+PD into that buffer ->
    //            MFMAs of step s
    constexpr int L = NA + NB;                               // load instructions per wave per step
#pragma unroll
    for (int d = 0; d < PD; ++d) if (d < nsteps) issue(d, d);
    const int frow = lane & 31, fhalf = lane >> 5;
    int cur = 0;
    for (int step = 0; step < nsteps; ++step) {
        // wait until only the later steps' loads (at most PD-1 of them) are still in flight
        const int ahead = min(PD - 1, nsteps - 1 - step);
        if (ahead >= 4) wait_vm<(PD >= 5 ? 4 : 0) * L>();
        else if (ahead == 3) wait_vm<(PD >= 4 ? 3 : 0) * L>();
        else if (ahead == 2) wait_vm<(PD >= 3 ? 2 : 0) * L>();
        else if (ahead == 1) wait_vm<L>();
        else wait_vm<0>();
        // WAR: the buffer restaged right after this barrier was read in the previous step; those ds_reads must have RETIRED
        // (not merely issued) before any wave passes the barrier.  The compiler otherwise sinks the previous step's last MFMA
        // -- and the lgkmcnt wait it needs -- below the barrier (seen in the ISA of the fully unrolled 4-step stem variant:
        // sporadic wrong tiles, run-to-run different).  cdna_hip_programming.md: "raw s_barrier + lgkmcnt(0)".
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (step + PD < nsteps) { int nb = cur + PD; if (nb >= NBUF) nb -= NBUF; issue(step + PD, nb); }
        const unsigned char* sa = smem + cur * BUFSZ;
        const unsigned char* sb = sa + BM * 128;
        auto read_frags = [&](int kk, uint4* fa, uint4* fb) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int r = (wave_m * TM + i) * 32 + frow;
                fa[i] = *(const uint4*)(sa + r * 128 + (((kk * 2 + fhalf) ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int r = (wave_n * TN + j) * 32 + frow;
                fb[j] = *(const uint4*)(sb + r * 128 + (((kk * 2 + fhalf) ^ ((r >> 1) & 7)) << 4));
            }
        };
        if constexpr (X3) {
            // slots 0,1: hi k-slices (16 channels each); slots 2,3: the lo planes of the same channels
            uint4 fa[4][TM], fb[4][TN];
            read_frags(0, fa[0], fb[0]); read_frags(2, fa[2], fb[2]);
            read_frags(1, fa[1], fb[1]); read_frags(3, fa[3], fb[3]);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, fa[k2][i]), al = __builtin_bit_cast(bf16x8, fa[k2 + 2][i]);
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, fb[k2][j]), bl = __builtin_bit_cast(bf16x8, fb[k2 + 2][j]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
                        accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, accx[i][j], 0, 0, 0);
                        accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, accx[i][j], 0, 0, 0);
                    }
        } else {
        // software-pipelined fragments: the ds_reads of k-slice kk+1 are in flight while the MFMAs of kk issue
        uint4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
        read_frags(0, fa0, fb0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            uint4* fa = (kk & 1) ? fa1 : fa0; uint4* fb = (kk & 1) ? fb1 : fb0;
            if (kk < 3) read_frags(kk + 1, (kk & 1) ? fa0 : fa1, (kk & 1) ? fb0 : fb1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                       __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
        }
        }
        if (++cur == NBUF) cur = 0;
    }
    __syncthreads();

    if constexpr (X3) {
        // ---- fp32 epilogue of the split-bf16 launches (same flow as below, 4-byte elements, 16-byte row vectors)
        float* __restrict__ OutF = (float*)g.Out;
        const float* __restrict__ AddF = (const float*)g.addend;
        constexpr int SPF = BN * 4 + 16;
        static_assert(BM * SPF <= RINGSZ, "fp32 staging tile must fit in the K-loop buffers");
        float csum[TN], csq[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) { csum[j] = 0.f; csq[j] = 0.f; }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cl = (wave_n * TN + j) * 32 + (lane & 31);
            const int col = n0 + cl;
            const bool cok = col < g.Cn;
            const float bj = (g.bias && cok) ? g.bias[col] : 0.f;
            const float scj = (g.ep_scale && cok) ? g.ep_scale[col] : 1.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wave_m * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                    float v = acc[i][j][r] + accx[i][j][r];
                    v = g.ep_scale ? v * scj + bj : v + bj;          // (bn_apply's own expression: bit-identical to conv + apply pass)
                    if (g.relu) v = fmaxf(v, 0.f);
                    *(float*)(smem + row * SPF + cl * 4) = v;
                    if (cok && s_outpix[row] >= 0) { csum[j] += v; csq[j] += v * v; }
                }
            }
        }
        __syncthreads();
        float bfs[4] = {0.f, 0.f, 0.f, 0.f}, bfq[4] = {0.f, 0.f, 0.f, 0.f};
        float e_sc[4] = {}, e_sh[4] = {}, e_mean[4] = {}, e_istd[4] = {};
        if (g.bn_y) {
            const int ccol = n0 + (tid % (BN / 4)) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) { e_sc[k] = g.bnp[ccol + k]; e_sh[k] = g.bnp[g.Cn + ccol + k]; e_mean[k] = g.bnp[2 * g.Cn + ccol + k]; e_istd[k] = g.bnp[3 * g.Cn + ccol + k]; }
        }
        {
            constexpr int CPRF = BN / 4;
            const bool vec_ok = (g.Cn & 3) == 0;
            for (int id = tid; id < BM * CPRF; id += NT) {
                const int row = id / CPRF, c4 = id - row * CPRF;
                const int op = s_outpix[row], col = n0 + c4 * 4;
                if (op < 0 || col >= g.Cn) continue;
                const long o = (long)op * g.Cn + col;
                float4 v = *(const float4*)(smem + row * SPF + c4 * 16);
                if (g.out_hi) {          // eval-mode fold: the next convolution's operand planes, no fp32 tensor (Cn % 4 == 0 checked by the host)
                    uint2 h, l;
                    h.x = pack_bf16x2(v.x, v.y); h.y = pack_bf16x2(v.z, v.w);
                    l.x = pack_bf16x2(v.x - __uint_as_float(h.x << 16), v.y - __uint_as_float(h.x & 0xffff0000u));
                    l.y = pack_bf16x2(v.z - __uint_as_float(h.y << 16), v.w - __uint_as_float(h.y & 0xffff0000u));
                    *(uint2*)((bf16_t*)g.out_hi + o) = h; *(uint2*)((bf16_t*)g.out_lo + o) = l;
                } else if (g.bn_y) {     // (host: Cn % 4 == 0, BN divides Cn, no addend -- a thread keeps its four channels: NT % CPRF == 0)
                    const float4 y4 = *(const float4*)(g.bn_y + o);
                    float vv[4] = {v.x, v.y, v.z, v.w};
                    const float yy[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool dead = !(yy[k] * e_sc[k] + e_sh[k] > 0.f);
                        vv[k] = dead ? 0.f : vv[k];
                        bfs[k] += vv[k]; bfq[k] += vv[k] * ((yy[k] - e_mean[k]) * e_istd[k]);
                    }
                    *(float4*)(OutF + o) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                } else if (vec_ok) {
                    if (AddF) { const float4 a = *(const float4*)(AddF + o); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
                    *(float4*)(OutF + o) = v;
                } else {
                    const float vv[4] = {v.x, v.y, v.z, v.w};
                    for (int k = 0; k < 4 && col + k < g.Cn; ++k) OutF[o + k] = vv[k] + (AddF ? AddF[o + k] : 0.f);
                }
            }
        }
        __syncthreads();
        if (g.bn_y) {
            constexpr int CPRF = BN / 4;
            static_assert(NT % CPRF == 0 && (NT / CPRF) * BN * 8 <= RINGSZ, "BatchNorm-backward partials: one channel group per thread, scratch over the staging tile");
            float* sp = (float*)smem;                          // [NT / CPRF][BN][2]
            const int rg = tid / CPRF, cb = (tid % CPRF) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) { sp[(rg * BN + cb + k) * 2] = bfs[k]; sp[(rg * BN + cb + k) * 2 + 1] = bfq[k]; }
            __syncthreads();
            for (int c = tid; c < BN; c += NT) {
                const int col = n0 + c;
                if (col < g.Cn) {
                    float s2 = 0.f, q2 = 0.f;
                    for (int r = 0; r < NT / CPRF; ++r) { s2 += sp[(r * BN + c) * 2]; q2 += sp[(r * BN + c) * 2 + 1]; }
                    g.bn_part[((long)stat_row * g.Cn + col) * 2] = s2;
                    g.bn_part[((long)stat_row * g.Cn + col) * 2 + 1] = q2;
                }
            }
            return;
        }
        if (g.stats) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float s2 = csum[j] + __shfl_xor(csum[j], 32, 64);
                float q2 = csq[j] + __shfl_xor(csq[j], 32, 64);
                if (lane < 32) {
                    int cl = (wave_n * TN + j) * 32 + lane;
                    s_stat[(wave_m * BN + cl) * 2] = s2;
                    s_stat[(wave_m * BN + cl) * 2 + 1] = q2;
                }
            }
            __syncthreads();
            for (int c = tid; c < BN; c += NT) {
                int col = n0 + c;
                if (col < g.Cn) {
                    float s2 = 0.f, q2 = 0.f;
#pragma unroll
                    for (int wm = 0; wm < WM; ++wm) { s2 += s_stat[(wm * BN + c) * 2]; q2 += s_stat[(wm * BN + c) * 2 + 1]; }
                    g.stats[((long)stat_row * g.Cn + col) * 2] = s2;
                    g.stats[((long)stat_row * g.Cn + col) * 2 + 1] = q2;
                }
            }
        }
        return;
    }

    // ---- epilogue: bias / ReLU / BN partials from the f32 accumulators, then the bf16 tile is transposed through LDS
    // (free after the K loop) and written with 16-byte vectors (the residual-gradient addend is folded in there).
    bf16_t* __restrict__ Out = (bf16_t*)g.Out;
    const bf16_t* __restrict__ Add = (const bf16_t*)g.addend;
    constexpr int SPITCH = BN * 2 + 16;
    static_assert(BM * SPITCH <= NBUF * BUFSZ, "staging tile must fit in the K-loop buffers");
    float csum[TN], csq[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { csum[j] = 0.f; csq[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int cl = (wave_n * TN + j) * 32 + (lane & 31);
        const int col = n0 + cl;
        const bool cok = col < g.Cn;
        const float bj = (g.bias && cok) ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = (wave_m * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                float v = acc[i][j][r] + bj;
                if (g.relu) v = fmaxf(v, 0.f);
                *(bf16_t*)(smem + row * SPITCH + cl * 2) = f32_to_bf16(v);
                if (cok && s_outpix[row] >= 0) { csum[j] += v; csq[j] += v * v; }
            }
        }
    }
    __syncthreads();
    {
        constexpr int CPR = BN / 8;
        const bool vec_ok = (g.Cn & 7) == 0;
        for (int id = tid; id < BM * CPR; id += NT) {
            int row = id / CPR, c8 = id - row * CPR;
            int op = s_outpix[row], col = n0 + c8 * 8;
            if (op < 0 || col >= g.Cn) continue;
            long o = (long)op * g.Cn + col;
            uint4 v = *(const uint4*)(smem + row * SPITCH + c8 * 16);
            if (vec_ok) {
                if (Add) {
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
            } else {      // ragged channel count: element-wise tail
                const bf16_t* sv = (const bf16_t*)(smem + row * SPITCH + c8 * 16);
                for (int k = 0; k < 8 && col + k < g.Cn; ++k) {
                    float x = bf16_to_f32(sv[k]);
                    if (Add) x += bf16_to_f32(Add[o + k]);
                    Out[o + k] = f32_to_bf16(x);
                }
            }
        }
    }
    __syncthreads();
    if (g.stats) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = csum[j] + __shfl_xor(csum[j], 32, 64);
            float q = csq[j] + __shfl_xor(csq[j], 32, 64);
            if (lane < 32) {
                int cl = (wave_n * TN + j) * 32 + lane;
                s_stat[(wave_m * BN + cl) * 2] = s;
                s_stat[(wave_m * BN + cl) * 2 + 1] = q;
            }
        }
        __syncthreads();
        for (int c = tid; c < BN; c += NT) {
            int col = n0 + c;
            if (col < g.Cn) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int wm = 0; wm < WM; ++wm) { s += s_stat[(wm * BN + c) * 2]; q += s_stat[(wm * BN + c) * 2 + 1]; }
                g.stats[((long)stat_row * g.Cn + col) * 2] = s;
                g.stats[((long)stat_row * g.Cn + col) * 2 + 1] = q;
            }
        }
    }
}

static void pick_tile2(int M, int Cn, int nsteps, int* bm, int* bn) {
    *bn = (Cn > 64) ? 128 : 64;
    *bm = 128;
    long tiles = (long)((M + 127) / 128) * ((Cn + *bn - 1) / *bn);
    if (tiles < 1024) { *bm = 64; }      // keep >= 2 workgroups per CU in flight
    // short K loops (1x1 convs): the fill / drain of a workgroup is not amortised, so favour LDS footprints that let
    // 2-3 workgroups share a CU and overlap each other's prologue and epilogue
    if (nsteps <= 8) { *bm = 64; }
    static const int force_bm = getenv("AB_GEMM2_BM") ? atoi(getenv("AB_GEMM2_BM")) : 0;
    if (force_bm) *bm = force_bm;
}

int conv_gemm2_mtiles(int M, int Cn, int nsteps) {
    int bm, bn; pick_tile2(M, Cn, nsteps, &bm, &bn);
    return (M + bm - 1) / bm;
}

// bf16 stem (see the STEM note above the kernel): g as set up by ab_conv2d_stem_fwd (Ca = 4, a_sh = a_sw = 2, ktot = 224)
int conv_gemm2_stem_mtiles(int M) { return (M + 127) / 128; }
int conv_gemm2_stem_run(ConvGemmArgs& g, hipStream_t st) {
    if (g.Cn != 64 || g.ktot != 224 || getenv("AB_STEM_V1")) return AB_ESHAPE;
    g.nclass = 0; g.nmajor = 0; g.ntaps = 4; g.cpt = 1;
    int tiles = (g.M + 127) / 128;
    // two ring stages: the four K steps of a tile are latency-bound, and 49 KB of LDS lets three workgroups share a CU
    // (110 us at B = 64, 256x256 vs 125 us with three stages and 130 us for the register-staged kernel)
    conv_gemm2_kernel<128, 64, 4, 2, 2, true><<<tiles, 512, 0, st>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int conv_gemm2_run(ConvGemmArgs& g, hipStream_t st) {
    if (g.Ca % 64) return AB_ESHAPE;
    int bm, bn; pick_tile2(g.M, g.Cn, g.ntaps * g.cpt, &bm, &bn);
    int tiles = ((g.M + bm - 1) / bm) * ((g.Cn + bn - 1) / bn) * (g.nclass > 1 ? g.nclass : 1);
    // Tile order inside an XCD's contiguous id range.  M-major (default): an XCD streams a band of pixels against ALL the
    // weights -- right when the weights fit its 4 MiB L2.  When they do not (l4: 512x4608 bf16 = 4.7 MB) every XCD thrashes
    // on them; N-major gives each XCD one or two N tiles (1.2 MB of weights) and the whole, small, activation tensor.
    static const int nm_env = getenv("AB_GEMM2_NMAJOR") ? atoi(getenv("AB_GEMM2_NMAJOR")) : -1;
    const long wbytes = (long)g.Cn * g.ktot * 2, abytes = (long)g.N * g.Ha * g.Wa * g.Ca * 2;
    g.nmajor = nm_env >= 0 ? nm_env : (wbytes > (3L << 20) && abytes <= (8L << 20) && g.Cn > bn);
    static const int deep_max = getenv("AB_GEMM2_DEEP") ? atoi(getenv("AB_GEMM2_DEEP")) : 0;
    const bool deep = tiles <= deep_max && g.ntaps * g.cpt >= 8;     // one workgroup per CU: deeper prefetch ring
    static const int w8 = getenv("AB_GEMM2_W8") ? atoi(getenv("AB_GEMM2_W8")) : 1;
    if (bm == 256 && bn == 64) conv_gemm2_kernel<256, 64, 4, 1><<<tiles, 256, 0, st>>>(g);
    else if (bm == 128 && bn == 128) {
        if (w8) conv_gemm2_kernel<128, 128, 4, 2><<<tiles, 512, 0, st>>>(g);
        else conv_gemm2_kernel<128, 128, 2, 2><<<tiles, 256, 0, st>>>(g);
    } else if (bm == 128 && bn == 64) {
        if (w8) conv_gemm2_kernel<128, 64, 4, 2><<<tiles, 512, 0, st>>>(g);
        else conv_gemm2_kernel<128, 64, 2, 2><<<tiles, 256, 0, st>>>(g);
    } else if (bm == 64 && bn == 128) {
        if (w8) conv_gemm2_kernel<64, 128, 2, 4><<<tiles, 512, 0, st>>>(g);
        else if (deep) conv_gemm2_kernel<64, 128, 2, 2, 5><<<tiles, 256, 0, st>>>(g);
        else conv_gemm2_kernel<64, 128, 2, 2><<<tiles, 256, 0, st>>>(g);
    } else {
        if (deep) conv_gemm2_kernel<64, 64, 2, 2, 5><<<tiles, 256, 0, st>>>(g);
        else conv_gemm2_kernel<64, 64, 2, 2><<<tiles, 256, 0, st>>>(g);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// ---- split-bf16 ("bf16x3") launches: g.A / g.A_lo and g.Bw / g.Bw_lo are the bf16 planes, g.cpt counts 32-channel
// chunks, Out / addend / stats are fp32.  Tile choice follows the bf16 path (2x the K steps of the same shape).
// Tile choice of the split-bf16 launches.  A K step carries 3x the MFMA work of a bf16 step on the same bytes, so these
// launches are bound by barriers / prologue / epilogue rather than by the fill: prefer the 128-pixel tile (twice the MFMAs per
// barrier) whenever it still gives every CU two workgroups, counting all parity classes of a one-grid stride-2 data gradient.
static void pick_tile2x(int M, int Cn, int nsteps, int nclass, int* bm, int* bn) {
    static const int mode = getenv("AB_G2X_MODE") ? atoi(getenv("AB_G2X_MODE")) : 1;
    if (mode == 0) { pick_tile2(M, Cn, nsteps, bm, bn); return; }
    *bn = (Cn > 64) ? 128 : 64;
    const long t128 = (long)((M + 127) / 128) * ((Cn + *bn - 1) / *bn) * (nclass > 1 ? nclass : 1);
    *bm = t128 >= 512 ? 128 : 64;
    // (a 256 x 128 tile -- 64 x 64 wave tiles, 0.67 fragment reads per MFMA instead of 1 -- needs 140 KB of LDS for its fp32
    // epilogue staging, i.e. one workgroup per CU, and measured slower: final 1x1 layer 113 -> 129 us, its data gradient 87 -> 92;
    // on a three-stage ring -- two steps in flight, 96 KB of loads per CU instead of 2 x 32 KB -- 119 -> 137 and 90 -> 97)
}

int conv_gemm2_x3_mtiles(int M, int Cn, int nsteps, int nclass) {
    int bm, bn; pick_tile2x(M, Cn, nsteps, nclass, &bm, &bn);
    return (M + bm - 1) / bm;
}

int conv_gemm2_x3_run(ConvGemmArgs& g, hipStream_t st) {
    if (g.Ca % 32 || !g.A_lo || !g.Bw_lo) return AB_ESHAPE;
    const int nsteps = g.ntaps * g.cpt;
    int bm, bn; pick_tile2x(g.M, g.Cn, nsteps, g.nclass, &bm, &bn);
    int tiles = ((g.M + bm - 1) / bm) * ((g.Cn + bn - 1) / bn) * (g.nclass > 1 ? g.nclass : 1);
    const long wbytes = (long)g.Cn * g.ktot * 4, abytes = (long)g.N * g.Ha * g.Wa * g.Ca * 4;
    g.nmajor = (wbytes > (3L << 20) && abytes <= (8L << 20) && g.Cn > bn);
    // two ring stages for short K loops and for the 128x128 tile: 64 KB instead of 96, two workgroups per CU
    static const int nb2 = getenv("AB_G2X_NBUF2") ? atoi(getenv("AB_G2X_NBUF2")) : 1;
    const bool two = nb2 && (nsteps <= 16 || (bm == 128 && bn == 128));
#define G2X_LAUNCH(ALT_)                                                                                                     \
    if (bm == 128 && bn == 128) {                                                                                            \
        if (two) conv_gemm2_kernel<128, 128, 4, 2, 2, false, 1, ALT_><<<tiles, 512, 0, st>>>(g);                              \
        else conv_gemm2_kernel<128, 128, 4, 2, 3, false, 1, ALT_><<<tiles, 512, 0, st>>>(g);                                  \
    } else if (bm == 128 && bn == 64) {                                                                                      \
        if (two) conv_gemm2_kernel<128, 64, 4, 2, 2, false, 1, ALT_><<<tiles, 512, 0, st>>>(g);                               \
        else conv_gemm2_kernel<128, 64, 4, 2, 3, false, 1, ALT_><<<tiles, 512, 0, st>>>(g);                                   \
    } else if (bm == 64 && bn == 128) {                                                                                      \
        if (two) conv_gemm2_kernel<64, 128, 2, 4, 2, false, 1, ALT_><<<tiles, 512, 0, st>>>(g);                               \
        else conv_gemm2_kernel<64, 128, 2, 4, 3, false, 1, ALT_><<<tiles, 512, 0, st>>>(g);                                   \
    } else {                                                                                                                 \
        if (two) conv_gemm2_kernel<64, 64, 2, 2, 2, false, 1, ALT_><<<tiles, 256, 0, st>>>(g);                                \
        else conv_gemm2_kernel<64, 64, 2, 2, 3, false, 1, ALT_><<<tiles, 256, 0, st>>>(g);                                    \
    }
    if (g.alt_tap1) { G2X_LAUNCH(true) }      // the paired data gradient of a down-sampling block (ab_conv2d_dgrad_x3_pair)
    else { G2X_LAUNCH(false) }
#undef G2X_LAUNCH
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// split-bf16 stem (7x7/2 on the zero-bordered NHWC4 image planes; weights [Cout][7][8][4] planes): seven K steps of one
// kernel row each
int conv_gemm2_x3_stem_mtiles(int M) { return (M + 127) / 128; }
int conv_gemm2_x3_stem_run(ConvGemmArgs& g, hipStream_t st) {
    if (g.Cn != 64 || g.ktot != 224 || !g.A_lo || !g.Bw_lo) return AB_ESHAPE;
    g.nclass = 0; g.nmajor = 0; g.ntaps = 7; g.cpt = 1;
    int tiles = (g.M + 127) / 128;
    conv_gemm2_kernel<128, 64, 4, 2, 2, true, 1><<<tiles, 512, 0, st>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
