// What the inner step of the split-bf16 3x3 kernel can reach, without anything else around it: a workgroup of NW waves
// loops over steps of [optional s_barrier] -> NRD ds_read_b128 of conflict-free fragments -> NMF MFMAs that consume them
// (main + cross accumulator chains).  Variants: reads issued up front (as the kernel does) or interleaved one pair ahead
// of the MFMAs that use them.  Compare the MFMA rate with the nominal 2.5 PFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_step.hip -o tools/probe_step && tools/probe_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;

// TM x TN wave tile as in conv3x3 X3: per step 4*(TM+TN) fragment reads, 6*TM*TN MFMAs
// AMODE 1: the pixel-side fragments are read from LDS on every third step only and derived by a DPP row shift (one v_mov_dpp
// per dword) on the two steps in between -- what serving the three horizontal taps of a 3x3 window from ONE fragment read
// would cost; AMODE 2: never re-read (upper bound: the weight-side reads alone).
// CH1: the three products of a tile go to ONE accumulator (interleaved over the TN tiles) instead of main + cross chains
template <int NW, int TM, int TN, int BARRIER, int INTERLEAVE, int AMODE = 0, int CH1 = 0>
__global__ __launch_bounds__(64 * NW) void step_loop(int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 96 * 1024 / 4; i += 64 * NW) ((unsigned*)smem)[i] = 0x3f803f80u + (i & 255);
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)smem;
    // conflict-free b128 pattern of the kernel: row r at r*128, slot (kk*2+half) ^ ((r>>1)&7)
    const int l32 = lane & 31, fh = lane >> 5;
    unsigned a_rel[TM][4], b_rel[TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) { int r = ((wave % 4) * TM + i) * 32 + l32; a_rel[i][k] = lds0 + 49152 + r * 128 + ((((k * 2 + fh)) ^ ((r >> 1) & 7)) << 4); }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) { int r = ((wave / 4) * TN + j) * 32 + l32; b_rel[j][k] = lds0 + r * 128 + ((((k * 2 + fh)) ^ ((r >> 1) & 7)) << 4); }
    f32x16 acc[TM][TN], accx[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accx[i][j][r] = 0.f; }
    u32x4 fa[4][TM];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[k][i] = *(const lds_u32x4*)(a_rel[i][k]);
    for (int it = 0; it < iters; ++it) {
        if (BARRIER) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
        const unsigned off = (it % 3) * 16384;
        u32x4 fb[4][TN];
        if (!INTERLEAVE) {
            const bool rd = AMODE == 0 || ((AMODE == 1 || AMODE == 3) && it % 3 == 0);      // AMODE 3: re-read every third step, reuse as is
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (rd) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[k][i] = *(const lds_u32x4*)(a_rel[i][k] + (off >> 1));
                } else if (AMODE == 1) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            fa[k][i][c] = (unsigned)__builtin_amdgcn_update_dpp((int)fa[k][i][c], (int)fa[k][i][c], 0x111, 0xf, 0xf, false);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[k][j] = *(const lds_u32x4*)(b_rel[j][k] + off);
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            if (INTERLEAVE) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[k2 + 2 * h][i] = *(const lds_u32x4*)(a_rel[i][k2 + 2 * h] + (off >> 1));
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[k2 + 2 * h][j] = *(const lds_u32x4*)(b_rel[j][k2 + 2 * h] + off);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (CH1) {
#pragma unroll
                    for (int p = 0; p < 3; ++p)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const bf16x8 b = __builtin_bit_cast(bf16x8, fb[p == 2 ? k2 + 2 : k2][j]);
                            const bf16x8 a = __builtin_bit_cast(bf16x8, fa[p == 1 ? k2 + 2 : k2][i]);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[i][j], 0, 0, 0);
                        }
                } else
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, fb[k2][j]), bl = __builtin_bit_cast(bf16x8, fb[k2 + 2][j]);
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, fa[k2][i]), al = __builtin_bit_cast(bf16x8, fa[k2 + 2][i]);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, acc[i][j], 0, 0, 0);
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al, accx[i][j], 0, 0, 0);
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah, accx[i][j], 0, 0, 0);
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) s += acc[i][j][0] + accx[i][j][5];
    if (s == 12345.678f) sink[0] = s;
}

// Opposite phases: the upper half of the waves (the second wave of every SIMD) computes one interval late -- inside a barrier
// interval group A does [read fragments of stage k -> MFMAs(k)] while group B does [MFMAs(k-1) from registers -> read stage k],
// so the LDS phase of one wave of a SIMD lies under the matrix phase of the other.  PHASED = 0: same loop, both groups in phase.
template <int NW, int TM, int TN, int PHASED>
__global__ __launch_bounds__(64 * NW) void step_loop_phased(int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 96 * 1024 / 4; i += 64 * NW) ((unsigned*)smem)[i] = 0x3f803f80u + (i & 255);
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)smem;
    const int l32 = lane & 31, fh = lane >> 5;
    unsigned a_rel[TM][4], b_rel[TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) { int r = ((wave % 4) * TM + i) * 32 + l32; a_rel[i][k] = lds0 + 49152 + r * 128 + ((((k * 2 + fh)) ^ ((r >> 1) & 7)) << 4); }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) { int r = ((wave / 4) * TN + j) * 32 + l32; b_rel[j][k] = lds0 + r * 128 + ((((k * 2 + fh)) ^ ((r >> 1) & 7)) << 4); }
    f32x16 acc[TM][TN], accx[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accx[i][j][r] = 0.f; }
    u32x4 fa[4][TM], fb[4][TN];
    auto rd = [&](int it) {
        const unsigned off = (it % 3) * 16384;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[k][i] = *(const lds_u32x4*)(a_rel[i][k] + (off >> 1));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[k][j] = *(const lds_u32x4*)(b_rel[j][k] + off);
        }
    };
    auto mm = [&]() {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
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
    };
    const bool late = PHASED && wave >= NW / 2;          // wave-uniform
    if (late) rd(0);
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
        if (late) { mm(); asm volatile("" ::: "memory"); rd(it); }
        else { rd(it); mm(); }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) s += acc[i][j][0] + accx[i][j][5];
    if (s == 12345.678f) sink[0] = s;
}

template <int NW, int TM, int TN, int PHASED>
static void run_phased(float* sink, const char* tag) {
    const int iters = 4000, blocks = 256;
    auto k = step_loop_phased<NW, TM, TN, PHASED>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<<<blocks, 64 * NW, 96 * 1024>>>(iters / 10, sink); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k<<<blocks, 64 * NW, 96 * 1024>>>(iters, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double flops = (double)blocks * NW * iters * (6.0 * TM * TN) * 32768.0;
    printf("%-44s waves %2d tile %dx%d phased %d                : %7.1f TFLOP/s MFMA  (%.0f ns/step)\n", tag, NW, TM, TN, PHASED,
           flops / ms / 1e9, ms * 1e6 / iters);
}

template <int NW, int TM, int TN, int BARRIER, int INTERLEAVE, int AMODE = 0, int CH1 = 0>
static void run(float* sink, const char* tag) {
    const int iters = 4000, blocks = 256;
    auto k = step_loop<NW, TM, TN, BARRIER, INTERLEAVE, AMODE, CH1>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<<<blocks, 64 * NW, 96 * 1024>>>(iters / 10, sink); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k<<<blocks, 64 * NW, 96 * 1024>>>(iters, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double flops = (double)blocks * NW * iters * (6.0 * TM * TN) * 32768.0;
    printf("%-44s waves %2d tile %dx%d barrier %d interleave %d : %7.1f TFLOP/s MFMA  (%.0f ns/step)\n", tag, NW, TM, TN, BARRIER, INTERLEAVE,
           flops / ms / 1e9, ms * 1e6 / iters);
}

template <int NW, int TM, int TN, int BARRIER, int INTERLEAVE, int AMODE = 0, int CH1 = 0>
static void hold(float* sink, double seconds, const char* tag) {
    auto k = step_loop<NW, TM, TN, BARRIER, INTERLEAVE, AMODE, CH1>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000; double total = 0; long n = 0;
    while (total < seconds * 1e3) {
        CK(hipEventRecord(e0)); k<<<256, 64 * NW, 96 * 1024>>>(iters, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); total += ms; ++n;
    }
    double flops = (double)256 * NW * iters * n * (6.0 * TM * TN) * 32768.0;
    printf("hold %-40s %.1f s : %7.1f TFLOP/s MFMA sustained\n", tag, total / 1e3, flops / total / 1e9);
}

int main(int argc, char** argv) {
    float* sink; CK(hipMalloc(&sink, 64));
    if (argc >= 4 && !strcmp(argv[1], "hold")) {      // one variant in a closed loop for N seconds (watch rocm-smi next to it)
        const int w = atoi(argv[2]); const double sec = atof(argv[3]);
        if (w == 0) hold<8, 1, 2, 1, 0>(sink, sec, "l3 shape");
        if (w == 1) hold<8, 1, 2, 0, 0>(sink, sec, "l3 shape, no barrier");
        if (w == 2) hold<8, 1, 2, 1, 0, 2>(sink, sec, "l3 shape, pixel frags never re-read");
        if (w == 3) hold<8, 2, 2, 1, 0>(sink, sec, "l2 shape");
        if (w == 4) hold<4, 2, 4, 1, 0>(sink, sec, "4 waves 2x4");
        if (w == 5) hold<8, 1, 2, 1, 0, 3>(sink, sec, "l3 shape, frags every 3rd step");
        return 0;
    }
    run<8, 1, 2, 1, 0>(sink, "l3 shape (128x128 tile, 8 waves)");
    run<8, 1, 2, 0, 0>(sink, "  no barrier");
    run<8, 1, 2, 1, 1>(sink, "  reads per k-slice");
    run<8, 2, 2, 1, 0>(sink, "l2 shape (256x128 tile, 8 waves)");
    run<8, 2, 2, 0, 0>(sink, "  no barrier");
    run<8, 1, 1, 1, 0>(sink, "l1/l4 shape (1x1 per wave, 8 waves)");
    run<4, 2, 2, 1, 0>(sink, "4 waves, 2x2 per wave");
    run<4, 2, 2, 0, 0>(sink, "  no barrier");
    run<4, 2, 4, 1, 0>(sink, "4 waves, 2x4 per wave");
    run<16, 1, 1, 1, 0>(sink, "16 waves, 1x1 per wave");
    run<8, 1, 2, 1, 0, 1>(sink, "l3 shape, pixel frags: 1 read + 2 DPP shifts");
    run<8, 1, 2, 1, 0, 2>(sink, "l3 shape, pixel frags never re-read");
    run<8, 2, 2, 1, 0, 1>(sink, "l2 shape, pixel frags: 1 read + 2 DPP shifts");
    run<8, 2, 2, 1, 0, 2>(sink, "l2 shape, pixel frags never re-read");
    run<8, 1, 1, 1, 0, 1>(sink, "l1/l4 shape, pixel frags: 1 read + 2 DPP shifts");
    run<8, 1, 1, 1, 0, 2>(sink, "l1/l4 shape, pixel frags never re-read");
    run<8, 1, 2, 1, 0, 3>(sink, "l3 shape, pixel frags read every 3rd step");
    run<8, 2, 2, 1, 0, 3>(sink, "l2 shape, pixel frags read every 3rd step");
    run<8, 1, 1, 1, 0, 3>(sink, "l1/l4 shape, pixel frags read every 3rd step");
    run<4, 2, 2, 1, 0, 3>(sink, "4 waves 2x2, pixel frags read every 3rd step");
    run<4, 2, 2, 1, 0, 2>(sink, "4 waves 2x2, pixel frags never re-read");
    run<4, 1, 2, 1, 0, 3>(sink, "4 waves 1x2, pixel frags read every 3rd step");
    run<4, 2, 1, 1, 0, 3>(sink, "4 waves 2x1, pixel frags read every 3rd step");
    run_phased<8, 1, 2, 0>(sink, "l3 shape, phased kernel, in phase");
    run_phased<8, 1, 2, 1>(sink, "l3 shape, upper waves one interval late");
    run_phased<8, 2, 2, 0>(sink, "l2 shape, phased kernel, in phase");
    run_phased<8, 2, 2, 1>(sink, "l2 shape, upper waves one interval late");
    run_phased<8, 1, 1, 0>(sink, "l1/l4 shape, phased kernel, in phase");
    run_phased<8, 1, 1, 1>(sink, "l1/l4 shape, upper waves one interval late");
    run<8, 1, 2, 1, 0>(sink, "l3 shape again (order check)");
    run<8, 1, 2, 1, 0, 3>(sink, "l3 shape, every 3rd step again");
    run<8, 1, 2, 1, 0>(sink, "l3 shape again (order check)");
    run<8, 1, 2, 1, 0, 3, 1>(sink, "l3 shape, every 3rd step, ONE accumulator chain");
    run<8, 1, 2, 1, 0, 0, 1>(sink, "l3 shape, every step, ONE accumulator chain");
    return 0;
}
