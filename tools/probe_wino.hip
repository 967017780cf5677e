// Upper bound for a fused Winograd F(2x2,3x3) split-bf16 convolution on gfx950 (round-3 review item 2), timing only.
//
// F(2x2,3x3): 16 products per 4 outputs instead of 36 (2.25x fewer MFMAs); the price is (a) sixteen [tiles x C] x [C x Cout]
// GEMMs whose accumulators must ALL stay live over the channel loop, (b) a weight stream of 16/9 the bytes, (c) the input
// transform B^T d B + re-split to (hi, lo) on the VALU, written to LDS with ds_write (the direct kernel DMA's its patch).
//
// The largest tile the register file allows: 64 tiles (16 x 16 pixels) x 64 output channels x 16 positions = 65 536 fp32
// accumulators = 256 per lane on 4 waves x 512 registers (one wave per SIMD, wave a owns position row a = 4 positions x 2 x 2
// blocks of 32 x 32).  Layer 3 of ResNet-34 at B = 64 (16 x 16 images, 256 -> 256 channels): 64 images x 4 channel tiles = 256
// workgroups = one round, 16 steps of 16 channels each.  Per step and wave:
//   U fragments : 4 positions x 2 channel blocks x (hi, lo) = 16 global_load_dwordx4, fragment-ordered (1 KiB contiguous per wave
//                 instruction), straight to VGPRs: no other wave needs them.  268 MB per launch from a 4.2 MB L2-resident tensor.
//   V fragments : 4 x 2 x (hi, lo) = 16 ds_read_b128 (conflict-free [position][plane][half][tile][16 B] rows)
//   MFMA        : 4 positions x 4 blocks x 3 products = 48 v_mfma_f32_32x32x16_bf16
//   V transform : MODE >= 2.  One (tile, 4 channels) item per lane: 32 ds_read_b64 of the 4 x 4 patch (hi, lo), ~2 ops per value to
//                 rebuild fp32, 32 adds per channel for B^T d B, ~2.5 ops per value to re-split, 32 ds_write_b64: ~480 VALU, 64 DS.
//                 The arithmetic here is a stand-in with the same instruction counts and the same LDS traffic (values are garbage).
// MODE 0: MFMA + V fragment reads only.  MODE 1: + the U stream.  MODE 2: + the transform's DS traffic.  MODE 3: + its VALU.
// Compare with the direct kernel's measured layer-3 launch (conv3x3_kernel<128,16,128,4,2,0,1>: 50.7 us, 19.33 GFLOP) and the
// kill criterion of the review (1.3x = 39 us): prologue, epilogue (output transform through LDS, BatchNorm partials, stores) and
// the exposed patch DMA come ON TOP of what this loop measures.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_wino.hip -o tools/probe_wino && tools/probe_wino
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) u32x2 lds_u32x2;

constexpr int NSTEP = 16;          // 256 channels / 16
constexpr int V_BYTES = 16 * 2 * 2 * 64 * 16;      // [16 positions][2 planes][2 halves][64 tiles][16 B] = 64 KiB
// patch: 18 x 18 pixels x (16 ch hi | 16 ch lo) x 2 B, pixel pitch 80 B and row pitch 386 dwords: with these pitches the 32 lanes of
// a ds_read_b64 group (16 tiles x 2 channel quads: 8 x-positions 2 pixels apart, 2 tile rows) fall on 64 distinct banks
constexpr int PIX_PITCH = 80, ROW_PITCH = 386 * 4;
constexpr int PATCH_BYTES = 18 * ROW_PITCH;

template <int MODE>
__global__ __launch_bounds__(256, 1) void wino_loop(const u32x4* __restrict__ U, float* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (2 * V_BYTES + PATCH_BYTES) / 4; i += 256) ((unsigned*)smem)[i] = 0x3f803f80u + (i & 255);
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)smem;
    // V fragment of (position p, plane, tile block tb): lanes 0-31 read half 0 of tiles tb*32 .. +31, lanes 32-63 half 1
    const int l32 = lane & 31, fh = lane >> 5;
    unsigned v_rel[4][2][2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
                v_rel[p][pl][tb] = lds0 + ((((wave * 4 + p) * 2 + pl) * 2 + fh) * 64 + tb * 32 + l32) * 16;
    // transform item of this lane: tile t (ty = t >> 3, tx = t & 7) and channel quad cq, mapped so that a wave's 64 lanes write one
    // contiguous 512-byte run per (position, plane): t = (wave & 1) * 32 + (lane >> 1), cq = (wave >> 1) * 2 + (lane & 1)
    const int t = (wave & 1) * 32 + (lane >> 1), cq = (wave >> 1) * 2 + (lane & 1);
    const unsigned patch0 = lds0 + 2 * V_BYTES + ((t >> 3) * 2) * ROW_PITCH + ((t & 7) * 2) * PIX_PITCH + cq * 8;
    const unsigned vw0 = lds0 + (((cq >> 1) * 64 + t) * 16 + (cq & 1) * 8);     // + position * 4096 + plane * 2048 (+ buffer)
    f32x16 acc[4][2][2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][a][b][r] = 0.f;
    // U: [channel tile (blockIdx & 3)][step][wave][16 fragments][64 lanes] x 16 B
    const u32x4* ub = U + ((long)(blockIdx.x & 3) * NSTEP * 4 + wave) * 16 * 64 + lane;
    u32x4 fu[4][2][2];
    auto load_u = [&](int step) {
        const u32x4* s = ub + (long)step * 4 * 16 * 64;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) fu[p][cb][pl] = s[((p * 2 + cb) * 2 + pl) * 64];
    };
    if (MODE >= 1) load_u(0);
    else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) fu[p][cb][pl] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    }
    float keep = 0.f;
    for (int rep = 0; rep < reps; ++rep)
        for (int step = 0; step < NSTEP; ++step) {
            const unsigned vbuf = (step & 1) * V_BYTES, vnext = ((step + 1) & 1) * V_BYTES;
            u32x4 fv[4][2][2];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb) fv[p][pl][tb] = *(const lds_u32x4*)(v_rel[p][pl][tb] + vbuf);
            // ---- the transform of the NEXT step's 16 channels (DS traffic and VALU of one (tile, 4 channels) item per lane)
            float d[4][4];      // running 4 x 4 tile of ONE channel at a time keeps the stand-in's register use honest
            u32x2 ph[16], plo[16];
            if (MODE >= 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    ph[i] = *(const lds_u32x2*)(patch0 + (i >> 2) * ROW_PITCH + (i & 3) * PIX_PITCH);
                    plo[i] = *(const lds_u32x2*)(patch0 + (i >> 2) * ROW_PITCH + (i & 3) * PIX_PITCH + 32);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb) {
                        const bf16x8 uh = __builtin_bit_cast(bf16x8, fu[p][cb][0]), ul = __builtin_bit_cast(bf16x8, fu[p][cb][1]);
                        const bf16x8 vh = __builtin_bit_cast(bf16x8, fv[p][0][tb]), vl = __builtin_bit_cast(bf16x8, fv[p][1][tb]);
                        acc[p][cb][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh, vh, acc[p][cb][tb], 0, 0, 0);
                        acc[p][cb][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh, vl, acc[p][cb][tb], 0, 0, 0);
                        acc[p][cb][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ul, vh, acc[p][cb][tb], 0, 0, 0);
                    }
                if (MODE >= 1 && step + 1 < NSTEP) {      // position p's U registers are free: refill them for the next step
                    const u32x4* s = ub + (long)(step + 1) * 4 * 16 * 64;
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) fu[p][cb][pl] = s[((p * 2 + cb) * 2 + pl) * 64];
                }
                if (MODE >= 2) {
                    // channel p of this lane's quad: rebuild fp32 (MODE 3), B^T d B, re-split, store the 16 positions' (hi, lo)
                    // halves -- written as one ds_write_b64 per (position, plane) once all four channels are done (below)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const unsigned hw = (p < 2) ? ph[i].x : ph[i].y, lw = (p < 2) ? plo[i].x : plo[i].y;
                        if (MODE >= 3) {
                            const float h = __uint_as_float((p & 1) ? (hw & 0xffff0000u) : (hw << 16));
                            const float l = __uint_as_float((p & 1) ? (lw & 0xffff0000u) : (lw << 16));
                            d[i >> 2][i & 3] = h + l;
                        } else d[i >> 2][i & 3] = __uint_as_float(hw ^ lw);
                    }
                    if (MODE >= 3) {
                        float e[4][4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {      // B^T d
                            e[0][c] = d[0][c] - d[2][c]; e[1][c] = d[1][c] + d[2][c]; e[2][c] = d[2][c] - d[1][c]; e[3][c] = d[1][c] - d[3][c];
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {      // (B^T d) B
                            d[r][0] = e[r][0] - e[r][2]; d[r][1] = e[r][1] + e[r][2]; d[r][2] = e[r][2] - e[r][1]; d[r][3] = e[r][1] - e[r][3];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float v = d[i >> 2][i & 3];
                        unsigned hb, lb;
                        if (MODE >= 3) {
                            const __bf16 hh = (__bf16)v;
                            const float back = __uint_as_float((unsigned)__builtin_bit_cast(unsigned short, hh) << 16);
                            const __bf16 ll = (__bf16)(v - back);
                            hb = __builtin_bit_cast(unsigned short, hh); lb = __builtin_bit_cast(unsigned short, ll);
                        } else { hb = __float_as_uint(v) >> 16; lb = __float_as_uint(v) & 0xffffu; }
                        // pack channel p into the quad's (hi, lo) dwords: reuse ph / plo as the outgoing registers
                        if (p == 0) { ph[i].x = hb; plo[i].x = lb; }
                        else if (p == 1) { ph[i].x |= hb << 16; plo[i].x |= lb << 16; }
                        else if (p == 2) { ph[i].y = hb; plo[i].y = lb; }
                        else { ph[i].y |= hb << 16; plo[i].y |= lb << 16; }
                    }
                }
            }
            if (MODE >= 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    *(lds_u32x2*)(vw0 + vnext + i * 4096) = ph[i];
                    *(lds_u32x2*)(vw0 + vnext + i * 4096 + 2048) = plo[i];
                }
            } else keep += d[0][0] * 0.f;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    float s = keep;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) s += acc[p][a][b][0] + acc[p][a][b][7];
    if (s == 12345.678f) sink[0] = s;
}

template <int MODE>
static void run(const u32x4* U, float* sink, const char* tag) {
    const int lds = 2 * V_BYTES + PATCH_BYTES, blocks = 256, reps = 40;
    auto k = wino_loop<MODE>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<<<blocks, 256, lds>>>(U, sink, 4); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k<<<blocks, 256, lds>>>(U, sink, reps); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;                                  // one layer-3 launch worth of K loop
    const double mfma = 256.0 * 4 * NSTEP * 48 * 32768.0 * 2 / 2;      // MFMA flops per launch (2 * MAC; 16384 MAC per instruction)
    printf("%-58s : %6.1f us per layer-3 K loop  (%.2f PFLOP/s of MFMA work; U stream %.1f TB/s)\n", tag, us, mfma * 2 / us / 1e9 / 2,
           MODE >= 1 ? 256.0 * NSTEP * 4 * 16 * 1024 / us / 1e6 : 0.0);
}

int main() {
    float* sink; CK(hipMalloc(&sink, 64));
    u32x4* U; const size_t ub = (size_t)4 * NSTEP * 4 * 16 * 64 * 16;  // 4 channel tiles x 16 steps x 4 waves x 16 fragments x 1 KiB = 4 MiB
    CK(hipMalloc(&U, ub)); CK(hipMemset(U, 0x3f, ub));
    for (int round = 0; round < 2; ++round) {
        run<0>(U, sink, "MODE 0: MFMA + V fragment reads");
        run<1>(U, sink, "MODE 1: + U fragment stream (global -> VGPR)");
        run<2>(U, sink, "MODE 2: + transform DS traffic (32 rd b64 + 32 wr b64 / lane)");
        run<3>(U, sink, "MODE 3: + transform VALU (rebuild, B^T d B, re-split)");
    }
    printf("reference: direct conv3x3_kernel<128,16,128,4,2,0,1> layer-3 launch 50.7 us whole (profiles/round2_h_step_trace.txt); 1.3x = 39 us whole launch\n");
    return 0;
}
