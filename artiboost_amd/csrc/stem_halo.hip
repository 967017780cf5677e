// Stem convolution (7x7 / stride 2 / pad 3, 3 -> 64 channels; anakin/models/resnet.py:154) with an LDS-resident input halo.
//
// The tap-table kernel (conv_gemm2.hip, STEM) fills 96 KB of LDS per 128 output pixels -- every input pixel is fetched ~10x
// (7 kernel rows x overlapping 8-pixel runs) and the 28 KB of weights once per tile -- and is bound by that fill.  Here a
// persistent workgroup keeps its weight fragments in registers (each wave's 32 output channels x 224 = 14 B fragments, 56
// VGPRs, loaded once) and owns TH x TW = 8 x 16 output pixels at a time: their input
// patch (21 rows x 40 pixels of the zero-bordered NHWC4 image, 6.7 KB) is DMA'd once, double-buffered against the MFMAs of
// the previous tile, and every A fragment is a 16-byte LDS read straight from a patch row: output pixel (p, q), kernel row
// kh, K elements [j*16 + half*8, +8) are the bytes  (2p + kh) * 320 + 16 q + 32 j + 16 half  of the patch (pixel pitch
// 8 bytes, so consecutive q are consecutive 16-byte slots: conflict-free).  K = 7 x 32 = 224 = 14 MFMA k-slices.
// One barrier per tile: the staging tile and the BN partial buffer are double-buffered like the patch, so the 16-byte output
// stores of tile t retire under the MFMAs of tile t + 1 (on gfx9 a wave's vmcnt also counts its stores: waiting for the
// next patch right after issuing the stores would serialise every tile on the store acknowledgements).
// What is left is the 134 MB output write of the launch.
#include "conv_common.h"

#define SH_TH 8
#define SH_TW 16
#define SH_PROW 320                 // patch row pitch: 40 pixels x 8 bytes
#define SH_PSLOTS 420               // 21 rows x 20 sixteen-byte slots
#define SH_PATCH 7168               // 7 wave instructions of 1 KiB
#define SH_SPITCH 144               // staging row pitch (64 channels x 2 bytes + 16)
#define SH_STAGE (128 * SH_SPITCH)  // 18432
#define SH_STAT (4 * 64 * 2 * 4)    // 2048: [wave_m][channel][sum, sumsq]
#define SH_OFF_P 0
#define SH_OFF_S (2 * SH_PATCH)                         // 14336
#define SH_OFF_T (SH_OFF_S + 2 * SH_STAGE)              // 51200
#define SH_LDS (SH_OFF_T + 2 * SH_STAT)                 // 55296

struct StemArgs {
    const void* X; const void* Wt; void* Out; float* stats;
    int Ha, Wa, Ho, Wo, tiles_x, tiles_per_img, ntiles;
};

__global__ __launch_bounds__(512) void stem_halo_kernel(StemArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int l32 = lane & 31, fhalf = lane >> 5;
    const bf16_t* __restrict__ X = (const bf16_t*)g.X;
    const bf16_t* __restrict__ Wt = (const bf16_t*)g.Wt;
    bf16_t* __restrict__ Out = (bf16_t*)g.Out;
    const unsigned lds0 = lds_addr_of(smem);

    // B fragments of this wave's 32 output channels, all 14 k-slices, resident in registers for the life of the workgroup
    uint4 fb[14];
#pragma unroll
    for (int kk = 0; kk < 14; ++kk) fb[kk] = *(const uint4*)(Wt + (wave_n * 32 + l32) * 224 + kk * 16 + fhalf * 8);

    auto issue_patch = [&](int tile, int buf) {          // 420 sixteen-byte slots: waves 0..6, one instruction each
        if (wave < 7) {
            const int s = wave * 64 + lane;
            if (s < SH_PSLOTS) {
                const int img = tile / g.tiles_per_img, rem = tile - img * g.tiles_per_img;
                const int ty0 = (rem / g.tiles_x) * SH_TH, tx0 = (rem % g.tiles_x) * SH_TW;
                const int row = s / 20, c16 = s - row * 20;
                const bf16_t* src = X + ((((long)img * g.Ha + 2 * ty0 + row) * g.Wa + 2 * tx0 + c16 * 2) * 4);
                glds16(src, __builtin_amdgcn_readfirstlane(lds0 + SH_OFF_P + buf * SH_PATCH + wave * 1024));
            }
        }
    };

    // this lane's fragment bases: MFMA row m = l32 is output pixel (p, q) = (wave_m * 2 + m / 16, m % 16) of the tile
    const int pq = l32 & 15, pp = wave_m * 2 + (l32 >> 4);
    const unsigned a_off = (unsigned)(2 * pp * SH_PROW + pq * 16 + fhalf * 16);

    float wg_sum = 0.f, wg_sq = 0.f;                     // threads 0..63: this workgroup's BN partial of channel tid
    int tile = blockIdx.x, buf = 0;
    if (tile < g.ntiles) issue_patch(tile, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (; tile < g.ntiles; tile += gridDim.x) {
        // buffer parity `buf` of patch, staging and stat buffers belongs to this tile; the other parity's readers (MFMAs and
        // output stores of the previous tile) all passed the previous barrier with lgkmcnt(0)
        const int next = tile + gridDim.x;
        if (next < g.ntiles) issue_patch(next, buf ^ 1);

        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const unsigned char* pa = smem + SH_OFF_P + buf * SH_PATCH + a_off;
#pragma unroll
        for (int kk = 0; kk < 14; ++kk) {
            const uint4 fa = *(const uint4*)(pa + (kk >> 1) * SH_PROW + (kk & 1) * 32);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb[kk]), acc, 0, 0, 0);
        }

        // ---- epilogue: bf16 tile through LDS -> 16-byte stores; BN partial sums of this tile
        const int cl = wave_n * 32 + l32;
        unsigned char* stg = smem + SH_OFF_S + buf * SH_STAGE;
        float* s_stat = (float*)(smem + SH_OFF_T + buf * SH_STAT);
        float csum = 0.f, csq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
            const float v = acc[r];
            *(bf16_t*)(stg + row * SH_SPITCH + cl * 2) = f32_to_bf16(v);
            csum += v; csq += v * v;
        }
        if (g.stats) {
            const float s = csum + __shfl_xor(csum, 32, 64), q = csq + __shfl_xor(csq, 32, 64);
            if (lane < 32) { s_stat[(wave_m * 64 + cl) * 2] = s; s_stat[(wave_m * 64 + cl) * 2 + 1] = q; }
        }
        // the one barrier of the tile: staging + partials visible, next patch landed (the older output stores of the previous
        // tile retired long ago), and every LDS read of this tile's patch has returned (WAR against the DMA after next)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int img = tile / g.tiles_per_img, rem = tile - img * g.tiles_per_img;
        const int ty0 = (rem / g.tiles_x) * SH_TH, tx0 = (rem % g.tiles_x) * SH_TW;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int id = tid + u * 512;                  // 128 rows x 8 sixteen-byte chunks
            const int row = id >> 3, c8 = id & 7;
            const int p = row >> 4, q = row & 15;
            const uint4 v = *(const uint4*)(stg + row * SH_SPITCH + c8 * 16);
            uint4* dst = (uint4*)(Out + ((((long)img * g.Ho + ty0 + p) * g.Wo + tx0 + q) * 64 + c8 * 8));
            *dst = v;
        }
        if (g.stats && tid < 64) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int wm = 0; wm < 4; ++wm) { s += s_stat[(wm * 64 + tid) * 2]; q += s_stat[(wm * 64 + tid) * 2 + 1]; }
            wg_sum += s; wg_sq += q;
        }
        buf ^= 1;
    }
    // one partial row per workgroup (its tiles in launch order): 16x fewer rows for the finalize launch to reduce
    if (g.stats && tid < 64) {
        g.stats[((long)blockIdx.x * 64 + tid) * 2] = wg_sum;
        g.stats[((long)blockIdx.x * 64 + tid) * 2 + 1] = wg_sq;
    }
}

// bf16 stem forward; H, W = image size (output H/2 x W/2).  Returns AB_ESHAPE when the tiling does not fit (caller falls back).
static int stem_halo_ntiles(int N, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    if (Ho % SH_TH || Wo % SH_TW) return 0;
    return N * (Ho / SH_TH) * (Wo / SH_TW);
}
static int stem_halo_grid(int ntiles) {
    static const int per_cu = getenv("AB_STEM_HALO_WGS") ? atoi(getenv("AB_STEM_HALO_WGS")) : 2;
    return ntiles < 256 * per_cu ? ntiles : 256 * per_cu;
}
// number of BN-partial rows the launch writes (one per workgroup); 0 = shape not handled
int stem_halo_tiles(int N, int H, int W) { return stem_halo_grid(stem_halo_ntiles(N, H, W)); }
int stem_halo_run(const void* xpad, const void* w, void* y, int N, int H, int W, int Cout, float* stats, hipStream_t st) {
    const int ntiles = stem_halo_ntiles(N, H, W);
    if (!ntiles || Cout != 64) return AB_ESHAPE;
    StemArgs g;
    g.X = xpad; g.Wt = w; g.Out = y; g.stats = stats;
    g.Ha = H + 6; g.Wa = W + 8; g.Ho = H / 2; g.Wo = W / 2;
    g.tiles_x = g.Wo / SH_TW; g.tiles_per_img = g.tiles_x * (g.Ho / SH_TH); g.ntiles = ntiles;
    const int grid = stem_halo_grid(ntiles);
    stem_halo_kernel<<<grid, 512, SH_LDS, st>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x3") stem: image and weights as (hi, lo) bf16 planes, fp32 output and BN partials.  Same persistent-tile
// structure; both planes of the patch are DMA'd per tile (2 x 6.7 KB), the weight fragments of both planes stay in registers
// (28 B fragments = 112 VGPRs), a k-slice is three MFMAs (hi*hi on the main chain, hi*lo + lo*hi on the cross chain), the
// staging tile is fp32 (two buffers of 34 KB: one workgroup per CU, the stores of tile t under the MFMAs of tile t + 1).
#define SX_SPITCH 272               // 64 channels x 4 bytes + 16
#define SX_STAGE (128 * SX_SPITCH)  // 34816
#define SX_OFF_P 0                                      // [buf][plane] patches of SH_PATCH bytes
#define SX_OFF_S (4 * SH_PATCH)                         // 28672
#define SX_OFF_T (SX_OFF_S + 2 * SX_STAGE)              // 98304
#define SX_LDS (SX_OFF_T + 2 * SH_STAT)                 // 102400

struct StemArgsX3 {
    const void* X; const void* Xlo; const void* Wt; const void* Wtlo; float* Out; float* stats;
    int Ha, Wa, Ho, Wo, tiles_x, tiles_per_img, ntiles;
};

// U8N: the image is ONE plane of odd integers n = 2 v - 255 (v the uint8 pixel; exact in bf16) -- what the loaders' warp / jitter pass writes
// with output code AB_DT_U8N -- and the convolution's input is n / 510 = v / 255 - 0.5 (anakin/datasets/hodata.py:446 after func.to_tensor):
// a k-slice is TWO MFMAs (n . w_hi, n . w_lo: no rounding of the image at all, where the split of the fp32 image dropped its lo . lo term),
// one patch plane is DMA'd, the factor 1 / 510 rides in the epilogue.
template <bool U8N>
__global__ __launch_bounds__(512) void stem_halo_x3_kernel(StemArgsX3 g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int l32 = lane & 31, fhalf = lane >> 5;
    const bf16_t* __restrict__ Xh = (const bf16_t*)g.X;
    const bf16_t* __restrict__ Xl = (const bf16_t*)g.Xlo;
    const bf16_t* __restrict__ Wh = (const bf16_t*)g.Wt;
    const bf16_t* __restrict__ Wl = (const bf16_t*)g.Wtlo;
    float* __restrict__ Out = g.Out;
    const unsigned lds0 = lds_addr_of(smem);

    uint4 fbh[14], fbl[14];
#pragma unroll
    for (int kk = 0; kk < 14; ++kk) {
        fbh[kk] = *(const uint4*)(Wh + (wave_n * 32 + l32) * 224 + kk * 16 + fhalf * 8);
        fbl[kk] = *(const uint4*)(Wl + (wave_n * 32 + l32) * 224 + kk * 16 + fhalf * 8);
    }

    auto issue_patch = [&](int tile, int buf) {          // per plane 420 sixteen-byte slots: waves 0..6 one instruction per plane
        if (wave < 7) {
            const int s = wave * 64 + lane;
            if (s < SH_PSLOTS) {
                const int img = tile / g.tiles_per_img, rem = tile - img * g.tiles_per_img;
                const int ty0 = (rem / g.tiles_x) * SH_TH, tx0 = (rem % g.tiles_x) * SH_TW;
                const int row = s / 20, c16 = s - row * 20;
                const long e = (((long)img * g.Ha + 2 * ty0 + row) * g.Wa + 2 * tx0 + c16 * 2) * 4;
                glds16(Xh + e, __builtin_amdgcn_readfirstlane(lds0 + SX_OFF_P + (buf * 2) * SH_PATCH + wave * 1024));
                if constexpr (!U8N) glds16(Xl + e, __builtin_amdgcn_readfirstlane(lds0 + SX_OFF_P + (buf * 2 + 1) * SH_PATCH + wave * 1024));
            }
        }
    };

    const int pq = l32 & 15, pp = wave_m * 2 + (l32 >> 4);
    const unsigned a_off = (unsigned)(2 * pp * SH_PROW + pq * 16 + fhalf * 16);

    float wg_sum = 0.f, wg_sq = 0.f;
    int tile = blockIdx.x, buf = 0;
    if (tile < g.ntiles) issue_patch(tile, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (; tile < g.ntiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        if (next < g.ntiles) issue_patch(next, buf ^ 1);

        f32x16 acc, accx;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accx[r] = 0.f; }
        const unsigned char* pah = smem + SX_OFF_P + (buf * 2) * SH_PATCH + a_off;
        const unsigned char* pal = pah + SH_PATCH;
#pragma unroll
        for (int kk = 0; kk < 14; ++kk) {
            const uint4 fah = *(const uint4*)(pah + (kk >> 1) * SH_PROW + (kk & 1) * 32);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fah), __builtin_bit_cast(bf16x8, fbh[kk]), acc, 0, 0, 0);
            accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fah), __builtin_bit_cast(bf16x8, fbl[kk]), accx, 0, 0, 0);
            if constexpr (!U8N) {
                const uint4 fal = *(const uint4*)(pal + (kk >> 1) * SH_PROW + (kk & 1) * 32);
                accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fal), __builtin_bit_cast(bf16x8, fbh[kk]), accx, 0, 0, 0);
            }
        }

        const int cl = wave_n * 32 + l32;
        unsigned char* stg = smem + SX_OFF_S + buf * SX_STAGE;
        float* s_stat = (float*)(smem + SX_OFF_T + buf * SH_STAT);
        float csum = 0.f, csq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
            const float v = U8N ? (acc[r] + accx[r]) * (1.0f / 510.0f) : acc[r] + accx[r];
            *(float*)(stg + row * SX_SPITCH + cl * 4) = v;
            csum += v; csq += v * v;
        }
        if (g.stats) {
            const float s = csum + __shfl_xor(csum, 32, 64), q = csq + __shfl_xor(csq, 32, 64);
            if (lane < 32) { s_stat[(wave_m * 64 + cl) * 2] = s; s_stat[(wave_m * 64 + cl) * 2 + 1] = q; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int img = tile / g.tiles_per_img, rem = tile - img * g.tiles_per_img;
        const int ty0 = (rem / g.tiles_x) * SH_TH, tx0 = (rem % g.tiles_x) * SH_TW;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id = tid + u * 512;                  // 128 rows x 16 sixteen-byte chunks (4 channels each)
            const int row = id >> 4, c4 = id & 15;
            const int p = row >> 4, q = row & 15;
            const float4 v = *(const float4*)(stg + row * SX_SPITCH + c4 * 16);
            *(float4*)(Out + ((((long)img * g.Ho + ty0 + p) * g.Wo + tx0 + q) * 64 + c4 * 4)) = v;
        }
        if (g.stats && tid < 64) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int wm = 0; wm < 4; ++wm) { s += s_stat[(wm * 64 + tid) * 2]; q += s_stat[(wm * 64 + tid) * 2 + 1]; }
            wg_sum += s; wg_sq += q;
        }
        buf ^= 1;
    }
    if (g.stats && tid < 64) {
        g.stats[((long)blockIdx.x * 64 + tid) * 2] = wg_sum;
        g.stats[((long)blockIdx.x * 64 + tid) * 2 + 1] = wg_sq;
    }
}

static int stem_halo_x3_grid(int ntiles) { return ntiles < 256 ? ntiles : 256; }      // 100 KB of LDS: one workgroup per CU
int stem_halo_x3_tiles(int N, int H, int W) {
    if (getenv("AB_STEM_HALO_X3") && !atoi(getenv("AB_STEM_HALO_X3"))) return 0;
    return stem_halo_x3_grid(stem_halo_ntiles(N, H, W));
}
int stem_halo_x3_run(const void* xpad_hi, const void* xpad_lo, const void* w_hi, const void* w_lo, float* y, int N, int H, int W,
                     int Cout, float* stats, hipStream_t st) {
    const int ntiles = stem_halo_ntiles(N, H, W);
    if (!ntiles || Cout != 64) return AB_ESHAPE;
    StemArgsX3 g;
    g.X = xpad_hi; g.Xlo = xpad_lo; g.Wt = w_hi; g.Wtlo = w_lo; g.Out = y; g.stats = stats;
    g.Ha = H + 6; g.Wa = W + 8; g.Ho = H / 2; g.Wo = W / 2;
    g.tiles_x = g.Wo / SH_TW; g.tiles_per_img = g.tiles_x * (g.Ho / SH_TH); g.ntiles = ntiles;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)stem_halo_x3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SX_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)stem_halo_x3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SX_LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    if (!xpad_lo) stem_halo_x3_kernel<true><<<stem_halo_x3_grid(ntiles), 512, SX_LDS, st>>>(g);      // the integer image plane (AB_DT_U8N)
    else stem_halo_x3_kernel<false><<<stem_halo_x3_grid(ntiles), 512, SX_LDS, st>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
