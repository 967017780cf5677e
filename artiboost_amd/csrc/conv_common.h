// Shared by conv_gemm.hip (register-staged, any dtype) and conv_gemm2.hip (direct-to-LDS bf16 fast path).
#pragma once
#include "common.h"

#define CG_MAXTAPS 16

struct ConvGemmArgs {
    const void* A; const void* Bw; void* Out;
    const void* A_lo; const void* Bw_lo;        // split-bf16 (X3) launches of conv_gemm2: the low-order planes of A and Bw
    const float* bias; const void* addend; float* stats;   // stats: [gridM][Cn][2]
    int N, Ha, Wa, Ca;          // A tensor dims (Ca = channel pitch in elements)
    int P, Q;                   // output sub-grid
    int Ho, Wo, Cn;             // full output tensor dims
    int out_sh, out_sw, out_oh, out_ow;
    int a_sh, a_sw;
    int ntaps, cpt;             // cpt = K-steps per tap (Ca / BK, or 1 for the padded stem rows)
    int ktot;                   // Bw row length in elements
    int M;
    int relu;                   // apply ReLU in the epilogue (linear layers)
    int nmajor;                 // conv_gemm2: walk the tiles N-major inside an XCD's range (weights > L2, see conv_gemm2_run)
    // conv_gemm2 only: nclass > 1 runs the output-parity classes of a strided data gradient in ONE grid.  Class c owns the
    // M tiles [c*tiles_m, (c+1)*tiles_m), writes output pixels (out_oh, out_ow) = (cls_oh[c], cls_ow[c]) and uses the taps
    // [4c, 4c + cls_ntaps[c]) of dh/dw/koff (0 taps: the class receives no gradient -> zeros / the addend).
    int nclass;
    int cls_ntaps[4]; int cls_oh[4], cls_ow[4];
    int8_t dh[CG_MAXTAPS], dw[CG_MAXTAPS];
    int koff[CG_MAXTAPS];
    // conv_gemm2 X3 only: tap `alt_tap1 - 1` (absolute index into dh/dw/koff; 0: none) reads a SECOND operand pair -- A2 / Bw2 with
    // weight row length ktot2 -- instead of A / Bw: the 1x1/s2 downsample branch rides in the data gradient of the stride-2 3x3
    // convolution next to it as one more tap of parity class (even, even).
    const void* A2; const void* A2_lo; const void* Bw2; const void* Bw2_lo;
    int alt_tap1, ktot2;
    // conv_gemm2 X3 only, eval-mode forwards: per-channel affine of the BatchNorm that follows -- out = relu?(acc * ep_scale[c] + bias[c])
    // (`bias` carries the shift) -- and, when out_hi != NULL, the result leaves as (hi, lo) bf16 planes instead of fp32 `Out`
    const float* ep_scale; void* out_hi; void* out_lo;
    // conv_gemm2 X3 only, data gradients arriving at relu(bn(bn_y)) (no residual: the mask is recomputed from bn_y): Out receives the MASKED
    // gradient and bn_part [M tiles][Cn][2] the tile's (sum dz, sum dz * xhat) -- conv3x3.hip's X3 = 2 epilogue on the generic kernel
    const float* bn_y; const float* bnp; float* bn_part;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// K bytes per step: 128 (64 bf16) on the fast path -- full 128-byte rows per pixel keep the gather on whole cache
// lines and put 16 MFMAs between barriers; 64 (16 f32) on the exact parity path.  LDS row pitch = KB + 16 bytes:
// 16 consecutive rows then land on 16 distinct 16-byte slots (conflict-free ds_read_b128 / ds_write_b128).

// Direct-to-LDS 16-byte load (1 KiB per wave instruction: LDS destination = wave-uniform base + lane*16).
// Issued as inline asm on purpose: hipcc treats a __builtin_amdgcn_global_load_lds as an LDS store that may alias every
// later ds_read and drains vmcnt to 0 before the first fragment read of each K step (seen in the ISA), which destroys
// the multi-step prefetch.  As asm the load is invisible to the compiler's wait-count bookkeeping; the kernels place
// their own counted s_waitcnt vmcnt(N) + s_barrier (cdna_hip_programming.md section 5.7).  M0 is saved/restored.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}

// (tile, slice) of a workgroup in a (tiles, slices) grid, renumbered so that each XCD (dispatch order mod 8) owns a contiguous
// range of slices with all their tiles: the tiles of a slice stream the same operand rows, which then come out of that XCD's L2
// once instead of crossing the fabric once per XCD that happens to host one of them.  Bijective for any grid size.
// Measured on the weight-gradient kernels (AB_WG_XCD=1): 2 171 vs 2 169 us per step over all their launches (round 2) and 8.975 vs 8.992 ms per
// step (round 5) -- they are not fabric-bound -- but 2.3 GB per step less crosses the fabric (FETCH_SIZE of the conv stack 16.6 -> 14.2 GB:
// wgrad3x3 3.27 -> 1.81 GB, wgrad_gemm2 1.88 -> 1.00 GB); on by default since round 5.
__device__ __forceinline__ void xcd_slice_major(int on, int& tile, int& slice) {
    tile = blockIdx.x; slice = blockIdx.y;
    if (on) {
        const int nx = gridDim.x, nblk = nx * gridDim.y, bid = slice * nx + tile;
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, within = bid >> 3;
        const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
        slice = logical / nx; tile = logical - slice * nx;
    }
}

int conv_gemm2_run(ConvGemmArgs& g, hipStream_t st);     // conv_gemm2.hip; returns AB_ESHAPE when the shape is unsupported
