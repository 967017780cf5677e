// Shared by conv3x3.hip (LDS weight ring) and conv3x3v.hip (weights streamed L2 -> VGPR in MFMA fragment order).
#pragma once
#include "conv_common.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;

struct Conv3Args {
    const void* X; const void* Wt; void* Out; const void* addend; float* stats;
    const void* X_lo; unsigned wlo_delta;   // X3 (split-bf16) launches: low-order plane of X; byte distance Wt_lo - Wt
    int N, H, W, C;          // input  [N,H,W,C]  (C % 64 == 0)
    int Cn;                  // output [N,H,W,Cn]
    int ktot;                // weight row length (9*C)
    int tiles_x, tiles_y;    // tiles per image
    // optional fused BatchNorm-backward reduction over the OUTPUT of this (data-gradient) launch: see ab_conv2d_dgrad_bnstats
    const void* bn_y; const void* bn_out; const float* bnp; float* bn_part;
    // X3 = 3 (eval-mode forward with the BatchNorm that follows folded in): Out / Out_lo are the (hi, lo) planes of
    // relu?(acc * bnp[c] + bnp[Cn + c] + residual); residual = res_hi + res_lo planes, or the fp32 `addend`; OutF (optional) = the fp32 value
    void* Out_lo; const void* res_hi; const void* res_lo; float* OutF; int ep_relu;
    unsigned long long* dbg; // conv3x3v.hip: per-wave s_memtime stamps (ab_c3v_debug_buffer; NULL: off)
    const void* Wf;          // conv3x3v.hip: the weights in MFMA fragment order (c3v_pack_kernel), both planes
    int flip;                // 0: tap t reads input (t/3-1, t%3-1); 1 (data gradient): (1-t/3, 1-t%3).  Weight K offset = t*C.
    // tools/probe_c3fold.hip only (C3_FOLD_PROBE): the input as the fp32 conv output `fold_y` of the layer below + its BatchNorm (scale | shift) in
    // fold_bnp [2][C] -- the patch is built through registers as the planes of relu(fold_y * scale + shift) instead of being DMA'd from X / X_lo
    const float* fold_y; const float* fold_bnp;
};                           // (no per-tap tables: a dynamically indexed kernarg array becomes a VMEM load inside the K loop,
                             //  and the vmcnt wait for it would drain the in-flight LDS-DMA prefetch)


struct C3EvalBn { void* out_hi; void* out_lo; float* out_f32; const void* res_hi; const void* res_lo; int relu; };
