// Shared by conv3x3.hip (LDS weight ring) and conv3x3r.hip (weights resident in registers).
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
    int flip;                // 0: tap t reads input (t/3-1, t%3-1); 1 (data gradient): (1-t/3, 1-t%3).  Weight K offset = t*C.
};                           // (no per-tap tables: a dynamically indexed kernarg array becomes a VMEM load inside the K loop,
                             //  and the vmcnt wait for it would drain the in-flight LDS-DMA prefetch)


struct C3EvalBn { void* out_hi; void* out_lo; float* out_f32; const void* res_hi; const void* res_lo; int relu; };
