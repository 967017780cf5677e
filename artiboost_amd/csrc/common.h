// Shared device helpers for the artiboost_hip kernels (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/artiboost_hip.h"

#define AB_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

typedef uint16_t bf16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, NaN preserved (same as torch's float->bfloat16)
// round-to-nearest-even; on gfx950 the conversion is one v_cvt_pk_bf16_f32 (the bit-twiddled form is 7 VALU ops, which made
// the fused BN + ReLU + max-pool kernel VALU-bound)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
// two floats -> packed bf16 pair (lo in bits 0..15): one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
template <typename T> __device__ __forceinline__ float ld_f32(const T* p);
template <> __device__ __forceinline__ float ld_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_f32<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void st_f32(T* p, float v);
template <> __device__ __forceinline__ void st_f32<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_f32<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
