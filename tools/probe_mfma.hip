// Sustained rate of v_mfma_f32_32x32x16_bf16 with nothing else going on: every wave of the chip loops over CH independent
// accumulator chains.  Compare with the nominal 2.5 PFLOP/s (dense bf16) to see what the clocks under MFMA load allow.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma.hip -o tools/probe_mfma && tools/probe_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CH>
__global__ __launch_bounds__(256) void mfma_loop(int iters, float* sink) {
    f32x16 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = (float)(threadIdx.x + c);
    uint4 ua = make_uint4(0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u), ub = ua;
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][15];
    if (s == 12345.678f) sink[0] = s;
}

template <int CH>
static void run(int waves_per_cu, float* sink) {
    const int iters = 20000, blocks = 256 * waves_per_cu / 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    mfma_loop<CH><<<blocks, 256>>>(iters / 10, sink); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); mfma_loop<CH><<<blocks, 256>>>(iters, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double flops = (double)blocks * 4 * iters * CH * 32768.0;
    printf("chains/wave %d  waves/CU %2d : %7.1f TFLOP/s  (%.1f ms)\n", CH, waves_per_cu, flops / ms / 1e9, ms);
}

int main() {
    float* sink; CK(hipMalloc(&sink, 64));
    for (int w : {4, 8, 16}) { run<1>(w, sink); run<2>(w, sink); run<4>(w, sink); }
    return 0;
}
