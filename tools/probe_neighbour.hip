// Probe (round 6): which kind of co-resident workgroup of ANOTHER kernel disturbs a VALU kernel (tools/render_race_debug.py)?  Answer: any wave that
// executes v_mfma_f32_32x32x16_bf16 on the same SIMD corrupts packed-fp32 VALU results (v_pk_mul/fma/add_f32) of its neighbours; LDS traffic does not.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/libprobe_neighbour.so tools/probe_neighbour.hip
//   mode 0: allocate `lds` bytes of dynamic LDS, never touch it, spin;  mode 1: fill the allocation with ds_write;  mode 2: fill it with LDS-DMA
#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
__global__ __launch_bounds__(512) void lds_neighbour(const uint4* src, int lds_bytes, int mode, long spin, unsigned* sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)smem;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned acc = 0;
    for (long it = 0; it < spin; ++it) {
        if (mode == 1) for (int o = threadIdx.x * 16; o + 16 <= lds_bytes; o += 512 * 16) *(uint4*)(smem + o) = make_uint4(0xdeadbeefu, it, o, 7);
        if (mode == 2) {      // stream: every fill from another 1 KiB of a 16 MB source (hits L2 / HBM, not one hot line set)
            for (int o = wave * 1024; o + 1024 <= lds_bytes; o += 8 * 1024)
                glds16(src + ((((long)blockIdx.x * 131 + it * 977 + o / 1024) * 64 + lane) & 0xfffff), __builtin_amdgcn_readfirstlane(lds0 + o));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (mode == 5) {      // the same stream as ordinary 16-byte loads into registers
            for (int o = wave * 1024; o + 1024 <= lds_bytes; o += 8 * 1024) {
                const uint4 v = src[(((long)blockIdx.x * 131 + it * 977 + o / 1024) * 64 + lane) & 0xfffff];
                acc += v.x ^ v.w;
            }
        }
        __syncthreads();
        acc += mode ? *(unsigned*)(smem + ((threadIdx.x * 64 + it) % (lds_bytes > 4 ? lds_bytes - 4 : 4) & ~3)) : (unsigned)it;
        if (mode < 2) __builtin_amdgcn_s_sleep(8);
    }
    if (mode == 6 || mode == 7) {      // MFMA only (registers; no memory, no LDS): 32x32x16 (6) or 16x16x32 (7) bf16
        typedef float f16v __attribute__((ext_vector_type(16)));
        typedef float f4v __attribute__((ext_vector_type(4)));
        typedef __bf16 b8v __attribute__((ext_vector_type(8)));
        b8v a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(float)(lane + k); b[k] = (__bf16)(float)(wave - k); }
        f16v c0 = {}, c1 = {}; f4v d0 = {}, d1 = {};
        for (long it = 0; it < spin * 64; ++it) {
            if (mode == 6) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0); }
            else { d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, d1, 0, 0, 0); }
        }
        if (c0[0] + c1[3] + d0[1] + d1[2] == 123.456f) sink[1] = 1;
        return;
    }
    if (mode == 3) {      // leave with LDS-DMA fills in flight: does their data land in LDS that has been handed to another workgroup by then?
        for (int rep = 0; rep < 4; ++rep)
            for (int o = wave * 1024; o + 1024 <= lds_bytes; o += 8 * 1024)
                glds16(src + (((o / 16 + lane) * 97 + rep * 4001 + blockIdx.x * 13) & 0xfffff), __builtin_amdgcn_readfirstlane(lds0 + o));
        return;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
extern "C" int lds_neighbour_launch(const void* src, int blocks, int lds_bytes, int mode, long spin, void* sink, void* stream) {
    static bool done = false;
    if (!done) { if (hipFuncSetAttribute((const void*)lds_neighbour, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1; done = true; }
    lds_neighbour<<<blocks, 512, lds_bytes, (hipStream_t)stream>>>((const uint4*)src, lds_bytes, mode, spin, (unsigned*)sink);
    return (int)hipGetLastError();
}
