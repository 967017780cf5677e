// Micro-benchmark: how fast can a CU pull L2-resident data into LDS?  (a) LDS-DMA (global_load_lds_dwordx4),
// (b) global_load_dwordx4 -> VGPR -> ds_write_b128, (c) global_load_dwordx4 only.  Footprint and waves/CU are swept.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_fill.hip -o tools/probe_fill && tools/probe_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

// each wave moves DEPTH x 1 KiB per iteration, DEPTH loads in flight
template <int MODE, int DEPTH = 8>
__global__ __launch_bounds__(256) void fill_kernel(const uint4* __restrict__ src, size_t footprint_vec, int iters, uint4* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t pos = ((size_t)blockIdx.x * 4 + wave) * 64 * DEPTH;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                size_t v = (pos + j * 64 + lane) % footprint_vec;
                glds16(src + v, __builtin_amdgcn_readfirstlane(lds0 + wave * DEPTH * 1024 + j * 1024));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            uint4 r[DEPTH];
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) { size_t v = (pos + j * 64 + lane) % footprint_vec; r[j] = src[v]; }
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < DEPTH; ++j) *(uint4*)(smem + wave * DEPTH * 1024 + j * 1024 + lane * 16) = r[j];
            } else {
#pragma unroll
                for (int j = 0; j < DEPTH; ++j) { acc.x ^= r[j].x; acc.y ^= r[j].y; acc.z ^= r[j].z; acc.w ^= r[j].w; }
            }
        }
        pos += (size_t)gridDim.x * 4 * 64 * DEPTH;
    }
    if (MODE == 2 && acc.x == 0x12345678u) sink[0] = acc;
    if (MODE != 2 && iters < 0) sink[0] = *(uint4*)(smem + lane * 16);
}

int main() {
    const size_t maxbytes = 512ull << 20;
    uint4 *src, *sink;
    CK(hipMalloc(&src, maxbytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 1, maxbytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[3] = {"lds-dma (global_load_lds_dwordx4)", "global_load_dwordx4 + ds_write_b128", "global_load_dwordx4 only"};
    for (size_t fp_mb : {4, 16, 128, 512}) {
        for (int wg_per_cu : {1, 2, 4}) {
            for (int mode = 0; mode < 3; ++mode) {
                const int blocks = 256 * wg_per_cu, iters = 400;
                const size_t fpv = (fp_mb << 20) / 16;
                auto launch = [&]() {
                    if (mode == 0) fill_kernel<0><<<blocks, 256, 32768>>>(src, fpv, iters, sink);
                    else if (mode == 1) fill_kernel<1><<<blocks, 256, 32768>>>(src, fpv, iters, sink);
                    else fill_kernel<2><<<blocks, 256, 32768>>>(src, fpv, iters, sink);
                };
                launch(); CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                double bytes = (double)blocks * 4 * iters * 8 * 1024;
                printf("footprint %4zu MB  %d wg/CU (%2d waves)  %-40s %7.2f TB/s  (%.1f GB/s per CU)\n", fp_mb, wg_per_cu, 4 * wg_per_cu,
                       names[mode], bytes / ms / 1e9, bytes / ms / 1e6 / 256);
            }
        }
    }
    printf("\n-- depth sweep: 1 workgroup (4 waves) per CU, 16 MB footprint, LDS-DMA\n");
    for (int depth : {4, 8, 16, 32}) {
        const int blocks = 256, iters = 400; const size_t fpv = (16ull << 20) / 16;
        auto launch = [&]() {
            if (depth == 4) fill_kernel<0, 4><<<blocks, 256, 4 * 4096>>>(src, fpv, iters, sink);
            else if (depth == 8) fill_kernel<0, 8><<<blocks, 256, 4 * 8192>>>(src, fpv, iters, sink);
            else if (depth == 16) fill_kernel<0, 16><<<blocks, 256, 4 * 16384>>>(src, fpv, iters, sink);
            else { CK(hipFuncSetAttribute((const void*)fill_kernel<0, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
                   fill_kernel<0, 32><<<blocks, 256, 4 * 32768>>>(src, fpv, iters, sink); }
        };
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double bytes = (double)blocks * 4 * iters * depth * 1024;
        printf("depth %2d KiB in flight per wave: %7.2f TB/s  (%.1f GB/s per CU)\n", depth, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
    }
    return 0;
}
