// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (which LDS element lands in which lane/slot).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short short4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void probe(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    int l = threadIdx.x;
    for (int i = l; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    int addr_elems;
    if (mode == 0) addr_elems = l * 4;
    else if (mode == 1) addr_elems = (l & 15) * 64 + (l >> 4) * 4;
    else addr_elems = (l >> 4) * 64 + (l & 15) * 4;
    short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(lds + addr_elems));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
    uint16_t* d; CK(hipMalloc(&d, 64 * 4 * 2));
    uint16_t h[256];
    for (int mode = 0; mode < 3; ++mode) {
        probe<<<1, 64>>>(d, mode);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
