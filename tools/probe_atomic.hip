// Throughput of device-scope 64-bit integer atomic adds without return (global_atomic_add_x2), the primitive of an
// order-independent (hence bit-reproducible) fixed-point BatchNorm reduction: every workgroup adds one value to each of
// NADDR shared addresses, as a conv epilogue / BN-backward reduction would (NADDR = 2 sums x C channels).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_atomic.hip -o tools/probe_atomic && tools/probe_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void adds(unsigned long long* acc, int naddr, int reps) {
    for (int r = 0; r < reps; ++r)
        for (int a = threadIdx.x; a < naddr; a += 256)
            atomicAdd(acc + a, (unsigned long long)(blockIdx.x + a + r + 1));
}
__global__ void empty() {}

int main() {
    unsigned long long* acc; CK(hipMalloc(&acc, 1 << 20)); CK(hipMemset(acc, 0, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int naddr : {128, 512, 1024})
        for (int blocks : {256, 1024, 2048}) {
            adds<<<blocks, 256>>>(acc, naddr, 1); CK(hipDeviceSynchronize());
            const int reps = 8;
            CK(hipEventRecord(e0)); adds<<<blocks, 256>>>(acc, naddr, reps); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double n = (double)blocks * naddr * reps;
            printf("addresses %5d  workgroups %5d : %8.1f G atomics/s   (%.1f us for one round of %d)\n", naddr, blocks, n / ms / 1e6,
                   ms * 1e3 / reps, blocks * naddr);
        }
    // determinism: two runs give identical sums
    CK(hipMemset(acc, 0, 8192)); adds<<<2048, 256>>>(acc, 1024, 3); CK(hipDeviceSynchronize());
    unsigned long long h0[1024], h1[1024]; CK(hipMemcpy(h0, acc, 8192, hipMemcpyDeviceToHost));
    CK(hipMemset(acc, 0, 8192)); adds<<<2048, 256>>>(acc, 1024, 3); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h1, acc, 8192, hipMemcpyDeviceToHost));
    int same = 1; for (int i = 0; i < 1024; ++i) same &= (h0[i] == h1[i]);
    unsigned long long want = 0; for (int b = 0; b < 2048; ++b) for (int r = 0; r < 3; ++r) want += (unsigned long long)(b + 5 + r + 1);
    printf("repeatable: %d   exact: %d\n", same, (int)(h0[5] == want));
    CK(hipEventRecord(e0)); for (int i = 0; i < 100; ++i) empty<<<1, 64>>>(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("empty kernel back to back: %.2f us\n", ms * 10);
    return 0;
}
