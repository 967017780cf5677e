// Ablation timings of gemm_rw_kernel (artiboost_amd/csrc/gemm_rw.hip) at the benchmark head geometry, no torch:
//   for a in 0 1 2 4 8; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGRW_ABL=$a tools/probe_grw.hip -o tools/probe_grw_$a; done
// GRW_ABL bits: 1 no output stores, 2 no MFMAs (fragments still read), 4 no fragment reads, 8 no fills.  Results are garbage with any bit set.
#include "../artiboost_amd/csrc/gemm_rw.hip"
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void fill_bf16(uint16_t* p, long n, unsigned seed) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (uint16_t)(0x3c00u + (h & 0x3ffu)) | (uint16_t)((h >> 16) & 0x8000u);      // +-[0.0078, 0.0156): random mantissas and signs
    }
}

int main(int argc, char** argv) {
    const int sam = argc > 1 ? atoi(argv[1]) : 0, iters = argc > 2 ? atoi(argv[2]) : 50;
    const long M = 65536; const int N = 704, K = 256;
    uint16_t *ah, *al, *wh, *wl; float *bias, *out, *part;
    CK(hipMalloc(&ah, M * K * 2)); CK(hipMalloc(&al, M * K * 2)); CK(hipMalloc(&wh, (long)N * K * 2)); CK(hipMalloc(&wl, (long)N * K * 2));
    CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&out, M * N * 4)); CK(hipMalloc(&part, 64 * 16 * 22 * 8 * 4));
    fill_bf16<<<4096, 256>>>(ah, M * K, 1); fill_bf16<<<4096, 256>>>(al, M * K, 2);
    fill_bf16<<<256, 256>>>(wh, (long)N * K, 3); fill_bf16<<<256, 256>>>(wl, (long)N * K, 4);
    CK(hipMemset(bias, 0, N * 4));
    GemmRwSam s = {part, 22, 28, 32, 32};
    for (int i = 0; i < 5; ++i) if (gemm_rw_run(ah, al, wh, wl, bias, out, M, N, K, sam ? &s : nullptr, 0)) { printf("launch failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) gemm_rw_run(ah, al, wh, wl, bias, out, M, N, K, sam ? &s : nullptr, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, fl = 2.0 * M * N * K;
    {   // one more launch with per-wave stamps: shader cycles and wall time of every wave's life
        unsigned long long* d; CK(hipMalloc(&d, 256 * 8 * 2 * 8)); CK(hipMemset(d, 0, 256 * 8 * 2 * 8));
        g_grw_dbg = d;
        gemm_rw_run(ah, al, wh, wl, bias, out, M, N, K, sam ? &s : nullptr, 0);
        CK(hipDeviceSynchronize());
        g_grw_dbg = nullptr;
        static unsigned long long h[256 * 8 * 2];
        CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        double cyc = 0, rt = 0, cmax = 0, rmax = 0; int n = 0;
        for (int i = 0; i < 256 * 8; ++i) if (h[2 * i]) { cyc += h[2 * i]; rt += h[2 * i + 1]; ++n; if (h[2 * i] > cmax) cmax = h[2 * i]; if (h[2 * i + 1] > rmax) rmax = h[2 * i + 1]; }
        {   // by role: block b -> XCD b & 7, j = b >> 3; j < 30: tuple member of channel group j % 3, else spare
            double rc[4] = {0, 0, 0, 0}, rm[4] = {0, 0, 0, 0}; int rn[4] = {0, 0, 0, 0};
            for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) {
                const int j = b >> 3, role = j < 30 ? j % 3 : 3; const double c = (double)h[(b * 8 + w) * 2];
                if (c > 0) { rc[role] += c; ++rn[role]; if (c > rm[role]) rm[role] = c; }
            }
            for (int r = 0; r < 4; ++r) if (rn[r]) printf("  role %d (%s): %d waves, mean %.0f max %.0f cycles\n", r, r < 3 ? "tuple member of group" : "spare", rn[r], rc[r] / rn[r], rm[r]);
        }
        if (n) printf("  stamps: %d waves, mean %.0f shader cycles in %.2f us (max %.0f / %.2f us) -> %.3f GHz; MFMA pipe floor 12 units x 192 MFMAs x 32 cyc = 73728 cyc -> busy %.2f\n",
                      n, cyc / n, rt / n / 100.0, cmax, rmax / 100.0, (cyc / n) / (rt / n / 100.0) / 1e3, 73728.0 / (cyc / n));
    }
    printf("GRW_ABL=%d sam=%d: %.1f us per launch, %.1f TFLOP/s (%.3f of 833)\n", GRW_ABL, sam, us, fl / us / 1e6, fl / us / 1e6 / 833.0);
    return 0;
}
