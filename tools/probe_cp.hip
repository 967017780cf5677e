// Knock-out timings of convp_kernel (artiboost_amd/csrc/convp.hip) at the three stage-entry geometries of the benchmark, no torch:
//   for a in 0 1 2 4 6 8; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DCP_ABL=$a tools/probe_cp.hip -o tools/probe_cp_$a; done
// CP_ABL bits: 1 no MFMAs, 2 no patch requests inside the loop, 4 no weight requests inside the loop, 8 no fragment reads (results are garbage).
#include "../artiboost_amd/csrc/convp.hip"
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void fill_bf16(uint16_t* p, long n, unsigned seed) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (uint16_t)(0x3c00u + (h & 0x3ffu)) | (uint16_t)((h >> 16) & 0x8000u);
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 30;
    const int N = 64;
    const int geo[3][3] = {{64, 64, 128}, {32, 128, 256}, {16, 256, 512}};      // H, Cin, Cout
    for (int k = 0; k < 3; ++k) {
        const int H = geo[k][0], Ci = geo[k][1], Co = geo[k][2];
        const long nx = (long)N * H * H * Ci, ny = (long)N * (H / 2) * (H / 2) * Co, nw = (long)Co * 9 * Ci, nw2 = (long)Co * Ci;
        uint16_t *xh, *xl, *yh, *yl, *y2h, *y2l, *wh, *wl, *w2h, *w2l; float *out, *dx, *stats;
        CK(hipMalloc(&xh, nx * 2)); CK(hipMalloc(&xl, nx * 2)); CK(hipMalloc(&yh, ny * 2)); CK(hipMalloc(&yl, ny * 2)); CK(hipMalloc(&y2h, ny * 2)); CK(hipMalloc(&y2l, ny * 2));
        CK(hipMalloc(&wh, nw * 4)); wl = wh + nw; CK(hipMalloc(&w2h, nw2 * 4)); w2l = w2h + nw2;      // (hi, lo) planes of one allocation, as the library's callers hold them
        CK(hipMalloc(&out, ny * 4)); CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&stats, (long)N * 16 * Co * 8));
        fill_bf16<<<1024, 256>>>(xh, nx, 1); fill_bf16<<<1024, 256>>>(xl, nx, 2); fill_bf16<<<1024, 256>>>(yh, ny, 3); fill_bf16<<<1024, 256>>>(yl, ny, 4);
        fill_bf16<<<1024, 256>>>(y2h, ny, 5); fill_bf16<<<1024, 256>>>(y2l, ny, 6); fill_bf16<<<256, 256>>>(wh, nw, 7); fill_bf16<<<256, 256>>>(wl, nw, 8);
        fill_bf16<<<256, 256>>>(w2h, nw2, 9); fill_bf16<<<256, 256>>>(w2l, nw2, 10);
        float us[2];
        for (int mode = 0; mode < 2; ++mode) {
            auto run = [&]() {
                return mode == 0 ? convp_s2fwd_run(xh, xl, wh, wl, out, N, H, H, Ci, Co, stats, 0, nullptr, nullptr, 0, nullptr, nullptr)
                                 : convp_s2dgrad_run(yh, yl, wh, wl, y2h, y2l, w2h, w2l, dx, N, H, H, Ci, Co, 0, nullptr, nullptr, nullptr, nullptr);
            };
            int rc = run();
            if (rc == AB_ESHAPE) { us[mode] = 0.f; continue; }        // shape left to the tap-by-tap kernel
            if (rc) { printf("launch failed: mode %d rc %d (%s)\n", mode, rc, rc > 0 ? hipGetErrorString((hipError_t)rc) : "ab code"); return 1; }
            for (int i = 0; i < 2; ++i) run();
            CK(hipDeviceSynchronize());
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) run();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            us[mode] = ms * 1e3f / iters;
        }
        printf("CP_ABL=%d  %3d^2 %3d->%3d: forward %6.1f us, data gradient (+ downsample) %6.1f us\n", CP_ABL, H, Ci, Co, us[0], us[1]);
        hipFree(xh); hipFree(xl); hipFree(yh); hipFree(yl); hipFree(y2h); hipFree(y2l); hipFree(wh); hipFree(w2h); hipFree(out); hipFree(dx); hipFree(stats);
    }
    return 0;
}
