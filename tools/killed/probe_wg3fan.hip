// Review item 3 of round 4 as a measurement: the slab reduction of the 3x3 weight gradient fanned into its producer (WG3_FANIN_PROBE in
// wgrad3x3.hip: the last-arriving workgroup of a 64 x 64 x 9 output tile sums the tile's pixel slices) against wgrad3x3 + wgrad_reduce,
// at the four stage geometries of the benchmark (B = 64).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w tools/probe_wg3fan.hip -o tools/probe_wg3fan
#define WG3_FANIN_PROBE 1
#include "../artiboost_amd/csrc/wgrad3x3.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK_(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void fill_bf16(uint16_t* p, long n, unsigned seed) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (uint16_t)(0x3c00u + (h & 0x3ffu)) | (uint16_t)((h >> 16) & 0x8000u);
    }
}
// the reduction launch of the product path (conv_wgrad.hip's wgrad_reduce<16> without its stem / accumulate options)
__global__ __launch_bounds__(256) void reduce16(const float* __restrict__ slabs, int nslices, long slab_elems, float* __restrict__ dst) {
    constexpr int KY = 16, QX = 16;
    __shared__ float4 part[KY][QX + 1];
    const int qx = threadIdx.x % QX, ky = threadIdx.x / QX;
    const long e = ((long)blockIdx.x * QX + qx) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < slab_elems) for (int k = ky; k < nslices; k += KY) { const float4 v = *(const float4*)(slabs + (long)k * slab_elems + e); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    part[ky][qx] = s;
    __syncthreads();
    if (ky == 0 && e < slab_elems) {
        float4 t = part[0][qx];
        for (int k = 1; k < KY; ++k) { const float4 v = part[k][qx]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *(float4*)(dst + e) = t;
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 30;
    const int N = 64;
    const int geo[4][2] = {{64, 64}, {32, 128}, {16, 256}, {8, 512}};
    for (int k = 0; k < 4; ++k) {
        const int H = geo[k][0], C = geo[k][1];
        const long nx = (long)N * H * H * C, nw = (long)C * 9 * C;
        const int ns = wgrad3x3_x3_slices(N, H, H, C, C);
        uint16_t *xh, *xl, *dh, *dl; float *slabs, *dw0, *dw1; int* cnt;
        CK_(hipMalloc(&xh, nx * 2)); CK_(hipMalloc(&xl, nx * 2)); CK_(hipMalloc(&dh, nx * 2)); CK_(hipMalloc(&dl, nx * 2));
        CK_(hipMalloc(&slabs, (long)ns * nw * 4)); CK_(hipMalloc(&dw0, nw * 4)); CK_(hipMalloc(&dw1, nw * 4)); CK_(hipMalloc(&cnt, 4096)); CK_(hipMemset(cnt, 0, 4096));
        fill_bf16<<<1024, 256>>>(xh, nx, 1); fill_bf16<<<1024, 256>>>(xl, nx, 2); fill_bf16<<<1024, 256>>>(dh, nx, 3); fill_bf16<<<1024, 256>>>(dl, nx, 4);
        float us[2];
        for (int mode = 0; mode < 2; ++mode) {
            g_wg3_fan_counter = mode ? cnt : nullptr; g_wg3_fan_out = mode ? dw1 : nullptr;
            auto run = [&]() {
                int rc = wgrad3x3_x3_run(xh, xl, dh, dl, slabs, N, H, H, C, C, 0);
                if (!mode) reduce16<<<(unsigned)((nw / 4 + 15) / 16), 256>>>(slabs, ns, nw, dw0);
                return rc;
            };
            if (run()) { printf("launch failed\n"); return 1; }
            for (int i = 0; i < 3; ++i) run();
            CK_(hipDeviceSynchronize());
            hipEvent_t e0, e1; CK_(hipEventCreate(&e0)); CK_(hipEventCreate(&e1));
            CK_(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) run();
            CK_(hipEventRecord(e1)); CK_(hipEventSynchronize(e1));
            float ms; CK_(hipEventElapsedTime(&ms, e0, e1));
            us[mode] = ms * 1e3f / iters;
        }
        std::vector<float> a(nw), b(nw);
        CK_(hipMemcpy(a.data(), dw0, nw * 4, hipMemcpyDeviceToHost)); CK_(hipMemcpy(b.data(), dw1, nw * 4, hipMemcpyDeviceToHost));
        double mx = 0, amax = 0;
        for (long i = 0; i < nw; ++i) { double d = fabs((double)a[i] - b[i]); if (d > mx) mx = d; if (fabs(a[i]) > amax) amax = fabs(a[i]); }
        printf("%3d^2 x %3d channels, %3d slices x %3d tiles: wgrad3x3 + reduction launch %6.1f us; last arriver of a tile sums its slices %6.1f us   (max |diff| %.3g of %.3g)\n",
               H, C, ns, (C / 64) * (C / 64), us[0], us[1], mx, amax);
        hipFree(xh); hipFree(xl); hipFree(dh); hipFree(dl); hipFree(slabs); hipFree(dw0); hipFree(dw1); hipFree(cnt);
    }
    return 0;
}
