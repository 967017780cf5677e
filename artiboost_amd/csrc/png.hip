// PNG frames on the device: scanline reconstruction ("unfiltering") + RGB(X) packing of inflated IDAT streams, bit-identical to what
// `Image.open(path).convert("RGB")` returns in the reference's DataLoader workers for HO3D v2's rgb/NNNN.png frames
// (anakin/datasets/ho3d.py:181,228-231; Pillow: zlib inflate + libImaging/ZipDecode.c = the PNG specification's five filters).
// The inflate itself stays on the host (a thread pool of zlib calls, artiboost_amd/png.py): LZ77 + Huffman of one stream is a serial
// chain and the batch holds only 40 - 160 streams, while the reconstruction and the RGBX scatter are the part that parallelises.
//
// Dependencies of a byte: left (same line, one pixel back), up, up-left -- and only bytes of the same position inside the pixel, so the
// three colour samples are three independent chains and alpha / low-order bytes of 16-bit samples are never touched.  One wave per image
// walks a band of 64 lines as a skewed wavefront: lane l owns line 64 band + l and, at step s, rebuilds pixel x = s - l from its own
// previous pixel (left), the pixel lane l - 1 produced one step earlier (up: a lane shift, no memory) and the one it received the step
// before (up-left).  Lane 0's line above is the previous band's last line, kept in LDS.  HBM traffic: each filtered byte read once,
// each output pixel written once.
#include "common.h"

#define AB_PNG_DESC_INTS_ 8
// desc int32 [n][8]: 0 / 1 byte offset of the image's scanlines in raw (low / high word), 2 width, 3 height, 4 bytes per pixel,
//                    5 sample byte offsets c0 | c1 << 8 | c2 << 16, 6 output offset (pixels), 7 output row pitch (pixels)

// one sample, branch-free: f filtered byte, a left, b up, c up-left (all 0..255) -> rebuilt byte.  is1..is4: this lane's LINE carries filter
// type Sub / Up / Average / Paeth (constant over the band, so the selects below never diverge into branches).
__device__ __forceinline__ unsigned png_rebuild(unsigned f, unsigned a, unsigned b, unsigned c, bool is1, bool is2, bool is3, bool is4) {
    const unsigned pa = __builtin_amdgcn_sad_u16(b, c, 0u), pb = __builtin_amdgcn_sad_u16(a, c, 0u);      // |b - c|, |a - c|
    const int t = (int)(a + b) - (int)(c + c);
    const unsigned pc = (unsigned)max(t, -t);                                                               // |a + b - 2c|
    unsigned pred = (pb <= pc) ? b : c;
    pred = (pa <= pb && pa <= pc) ? a : pred;            // Paeth: ties in the order a, b, c
    pred = is4 ? pred : 0u;
    pred = is3 ? (a + b) >> 1 : pred;
    pred = is2 ? b : pred;
    pred = is1 ? a : pred;
    return (f + pred) & 255u;
}

template <int OC, bool WIDE>      // WIDE: some image of the batch has more than 4 bytes per pixel (16-bit samples): 8-byte sample fetches
__global__ __launch_bounds__(64) void png_unfilter_kernel(const uint8_t* __restrict__ raw, const int32_t* __restrict__ desc, uint8_t* __restrict__ out,
                                                          int* __restrict__ status) {
    extern __shared__ unsigned lastline[];                    // [width]: packed R | G << 8 | B << 16 of the band's last line
    const int32_t* d = desc + (long)blockIdx.x * AB_PNG_DESC_INTS_;
    const uint8_t* img = raw + (((long)(uint32_t)d[1] << 32) | (uint32_t)d[0]);
    const int W = d[2], H = d[3], bpp = d[4];
    const int c0 = d[5] & 255, c1 = (d[5] >> 8) & 255, c2 = (d[5] >> 16) & 255;
    const long opix = d[6];
    const int opitch = d[7];
    const int lane = threadIdx.x;
    const long stride = 1 + (long)W * bpp;
    bool bad = false;
    for (int band = 0; band * 64 < H; ++band) {
        const int r = band * 64 + lane;
        const bool rowok = r < H;
        const uint8_t* line = img + (long)(rowok ? r : 0) * stride;
        const int ft = rowok ? line[0] : 0;
        bad |= ft > 4;
        const bool is1 = ft == 1, is2 = ft == 2, is3 = ft == 3, is4 = ft == 4;
        unsigned l0 = 0, l1 = 0, l2 = 0, ul0 = 0, ul1 = 0, ul2 = 0, cu0 = 0, cu1 = 0, cu2 = 0;      // left, up-left, this lane's last pixel: per sample
        // The filtered samples of a pixel are fetched PD steps before the pixel is rebuilt (one unaligned 4- or 8-byte load per pixel into a
        // register ring): a lane walks its own line, so on every step some lane of the wave crosses into a new cache line -- with the fetch
        // one step ahead the whole wave paid an L2 round trip per step.  The step itself is branch-free apart from the guarded stores
        // (210 -> ~90 instructions per step; the kernel is issue-bound: one wave per image, 64 lines in flight).
        constexpr int PD = 16;
        uint2 ring[PD];
        const uint8_t* line1 = line + 1;
        auto fetch = [&](int x) -> uint2 {
            uint2 v = make_uint2(0u, 0u);
            const int xc = min(max(x, 0), W - 1);                     // clamped: always a valid address of this line, the value is unused when x is outside
            const uint8_t* q = line1 + (long)xc * bpp;
            if (WIDE) __builtin_memcpy(&v, q, 8); else __builtin_memcpy(&v.x, q, 4);
            return v;
        };
#pragma unroll
        for (int k = 0; k < PD; ++k) ring[k] = fetch(k - lane);
        const bool has_above = band > 0;
        unsigned up0 = has_above ? lastline[0] : 0u;
        const int nsteps = (W + 63 + PD - 1) / PD * PD;
        const unsigned sh0 = 8u * c0, sh1 = 8u * c1, sh2 = 8u * c2;
        const bool first = lane == 0, last = lane == 63;
        for (int sb = 0; sb < nsteps; sb += PD) {
#pragma unroll
            for (int k = 0; k < PD; ++k) {
                const int s = sb + k, x = s - lane;
                // the pixel lane l - 1 rebuilt one step ago: whole-wave shifts by one lane in the data path (DPP wave_shr:1; lane 0 receives 0)
                unsigned u0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)cu0, 0x138, 0xf, 0xf, false);
                unsigned u1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)cu1, 0x138, 0xf, 0xf, false);
                unsigned u2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)cu2, 0x138, 0xf, 0xf, false);
                u0 = first ? (up0 & 255u) : u0; u1 = first ? ((up0 >> 8) & 255u) : u1; u2 = first ? ((up0 >> 16) & 255u) : u2;
                const bool act = rowok && x >= 0 && x < W;
                const unsigned long long fw = ((unsigned long long)ring[k].y << 32) | ring[k].x;
                ring[k] = fetch(x + PD);
                up0 = has_above ? lastline[min(s + 1, W - 1)] : 0u;          // (every lane reads the same word: a broadcast)
                const unsigned g0 = (unsigned)(fw >> sh0) & 255u, g1 = (unsigned)(fw >> sh1) & 255u, g2 = (unsigned)(fw >> sh2) & 255u;
                const unsigned r0 = png_rebuild(g0, l0, u0, ul0, is1, is2, is3, is4);
                const unsigned r1 = png_rebuild(g1, l1, u1, ul1, is1, is2, is3, is4);
                const unsigned r2 = png_rebuild(g2, l2, u2, ul2, is1, is2, is3, is4);
                cu0 = act ? r0 : cu0; cu1 = act ? r1 : cu1; cu2 = act ? r2 : cu2;
                l0 = cu0; l1 = cu1; l2 = cu2;
                ul0 = act ? u0 : ul0; ul1 = act ? u1 : ul1; ul2 = act ? u2 : ul2;
                if (act) {
                    const unsigned px = r0 | (r1 << 8) | (r2 << 16);
                    uint8_t* o = out + (opix + (long)r * opitch + x) * OC;
                    if (OC == 4) *(unsigned*)o = px;
                    else { o[0] = (uint8_t)r0; o[1] = (uint8_t)r1; o[2] = (uint8_t)r2; }
                    if (last) lastline[x] = px;
                }
            }
        }
        __syncthreads();                                                 // lane 63's line is complete before lane 0 of the next band reads it
    }
    if (__any(bad) && lane == 0) atomicOr(status, 1);
}

// raw: the inflated scanlines of n images (device; readable for 8 bytes past the last line: the sample fetch is 4 / 8 bytes wide), desc as above; out uint8 [.][out_channels] (3: RGB, 4: RGBX with X = 0);
// max_bpp: the largest desc field 4 of the batch;  status (device int32, optional): bit 0 is set when a line carries a filter type above 4 (a corrupt stream; its output is undefined)
extern "C" int ab_png_unfilter_batch(const void* raw, const int32_t* desc, int n, int max_width, int max_bpp, int out_channels, void* out,
                                     int* status, void* stream) {
    if (!raw || !desc || !out || n < 0 || max_width <= 0 || max_bpp < 1 || max_bpp > 8) return AB_EINVAL;
    if (out_channels != 3 && out_channels != 4) return AB_ESHAPE;
    if ((long)max_width * 4 > 64 * 1024) return AB_ESHAPE;
    if (n == 0) return 0;
    const size_t lds = (size_t)max_width * 4;
    const uint8_t* r = (const uint8_t*)raw;
    uint8_t* o = (uint8_t*)out;
    hipStream_t st = as_stream(stream);
    if (out_channels == 4) {
        if (max_bpp > 4) png_unfilter_kernel<4, true><<<n, 64, lds, st>>>(r, desc, o, status);
        else png_unfilter_kernel<4, false><<<n, 64, lds, st>>>(r, desc, o, status);
    } else {
        if (max_bpp > 4) png_unfilter_kernel<3, true><<<n, 64, lds, st>>>(r, desc, o, status);
        else png_unfilter_kernel<3, false><<<n, 64, lds, st>>>(r, desc, o, status);
    }
    AB_LAUNCH_CHECK(); return 0;
}
