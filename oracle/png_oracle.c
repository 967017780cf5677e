/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the PNG scanline reconstruction that stands behind the reference's
 * `Image.open(path).convert("RGB")` on a .png frame (anakin/datasets/ho3d.py:181,228-231: HO3D v2 stores rgb/NNNN.png).
 *
 * The decoder itself is a third-party dependency of the reference, absent from /root/reference: Pillow==8.0.1 (requirements.txt:94),
 * whose PngImagePlugin hands the IDAT stream to zlib and the inflated scanlines to libImaging/ZipDecode.c.  Its algorithm is the PNG
 * specification's (ISO/IEC 15948 section 9, "Filtering"): every scanline carries a filter-type byte; a byte x of the line is rebuilt from
 * the filtered byte and the already rebuilt bytes a (bpp to the left), b (above), c (above-left), all arithmetic modulo 256:
 *     0 None  x = f        1 Sub  x = f + a        2 Up  x = f + b        3 Average  x = f + floor((a + b) / 2)
 *     4 Paeth x = f + (the one of a, b, c closest to p = a + b - c; ties in the order a, b, c)
 * with a = c = 0 left of the line and b = c = 0 above the image.  `convert("RGB")` then keeps, per pixel: R,G,B of an 8-bit RGB / RGBA
 * file (alpha dropped, no compositing), the HIGH byte of each 16-bit sample, the grey value three times for 8-bit greyscale.
 * Pinned against the real Pillow on committed files (tests/golden/png_cases.npz) and on a live sweep (tests/test_png_host.py).
 *
 * png_oracle_unfilter: inflated scanlines `raw` (h rows of 1 + w * bpp bytes) -> rgb uint8 [h][w][3]; c0,c1,c2 = byte offsets of the
 * R, G, B samples inside a pixel (0,1,2 | 0,2,4 | 0,0,0).  Returns 0, or -1 for an invalid filter byte / argument.                     */
#include <stdlib.h>
#include <string.h>

static int paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}

int png_oracle_unfilter(const unsigned char* raw, int w, int h, int bpp, int c0, int c1, int c2, unsigned char* rgb) {
    if (w <= 0 || h <= 0 || bpp <= 0 || bpp > 8 || c0 >= bpp || c1 >= bpp || c2 >= bpp) return -1;
    const long stride = (long)w * bpp;
    unsigned char* prev = (unsigned char*)calloc(stride, 1);
    unsigned char* cur = (unsigned char*)malloc(stride);
    if (!prev || !cur) { free(prev); free(cur); return -1; }
    for (int y = 0; y < h; ++y) {
        const unsigned char* f = raw + (long)y * (stride + 1);
        const int ft = f[0];
        if (ft > 4) { free(prev); free(cur); return -1; }
        for (long i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) pred = paeth(a, b, c);
            cur[i] = (unsigned char)(f[1 + i] + pred);
        }
        for (int x = 0; x < w; ++x) {
            unsigned char* o = rgb + ((long)y * w + x) * 3;
            o[0] = cur[(long)x * bpp + c0]; o[1] = cur[(long)x * bpp + c1]; o[2] = cur[(long)x * bpp + c2];
        }
        unsigned char* t = prev; prev = cur; cur = t;
    }
    free(prev); free(cur);
    return 0;
}
