/* TEST INFRASTRUCTURE ONLY -- sequential CPU restatement of the JPEG decode the reference gets from
 * `Image.open(path).convert("RGB")` (anakin/datasets/ho3d.py:228-231, dexycb.py:226-229, fhb.py:257-260): Pillow's JpegDecode over
 * libjpeg-turbo with libjpeg's defaults (dct_method JDCT_ISLOW, do_fancy_upsampling TRUE, no draft mode).
 *
 * The decoder is a third-party dependency that is NOT in /root/reference: requirements.txt:94 pins Pillow==8.0.1 (bundles
 * libjpeg-turbo 2.0.x); this image has Pillow 12.2.0 / libjpeg-turbo 3.1.4.1.  What is restated here is the published algorithm
 * (ITU T.81 Huffman/baseline sequential DCT; libjpeg's jidctint.c "islow" integer IDCT, jdsample.c "fancy" triangle up-sampling
 * h2v1 / h2v2 / h1v2 and replication otherwise, jdcolor.c fixed-point YCbCr -> RGB, jdmainct.c edge replication of the context rows),
 * and it is PINNED against the real library: tests/test_jpeg_oracle.py decodes the same files with Pillow here, bit-exact, and
 * tests/golden/jpeg_cases.npz holds files + Pillow's pixels (oracle/gen_jpeg_golden.py).
 *
 * Supported: baseline / extended-sequential Huffman (SOF0, SOF1), 8-bit, grey or YCbCr with 4:4:4 / 4:2:2 / 4:2:0 sampling (luma 1x1, 2x1 or 2x2
 * over 1x1 chroma: what Pillow can write, so what could be pinned), restart intervals, one interleaved scan.  Not supported (returns < 0):
 * progressive, arithmetic, 12-bit, CMYK / Adobe-RGB, 4:4:0 / 4:1:1, multi-scan files.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const unsigned char ZZ[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

typedef struct { int ok; int mincode[17], maxcode[18], valptr[17]; unsigned char vals[256]; } HT;
typedef struct { int id, h, v, tq, td, ta; int bw, bh; /* plane size in blocks */ int dw, dh; /* real down-sampled size */ unsigned char* plane; int pred; } Comp;

typedef struct { const unsigned char* p; long pos, end; uint32_t acc; int n, hit; } BR;

static int br_bit(BR* b) {
    if (b->n == 0) {
        int c = 0;
        if (!b->hit && b->pos < b->end) {
            c = b->p[b->pos++];
            if (c == 0xFF) {
                int d = b->pos < b->end ? b->p[b->pos] : 0xD9;
                if (d == 0) b->pos++;
                else { b->pos--; b->hit = 1; c = 0; }       /* a marker: stay in front of it; libjpeg feeds zero bits from here */
            }
        }
        b->acc = (uint32_t)c; b->n = 8;
    }
    b->n--;
    return (b->acc >> b->n) & 1;
}
static int br_bits(BR* b, int s) { int v = 0; while (s--) v = (v << 1) | br_bit(b); return v; }

static void ht_build(HT* t, const unsigned char* counts, const unsigned char* vals, int nvals) {
    int code = 0, k = 0;
    memcpy(t->vals, vals, nvals);
    for (int l = 1; l <= 16; l++) {
        t->valptr[l] = k; t->mincode[l] = code;
        code += counts[l - 1]; k += counts[l - 1];
        t->maxcode[l] = counts[l - 1] ? code - 1 : -1;
        code <<= 1;
    }
    t->ok = 1;
}
static int ht_decode(const HT* t, BR* b) {
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | br_bit(b);
        if (t->maxcode[l] >= 0 && code <= t->maxcode[l] && code >= t->mincode[l]) return t->vals[t->valptr[l] + code - t->mincode[l]];
    }
    return 0;     /* invalid code: libjpeg warns and returns 0 */
}
static inline int extend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

/* libjpeg's range_limit table addressed with (x & RANGE_MASK) after the IDCT, as a function (jdmaster.c prepare_range_limit_table) */
static inline int idct_range(int x) {
    x &= 1023;
    if (x < 128) return x + 128;
    if (x < 512) return 255;
    if (x < 896) return 0;
    return x - 896;
}
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))
#define F_0_298 2446
#define F_0_390 3196
#define F_0_541 4433
#define F_0_765 6270
#define F_0_899 7373
#define F_1_175 9633
#define F_1_501 12299
#define F_1_847 15137
#define F_1_961 16069
#define F_2_053 16819
#define F_2_562 20995
#define F_3_072 25172
static inline void idct_1d(const int* in, int stride, int* o, int shift_even) {
    /* the shared butterfly of both passes (jidctint.c); `in` already de-quantised; results UNscaled in o[0..7] */
    int z2 = in[2 * stride], z3 = in[6 * stride];
    int z1 = (z2 + z3) * F_0_541;
    int tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
    z2 = in[0]; z3 = in[4 * stride];
    int tmp0 = (z2 + z3) * (1 << shift_even), tmp1 = (z2 - z3) * (1 << shift_even);
    int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7 * stride]; tmp1 = in[5 * stride]; tmp2 = in[3 * stride]; tmp3 = in[1 * stride];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3, z5 = (z3 + z4) * F_1_175;
    tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
    z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    o[0] = tmp10 + tmp3; o[7] = tmp10 - tmp3; o[1] = tmp11 + tmp2; o[6] = tmp11 - tmp2;
    o[2] = tmp12 + tmp1; o[5] = tmp12 - tmp1; o[3] = tmp13 + tmp0; o[4] = tmp13 - tmp0;
}
static void idct_islow(const short* coef, const unsigned short* q, unsigned char* out, int pitch) {
    int ws[64], in[64], o[8];
    for (int i = 0; i < 64; i++) in[i] = (int)coef[i] * (int)q[i];
    for (int c = 0; c < 8; c++) {               /* pass 1: columns, scaled up by 2^PASS1_BITS */
        idct_1d(in + c, 8, o, 13);
        for (int r = 0; r < 8; r++) ws[r * 8 + c] = DESCALE(o[r], 13 - 2);
    }
    for (int r = 0; r < 8; r++) {               /* pass 2: rows */
        idct_1d(ws + r * 8, 1, o, 13);
        for (int c = 0; c < 8; c++) out[r * pitch + c] = (unsigned char)idct_range(DESCALE(o[c], 13 + 2 + 3));
    }
}

static inline int clamp255(int x) { return x < 0 ? 0 : x > 255 ? 255 : x; }

/* one up-sampled chroma (or luma) value at full-resolution (x, y): jdsample.c's method for this component */
static int upsampled(const Comp* c, int hmax, int vmax, int x, int y) {
    const unsigned char* P = c->plane; const int pitch = c->bw * 8, dw = c->dw, dh = c->dh;
    const int hr = hmax / c->h, vr = vmax / c->v;
    if (hr == 1 && vr == 1) return P[y * pitch + x];
    const int fancy = dw > 2;                      /* do_fancy && downsampled_width > 2 (h2v1, h2v2); h1v2 has no width condition */
    if (hr == 2 && vr == 1 && fancy) {
        const unsigned char* r = P + y * pitch; const int i = x >> 1;
        if (x & 1) return i == dw - 1 ? r[i] : (r[i] * 3 + r[i + 1] + 2) >> 2;
        return i == 0 ? r[i] : (r[i] * 3 + r[i - 1] + 1) >> 2;
    }
    if (hr == 1 && vr == 2) {
        const int ir = y >> 1; int nr = (y & 1) ? ir + 1 : ir - 1;
        nr = nr < 0 ? 0 : nr > dh - 1 ? dh - 1 : nr;
        return (P[ir * pitch + x] * 3 + P[nr * pitch + x] + ((y & 1) ? 2 : 1)) >> 2;
    }
    if (hr == 2 && vr == 2 && fancy) {
        const int ir = y >> 1; int nr = (y & 1) ? ir + 1 : ir - 1;
        nr = nr < 0 ? 0 : nr > dh - 1 ? dh - 1 : nr;
        const unsigned char *r0 = P + ir * pitch, *r1 = P + nr * pitch; const int i = x >> 1;
        const int cur = r0[i] * 3 + r1[i];
        if (x & 1) return i == dw - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + r0[i + 1] * 3 + r1[i + 1] + 7) >> 4;
        return i == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + r0[i - 1] * 3 + r1[i - 1] + 8) >> 4;
    }
    return P[(y / vr) * pitch + x / hr];           /* replication (h2v1_upsample / h2v2_upsample / int_upsample) */
}

/* Decodes one JFIF/JPEG file to interleaved RGB (grey files: R = G = B, as Image.convert("RGB")).
 * out: [cap] bytes; returns 0 and sets *W, *H, or a negative code (-1 malformed, -2 unsupported, -3 out too small). */
int jpeg_oracle_decode(const unsigned char* d, long len, unsigned char* out, long cap, int* W, int* H) {
    unsigned short qt[4][64]; int qt_ok[4] = {0, 0, 0, 0};
    HT dc[4], ac[4]; memset(dc, 0, sizeof dc); memset(ac, 0, sizeof ac);
    Comp comp[3]; int nf = 0, ri = 0, w = 0, h = 0, adobe_tf = -1;
    long p = 2;
    if (len < 4 || d[0] != 0xFF || d[1] != 0xD8) return -1;
    long sos = -1;
    while (p + 4 <= len) {
        if (d[p] != 0xFF) return -1;
        int m = d[p + 1];
        if (m == 0xFF) { p++; continue; }
        long L = (d[p + 2] << 8) | d[p + 3];
        const unsigned char* s = d + p + 4; long n = L - 2;
        if (p + 2 + L > len) return -1;
        if (m == 0xDB) {
            while (n > 0) {
                int pq = s[0] >> 4, tq = s[0] & 15; s++; n--;
                if (tq > 3) return -1;
                for (int i = 0; i < 64; i++) { qt[tq][ZZ[i]] = pq ? (unsigned short)((s[0] << 8) | s[1]) : s[0]; s += pq ? 2 : 1; n -= pq ? 2 : 1; }
                qt_ok[tq] = 1;
            }
        } else if (m == 0xC4) {
            while (n > 0) {
                int tc = s[0] >> 4, th = s[0] & 15, tot = 0;
                if (th > 3 || tc > 1) return -1;
                for (int i = 0; i < 16; i++) tot += s[1 + i];
                if (tot > 256) return -1;
                ht_build(tc ? &ac[th] : &dc[th], s + 1, s + 17, tot);
                s += 17 + tot; n -= 17 + tot;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (s[0] != 8) return -2;
            h = (s[1] << 8) | s[2]; w = (s[3] << 8) | s[4]; nf = s[5];
            if (nf != 1 && nf != 3) return -2;
            for (int i = 0; i < nf; i++) { comp[i].id = s[6 + 3 * i]; comp[i].h = s[7 + 3 * i] >> 4; comp[i].v = s[7 + 3 * i] & 15; comp[i].tq = s[8 + 3 * i]; }
        } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC) || m == 0xC3) {
            return -2;                               /* progressive, lossless, arithmetic, hierarchical */
        } else if (m == 0xDD) {
            ri = (s[0] << 8) | s[1];
        } else if (m == 0xEE && n >= 12 && !memcmp(s, "Adobe", 5)) {
            adobe_tf = s[11];
        } else if (m == 0xDA) {
            if (!nf || s[0] != nf) return -2;        /* one interleaved scan with every component */
            for (int i = 0; i < nf; i++) {
                int k = -1;
                for (int j = 0; j < nf; j++) if (comp[j].id == s[1 + 2 * i]) k = j;
                if (k != i) return -2;
                comp[i].td = s[2 + 2 * i] >> 4; comp[i].ta = s[2 + 2 * i] & 15;
            }
            if (s[1 + 2 * nf] != 0 || s[2 + 2 * nf] != 63) return -2;
            sos = p + 2 + L;
            break;
        }
        p += 2 + L;
    }
    if (sos < 0 || !w || !h) return -1;
    if (nf == 3 && (adobe_tf == 0 || (comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B'))) return -2;
    int hmax = 1, vmax = 1;
    for (int i = 0; i < nf; i++) {
        if (comp[i].h < 1 || comp[i].h > 2 || comp[i].v < 1 || comp[i].v > 2 || !qt_ok[comp[i].tq & 3] || !dc[comp[i].td & 3].ok || !ac[comp[i].ta & 3].ok) return -2;
        if (comp[i].h > hmax) hmax = comp[i].h;
        if (comp[i].v > vmax) vmax = comp[i].v;
    }
    for (int i = 0; i < nf; i++)                                          /* 4:4:0 (h1v2) cannot be written by Pillow, so it could not be pinned */
        if (nf == 3 && (comp[i].h == hmax) && comp[i].v * 2 == vmax && hmax == 1) return -2;
    if (nf == 3 && !(comp[0].h == hmax && comp[0].v == vmax && comp[1].h == 1 && comp[1].v == 1 && comp[2].h == 1 && comp[2].v == 1)) return -2;
    if (nf == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }       /* a single-component scan is never interleaved: 1 block per MCU */
    const int mcux = (w + 8 * hmax - 1) / (8 * hmax), mcuy = (h + 8 * vmax - 1) / (8 * vmax);
    for (int i = 0; i < nf; i++) {
        Comp* c = &comp[i];
        c->bw = mcux * c->h; c->bh = mcuy * c->v;
        c->dw = (w * c->h + hmax - 1) / hmax; c->dh = (h * c->v + vmax - 1) / vmax;
        c->plane = (unsigned char*)malloc((size_t)c->bw * c->bh * 64);
        c->pred = 0;
    }
    if ((long)w * h * 3 > cap) { for (int i = 0; i < nf; i++) free(comp[i].plane); return -3; }
    /* ---- entropy decode + IDCT, MCU by MCU */
    BR br = {d, sos, len, 0, 0, 0};
    short blk[64];
    long nmcu = (long)mcux * mcuy;
    for (long m = 0; m < nmcu; m++) {
        if (ri && m && m % ri == 0) {              /* restart: byte-align, skip the RSTn marker, reset the predictors */
            br.n = 0; br.hit = 0;
            long q = br.pos;
            while (q + 1 < len && !(d[q] == 0xFF && d[q + 1] >= 0xD0 && d[q + 1] <= 0xD7)) q++;
            br.pos = q + 2 < len ? q + 2 : len;
            for (int i = 0; i < nf; i++) comp[i].pred = 0;
        }
        const int mx = (int)(m % mcux), my = (int)(m / mcux);
        for (int i = 0; i < nf; i++) {
            Comp* c = &comp[i];
            for (int yy = 0; yy < c->v; yy++) for (int xx = 0; xx < c->h; xx++) {
                memset(blk, 0, sizeof blk);
                int s = ht_decode(&dc[c->td], &br);
                c->pred += extend(br_bits(&br, s), s);
                blk[0] = (short)c->pred;
                for (int k = 1; k < 64;) {
                    int rs = ht_decode(&ac[c->ta], &br), r = rs >> 4; s = rs & 15;
                    if (s == 0) { if (r == 15) { k += 16; continue; } break; }
                    k += r;
                    int v = extend(br_bits(&br, s), s);
                    if (k < 64) blk[ZZ[k]] = (short)v;
                    k++;
                }
                const int bx = mx * c->h + xx, by = my * c->v + yy;
                idct_islow(blk, qt[c->tq], c->plane + ((size_t)by * 8 * c->bw + bx) * 8, c->bw * 8);
            }
        }
    }
    /* ---- up-sample + colour */
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
        unsigned char* o = out + ((size_t)y * w + x) * 3;
        const int Y = upsampled(&comp[0], hmax, vmax, x, y);
        if (nf == 1) { o[0] = o[1] = o[2] = (unsigned char)Y; continue; }
        const int cb = upsampled(&comp[1], hmax, vmax, x, y) - 128, cr = upsampled(&comp[2], hmax, vmax, x, y) - 128;
        o[0] = (unsigned char)clamp255(Y + ((91881 * cr + 32768) >> 16));                         /* FIX(1.40200) */
        o[1] = (unsigned char)clamp255(Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));           /* FIX(0.34414), FIX(0.71414) */
        o[2] = (unsigned char)clamp255(Y + ((116130 * cb + 32768) >> 16));                        /* FIX(1.77200) */
    }
    for (int i = 0; i < nf; i++) free(comp[i].plane);
    *W = w; *H = h;
    return 0;
}
