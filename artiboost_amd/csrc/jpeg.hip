// Device-side baseline JPEG decode of a batch of files (the decode in front of the real-data augmentation chain;
// reference: Image.open(path).convert("RGB") in a DataLoader worker, anakin/datasets/ho3d.py:228-231, dexycb.py:226-229 -- Pillow over
// libjpeg-turbo, JDCT_ISLOW + fancy up-sampling).  Results are bit-identical to that decoder (tests/test_gpu_jpeg.py).
//
// A Huffman-coded scan has no markers the decoder could start from, so the entropy decode is the self-synchronising parallel scheme
// (Klein & Wiseman 2003; Weissenberger & Schmidt 2018/2021): the scan of every image (every restart interval, when the file has them)
// is cut into `sub_bytes`-byte subsequences; one thread per subsequence starts decoding at its first bit as if a block began there,
// keeps decoding into the following subsequences and stops as soon as its state (bit position, block-in-MCU, zig-zag index) at a
// subsequence boundary equals what the thread that started there recorded -- Huffman streams re-synchronise after a few dozen symbols.
// The recorded states then are the true ones (the first thread of a segment started right), a prefix sum of the blocks completed per
// subsequence gives every thread its output position, and a second pass over its own subsequence writes the coefficients.
//   jpeg_entropy_kernel  one workgroup per image: decode tables -> LDS, synchronisation rounds, block-count scan, coefficient pass, DC prediction scan
//   jpeg_idct_kernel     de-quantise + jidctint.c "islow" 8x8 integer IDCT, 8 lanes per block, planes of uint8 samples
//   jpeg_color_kernel    jdsample.c fancy h2v1 / h2v2 up-sampling (replication for widths <= 2) + jdcolor.c YCbCr -> RGB, RGB(X) rows out
// Byte stuffing (FF 00) is handled by the bit reader; the JFIF header (a few hundred bytes) is parsed on the host (artiboost_amd/jpeg.py).
#include "common.h"

namespace {

__device__ const unsigned char ZZ[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                         35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// descriptor fields (int32 per image, AB_JPEG_DESC_INTS of them; filled by artiboost_amd/jpeg.py)
enum { D_OFF = 0, D_LEN, D_W, D_H, D_NCOMP, D_HMAX, D_VMAX, D_COMP0 /* h v tq td ta x3 */, D_RI = 22, D_SEG_OFF, D_NSEG, D_SUB_BASE, D_NSUB, D_BLK_BASE,
       D_NBLK, D_PLANE_BASE, D_OUT_OFF, D_OUT_PITCH, D_MCUX, D_MCUY, D_BPM, D_QT, D_HT };

constexpr int LUT_BITS = 10, NT = 1024;

struct Tables {
    uint16_t lut[8][1 << LUT_BITS];      // (length << 8) | symbol for codes of <= LUT_BITS bits; 0: longer
    int maxcode[8][17];                   // largest code of each length, -1: none
    int valoff[8][17];                    // vals index of the first code of that length minus that code
    unsigned char vals[8][256];
    unsigned char tb_dc[10], tb_ac[10];   // table slot (0-3 DC, 4-7 AC) of each block of the MCU
};

struct Reader {
    // The scan is read as aligned 8-byte words (one word ahead in flight), byte-swapped so that the next raw byte is the top byte of `cur`.
    // Decoded bits sit MSB-first in `acc` (the next bit is bit 63, n valid bits).  A fetched FF takes its stuffed 00 with it; `ff` remembers
    // which of the buffered bytes those were, so that position() does not depend on how far ahead the buffer was filled.
    const uint64_t* w; uint32_t pos, end, widx; uint64_t cur, nxt, acc; int cur_n, n; uint32_t ff;
    __device__ __forceinline__ uint64_t word(uint32_t i) const { return __builtin_bswap64(w[i]); }
    __device__ __forceinline__ uint32_t raw_at(uint32_t o) const { return (uint32_t)(word(o >> 3) >> (56 - 8 * (o & 7))) & 255u; }
    __device__ __forceinline__ void raw_advance(int k) {
        cur <<= 8 * k; cur_n -= k;
        if (cur_n == 0) { cur = nxt; cur_n = 8; nxt = word(widx++); }
    }
    __device__ __forceinline__ void fill() {
        while (n <= 56) {
            if (n <= 32 && cur_n >= 4 && pos + 4 <= end) {   // four bytes at once when none of them is FF
                const uint32_t x = (uint32_t)(cur >> 32), y = ~x;
                if (!((y - 0x01010101u) & ~y & 0x80808080u)) {
                    acc |= (uint64_t)x << (32 - n);
                    n += 32; ff <<= 4; pos += 4;
                    raw_advance(4);
                    continue;
                }
            }
            uint32_t c = 0, isff = 0;
            if (pos < end) {
                c = (uint32_t)(cur >> 56);
                raw_advance(1);
                if (c == 0xFF && pos + 1 < end && (uint32_t)(cur >> 56) == 0) { raw_advance(1); pos++; isff = 1; }
            }
            pos++;                                           // past the end: zero bytes, the position keeps counting
            acc |= (uint64_t)c << (56 - n);
            n += 8; ff = (ff << 1) | isff;
        }
    }
    // bit position of the next bit, in units that depend only on where it is in the file: 8 * (offset just past its byte, stuffing
    // included) - (bits of that byte not yet consumed)
    __device__ __forceinline__ uint32_t position() const {
        const int later = (n - 1) >> 3;
        return pos * 8u - (uint32_t)n - 8u * (uint32_t)__popc(ff & ((1u << later) - 1u));
    }
    __device__ __forceinline__ uint32_t peek(int k) const { return k ? (uint32_t)(acc >> (64 - k)) : 0u; }
    __device__ __forceinline__ void skip(int k) { acc <<= k; n -= k; }
    // start at bit position P (a value position() returned, or 8 * a byte offset for a fresh start at a subsequence boundary); offsets are
    // relative to `data`, which is 8-byte aligned
    __device__ __forceinline__ void seek(const unsigned char* data, uint32_t seg_start, uint32_t seg_end, uint32_t P) {
        w = reinterpret_cast<const uint64_t*>(data); end = seg_end; acc = 0; n = 0; ff = 0;
        const uint32_t q = P >> 3;
        pos = (q > seg_start && q < seg_end && raw_at(q) == 0 && raw_at(q - 1) == 0xFF) ? q - 1 : q;
        widx = pos >> 3;
        cur = word(widx) << (8 * (pos & 7)); cur_n = 8 - (int)(pos & 7);
        nxt = word(widx + 1); widx += 2;
        fill();
        skip((int)(P & 7u));
    }
};

__device__ __forceinline__ int extend(int v, int s) { return (s && v < (1 << (s - 1))) ? v - (1 << s) + 1 : v; }

// one Huffman symbol (+ its value bits) of state (b, k); returns true when it completed a block
template <bool WRITE>
__device__ __forceinline__ bool symbol(Reader& r, const Tables& T, int& b, int& k, int bpm, short* blk_out) {
    r.fill();
    const int t = k == 0 ? T.tb_dc[b] : T.tb_ac[b];
    const uint32_t pk = r.peek(16);
    int len, sym;
    const uint32_t e = T.lut[t][pk >> (16 - LUT_BITS)];
    if (e) { len = e >> 8; sym = e & 255; }
    else {
        len = 16; sym = 0;
        for (int l = LUT_BITS + 1; l <= 16; l++) {
            const int code = (int)(pk >> (16 - l));
            if (code <= T.maxcode[t][l]) { len = l; sym = T.vals[t][(T.valoff[t][l] + code) & 255]; break; }
        }
    }
    r.skip(len);
    if (k == 0) {
        const int s = sym & 15;
        const int v = extend((int)r.peek(s), s);
        r.skip(s);
        if (WRITE) blk_out[0] = (short)v;                    // the DC difference; the prediction scan below turns it into the value
        k = 1;
    } else {
        const int run = sym >> 4, s = sym & 15;
        if (s == 0) k = run == 15 ? k + 16 : 64;
        else {
            k += run;
            const int v = extend((int)r.peek(s), s);
            r.skip(s);
            if (WRITE && k < 64) blk_out[ZZ[k]] = (short)v;
            k++;
        }
    }
    if (k >= 64) { k = 0; b = b + 1 == bpm ? 0 : b + 1; return true; }
    return false;
}

struct EntropyArgs {
    const unsigned char* data; const int32_t* desc; const int32_t* segs; const unsigned char* htabs;
    short* coefs; uint32_t* sP; uint32_t* sS; int32_t* sN; uint32_t* cP; uint32_t* cS;
    int sub_bytes;
};

__device__ __forceinline__ int seg_of_sub(const int32_t* segs, int nseg, int u) {       // segs[s][2] = first subsequence of segment s
    int lo = 0, hi = nseg - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (segs[mid * 4 + 2] <= u) lo = mid; else hi = mid - 1; }
    return lo;
}

__global__ __launch_bounds__(NT) void jpeg_entropy_kernel(EntropyArgs a) {
    __shared__ Tables T;
    __shared__ int s_scan[NT];
    __shared__ int s_flag[NT];
    __shared__ int s_carry;
    const int tid = threadIdx.x;
    const int32_t* d = a.desc + (size_t)blockIdx.x * AB_JPEG_DESC_INTS;
    const int ncomp = d[D_NCOMP], bpm = d[D_BPM], nseg = d[D_NSEG], nsub = d[D_NSUB], ri = d[D_RI];
    const unsigned char* base = a.data;                      // every offset below is relative to `data` (8-byte aligned)
    const uint32_t off0 = (uint32_t)d[D_OFF];
    const int32_t* segs = a.segs + (size_t)d[D_SEG_OFF] * 4;
    // ---- decode tables of this image: 8 slots x (16 counts + 256 symbols)
    for (int i = tid; i < 8 * (1 << LUT_BITS); i += NT) (&T.lut[0][0])[i] = 0;
    const unsigned char* ht = a.htabs + (size_t)d[D_HT] * (8 * 272);
    for (int i = tid; i < 8 * 256; i += NT) T.vals[i >> 8][i & 255] = ht[(i >> 8) * 272 + 16 + (i & 255)];
    if (tid < 10) {
        int c = 0, acc = 0;
        for (; c < ncomp; c++) { const int nb = d[D_COMP0 + 5 * c] * d[D_COMP0 + 5 * c + 1]; if (tid < acc + nb) break; acc += nb; }
        c = c < ncomp ? c : 0;
        T.tb_dc[tid] = (unsigned char)(d[D_COMP0 + 5 * c + 3] & 3);
        T.tb_ac[tid] = (unsigned char)(4 + (d[D_COMP0 + 5 * c + 4] & 3));
    }
    __syncthreads();
    if (tid < 8) {
        const unsigned char* cnt = ht + tid * 272;
        int code = 0, kk = 0;
        for (int l = 1; l <= 16; l++) {
            T.valoff[tid][l] = kk - code;
            const int nl = cnt[l - 1];
            if (l <= LUT_BITS)
                for (int j = 0; j < nl; j++) {
                    const uint16_t e = (uint16_t)((l << 8) | cnt[16 + kk + j]);
                    const int first = (code + j) << (LUT_BITS - l);
                    if (first + (1 << (LUT_BITS - l)) <= (1 << LUT_BITS))
                        for (int f = 0; f < (1 << (LUT_BITS - l)); f++) T.lut[tid][first + f] = e;
                }
            code += nl; kk += nl;
            T.maxcode[tid][l] = nl ? code - 1 : -1;
            code <<= 1;
        }
    }
    __syncthreads();
    uint32_t* sP = a.sP + d[D_SUB_BASE]; uint32_t* sS = a.sS + d[D_SUB_BASE]; int32_t* sN = a.sN + d[D_SUB_BASE];
    uint32_t* cP = a.cP + d[D_SUB_BASE]; uint32_t* cS = a.cS + d[D_SUB_BASE];
    const uint32_t SB = (uint32_t)a.sub_bytes;
    // ---- round 0: every subsequence from its first bit, state (block 0 of the MCU, k = 0)
    for (int u = tid; u < nsub; u += NT) {
        const int s = seg_of_sub(segs, nseg, u);
        const uint32_t s0 = off0 + (uint32_t)segs[s * 4], s1 = s0 + (uint32_t)segs[s * 4 + 1];
        const uint32_t j = (uint32_t)(u - segs[s * 4 + 2]);
        const uint32_t endbits = min(s1, s0 + (j + 1) * SB) * 8u;
        Reader r; r.seek(base, s0, s1, (s0 + j * SB) * 8u);
        int b = 0, k = 0, nb = 0;
        r.fill();
        while (r.position() < endbits) { nb += symbol<false>(r, T, b, k, bpm, nullptr) ? 1 : 0; r.fill(); }
        const uint32_t P = r.position(), S = (uint32_t)b | ((uint32_t)k << 8);
        sP[u] = P; sS[u] = S; sN[u] = nb;
        cP[u] = P; cS[u] = S;                                // the chain of the thread that started at u, now at the end of u
    }
    __syncthreads();
    // ---- rounds r = 1, 2, ...: chain u continues through subsequence u + r until it meets the recorded state there
    for (int rnd = 1;; rnd++) {
        int active = 0;
        for (int u = tid; u < nsub; u += NT) {
            const uint32_t P0 = cP[u];
            if (P0 == 0xFFFFFFFFu) continue;
            const int v = u + rnd;
            const int s = seg_of_sub(segs, nseg, u);
            const int last = (s + 1 < nseg ? segs[(s + 1) * 4 + 2] : nsub) - 1;          // last subsequence of the segment
            if (v > last) { cP[u] = 0xFFFFFFFFu; continue; }
            const uint32_t s0 = off0 + (uint32_t)segs[s * 4], s1 = s0 + (uint32_t)segs[s * 4 + 1];
            const uint32_t j = (uint32_t)(v - segs[s * 4 + 2]);
            const uint32_t endbits = min(s1, s0 + (j + 1) * SB) * 8u;
            Reader r; r.seek(base, s0, s1, P0);
            int b = (int)(cS[u] & 255u), k = (int)(cS[u] >> 8), nb = 0;
            r.fill();
            while (r.position() < endbits) { nb += symbol<false>(r, T, b, k, bpm, nullptr) ? 1 : 0; r.fill(); }
            const uint32_t P = r.position(), S = (uint32_t)b | ((uint32_t)k << 8);
            // the chains reach v in order of decreasing start index, the last one (matching or not) carries the true state into v: its
            // count of blocks completed inside v is the one that stays
            sN[v] = nb;
            if (sP[v] == P && sS[v] == S) cP[u] = 0xFFFFFFFFu;                      // synchronised: from here on it is v's chain
            else { sP[v] = P; sS[v] = S; cP[u] = P; cS[u] = S; active = 1; }
        }
        if (!__syncthreads_or(active)) { if (tid == 0) cP[0] = (uint32_t)rnd; break; }      // diagnostic: rounds this image took
    }
    // ---- exclusive scan of the counts over the image's subsequences (restarted at segment starts in the next pass by subtraction)
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int u0 = 0; u0 < nsub; u0 += NT) {
        const int u = u0 + tid;
        const int v = u < nsub ? sN[u] : 0;
        s_scan[tid] = v;
        __syncthreads();
        for (int o = 1; o < NT; o <<= 1) {
            const int t = tid >= o ? s_scan[tid - o] : 0;
            __syncthreads();
            s_scan[tid] += t;
            __syncthreads();
        }
        const int carry = s_carry;
        if (u < nsub) sN[u] = carry + s_scan[tid] - v;
        __syncthreads();
        if (tid == NT - 1) s_carry = carry + s_scan[tid];
        __syncthreads();
    }
    // ---- coefficient pass
    short* coefs = a.coefs + (size_t)d[D_BLK_BASE] * 64;
    for (int u = tid; u < nsub; u += NT) {
        const int s = seg_of_sub(segs, nseg, u);
        const uint32_t s0 = off0 + (uint32_t)segs[s * 4], s1 = s0 + (uint32_t)segs[s * 4 + 1];
        const int jf = segs[s * 4 + 2];
        const uint32_t j = (uint32_t)(u - jf);
        const uint32_t endbits = min(s1, s0 + (j + 1) * SB) * 8u;
        const int seg_first_blk = segs[s * 4 + 3];
        const int seg_nblk = (s + 1 < nseg ? segs[(s + 1) * 4 + 3] : d[D_NBLK]) - seg_first_blk;
        Reader r; int b = 0, k = 0;
        if (u == jf) r.seek(base, s0, s1, s0 * 8u);
        else { r.seek(base, s0, s1, sP[u - 1]); b = (int)(sS[u - 1] & 255u); k = (int)(sS[u - 1] >> 8); }
        int nb = sN[u] - sN[jf];                             // blocks of this segment completed before this subsequence
        r.fill();
        while (r.position() < endbits && nb < seg_nblk) {
            nb += symbol<true>(r, T, b, k, bpm, coefs + (size_t)(seg_first_blk + nb) * 64) ? 1 : 0;
            r.fill();
        }
    }
    __syncthreads();
    // ---- DC prediction: running sum of the differences per component, in scan order, reset at restart intervals
    const int nmcu = d[D_MCUX] * d[D_MCUY];
    int off = 0;
    for (int c = 0; c < ncomp; c++) {
        const int nbc = d[D_COMP0 + 5 * c] * d[D_COMP0 + 5 * c + 1];
        const int count = nmcu * nbc, L = (count + NT - 1) / NT;
        const int q0 = min(tid * L, count), q1 = min(q0 + L, count);
        int run = 0, hasreset = 0;
        for (int q = q0; q < q1; q++) {
            const int m = q / nbc, w = q - m * nbc;
            if (ri && w == 0 && m % ri == 0) { run = 0; hasreset = 1; }
            run += coefs[(size_t)(m * bpm + off + w) * 64];
        }
        s_scan[tid] = run; s_flag[tid] = hasreset;
        __syncthreads();
        for (int o = 1; o < NT; o <<= 1) {                   // inclusive segmented scan: (sum, flag)
            int ts = 0, tf = 0;
            if (tid >= o) { ts = s_scan[tid - o]; tf = s_flag[tid - o]; }
            __syncthreads();
            if (tid >= o) { if (!s_flag[tid]) s_scan[tid] += ts; s_flag[tid] |= tf; }
            __syncthreads();
        }
        run = tid ? s_scan[tid - 1] : 0;                     // what precedes this thread's chunk (since the last reset)
        __syncthreads();
        for (int q = q0; q < q1; q++) {
            const int m = q / nbc, w = q - m * nbc;
            if (ri && w == 0 && m % ri == 0) run = 0;
            short* p = coefs + (size_t)(m * bpm + off + w) * 64;
            run += *p;
            *p = (short)run;
        }
        off += nbc;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))
__device__ __forceinline__ void idct_1d(const int* in, int stride, int* o) {         // jidctint.c butterfly, unscaled outputs
    int z2 = in[2 * stride], z3 = in[6 * stride];
    int z1 = (z2 + z3) * 4433;
    int tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
    z2 = in[0]; z3 = in[4 * stride];
    int tmp0 = (z2 + z3) * 8192, tmp1 = (z2 - z3) * 8192;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7 * stride]; tmp1 = in[5 * stride]; tmp2 = in[3 * stride]; tmp3 = in[stride];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * 9633;
    tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    o[0] = tmp10 + tmp3; o[7] = tmp10 - tmp3; o[1] = tmp11 + tmp2; o[6] = tmp11 - tmp2;
    o[2] = tmp12 + tmp1; o[5] = tmp12 - tmp1; o[3] = tmp13 + tmp0; o[4] = tmp13 - tmp0;
}
__device__ __forceinline__ int idct_range(int x) {          // range_limit[x & RANGE_MASK] of jdmaster.c, centred on 128
    x &= 1023;
    return x < 128 ? x + 128 : x < 512 ? 255 : x < 896 ? 0 : x - 896;
}

__global__ __launch_bounds__(256) void jpeg_idct_kernel(const int32_t* desc, const short* coefs, const unsigned short* qtabs, unsigned char* planes) {
    __shared__ int ws[32][64 + 1];
    const int32_t* d = desc + (size_t)blockIdx.y * AB_JPEG_DESC_INTS;
    const int l8 = threadIdx.x & 7, lb = threadIdx.x >> 3;
    const int g = blockIdx.x * 32 + lb;
    const bool live = g < d[D_NBLK];
    int c = 0, bx = 0, by = 0, pitch = 0; size_t pbase = 0;
    if (live) {
        const int bpm = d[D_BPM], m = g / bpm, jb = g - m * bpm;
        int acc = 0; size_t po = (size_t)d[D_PLANE_BASE];
        for (c = 0; c < d[D_NCOMP]; c++) {
            const int h = d[D_COMP0 + 5 * c], v = d[D_COMP0 + 5 * c + 1];
            if (jb < acc + h * v) {
                const int w = jb - acc, mx = m % d[D_MCUX], my = m / d[D_MCUX];
                bx = mx * h + w % h; by = my * v + w / h; pitch = d[D_MCUX] * h * 8; pbase = po;
                break;
            }
            acc += h * v; po += (size_t)d[D_MCUX] * h * 8 * d[D_MCUY] * v * 8;
        }
        const unsigned short* q = qtabs + ((size_t)d[D_QT] * 4 + (d[D_COMP0 + 5 * c + 2] & 3)) * 64 + l8 * 8;
        const short* cf = coefs + ((size_t)d[D_BLK_BASE] + g) * 64 + l8 * 8;
        const int4 raw = *reinterpret_cast<const int4*>(cf);
        const short* rs = reinterpret_cast<const short*>(&raw);
#pragma unroll
        for (int i = 0; i < 8; i++) ws[lb][l8 * 8 + i] = (int)rs[i] * (int)q[i];
    }
    __syncthreads();
    int o[8];
    if (live) {                                              // pass 1: column l8
        idct_1d(&ws[lb][l8], 8, o);
#pragma unroll
        for (int r = 0; r < 8; r++) ws[lb][r * 8 + l8] = DESCALE(o[r], 11);
    }
    __syncthreads();
    if (live) {                                              // pass 2: row l8
        idct_1d(&ws[lb][l8 * 8], 1, o);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            lo |= (uint32_t)idct_range(DESCALE(o[i], 18)) << (8 * i);
            hi |= (uint32_t)idct_range(DESCALE(o[4 + i], 18)) << (8 * i);
        }
        *reinterpret_cast<uint2*>(planes + pbase + ((size_t)by * 8 + l8) * pitch + bx * 8) = make_uint2(lo, hi);
    }
}

struct PlaneRef { const unsigned char* P; int pitch, dw, dh, hr, vr; };

__device__ __forceinline__ int upsampled(const PlaneRef& c, int x, int y) {           // jdsample.c, the method libjpeg picks for this component
    const unsigned char* P = c.P;
    if (c.hr == 1 && c.vr == 1) return P[(size_t)y * c.pitch + x];
    const bool fancy = c.dw > 2;
    if (c.hr == 2 && c.vr == 1 && fancy) {
        const unsigned char* r = P + (size_t)y * c.pitch; const int i = x >> 1;
        if (x & 1) return i == c.dw - 1 ? r[i] : (r[i] * 3 + r[i + 1] + 2) >> 2;
        return i == 0 ? r[i] : (r[i] * 3 + r[i - 1] + 1) >> 2;
    }
    if (c.hr == 2 && c.vr == 2 && fancy) {
        const int ir = y >> 1; int nr = (y & 1) ? ir + 1 : ir - 1;
        nr = nr < 0 ? 0 : nr > c.dh - 1 ? c.dh - 1 : nr;
        const unsigned char *r0 = P + (size_t)ir * c.pitch, *r1 = P + (size_t)nr * c.pitch; const int i = x >> 1;
        const int cur = r0[i] * 3 + r1[i];
        if (x & 1) return i == c.dw - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + r0[i + 1] * 3 + r1[i + 1] + 7) >> 4;
        return i == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + r0[i - 1] * 3 + r1[i - 1] + 8) >> 4;
    }
    return P[(size_t)(y / c.vr) * c.pitch + x / c.hr];
}
__device__ __forceinline__ int clamp255(int x) { return x < 0 ? 0 : x > 255 ? 255 : x; }

// four horizontally adjacent samples (x0 .. x0 + 3, x0 % 4 == 0) of one component at full resolution
__device__ __forceinline__ void upsampled4(const PlaneRef& c, int x0, int y, int* v) {
    if (c.hr == 1 && c.vr == 1) {
        const uint32_t q = *reinterpret_cast<const uint32_t*>(c.P + (size_t)y * c.pitch + x0);
        v[0] = q & 255; v[1] = (q >> 8) & 255; v[2] = (q >> 16) & 255; v[3] = q >> 24;
        return;
    }
    if (c.hr == 2 && c.dw > 2 && (c.vr == 1 || c.vr == 2)) {      // fancy h2v1 / h2v2: columns i0 - 1 .. i0 + 2 of one or two rows
        const int i0 = x0 >> 1, last = c.dw - 1;
        const int ja = max(i0 - 1, 0), jb = min(i0, last), jc = min(i0 + 1, last), jd = min(i0 + 2, last);
        if (c.vr == 1) {
            const unsigned char* r = c.P + (size_t)y * c.pitch;
            const int a = r[ja], b = r[jb], cc = r[jc], d = r[jd];
            v[0] = i0 == 0 ? b : (b * 3 + a + 1) >> 2;
            v[1] = i0 == last ? b : (b * 3 + cc + 2) >> 2;
            v[2] = (cc * 3 + b + 1) >> 2;
            v[3] = i0 + 1 >= last ? cc : (cc * 3 + d + 2) >> 2;
        } else {
            const int ir = y >> 1; int nr = (y & 1) ? ir + 1 : ir - 1;
            nr = nr < 0 ? 0 : nr > c.dh - 1 ? c.dh - 1 : nr;
            const unsigned char *r0 = c.P + (size_t)ir * c.pitch, *r1 = c.P + (size_t)nr * c.pitch;
            const int a = r0[ja] * 3 + r1[ja], b = r0[jb] * 3 + r1[jb], cc = r0[jc] * 3 + r1[jc], d = r0[jd] * 3 + r1[jd];
            v[0] = i0 == 0 ? (b * 4 + 8) >> 4 : (b * 3 + a + 8) >> 4;
            v[1] = i0 == last ? (b * 4 + 7) >> 4 : (b * 3 + cc + 7) >> 4;
            v[2] = (cc * 3 + b + 8) >> 4;
            v[3] = i0 + 1 >= last ? (cc * 4 + 7) >> 4 : (cc * 3 + d + 7) >> 4;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = upsampled(c, x0 + i, y);
}

__global__ __launch_bounds__(256) void jpeg_color_kernel(const int32_t* desc, const unsigned char* planes, unsigned char* out, int channels) {
    const int32_t* d = desc + (size_t)blockIdx.y * AB_JPEG_DESC_INTS;
    const int W = d[D_W], H = d[D_H], W4 = (W + 3) >> 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= W4 * H) return;
    const int y = i / W4, x0 = (i - y * W4) * 4;
    const int ncomp = d[D_NCOMP], hmax = d[D_HMAX], vmax = d[D_VMAX];
    PlaneRef pr[3];
    size_t po = (size_t)d[D_PLANE_BASE];
    for (int c = 0; c < ncomp; c++) {
        const int h = d[D_COMP0 + 5 * c], v = d[D_COMP0 + 5 * c + 1];
        pr[c].P = planes + po; pr[c].pitch = d[D_MCUX] * h * 8;
        pr[c].dw = (W * h + hmax - 1) / hmax; pr[c].dh = (H * v + vmax - 1) / vmax; pr[c].hr = hmax / h; pr[c].vr = vmax / v;
        po += (size_t)pr[c].pitch * d[D_MCUY] * v * 8;
    }
    int Y[4], Cb[4], Cr[4];
    upsampled4(pr[0], x0, y, Y);
    if (ncomp == 3) { upsampled4(pr[1], x0, y, Cb); upsampled4(pr[2], x0, y, Cr); }
    uint32_t px[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int R = Y[j], G = Y[j], B = Y[j];
        if (ncomp == 3) {
            const int cb = Cb[j] - 128, cr = Cr[j] - 128;
            R = clamp255(Y[j] + ((91881 * cr + 32768) >> 16));
            G = clamp255(Y[j] + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
            B = clamp255(Y[j] + ((116130 * cb + 32768) >> 16));
        }
        px[j] = (uint32_t)R | ((uint32_t)G << 8) | ((uint32_t)B << 16);
    }
    const size_t o0 = (size_t)d[D_OUT_OFF] + (size_t)y * d[D_OUT_PITCH] + x0;
    if (channels == 4 && x0 + 3 < W && !(o0 & 3)) { *reinterpret_cast<uint4*>(out + o0 * 4) = make_uint4(px[0], px[1], px[2], px[3]); return; }
    for (int j = 0; j < 4 && x0 + j < W; j++) {
        unsigned char* o = out + (o0 + j) * channels;
        if (channels == 4) *reinterpret_cast<uint32_t*>(o) = px[j];
        else { o[0] = (unsigned char)px[j]; o[1] = (unsigned char)(px[j] >> 8); o[2] = (unsigned char)(px[j] >> 16); }
    }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" long ab_jpeg_workspace_bytes(long total_blocks, long total_subseq, long plane_bytes) {
    return (long)(align256((size_t)total_blocks * 128) + 5 * align256((size_t)total_subseq * 4) + align256((size_t)plane_bytes));
}

extern "C" int ab_jpeg_decode_batch(const void* data, const int32_t* desc, const int32_t* segs, const void* qtabs, const void* htabs, int n,
                                    int sub_bytes, long total_blocks, long total_subseq, long plane_bytes, int max_blocks, int max_pixels,
                                    int out_channels, void* out, void* workspace, void* stream) {
    if (n <= 0) return 0;
    if (sub_bytes < 16 || (out_channels != 3 && out_channels != 4) || !data || !desc || !segs || !qtabs || !htabs || !out || !workspace) return -1;
    hipStream_t st = as_stream(stream);
    char* w = (char*)workspace;
    short* coefs = (short*)w; w += align256((size_t)total_blocks * 128);
    uint32_t* arr[5];
    for (int i = 0; i < 5; i++) { arr[i] = (uint32_t*)w; w += align256((size_t)total_subseq * 4); }
    unsigned char* planes = (unsigned char*)w;
    (void)plane_bytes;
    hipError_t e = hipMemsetAsync(coefs, 0, (size_t)total_blocks * 128, st);
    if (e != hipSuccess) return (int)e;
    EntropyArgs a{(const unsigned char*)data, desc, segs, (const unsigned char*)htabs, coefs, arr[0], arr[1], (int32_t*)arr[2], arr[3], arr[4], sub_bytes};
    hipLaunchKernelGGL(jpeg_entropy_kernel, dim3(n), dim3(NT), 0, st, a);
    AB_LAUNCH_CHECK();
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((max_blocks + 31) / 32, n), dim3(256), 0, st, desc, coefs, (const unsigned short*)qtabs, planes);
    AB_LAUNCH_CHECK();
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((max_pixels / 4 + 255) / 256, n), dim3(256), 0, st, desc, planes, (unsigned char*)out, out_channels);
    AB_LAUNCH_CHECK();
    return 0;
}
