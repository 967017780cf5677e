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
//   jpeg_tables_kernel   LUT + canonical-code tables of every distinct DHT set of the batch
//   jpeg_unstuff_kernel  FF 00 -> FF (one workgroup per image: count, scan, copy), unstuffed extent of every restart interval
//   jpeg_sync_kernel     one launch per round (R_MAX + 1 of them; a round in which an image has no live chain exits at once), one thread per subsequence
//   jpeg_finish_kernel   one workgroup per image: any rounds beyond R_MAX (never seen on photographs), block-count scan
//   jpeg_coef_kernel     second pass over every subsequence from its true state, coefficients (de-zigzagged) to their blocks
//   jpeg_dc_kernel       DC prediction: segmented running sum per component
//   jpeg_idct_kernel     de-quantise + jidctint.c "islow" 8x8 integer IDCT, 8 lanes per block, planes of uint8 samples
//   jpeg_color_kernel    jdsample.c fancy h2v1 / h2v2 up-sampling (replication for widths <= 2) + jdcolor.c YCbCr -> RGB, RGB(X) rows out
// The JFIF header (a few hundred bytes) is parsed on the host (artiboost_amd/jpeg.py).
#include <stdlib.h>
#include "common.h"

namespace {

__device__ const unsigned char ZZ[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                         35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// descriptor fields (int32 per image, AB_JPEG_DESC_INTS of them; filled by artiboost_amd/jpeg.py)
enum { D_OFF = 0, D_LEN, D_W, D_H, D_NCOMP, D_HMAX, D_VMAX, D_COMP0 /* h v tq td ta x3 */, D_RI = 22, D_SEG_OFF, D_NSEG, D_SUB_BASE, D_NSUB, D_BLK_BASE,
       D_NBLK, D_PLANE_BASE, D_OUT_OFF, D_OUT_PITCH, D_MCUX, D_MCUY, D_BPM, D_QT, D_HT };

constexpr int LUT_BITS = 9, LUT2 = 1024, NT = 1024, NW = 256, R_MAX = 16;

// decode tables of one set of DHT segments (8 slots: DC 0-3, AC 0-3), built once per batch; every decoding workgroup copies the two
// look-up levels to LDS.  Level 1: the first LUT_BITS bits of the 16-bit look-ahead.  Level 2: canonical codes longer than LUT_BITS bits are
// the numerically LARGEST look-aheads (>= base2), a few hundred values for the tables encoders write -- indexed directly.  A table whose long
// codes span more than LUT2 values falls back to the bit-by-bit canonical search in memory (slow2 == 0).
struct HuffLuts {
    uint16_t lut1[8][1 << LUT_BITS];     // (length << 8) | symbol; longer than LUT_BITS bits: 0x8000 | i (the code is in lut2[128 i ..]) or 0x4000
    uint16_t lut2[8][LUT2];               // (length << 8) | symbol for look-ahead values base2 + i
};
struct HuffTables {
    HuffLuts luts;
    int maxcode[8][17];                   // largest code of each length, -1: none
    int valoff[8][17];                    // vals index of the first code of that length minus that code
    unsigned char vals[8][256];
    uint32_t base2[8], ok2[8];
};
static_assert(sizeof(HuffLuts) % 16 == 0 && sizeof(HuffTables) % 16 == 0, "copied as uint4");

struct Lds {
    HuffLuts T;
    unsigned char zz[64];
};
// table slots of the blocks of one MCU, 3 bits each (<= 10 blocks): DC slots 0-3, AC slots 4-7
struct McuTables { uint32_t dc, ac; };
__device__ __forceinline__ McuTables mcu_tables(const int32_t* d) {
    McuTables m{0u, 0u};
    int j = 0;
    for (int c = 0; c < d[4 /* D_NCOMP */]; c++) {
        const int nb = d[7 + 5 * c] * d[7 + 5 * c + 1];
        for (int i = 0; i < nb && j < 10; i++, j++) {
            m.dc |= (uint32_t)(d[7 + 5 * c + 3] & 3) << (3 * j);
            m.ac |= (uint32_t)(4 + (d[7 + 5 * c + 4] & 3)) << (3 * j);
        }
    }
    return m;
}

// MSB-first bit reader over the UNSTUFFED scan (jpeg_unstuff_kernel), in aligned 32-bit words: the next bit is bit 63 of acc, n valid bits;
// bytes past `end` read as zero.  STAGED: the words of this thread's stretch were copied to LDS up front (lds[j * NW + thread], word w0 + j) --
// one batch of independent loads instead of a dependent load every five symbols; otherwise straight from memory, one word ahead.
constexpr int STAGE_MAX_SUB = 128, STAGE_WORDS = STAGE_MAX_SUB / 4 + 4;
template <bool STAGED>
struct Reader {
    const uint32_t* base; uint32_t widx, w0, end, pre; uint64_t acc; int n;
    __device__ __forceinline__ uint32_t load(uint32_t i) const { return STAGED ? base[(i - w0) * NW] : base[i]; }
    __device__ __forceinline__ uint32_t cooked(uint32_t raw, uint32_t i) const {          // big-endian, bytes past `end` zero
        const uint32_t x = __builtin_bswap32(raw), o = i * 4u;
        return o + 4u <= end ? x : o >= end ? 0u : x & (0xFFFFFFFFu << (8u * (4u - (end - o))));
    }
    __device__ __forceinline__ void fill() {                 // one refill per symbol: a symbol takes at most 16 + 15 bits
        if (n <= 32) {
            acc |= (uint64_t)cooked(pre, widx) << (32 - n);
            n += 32; widx++;
            pre = load(widx);                                // raw: not looked at before the next refill
        }
    }
    __device__ __forceinline__ uint32_t position() const { return widx * 32u - (uint32_t)n; }
    // b: the unstuffed scans as words (STAGED: this thread's LDS column, filled by stage()); P: bit position to start at
    __device__ __forceinline__ void seek(const uint32_t* b, uint32_t seg_end, uint32_t P) {
        base = b; end = seg_end; widx = P >> 5; w0 = widx; acc = 0; n = 0;
        pre = load(widx);
        fill();
        acc <<= (P & 31u); n -= (int)(P & 31u);
    }
};
// copies the words [P >> 5, (endbits + 95) >> 5] of the unstuffed scan to this thread's LDS column
__device__ __forceinline__ void stage(uint32_t* col, const uint32_t* words, uint32_t P, uint32_t endbits) {
    const uint32_t w0 = P >> 5, cnt = min(((endbits + 95u) >> 5) - w0 + 1u, (uint32_t)STAGE_WORDS);
    uint32_t v[STAGE_WORDS];
#pragma unroll
    for (int j = 0; j < STAGE_WORDS; j++) v[j] = (uint32_t)j < cnt ? words[w0 + j] : 0u;
#pragma unroll
    for (int j = 0; j < STAGE_WORDS; j++) col[j * NW] = v[j];
}

// one Huffman symbol (+ its value bits) in state (b, k); returns 1 when it completed a block
template <bool WRITE, bool STAGED>
__device__ __forceinline__ int symbol(Reader<STAGED>& r, const Lds& L, const HuffTables* G, const McuTables mt, int& b, int& k, int bpm, short* blk_out) {
    r.fill();
    const bool isdc = k == 0;
    const int t = (int)(((isdc ? mt.dc : mt.ac) >> (3 * b)) & 7u);
    const uint32_t pk = (uint32_t)(r.acc >> 48);
    uint32_t e = L.T.lut1[t][pk >> (16 - LUT_BITS)];
    if (e & 0xC000u) {
        if (e & 0x8000u) e = L.T.lut2[t][((e & 7u) << 7) | (pk & 127u)];
        else {
            e = 16u << 8;                                    // invalid code: 16 bits, symbol 0 (libjpeg: warning + 0)
            for (int l = LUT_BITS + 1; l <= 16; l++) {
                const int code = (int)(pk >> (16 - l));
                if (code <= G->maxcode[t][l]) { e = ((uint32_t)l << 8) | G->vals[t][(G->valoff[t][l] + code) & 255]; break; }
            }
        }
    }
    const int len = (int)(e >> 8), sym = (int)(e & 255u);
    const int s = sym & 15, run = isdc ? 0 : sym >> 4;
    r.acc <<= len;
    const int bits = (int)((r.acc >> 1) >> (63 - s));       // s == 0: 0
    r.acc <<= s; r.n -= len + s;
    const int v = (s && bits < (1 << (s - 1))) ? bits - (1 << s) + 1 : bits;
    const bool eob = !isdc && s == 0 && run != 15;
    const int kn = k + run;
    if (WRITE && s && kn < 64) blk_out[L.zz[kn]] = (short)v;        // DC: the difference; jpeg_dc_kernel turns it into the value
    k = eob ? 64 : kn + 1;
    const int done = k >= 64;
    k = done ? 0 : k;
    b = done ? (b + 1 == bpm ? 0 : b + 1) : b;
    return done;
}

struct EntropyArgs {
    const unsigned char* data; unsigned char* clean; const int32_t* desc; const int32_t* segs; int32_t* segc; int32_t* kp; const unsigned char* htabs;
    HuffTables* tables; short* coefs; uint32_t* sP; uint32_t* sS; int32_t* sN; uint32_t* cP; uint32_t* cS; int32_t* act;
    int sub_bytes, n_tables, no_lut2;
};

__device__ __forceinline__ int seg_of_sub(const int32_t* segs, int nseg, int u) {       // segs[s][2] = first subsequence of segment s
    int lo = 0, hi = nseg - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (segs[mid * 4 + 2] <= u) lo = mid; else hi = mid - 1; }
    return lo;
}

// ---- per batch: the decode tables of every distinct set of DHT segments
__global__ __launch_bounds__(256) void jpeg_tables_kernel(EntropyArgs a) {
    __shared__ unsigned char s_ht[8 * 272];
    __shared__ uint32_t s_base2[8], s_ok2[8];
    HuffTables& T = a.tables[blockIdx.x];
    const unsigned char* ht = a.htabs + (size_t)blockIdx.x * (8 * 272);
    const int tid = threadIdx.x;
    for (int i = tid; i < 8 * 272; i += 256) s_ht[i] = ht[i];
    for (int i = tid; i < 8 * (1 << LUT_BITS); i += 256) (&T.luts.lut1[0][0])[i] = (uint16_t)0x4000u;     // no code here: the canonical search says so
    for (int i = tid; i < 8 * LUT2; i += 256) (&T.luts.lut2[0][0])[i] = (uint16_t)(16u << 8);      // not a code: 16 bits, symbol 0
    __syncthreads();
    for (int i = tid; i < 8 * 256; i += 256) T.vals[i >> 8][i & 255] = s_ht[(i >> 8) * 272 + 16 + (i & 255)];
    if (tid < 8) {
        const unsigned char* cnt = s_ht + tid * 272;
        int code = 0;
        for (int l = 1; l <= LUT_BITS; l++) code = (code + cnt[l - 1]) << 1;          // first code of length LUT_BITS + 1
        const uint32_t b2 = (uint32_t)code << (16 - (LUT_BITS + 1));
        s_base2[tid] = b2; s_ok2[tid] = !a.no_lut2 && b2 <= 65536u && 65536u - b2 <= (uint32_t)LUT2;
        T.base2[tid] = b2; T.ok2[tid] = s_ok2[tid];
        // level-1 entries of the prefixes that belong to longer codes (canonical: the LARGEST prefixes, base2 is a multiple of 128)
        for (uint32_t p = b2 >> (16 - LUT_BITS); p < (1u << LUT_BITS); p++)
            T.luts.lut1[tid][p] = s_ok2[tid] ? (uint16_t)(0x8000u | (p - (b2 >> (16 - LUT_BITS)))) : (uint16_t)0x4000u;
    }
    __syncthreads();
    // thread j of a slot: the j-th symbol -- its code from the counts, then its LUT entries; threads 0..16 also the per-length tables
    for (int slot = 0; slot < 8; slot++) {
        const unsigned char* cnt = s_ht + slot * 272;
        if (tid <= 16) {
            if (tid >= 1) {
                int code = 0, kk = 0;
                for (int l = 1; l < tid; l++) { code = (code + cnt[l - 1]) << 1; kk += cnt[l - 1]; }
                T.valoff[slot][tid] = kk - code;
                T.maxcode[slot][tid] = cnt[tid - 1] ? code + cnt[tid - 1] - 1 : -1;
            } else { T.valoff[slot][0] = 0; T.maxcode[slot][0] = -1; }
        }
        int code = 0, kk = 0, l = 1;
        for (; l <= 16; l++) {                               // which length does symbol index tid have?
            if (tid < kk + cnt[l - 1]) break;
            code = (code + cnt[l - 1]) << 1; kk += cnt[l - 1];
        }
        if (l > 16) continue;
        const int c = code + (tid - kk);
        const uint16_t e = (uint16_t)((l << 8) | cnt[16 + tid]);
        if (l <= LUT_BITS) {
            const int first = c << (LUT_BITS - l), nfill = 1 << (LUT_BITS - l);
            if (first + nfill <= (1 << LUT_BITS))
                for (int f = 0; f < nfill; f++) T.luts.lut1[slot][first + f] = e;
        } else if (s_ok2[slot]) {
            const uint32_t first = ((uint32_t)c << (16 - l)) - s_base2[slot], nfill = 1u << (16 - l);
            if (first + nfill <= (uint32_t)LUT2)
                for (uint32_t f = 0; f < nfill; f++) T.luts.lut2[slot][first + f] = e;
        }
    }
}

// exclusive scan of one int per thread over a workgroup of NT threads (wave shuffles + one LDS hop); *total = the sum
__device__ __forceinline__ int block_exclusive_scan(int v, int* s_wave /* [NT / 64 + 1] */, int* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < NT / 64; i++) { const int t = s_wave[i]; s_wave[i] = run; run += t; }
        s_wave[NT / 64] = run;
    }
    __syncthreads();
    const int r = s_wave[wv] + inc - v;
    *total = s_wave[NT / 64];
    __syncthreads();
    return r;
}

// ---- per image: FF 00 -> FF over the whole scan (restart markers between the intervals stay where they fall), the unstuffed extent of
// every restart interval.  Tiles of NT x 16 bytes: aligned 16-byte loads, kept bytes compacted in LDS, coalesced byte stores; kp[chunk] =
// bytes kept in front of each 16-byte chunk (for the interval extents).
__global__ __launch_bounds__(NT) void jpeg_unstuff_kernel(EntropyArgs a) {
    __shared__ unsigned char s_out[NT * 16];
    __shared__ int s_wave[NT / 64 + 1];
    const int tid = threadIdx.x;
    const int32_t* d = a.desc + (size_t)blockIdx.x * AB_JPEG_DESC_INTS;
    const uint32_t off0 = (uint32_t)d[D_OFF], len = (uint32_t)d[D_LEN], A = off0 & ~15u, hi = off0 + len;
    const unsigned char* src = a.data;
    unsigned char* dst = a.clean + off0;
    int32_t* kp = a.kp + (A >> 4);
    uint32_t running = 0;
    for (uint32_t t0 = A; t0 < hi; t0 += NT * 16) {
        const uint32_t i = t0 + tid * 16;
        uint32_t keep = 0; uint4 q = make_uint4(0, 0, 0, 0);
        if (i < hi) {
            q = *reinterpret_cast<const uint4*>(src + i);
            unsigned prev = i > off0 ? src[i - 1] : 0u;
            const uint32_t wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const unsigned c = (wds[j >> 2] >> (8 * (j & 3))) & 255u;
                const bool in = i + j >= off0 && i + j < hi;
                keep |= (in && !(c == 0 && prev == 0xFF)) ? 1u << j : 0u;
                prev = c;
            }
        }
        int total;
        const int ex = block_exclusive_scan(__popc(keep), s_wave, &total);
        if (i < hi) {
            kp[(i - A) >> 4] = (int32_t)(running + ex);
            const uint32_t wds[4] = {q.x, q.y, q.z, q.w};
            int o = ex;
#pragma unroll
            for (int j = 0; j < 16; j++) if (keep >> j & 1u) s_out[o++] = (unsigned char)((wds[j >> 2] >> (8 * (j & 3))) & 255u);
        }
        __syncthreads();
        for (int o = tid; o < total; o += NT) dst[running + o] = s_out[o];
        running += (uint32_t)total;
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    // bytes kept in front of scan offset x (relative to off0)
    auto kept_before = [&](uint32_t x) {
        const uint32_t ab = off0 + x;
        if (ab >= hi) return running;
        const uint32_t cs = ab & ~15u;
        uint32_t r = (uint32_t)kp[(cs - A) >> 4];
        unsigned pv = cs > off0 ? src[cs - 1] : 0u;
        for (uint32_t i = cs; i < ab; i++) { const unsigned c = src[i]; r += (i >= off0 && !(c == 0 && pv == 0xFF)) ? 1u : 0u; pv = c; }
        return r;
    };
    const int32_t* segs = a.segs + (size_t)d[D_SEG_OFF] * 4;
    int32_t* segc = a.segc + (size_t)d[D_SEG_OFF] * 2;
    for (int sgi = tid; sgi < d[D_NSEG]; sgi += NT) {
        const uint32_t x0 = (uint32_t)segs[sgi * 4], x1 = x0 + (uint32_t)segs[sgi * 4 + 1];
        const uint32_t c0 = kept_before(x0), c1 = kept_before(x1);
        segc[sgi * 2] = (int32_t)(off0 + c0); segc[sgi * 2 + 1] = (int32_t)(c1 - c0);
    }
}

struct ImageCtx {
    const int32_t *d, *segs, *segc; int ncomp, bpm, nseg, nsub; uint32_t SB;
    uint32_t *sP, *sS, *cP, *cS; int32_t* sN;
};
__device__ __forceinline__ ImageCtx image_ctx(const EntropyArgs& a, int img) {
    ImageCtx c;
    c.d = a.desc + (size_t)img * AB_JPEG_DESC_INTS;
    c.segs = a.segs + (size_t)c.d[D_SEG_OFF] * 4; c.segc = a.segc + (size_t)c.d[D_SEG_OFF] * 2;
    c.ncomp = c.d[D_NCOMP]; c.bpm = c.d[D_BPM]; c.nseg = c.d[D_NSEG]; c.nsub = c.d[D_NSUB]; c.SB = (uint32_t)a.sub_bytes;
    const int sb = c.d[D_SUB_BASE];
    c.sP = a.sP + sb; c.sS = a.sS + sb; c.sN = a.sN + sb; c.cP = a.cP + sb; c.cS = a.cS + sb;
    return c;
}
__device__ __forceinline__ void load_lds(Lds& L, const EntropyArgs& a, const int32_t* d, int tid, int nthreads) {
    const uint4* src = reinterpret_cast<const uint4*>(&a.tables[d[D_HT]].luts);
    uint4* dst = reinterpret_cast<uint4*>(&L.T);
    for (int i = tid; i < (int)(sizeof(HuffLuts) / 16); i += nthreads) dst[i] = src[i];
    if (tid < 64) L.zz[tid] = ZZ[tid];
    __syncthreads();
}
// bit range [start, end) of subsequence v of segment s in the unstuffed scan, and the end of the segment; false: v is empty
struct SubRange { uint32_t startbits, endbits, seg_end; };
__device__ __forceinline__ bool sub_range(const ImageCtx& c, int s, int v, SubRange& r) {
    const uint32_t c0 = (uint32_t)c.segc[s * 2], c1 = c0 + (uint32_t)c.segc[s * 2 + 1];
    const uint32_t j = (uint32_t)(v - c.segs[s * 4 + 2]);
    const uint32_t st = c0 + j * c.SB;
    r.startbits = st * 8u; r.endbits = min(c1, st + c.SB) * 8u; r.seg_end = c1;
    return st < c1 || j == 0;
}
// decode (without writing) from (P, b, k) to the end of a subsequence; returns the blocks completed
template <bool STAGED>
__device__ __forceinline__ int run_sub(const uint32_t* words, uint32_t* col, const Lds& L, const HuffTables* G, const McuTables mt, const SubRange& sr, int bpm, uint32_t& P, int& b, int& k) {
    Reader<STAGED> r;
    if (STAGED) { stage(col, words, P, sr.endbits); r.seek(col, sr.seg_end, P); }
    else r.seek(words, sr.seg_end, P);
    int nb = 0;
    while (r.position() < sr.endbits) nb += symbol<false>(r, L, G, mt, b, k, bpm, nullptr);
    P = r.position();
    return nb;
}

// ---- round 0 (rnd == 0): every subsequence from its first bit in state (block 0 of the MCU, k = 0); rounds rnd >= 1: the chain that
// started at u continues through subsequence u + rnd until it meets the recorded state there.  grid (ceil(max nsub / NW), images).
template <bool STAGED>
__global__ __launch_bounds__(NW) void jpeg_sync_kernel(EntropyArgs a, int rnd) {
    __shared__ Lds L;
    __shared__ uint32_t s_stage[STAGED ? STAGE_WORDS * NW : 1];
    const int img = blockIdx.y, tid = threadIdx.x;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(a.clean);
    uint32_t* col = s_stage + (STAGED ? tid : 0);
    const ImageCtx c = image_ctx(a, img);
    if ((int)(blockIdx.x * NW) >= c.nsub) return;
    int32_t* act = a.act + (size_t)img * (R_MAX + 2);
    if (rnd > 0 && !act[rnd - 1]) return;                    // every chain of this image has synchronised already
    load_lds(L, a, c.d, tid, NW);
    const McuTables mt = mcu_tables(c.d);
    const int u = blockIdx.x * NW + tid;
    if (u >= c.nsub) return;
    const int s = seg_of_sub(c.segs, c.nseg, u);
    SubRange sr;
    if (rnd == 0) {
        const bool live = sub_range(c, s, u, sr);
        uint32_t P = sr.startbits; int b = 0, k = 0, nb = 0;
        if (live) nb = run_sub<STAGED>(words, col, L, a.tables + c.d[D_HT], mt, sr, c.bpm, P, b, k);
        const uint32_t S = (uint32_t)b | ((uint32_t)k << 8);
        c.sP[u] = P; c.sS[u] = S; c.sN[u] = nb;
        c.cP[u] = live ? P : 0xFFFFFFFFu; c.cS[u] = S;
        if (live) act[0] = 1;
        return;
    }
    uint32_t P = c.cP[u];
    if (P == 0xFFFFFFFFu) return;
    const int v = u + rnd;
    const int last = (s + 1 < c.nseg ? c.segs[(s + 1) * 4 + 2] : c.nsub) - 1;
    if (v > last || !sub_range(c, s, v, sr)) { c.cP[u] = 0xFFFFFFFFu; return; }
    int b = (int)(c.cS[u] & 255u), k = (int)(c.cS[u] >> 8);
    const int nb = run_sub<STAGED>(words, col, L, a.tables + c.d[D_HT], mt, sr, c.bpm, P, b, k);
    const uint32_t S = (uint32_t)b | ((uint32_t)k << 8);
    // the chains reach v in order of decreasing start index, the last one (matching or not) carries the true state into v: its count of
    // blocks completed inside v is the one that stays
    c.sN[v] = nb;
    if (c.sP[v] == P && c.sS[v] == S) c.cP[u] = 0xFFFFFFFFu;                        // synchronised: from here on it is v's chain
    else { c.sP[v] = P; c.sS[v] = S; c.cP[u] = P; c.cS[u] = S; act[rnd] = 1; }
}

// ---- per image: whatever R_MAX rounds left unsynchronised (normally nothing) is finished inside one workgroup; then the exclusive scan
// of the block counts over the image's subsequences
__global__ __launch_bounds__(NT) void jpeg_finish_kernel(EntropyArgs a) {
    __shared__ Lds L;
    __shared__ int s_scan[NT];
    __shared__ int s_carry;
    const int img = blockIdx.x, tid = threadIdx.x;
    const ImageCtx c = image_ctx(a, img);
    int32_t* act = a.act + (size_t)img * (R_MAX + 2);
    if (act[R_MAX]) {
        load_lds(L, a, c.d, tid, NT);
        const McuTables mt = mcu_tables(c.d);
        for (int rnd = R_MAX + 1;; rnd++) {
            int active = 0;
            for (int u = tid; u < c.nsub; u += NT) {
                uint32_t P = c.cP[u];
                if (P == 0xFFFFFFFFu) continue;
                const int v = u + rnd;
                const int s = seg_of_sub(c.segs, c.nseg, u);
                const int last = (s + 1 < c.nseg ? c.segs[(s + 1) * 4 + 2] : c.nsub) - 1;
                SubRange sr;
                if (v > last || !sub_range(c, s, v, sr)) { c.cP[u] = 0xFFFFFFFFu; continue; }
                int b = (int)(c.cS[u] & 255u), k = (int)(c.cS[u] >> 8);
                const int nb = run_sub<false>(reinterpret_cast<const uint32_t*>(a.clean), nullptr, L, a.tables + c.d[D_HT], mt, sr, c.bpm, P, b, k);
                const uint32_t S = (uint32_t)b | ((uint32_t)k << 8);
                c.sN[v] = nb;
                if (c.sP[v] == P && c.sS[v] == S) c.cP[u] = 0xFFFFFFFFu;
                else { c.sP[v] = P; c.sS[v] = S; c.cP[u] = P; c.cS[u] = S; active = 1; }
            }
            if (!__syncthreads_or(active)) { if (tid == 0) act[R_MAX + 1] = rnd; break; }     // diagnostic: rounds this image took
        }
    } else if (tid == 0) {
        int r = 0;
        while (r < R_MAX && act[r]) r++;
        act[R_MAX + 1] = r;
    }
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int u0 = 0; u0 < c.nsub; u0 += NT) {
        const int u = u0 + tid;
        const int v = u < c.nsub ? c.sN[u] : 0;
        s_scan[tid] = v;
        __syncthreads();
        for (int o = 1; o < NT; o <<= 1) {
            const int t = tid >= o ? s_scan[tid - o] : 0;
            __syncthreads();
            s_scan[tid] += t;
            __syncthreads();
        }
        const int carry = s_carry;
        if (u < c.nsub) c.sN[u] = carry + s_scan[tid] - v;
        __syncthreads();
        if (tid == NT - 1) s_carry = carry + s_scan[tid];
        __syncthreads();
    }
}

// ---- coefficient pass: every subsequence from its true start state, writing at its true block position.  grid as jpeg_sync_kernel
template <bool STAGED>
__global__ __launch_bounds__(NW) void jpeg_coef_kernel(EntropyArgs a) {
    __shared__ Lds L;
    __shared__ uint32_t s_stage[STAGED ? STAGE_WORDS * NW : 1];
    const int img = blockIdx.y, tid = threadIdx.x;
    const ImageCtx c = image_ctx(a, img);
    if ((int)(blockIdx.x * NW) >= c.nsub) return;
    load_lds(L, a, c.d, tid, NW);
    const int u = blockIdx.x * NW + tid;
    if (u >= c.nsub) return;
    const int s = seg_of_sub(c.segs, c.nseg, u);
    SubRange sr;
    if (!sub_range(c, s, u, sr)) return;
    const int jf = c.segs[s * 4 + 2];
    const int seg_first_blk = c.segs[s * 4 + 3];
    const int seg_nblk = (s + 1 < c.nseg ? c.segs[(s + 1) * 4 + 3] : c.d[D_NBLK]) - seg_first_blk;
    uint32_t P = sr.startbits; int b = 0, k = 0;
    if (u != jf) { P = c.sP[u - 1]; b = (int)(c.sS[u - 1] & 255u); k = (int)(c.sS[u - 1] >> 8); }
    int nb = c.sN[u] - c.sN[jf];                             // blocks of this segment completed before this subsequence
    short* coefs = a.coefs + ((size_t)c.d[D_BLK_BASE] + seg_first_blk) * 64;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(a.clean);
    Reader<STAGED> r;
    if (STAGED) { stage(s_stage + tid, words, P, sr.endbits); r.seek(s_stage + tid, sr.seg_end, P); }
    else r.seek(words, sr.seg_end, P);
    const McuTables mt = mcu_tables(c.d);
    while (r.position() < sr.endbits && nb < seg_nblk) nb += symbol<true>(r, L, a.tables + c.d[D_HT], mt, b, k, c.bpm, coefs + (size_t)nb * 64);
}

// ---- DC prediction: running sum of the differences per component, in scan order, reset at restart intervals.  One workgroup per image
__global__ __launch_bounds__(NT) void jpeg_dc_kernel(EntropyArgs a) {
    __shared__ int s_scan[NT];
    __shared__ int s_flag[NT];
    const int tid = threadIdx.x;
    const int32_t* d = a.desc + (size_t)blockIdx.x * AB_JPEG_DESC_INTS;
    const int ncomp = d[D_NCOMP], bpm = d[D_BPM], ri = d[D_RI];
    short* coefs = a.coefs + (size_t)d[D_BLK_BASE] * 64;
    const int nmcu = d[D_MCUX] * d[D_MCUY];
    int off = 0;
    for (int c = 0; c < ncomp; c++) {
        const int nbc = d[D_COMP0 + 5 * c] * d[D_COMP0 + 5 * c + 1];
        const int count = nmcu * nbc, Lq = (count + NT - 1) / NT;
        const int q0 = min(tid * Lq, count), q1 = min(q0 + Lq, count);
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

// grid (ceil(max width / 256), ceil(max height / 4), images), block (64, 4): four pixels of one row per thread.  The host parser only lets
// through luma at full resolution over 1 x 1 chroma (4:4:4 / 4:2:2 / 4:2:0) and grey.
__global__ __launch_bounds__(256) void jpeg_color_kernel(const int32_t* desc, const unsigned char* planes, unsigned char* out, int channels) {
    const int32_t* d = desc + (size_t)blockIdx.z * AB_JPEG_DESC_INTS;
    const int W = d[D_W], H = d[D_H];
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= W || y >= H) return;
    const int ncomp = d[D_NCOMP], hmax = d[D_HMAX], vmax = d[D_VMAX], mcux = d[D_MCUX], mcuy = d[D_MCUY];
    PlaneRef py;
    py.P = planes + (size_t)d[D_PLANE_BASE]; py.pitch = mcux * hmax * 8; py.dw = W; py.dh = H; py.hr = 1; py.vr = 1;
    int Y[4], Cb[4], Cr[4];
    upsampled4(py, x0, y, Y);
    if (ncomp == 3) {
        PlaneRef pc;
        pc.pitch = mcux * 8; pc.dw = (W + hmax - 1) >> (hmax - 1); pc.dh = (H + vmax - 1) >> (vmax - 1); pc.hr = hmax; pc.vr = vmax;
        pc.P = py.P + (size_t)py.pitch * mcuy * vmax * 8;
        upsampled4(pc, x0, y, Cb);
        pc.P += (size_t)pc.pitch * mcuy * 8;
        upsampled4(pc, x0, y, Cr);
    }
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

extern "C" long ab_jpeg_workspace_bytes(long total_blocks, long total_subseq, long plane_bytes, long data_bytes, long total_segs, int n, int n_tables) {
    return (long)(align256((size_t)total_blocks * 128) + 5 * align256((size_t)total_subseq * 4) + align256((size_t)plane_bytes) +
                  align256((size_t)data_bytes + 64) + align256((size_t)total_segs * 8) + align256((size_t)n * (R_MAX + 2) * 4) +
                  align256((size_t)n_tables * sizeof(HuffTables)) + align256(((size_t)data_bytes / 16 + 4) * 4));
}

extern "C" int ab_jpeg_decode_batch(const void* data, const int32_t* desc, const int32_t* segs, const void* qtabs, const void* htabs, int n,
                                    int n_tables, int sub_bytes, long total_blocks, long total_subseq, long plane_bytes, long data_bytes,
                                    long total_segs, int max_blocks, int max_width, int max_height, int max_subseq, int out_channels, void* out,
                                    void* workspace, void* stream) {
    if (n <= 0) return 0;
    if (sub_bytes < 16 || sub_bytes > STAGE_MAX_SUB || (out_channels != 3 && out_channels != 4) || !data || !desc || !segs || !qtabs || !htabs || !out || !workspace ||
        n_tables <= 0 || n > 65535 || max_subseq <= 0 || max_width <= 0 || max_height <= 0 || max_height > 4 * 65535) return -1;
    static const int no_lut2 = getenv("AB_JPEG_NO_LUT2") ? atoi(getenv("AB_JPEG_NO_LUT2")) : 0;    // test hook: canonical search for every long code
    hipStream_t st = as_stream(stream);
    char* w = (char*)workspace;
    short* coefs = (short*)w; w += align256((size_t)total_blocks * 128);
    uint32_t* arr[5];
    for (int i = 0; i < 5; i++) { arr[i] = (uint32_t*)w; w += align256((size_t)total_subseq * 4); }
    unsigned char* planes = (unsigned char*)w; w += align256((size_t)plane_bytes);
    unsigned char* clean = (unsigned char*)w; w += align256((size_t)data_bytes + 64);
    int32_t* segc = (int32_t*)w; w += align256((size_t)total_segs * 8);
    int32_t* act = (int32_t*)w; w += align256((size_t)n * (R_MAX + 2) * 4);
    HuffTables* tables = (HuffTables*)w; w += align256((size_t)n_tables * sizeof(HuffTables));
    int32_t* kp = (int32_t*)w;
    hipError_t e = hipMemsetAsync(coefs, 0, (size_t)total_blocks * 128, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(act, 0, (size_t)n * (R_MAX + 2) * 4, st);
    if (e != hipSuccess) return (int)e;
    EntropyArgs a{(const unsigned char*)data, clean, desc, segs, segc, kp, (const unsigned char*)htabs, tables, coefs,
                  arr[0], arr[1], (int32_t*)arr[2], arr[3], arr[4], act, sub_bytes, n_tables, no_lut2};
    hipLaunchKernelGGL(jpeg_tables_kernel, dim3(n_tables), dim3(256), 0, st, a);
    hipLaunchKernelGGL(jpeg_unstuff_kernel, dim3(n), dim3(NT), 0, st, a);
    const dim3 gw((max_subseq + NW - 1) / NW, n);
    for (int rnd = 0; rnd <= R_MAX; rnd++) hipLaunchKernelGGL(jpeg_sync_kernel<true>, gw, dim3(NW), 0, st, a, rnd);
    hipLaunchKernelGGL(jpeg_finish_kernel, dim3(n), dim3(NT), 0, st, a);
    hipLaunchKernelGGL(jpeg_coef_kernel<true>, gw, dim3(NW), 0, st, a);
    hipLaunchKernelGGL(jpeg_dc_kernel, dim3(n), dim3(NT), 0, st, a);
    AB_LAUNCH_CHECK();
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((max_blocks + 31) / 32, n), dim3(256), 0, st, desc, coefs, (const unsigned short*)qtabs, planes);
    AB_LAUNCH_CHECK();
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((max_width + 255) / 256, (max_height + 3) / 4, n), dim3(64, 4), 0, st, desc, planes, (unsigned char*)out, out_channels);
    AB_LAUNCH_CHECK();
    return 0;
}
