// Online synthesis on the training GPU: triangle setup -> tiled z-buffer rasteriser in LDS -> deferred shading +
// background composite -> PIL-semantics colour jitter -> nearest-neighbour affine crop, writing the network input
// (zero-bordered NHWC4) directly.  Replaces the reference's per-image pipeline
//   anakin/utils/renderer.py:101-136 (pyrender/OpenGL draw + np.putmask background), anakin/artiboost/render_infra.py
//   (two multiprocessing queue hops per image) and rendered_dataset.py:256-270 + utils/img_augment.py (PIL on CPU workers)
// with batched kernels: one launch renders all B images.
//
// Integer rules (vertex snapping 1/256 px, int64 edge functions, top-left rule, 24-bit depth, key = depth<<32|face)
// are DEFINED in oracle/render_oracle.c (the reference's rasteriser is the GL driver: parity unpinned); this file
// must match that oracle bit for bit.  Float work uses the same operation order and no FMA contraction.
//
// Rasteriser: the setup kernel snaps / culls every triangle and bins it into the 32x32-pixel tiles its bounding box touches
// (per-workgroup LDS counts, one global reservation per touched tile); one workgroup per tile then walks its own list:
// the tile's z-buffer is 1024 u64 keys in LDS, each lane takes one triangle at a time (the meshes are 1-20 px per face
// at this camera distance, so lane-per-triangle keeps the lanes busy), steps the edge functions and the depth numerator
// incrementally and resolves visibility with ds_min_u64; tiles with an empty list go straight to the background.  The
// HBM side is small (a 48-byte record per face, 4 bytes per pixel out).
#include "common.h"
#include <atomic>

#define HAND_FACES 1538
#define HAND_VERTS 778
#define NEAR_INV 20.0f
#define FAR_INV 0.01f
#define ZMAX 16777215.0f
// pyrender's mesh.frag: glTF metallic-roughness BRDF.  Material factors: the only ones the reference states
// (frender_utils.py:153-157: metallicFactor 0.2, roughnessFactor 0.8); light colour 0.9 (artiboost_loader.py:194);
// ambient 0.8 (renderer.py:76).  The point light sits at the camera, so l = v = h and the Fresnel term is its r0.
#define PBR_METALLIC 0.2f
#define PBR_ROUGHNESS 0.8f
#define PBR_F0 0.04f
#define PBR_LIGHT_COLOR 0.9f
#define PBR_AMBIENT 0.8f
#define PBR_INV_PI 0.31830987f
#define TILE 32
#define RS_THREADS 256   // raster/shade workgroup: one pixel of the 32x32 tile per thread in the shading pass
#define BIN_CAP 1024      // per-tile triangle list capacity (overflowing tiles fall back to scanning all records)

struct SceneDev {    // mirrors ab_scene (host struct of device pointers)
    const int32_t* hand_faces; const float* hand_normals; const float* hand_uv; const int32_t* hand_map; const uint8_t* hand_tex; int hts;
    const float* obj_verts; const float* obj_normals; const float* obj_uv; const int32_t* obj_faces;
    const int32_t* obj_vert_off; const int32_t* obj_face_off; const uint8_t* obj_tex; int ots;
    const uint8_t* bg; int bgs; const float* srgb2lin; const uint8_t* lin2srgb;
    float fx, fy, cx, cy; int W, H;
};
struct SampleDev {
    int32_t obj_id, hand_tex_id, bg_id, bg_x0, bg_y0, bg_w, bg_h; float light; float obj_pose[16];
};
struct TriRec { int32_t x[3], y[3]; uint32_t z[3]; int32_t valid; int16_t bx0, bx1, by0, by1; };   // 48 bytes

__device__ __forceinline__ int32_t snapf(float x) { return (int32_t)floorf(x * 256.0f + 0.5f); }
__device__ __forceinline__ uint32_t quant_z(float Z) {
    float inv = 1.0f / Z;
    float z01 = (inv - NEAR_INV) / (FAR_INV - NEAR_INV);
    float q = floorf(z01 * ZMAX + 0.5f);
    if (q < 0.f) q = 0.f;
    if (q > ZMAX) q = ZMAX;
    return (uint32_t)q;
}

__device__ __forceinline__ void face_verts(const SceneDev& sc, const SampleDev& sm, const float* hv, int gid,
                                           float P[3][3], int vid[3]) {
    if (gid < HAND_FACES) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int v = sc.hand_faces[gid * 3 + k];
            vid[k] = v;                                         // render vertex: uv / normal index
            const int pv = sc.hand_map ? sc.hand_map[v] : v;    // MANO vertex sharing its position (UV-seam duplicates)
            P[k][0] = hv[pv * 3]; P[k][1] = hv[pv * 3 + 1]; P[k][2] = hv[pv * 3 + 2];
        }
    } else {
        int o = sm.obj_id;
        int f = sc.obj_face_off[o] + (gid - HAND_FACES);
        const float* T = sm.obj_pose;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int v = sc.obj_vert_off[o] + sc.obj_faces[f * 3 + k];
            vid[k] = v;
            const float* p = sc.obj_verts + (size_t)v * 3;
            P[k][0] = (T[0] * p[0] + T[1] * p[1]) + (T[2] * p[2] + T[3]);
            P[k][1] = (T[4] * p[0] + T[5] * p[1]) + (T[6] * p[2] + T[7]);
            P[k][2] = (T[8] * p[0] + T[9] * p[1]) + (T[10] * p[2] + T[11]);
        }
    }
}

__device__ __forceinline__ void setup_tri(const SceneDev& sc, const SampleDev& sm, const float* hv, int gid, TriRec& t) {
    float P[3][3]; int vid[3];
    face_verts(sc, sm, hv, gid, P, vid);
    t.valid = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) if (!(P[k][2] > 0.05f)) return;
    if (gid >= HAND_FACES) {
        float e1[3] = {P[1][0] - P[0][0], P[1][1] - P[0][1], P[1][2] - P[0][2]};
        float e2[3] = {P[2][0] - P[0][0], P[2][1] - P[0][1], P[2][2] - P[0][2]};
        float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        float d = (n[0] * P[0][0] + n[1] * P[0][1]) + n[2] * P[0][2];
        if (!(d < 0.f)) return;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float x = (sc.fx * P[k][0]) / P[k][2] + sc.cx;
        float y = (sc.fy * P[k][1]) / P[k][2] + sc.cy;
        if (!(fabsf(x) < 1.0e6f) || !(fabsf(y) < 1.0e6f)) return;
        t.x[k] = snapf(x); t.y[k] = snapf(y); t.z[k] = quant_z(P[k][2]);
    }
    int64_t area = (int64_t)(t.x[1] - t.x[0]) * (t.y[2] - t.y[0]) - (int64_t)(t.y[1] - t.y[0]) * (t.x[2] - t.x[0]);
    if (area == 0) return;
    if (area < 0) {
        int32_t a = t.x[1]; t.x[1] = t.x[2]; t.x[2] = a;
        a = t.y[1]; t.y[1] = t.y[2]; t.y[2] = a;
        uint32_t b = t.z[1]; t.z[1] = t.z[2]; t.z[2] = b;
        t.valid = 2;
    } else t.valid = 1;
}

__device__ __forceinline__ int64_t edgef(int32_t ax, int32_t ay, int32_t bx, int32_t by, int32_t px, int32_t py) {
    return (int64_t)(bx - ax) * (py - ay) - (int64_t)(by - ay) * (px - ax);
}
__device__ __forceinline__ bool edge_incl(int32_t ax, int32_t ay, int32_t bx, int32_t by) {
    int32_t dx = bx - ax, dy = by - ay;
    return (dy > 0) || (dy == 0 && dx < 0);
}
__device__ __forceinline__ bool cover(const TriRec& t, int32_t px, int32_t py, int64_t w[3]) {
    w[0] = edgef(t.x[1], t.y[1], t.x[2], t.y[2], px, py);
    w[1] = edgef(t.x[2], t.y[2], t.x[0], t.y[0], px, py);
    w[2] = edgef(t.x[0], t.y[0], t.x[1], t.y[1], px, py);
    if (w[0] < 0 || w[1] < 0 || w[2] < 0) return false;
    if (w[0] == 0 && !edge_incl(t.x[1], t.y[1], t.x[2], t.y[2])) return false;
    if (w[1] == 0 && !edge_incl(t.x[2], t.y[2], t.x[0], t.y[0])) return false;
    if (w[2] == 0 && !edge_incl(t.x[0], t.y[0], t.x[1], t.y[1])) return false;
    return true;
}

// ------------------------------------------------------------------ kernel 1: triangle setup
// tri: [B][maxf] records; tails: their (valid, bbox) words; bin_count / bin_list: per-tile triangle lists (zeroed by the launcher)
__global__ __launch_bounds__(256) void raster_setup_kernel(SceneDev sc, const SampleDev* __restrict__ samples,
                                                           const float* __restrict__ hand_verts, int maxf,
                                                           TriRec* __restrict__ tri, int4* __restrict__ tails,
                                                           int* __restrict__ bin_count, int* __restrict__ bin_list) {
    extern __shared__ int s_bin[];                  // [3][ntile]: pairs counted, global base, pairs placed
    const int b = blockIdx.y;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const SampleDev sm = samples[b];
    const int nf = HAND_FACES + (sc.obj_face_off[sm.obj_id + 1] - sc.obj_face_off[sm.obj_id]);
    const int tiles_x = sc.W / TILE, ntile = tiles_x * (sc.H / TILE);
    int* lcount = s_bin; int* lbase = s_bin + ntile; int* lplaced = s_bin + 2 * ntile;
    for (int i = threadIdx.x; i < ntile; i += 256) { lcount[i] = 0; lplaced[i] = 0; }
    __syncthreads();
    TriRec t;
    t.valid = 0;
    int x0 = 0, x1 = -1, y0 = 0, y1 = -1;
    if (gid < nf) setup_tri(sc, sm, hand_verts + (size_t)b * HAND_VERTS * 3, gid, t);
    if (t.valid) {
        int32_t minx = min(t.x[0], min(t.x[1], t.x[2])), maxx = max(t.x[0], max(t.x[1], t.x[2]));
        int32_t miny = min(t.y[0], min(t.y[1], t.y[2])), maxy = max(t.y[0], max(t.y[1], t.y[2]));
        x0 = (minx - 128 + 255) >> 8; x1 = (maxx - 128) >> 8; y0 = (miny - 128 + 255) >> 8; y1 = (maxy - 128) >> 8;
        x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, sc.W - 1); y1 = min(y1, sc.H - 1);
        if (x0 > x1 || y0 > y1) t.valid = 0;
        else { t.bx0 = (int16_t)x0; t.bx1 = (int16_t)x1; t.by0 = (int16_t)y0; t.by1 = (int16_t)y1; }
    }
    if (gid < maxf) {
        tri[(size_t)b * maxf + gid] = t;
        // compact copy of (valid, bbox) = bytes 32..47 of the record: what a tile needs to clip a listed triangle
        tails[(size_t)b * maxf + gid] = make_int4((int)t.z[2], t.valid, (int)((uint16_t)t.bx0 | ((uint32_t)(uint16_t)t.bx1 << 16)),
                                                  (int)((uint16_t)t.by0 | ((uint32_t)(uint16_t)t.by1 << 16)));
    }
    // Bin the triangle into every tile its pixel bbox touches (append order is irrelevant: the z-test is a commutative
    // atomic min on (depth, face id)).  (triangle, tile) pairs are counted in LDS first so that a workgroup makes ONE global
    // reservation per tile it touches instead of one contended atomic per pair.
    if (t.valid)
        for (int ty = y0 / TILE; ty <= y1 / TILE; ++ty)
            for (int tx = x0 / TILE; tx <= x1 / TILE; ++tx) atomicAdd(&lcount[ty * tiles_x + tx], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < ntile; i += 256)
        if (lcount[i] > 0) lbase[i] = atomicAdd(&bin_count[b * ntile + i], lcount[i]);
    __syncthreads();
    if (t.valid)
        for (int ty = y0 / TILE; ty <= y1 / TILE; ++ty)
            for (int tx = x0 / TILE; tx <= x1 / TILE; ++tx) {
                const int tile = ty * tiles_x + tx;
                const int slot = lbase[tile] + atomicAdd(&lplaced[tile], 1);
                if (slot < BIN_CAP) bin_list[((size_t)b * ntile + tile) * BIN_CAP + slot] = gid;
            }
}

// ------------------------------------------------------------------ kernel 1b: dispatch order of the tiles
// One tile in five holds triangles, and such a tile lives ~10x longer than a background-only one (rasterise, barrier, shade: serial
// phases of a few busy waves).  In (sample, tile) launch order a CU holds one of them among four short-lived neighbours and the launch
// lasts (tiles with triangles per CU) x (their life).  Listed first -- all samples' triangle tiles, then all background tiles -- five of
// them share a CU and overlap each other's phases.  Which slot a tile gets depends on the order the samples' workgroups reach the
// cursor; what a workgroup computes for its (sample, tile) does not.  (sample << 16 | tile): <= 65 535 tiles per frame.
__global__ __launch_bounds__(256) void tile_order_kernel(const int* __restrict__ bin_count, int ntile, int* __restrict__ order_act,
                                                         int* __restrict__ order_bg, int* __restrict__ cursor) {
    __shared__ int s_cnt[2], s_base[2];
    const int b = blockIdx.x;
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    int slot[8];                                            // ntile <= 2048
    for (int t = threadIdx.x, k = 0; t < ntile; t += 256, ++k) {
        const int act = bin_count[b * ntile + t] > 0;
        slot[k] = atomicAdd(&s_cnt[act ? 0 : 1], 1);
    }
    __syncthreads();
    if (threadIdx.x < 2) s_base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], s_cnt[threadIdx.x]);
    __syncthreads();
    for (int t = threadIdx.x, k = 0; t < ntile; t += 256, ++k) {
        const int act = bin_count[b * ntile + t] > 0;
        (act ? order_act : order_bg)[s_base[act ? 0 : 1] + slot[k]] = (b << 16) | t;
    }
}

// ------------------------------------------------------------------ kernel 2: tile raster + shade
// Background = the sample's random crop resized to the render size with cv2's INTER_LINEAR fixed-point arithmetic
// (renderer.py:125-136; restated in oracle/render_oracle.c bg_pixel / lin_coef, which this must match bit for bit).
// The two coefficients and the source index of a destination column / row depend only on that column / row: the
// workgroup tabulates them for its 32 columns and 32 rows in LDS (bgc[0..31] columns, bgc[32..63] rows).
struct BgCoef { int s0, s1, a0, a1; };
__device__ __forceinline__ BgCoef bg_coef(int d, int dsize, int ssize) {
    const double scale = 1.0 / ((double)dsize / (double)ssize);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    BgCoef c;
    c.s0 = s; c.s1 = s + 1 < ssize ? s + 1 : ssize - 1;
    c.a0 = (int)rintf((1.f - f) * 2048.f); c.a1 = (int)rintf(f * 2048.f);
    return c;
}
__device__ __forceinline__ void bg_coef_fill(const SceneDev& sc, const SampleDev& sm, int tx0, int ty0, BgCoef* bgc) {
    if (threadIdx.x < TILE) bgc[threadIdx.x] = bg_coef(tx0 + threadIdx.x, sc.W, sm.bg_w);
    else if (threadIdx.x < 2 * TILE) bgc[threadIdx.x] = bg_coef(ty0 + threadIdx.x - TILE, sc.H, sm.bg_h);
}
__device__ __forceinline__ uint32_t bg_texels(const SceneDev& sc, const SampleDev& sm, const BgCoef& cx, const BgCoef& cy) {
    const uint32_t* img = (const uint32_t*)sc.bg + (size_t)sm.bg_id * sc.bgs * sc.bgs;      // RGBX texels
    const uint32_t* r0 = img + (size_t)(sm.bg_y0 + cy.s0) * sc.bgs + sm.bg_x0;
    const uint32_t* r1 = img + (size_t)(sm.bg_y0 + cy.s1) * sc.bgs + sm.bg_x0;
    const uint32_t p00 = r0[cx.s0], p01 = r0[cx.s1], p10 = r1[cx.s0], p11 = r1[cx.s1];
    uint32_t out = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int S0 = (int)((p00 >> (8 * c)) & 255u) * cx.a0 + (int)((p01 >> (8 * c)) & 255u) * cx.a1;
        const int S1 = (int)((p10 >> (8 * c)) & 255u) * cx.a0 + (int)((p11 >> (8 * c)) & 255u) * cx.a1;
        const int v = (((cy.a0 * (S0 >> 4)) >> 16) + ((cy.a1 * (S1 >> 4)) >> 16) + 2) >> 2;
        out |= (uint32_t)(v & 255) << (8 * c);
    }
    return out;
}
__device__ __forceinline__ void bg_sample(const SceneDev& sc, const SampleDev& sm, const BgCoef& cx, const BgCoef& cy, uint8_t o[4]) {
    const uint32_t q = bg_texels(sc, sm, cx, cy);
    o[0] = (uint8_t)q; o[1] = (uint8_t)(q >> 8); o[2] = (uint8_t)(q >> 16); o[3] = 0;
}

// recs: this sample's records as raster_setup_kernel stored them -- the same TriRec setup_tri() would rebuild for the winning face (six
// divisions, snapping, depth quantisation, orientation), read back as one 48-byte load instead
__device__ __forceinline__ void shade_pixel(const SceneDev& sc, const SampleDev& sm, const float* hv, uint64_t key, int x,
                                            int y, const BgCoef* bgc, uint8_t o[4], const TriRec* __restrict__ recs,
                                            const float* s2l, const uint8_t* l2s) {
    if (key == ~(uint64_t)0) { bg_sample(sc, sm, bgc[x & (TILE - 1)], bgc[TILE + (y & (TILE - 1))], o); return; }
    int gid = (int)(uint32_t)key;
    const TriRec t = recs[gid];
    float P[3][3]; int vid[3];
    face_verts(sc, sm, hv, gid, P, vid);
    if (t.valid == 2) {
        int a = vid[1]; vid[1] = vid[2]; vid[2] = a;
#pragma unroll
        for (int i = 0; i < 3; ++i) { float f = P[1][i]; P[1][i] = P[2][i]; P[2][i] = f; }
    }
    int64_t w[3]; cover(t, x * 256 + 128, y * 256 + 128, w);
    float ws = (float)(w[0] + w[1] + w[2]);
    float l0 = (float)w[0] / ws, l1 = (float)w[1] / ws, l2 = (float)w[2] / ws;
    float i0 = 1.0f / P[0][2], i1 = 1.0f / P[1][2], i2 = 1.0f / P[2][2];
    float d = (l0 * i0 + l1 * i1) + l2 * i2;
    float m0 = (l0 * i0) / d, m1 = (l1 * i1) / d, m2 = (l2 * i2) / d;
    float nn[3][3], uu[3][2];
    const uint8_t* tex; int ts;
    if (gid < HAND_FACES) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            nn[k][0] = sc.hand_normals[vid[k] * 3]; nn[k][1] = sc.hand_normals[vid[k] * 3 + 1]; nn[k][2] = sc.hand_normals[vid[k] * 3 + 2];
            uu[k][0] = sc.hand_uv[vid[k] * 2]; uu[k][1] = sc.hand_uv[vid[k] * 2 + 1];
        }
        ts = sc.hts; tex = sc.hand_tex + (size_t)sm.hand_tex_id * ts * ts * 3;
    } else {
        const float* T = sm.obj_pose;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float* n = sc.obj_normals + (size_t)vid[k] * 3;
            nn[k][0] = (T[0] * n[0] + T[1] * n[1]) + T[2] * n[2];
            nn[k][1] = (T[4] * n[0] + T[5] * n[1]) + T[6] * n[2];
            nn[k][2] = (T[8] * n[0] + T[9] * n[1]) + T[10] * n[2];
            uu[k][0] = sc.obj_uv[(size_t)vid[k] * 2]; uu[k][1] = sc.obj_uv[(size_t)vid[k] * 2 + 1];
        }
        ts = sc.ots; tex = sc.obj_tex + (size_t)sm.obj_id * ts * ts * 3;
    }
    float n[3], p[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        n[i] = (m0 * nn[0][i] + m1 * nn[1][i]) + m2 * nn[2][i];
        p[i] = (m0 * P[0][i] + m1 * P[1][i]) + m2 * P[2][i];
    }
    float u = (m0 * uu[0][0] + m1 * uu[1][0]) + m2 * uu[2][0], v = (m0 * uu[0][1] + m1 * uu[1][1]) + m2 * uu[2][1];
    float nl = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]); if (nl < 1e-20f) nl = 1e-20f;
    float d2 = (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2];
    float dl = sqrtf(d2);
    float ndl = -((n[0] * p[0] + n[1] * p[1]) + n[2] * p[2]) / (nl * dl);
    if (gid < HAND_FACES) ndl = fabsf(ndl);
    if (ndl < 0.f) ndl = 0.f;
    // ---- pyrender mesh.frag, one point light at the camera (l = v = h; vh = 1 -> F = r0 = specular colour):
     //   nl = clamp(n.l, 0.001, 1), nv = clamp(|n.v|, 0.001, 1), nh = clamp(n.h, 0, 1)
     //   G = aL * aV, a? = 2 n? / (n? + sqrt(a2 + (1 - a2) n?^2)),  D = a2 / (pi ((nh a2 - nh) nh + 1)^2),  a2 = roughness^4
     //   colour = nl * radiance * ((1 - F) diffuse / pi + F G D / (4 nl nv)) + base * ambient,  radiance = 0.9 I / d^2
     //   diffuse = base (1 - f0)(1 - metallic),  F = mix(f0, base, metallic)                 */
    const float a2 = (PBR_ROUGHNESS * PBR_ROUGHNESS) * (PBR_ROUGHNESS * PBR_ROUGHNESS);
    float nlc = ndl < 0.001f ? 0.001f : (ndl > 1.f ? 1.f : ndl);
    float nhc = ndl > 1.f ? 1.f : ndl;
    float att = (2.0f * nlc) / (nlc + sqrtf(a2 + (1.0f - a2) * (nlc * nlc)));
    float ff = (nhc * a2 - nhc) * nhc + 1.0f;
    float Dm = a2 / (3.14159274f * (ff * ff));
    float gd4 = ((att * att) * Dm) / ((4.0f * nlc) * nlc);
    float rad = (PBR_LIGHT_COLOR * sm.light) / d2;
    float nlrad = nlc * rad;
    u = u - floorf(u); v = v - floorf(v);
    int tx = (int)(u * (float)ts), ty = (int)(v * (float)ts);
    if (tx > ts - 1) tx = ts - 1;
    if (ty > ts - 1) ty = ts - 1;
    const uint8_t* texel = tex + ((size_t)ty * ts + tx) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float base = s2l[texel[c]];
        float F = PBR_F0 * (1.0f - PBR_METALLIC) + PBR_METALLIC * base;
        float dif = ((1.0f - F) * (((1.0f - PBR_F0) * (1.0f - PBR_METALLIC)) * base)) * PBR_INV_PI;
        float lin = base * PBR_AMBIENT + nlrad * (dif + F * gd4);
        if (lin < 0.f) lin = 0.f;
        if (lin > 1.f) lin = 1.f;
        o[c] = l2s[(int)(lin * 4095.0f + 0.5f)];
    }
    o[3] = 255;
}

__global__ __launch_bounds__(RS_THREADS) void raster_shade_kernel(SceneDev sc, const SampleDev* __restrict__ samples,
                                                           const float* __restrict__ hand_verts, int maxf,
                                                           const TriRec* __restrict__ tri, const int4* __restrict__ tails_g,
                                                           const int* __restrict__ bin_count, const int* __restrict__ bin_list,
                                                           uint8_t* __restrict__ rgbx, uint64_t* __restrict__ keys_out,
                                                           const int* __restrict__ order_act, const int* __restrict__ order_bg,
                                                           const int* __restrict__ cursor) {
    __shared__ unsigned long long zb[TILE * TILE];
    __shared__ BgCoef bgc[2 * TILE];
    __shared__ float l_s2l[256];                            // the two sRGB tables of the shading, copied per tile with triangles: the last two
    __shared__ uint32_t l_l2s[1024];                        // of a shaded pixel's five dependent look-ups stay inside the CU
    // workgroup -> (sample, tile) through tile_order_kernel's lists: every tile that holds triangles, of every sample, is dispatched
    // before the background-only ones (see there)
    const int tiles_x = sc.W / TILE;
    const int wg = blockIdx.x, nact = cursor[0];
    const int packed = wg < nact ? order_act[wg] : order_bg[wg - nact];
    const int b = packed >> 16, tile_id = packed & 0xffff;
    const int ty0 = (tile_id / tiles_x) * TILE, tx0 = (tile_id % tiles_x) * TILE;
    const SampleDev sm = samples[b];
    const float* hv = hand_verts + (size_t)b * HAND_VERTS * 3;
    for (int i = threadIdx.x; i < TILE * TILE; i += RS_THREADS) zb[i] = ~0ull;
    bg_coef_fill(sc, sm, tx0, ty0, bgc);
    __syncthreads();
    const int ntile = tiles_x * (sc.H / TILE);
    const int nbin = bin_count[b * ntile + tile_id];
    const bool active = nbin > 0;
    if (active) {
        l_s2l[threadIdx.x & 255] = sc.srgb2lin[threadIdx.x & 255];
        for (int i = threadIdx.x; i < 1024; i += RS_THREADS) l_l2s[i] = ((const uint32_t*)sc.lin2srgb)[i];
        const int nf = HAND_FACES + (sc.obj_face_off[sm.obj_id + 1] - sc.obj_face_off[sm.obj_id]);
        const TriRec* tb = tri + (size_t)b * maxf;
        const int4* tl = tails_g + (size_t)b * maxf;
        auto raster_tri = [&](const int gid, const int4 tail) {
            const int valid = tail.y;
            if (!valid) return;
            const int16_t qx0 = (int16_t)(tail.z & 0xffff), qx1 = (int16_t)((uint32_t)tail.z >> 16);
            const int16_t qy0 = (int16_t)(tail.w & 0xffff), qy1 = (int16_t)((uint32_t)tail.w >> 16);
            int x0 = max((int)qx0, tx0), x1 = min((int)qx1, tx0 + TILE - 1);
            int y0 = max((int)qy0, ty0), y1 = min((int)qy1, ty0 + TILE - 1);
            if (x0 > x1 || y0 > y1) return;
            const TriRec t = tb[gid];
            // Same integers as cover()/the oracle, evaluated incrementally: the edge functions are affine in the pixel
            // index, so one 64-bit add per edge per pixel replaces two 64-bit multiplies; the top-left rule is folded in
            // as a bias of 1 on the non-inclusive edges (w >= 0 and (w > 0 or inclusive)  <=>  w - bias >= 0); the depth
            // numerator sum(w_i z_i) steps the same way.  The denominator w0+w1+w2 is the triangle's doubled area.
            const int32_t ex[3] = {t.x[2] - t.x[1], t.x[0] - t.x[2], t.x[1] - t.x[0]};     // b - a per edge (1,2) (2,0) (0,1)
            const int32_t ey[3] = {t.y[2] - t.y[1], t.y[0] - t.y[2], t.y[1] - t.y[0]};
            const int32_t axv[3] = {t.x[1], t.x[2], t.x[0]}, ayv[3] = {t.y[1], t.y[2], t.y[0]};
            const int32_t px0 = x0 * 256 + 128, py0 = y0 * 256 + 128;
            int64_t wrow[3], sx[3], sy[3], nrow = 0, nsx = 0, nsy = 0, sum = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int64_t w = (int64_t)ex[k] * (py0 - ayv[k]) - (int64_t)ey[k] * (px0 - axv[k]);
                sx[k] = -(int64_t)ey[k] * 256; sy[k] = (int64_t)ex[k] * 256;
                sum += w;
                nrow += w * (int64_t)t.z[k]; nsx += sx[k] * (int64_t)t.z[k]; nsy += sy[k] * (int64_t)t.z[k];
                const bool incl = (ey[k] > 0) || (ey[k] == 0 && ex[k] < 0);
                wrow[k] = w - (incl ? 0 : 1);
            }
            const double inv = 1.0 / (double)sum;
            for (int y = y0; y <= y1; ++y) {
                int64_t w0 = wrow[0], w1 = wrow[1], w2 = wrow[2], num = nrow;
                for (int x = x0; x <= x1; ++x) {
                    if ((w0 | w1 | w2) >= 0) {
                        // exact floor(num / sum): double estimate (error < 1) + integer fix-up
                        int64_t q = (int64_t)((double)num * inv);
                        int64_t r = num - q * sum;
                        if (r < 0) --q; else if (r >= sum) ++q;
                        unsigned long long key = ((unsigned long long)q << 32) | (uint32_t)gid;
                        atomicMin(&zb[(y - ty0) * TILE + (x - tx0)], key);
                    }
                    w0 += sx[0]; w1 += sx[1]; w2 += sx[2]; num += nsx;
                }
                wrow[0] += sy[0]; wrow[1] += sy[1]; wrow[2] += sy[2]; nrow += nsy;
            }
        };
        if (nbin <= BIN_CAP) {
            // the tile's own list (typically ~100 triangles: one round of the lanes)
            const int* lst = bin_list + ((size_t)b * ntile + tile_id) * BIN_CAP;
            for (int i = threadIdx.x; i < nbin; i += RS_THREADS) { const int gid = lst[i]; raster_tri(gid, tl[gid]); }
        } else {
            // overflowed list: scan every record; four tails per lane are fetched before any is examined
            for (int gbase = threadIdx.x; gbase < nf; gbase += 4 * RS_THREADS) {
                int4 tails[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int gq = gbase + u * RS_THREADS; tails[u] = gq < nf ? tl[gq] : make_int4(0, 0, 0, 0); }
#pragma unroll
                for (int u = 0; u < 4; ++u) raster_tri(gbase + u * RS_THREADS, tails[u]);
            }
        }
    }
    __syncthreads();
    if (!active) {
        // background-only tile (4 of 5 at the benchmark geometry): all four pixels' texel loads are issued before any store
        constexpr int NP = TILE * TILE / RS_THREADS;
        uint32_t px[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = threadIdx.x + j * RS_THREADS;
            px[j] = bg_texels(sc, sm, bgc[i % TILE], bgc[TILE + i / TILE]);
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = threadIdx.x + j * RS_THREADS;
            const size_t pix = ((size_t)b * sc.H + (ty0 + i / TILE)) * sc.W + (tx0 + i % TILE);
            *(uint32_t*)(rgbx + pix * 4) = px[j];
            if (keys_out) keys_out[pix] = ~0ull;
        }
        return;
    }
    for (int i = threadIdx.x; i < TILE * TILE; i += RS_THREADS) {
        int y = ty0 + i / TILE, x = tx0 + i % TILE;
        uint64_t key = zb[i];
        uint8_t o[4];
        shade_pixel(sc, sm, hv, key, x, y, bgc, o, tri + (size_t)b * maxf, l_s2l, (const uint8_t*)l_l2s);
        size_t pix = ((size_t)b * sc.H + y) * sc.W + x;
        *(uint32_t*)(rgbx + pix * 4) = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
        if (keys_out) keys_out[pix] = key;
    }
}

// ------------------------------------------------------------------ PIL-semantics colour jitter (see the oracle)
// Branch-free forms (round 4): the kernels that run these per pixel are VALU-issue bound (jitter_stats: 44 M wave instructions per
// 64 x 512^2 launch, + 20 M scalar ones for the exec-mask bookkeeping of the divergent branches the first versions had).  Every
// function below returns, for every input, what its branching predecessor returned: checked over all 2^24 RGB triples against the
// oracle (tests/test_gpu_render.py) and, for the hue op, over all 2^24 triples x all 256 hue shifts (tests/hue_exhaustive.py).
__device__ __forceinline__ uint8_t blend8(int in1, int in2, float f) {
    // ImageEnhance: in1 + f (in2 - in1), truncated; clipped to 0..255 when f is outside [0, 1] -- inside it the value lies between
    // the two inputs, so clipping always is the same function.
    const float t = (float)in1 + f * (float)(in2 - in1);
    return (uint8_t)min((unsigned)fmaxf(t, 0.f), 255u);
}
__device__ __forceinline__ int luma8(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }
// Per-byte terms, tabulated once per workgroup with the very expressions PIL's C code evaluates per pixel:
//   hsv -> rgb: sector i = floor(h*6/255), fraction f = h*6/255 - i (as float), fs = s/255 (as float)
//   rgb -> hsv: rcp[d] = 1.0 / d as a double.  For integers 0 <= a <= 255, 1 <= d <= 255 the float quotient (float)a / (float)d
//   equals (float)((double)a * rcp[d]): a/d is never within 2^-33 (relative) of a float rounding boundary, the double product is
//   within 2^-52 of it.  Four IEEE float divisions per pixel become two 8-byte LDS reads and four double multiplies.
struct HueLut { const uint8_t* sect; const float* frac; const float* sat; const double* rcp; };
// The tables are computed once per process by hue_tab_init_kernel (four double divisions per entry: as a per-workgroup prologue they were
// a quarter of jitter_stats' time at 64 workgroups per sample) and copied to LDS by every workgroup.
struct HueTab { double rcp[256]; float frac[256], sat[256]; uint8_t sect[256]; };
static __device__ HueTab g_hue_tab;
__global__ void hue_tab_init_kernel() {
    const int v = threadIdx.x;
    const int i = (int)floor((double)(float)v * 6.0 / 255.0);
    g_hue_tab.sect[v] = (uint8_t)i;
    g_hue_tab.frac[v] = (float)((double)(float)v * 6.0 / 255.0 - (double)(float)i);
    g_hue_tab.sat[v] = (float)((double)(float)v / 255.0);
    g_hue_tab.rcp[v] = v ? 1.0 / (double)v : 0.0;
}
static void hue_tab_ready(hipStream_t st) {
    // First use: fill the tables on `st` and wait, so that launches on OTHER streams (the real-frame augmentation runs beside the render)
    // never read them half-written.  Inside a stream capture nothing may wait: the fill becomes a node in front of its reader and the
    // next un-captured call still does the one-time fill.
    // g_hue_tab is a PER-DEVICE symbol: the flag is kept per device (a process that renders on cuda:1 after cuda:0 -- tests, tools -- would
    // otherwise read an all-zero table there), and atomically (two host threads may race to the first call; a double fill is harmless).
    static std::atomic<bool> done[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (done[dev].load(std::memory_order_acquire)) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cs);
    hue_tab_init_kernel<<<1, 256, 0, st>>>();
    if (cs == hipStreamCaptureStatusNone) { (void)hipStreamSynchronize(st); done[dev].store(true, std::memory_order_release); }
}
__device__ __forceinline__ void hue_lut_fill(uint8_t* sect, float* frac, float* sat, double* rcp) {
    for (int v = threadIdx.x; v < 256; v += blockDim.x) {
        sect[v] = g_hue_tab.sect[v]; frac[v] = g_hue_tab.frac[v]; sat[v] = g_hue_tab.sat[v]; rcp[v] = g_hue_tab.rcp[v];
    }
}
#define HUE_LUT_DECL __shared__ uint8_t l_sect[256]; __shared__ float l_frac[256], l_sat[256]; __shared__ double l_rcp[256]; \
    hue_lut_fill(l_sect, l_frac, l_sat, l_rcp); __syncthreads(); const HueLut lut = {l_sect, l_frac, l_sat, l_rcp}

__device__ __forceinline__ void rgb2hsv8(const HueLut& lut, const uint8_t* in, uint8_t* out) {
    const int r = in[0], g = in[1], b = in[2];
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    const int d = maxc - minc;                              // gray pixels (d = 0): rcp[0] = 0, every term below is finite, h = s = 0 selected at the end
    const double rd = lut.rcp[d], rm = lut.rcp[maxc];
    const float s = (float)((double)d * rm);
    // PIL: rc = (max - r) / d .. as floats; h = bc - gc | 2 + rc - bc | 4 + gc - rc for the first of r, g, b that equals the maximum.  Only the
    // two terms of the taken case are formed; the float difference bc - gc equals (float)(0.0 + (double)bc - (double)gc) (the double difference
    // of two floats in [0, 1] is exact), so one double expression serves all three cases.
    const bool rmax = r == maxc, gmax = g == maxc;
    const int nx = maxc - (rmax ? b : gmax ? r : g), ny = maxc - (rmax ? g : gmax ? b : r);
    const double base = rmax ? 0.0 : gmax ? 2.0 : 4.0;
    const float xf = (float)((double)nx * rd), yf = (float)((double)ny * rd);
    float h = (float)(base + (double)xf - (double)yf);
    {   // (double)h / 6.0, correctly rounded: reciprocal estimate + one residual step (exact for every h this function produces: checked
        // exhaustively), then the fmod-free wrap: hv6 is in [5/6, 11/6]
        const double hd = (double)h, r6 = 1.0 / 6.0;
        double q = hd * r6;
        q = fma(fma(-6.0, q, hd), r6, q);
        const double hv6 = q + 1.0;
        h = (float)(hv6 - floor(hv6));
    }
    const int ih = (int)((double)h * 255.0), is = (int)((double)s * 255.0);
    out[0] = d ? (uint8_t)min(max(ih, 0), 255) : 0;
    out[1] = d ? (uint8_t)min(max(is, 0), 255) : 0;
    out[2] = (uint8_t)maxc;
}
// round() for x >= 0 (half away from zero), exact: x - trunc(x) is exact for |x| < 2^52
__device__ __forceinline__ int round_pos(double x) { const double t = trunc(x); return (int)(x - t >= 0.5 ? t + 1.0 : t); }

__device__ __forceinline__ void hsv2rgb8(const HueLut& lut, const uint8_t* in, uint8_t* out) {
    const uint8_t h = in[0], s = in[1], v = in[2];
    const int i = lut.sect[h];                              // 0 .. 6 (h = 255 -> 6 -> sector 0 below, as i % 6)
    const double f = (double)lut.frac[h], fs = (double)lut.sat[s], vd = (double)(float)v;
    const int p = min(round_pos(vd * (1.0 - fs)), 255);
    // odd sectors use q = v (1 - fs f), even ones t = v (1 - fs (1 - f)), never both: one product, named for both
    const int u = min(round_pos(vd * (1.0 - fs * ((i & 1) ? f : 1.0 - f))), 255);
    const int q = u, t = u;
    // sector:  0: v t p   1: q v p   2: p v t   3: p q v   4: t p v   5: v p q      (s = 0: v v v)
    const bool z = s == 0;
    const int o0 = (i == 0 || i >= 5) ? v : (i == 1) ? q : (i == 4) ? t : p;
    const int o1 = (i == 1 || i == 2) ? v : (i == 0 || i == 6) ? t : (i == 3) ? q : p;
    const int o2 = (i == 3 || i == 4) ? v : (i == 2) ? t : (i == 5) ? q : p;
    out[0] = (uint8_t)(z ? v : o0); out[1] = (uint8_t)(z ? v : o1); out[2] = (uint8_t)(z ? v : o2);
}
__device__ __forceinline__ void jitter_op(const HueLut& lut, int op, float f, int mean_gray, uint8_t* px) {
    if (op == 2) { uint8_t hsv[3]; rgb2hsv8(lut, px, hsv); hsv[0] = (uint8_t)(hsv[0] + (uint8_t)(int)(f * 255.0f)); hsv2rgb8(lut, hsv, px); return; }
    // brightness (towards 0), saturation (towards the pixel's luma), contrast (towards the image's mean luma): one blend, the op picks the target
    const int target = op == 0 ? 0 : op == 1 ? luma8(px[0], px[1], px[2]) : mean_gray;
#pragma unroll
    for (int c = 0; c < 3; ++c) px[c] = blend8(target, px[c], f);
}

// ------------------------------------------------------------------ PIL GaussianBlur (rendered_dataset.py:257-258)
// ImageFilter.GaussianBlur(radius) = libImaging/BoxBlur.c: three box-blur passes of fractional radius
// _gaussian_blur_radius(radius, 3) along x, then three along y; for a box radius < 1 (Gaussian radius < 1.41; the
// reference draws radius <= 0.1) one pass is  out = (c * ww + (left + right) * fw + 2^23) >> 24  with the neighbours
// clamped to the line.  Restated (and pinned against the real Pillow) in oracle/render_oracle.c ro_gaussian_blur.
// With ww = 2^24 - 2 fw - e (e = 0 | 1) a pass is the identity whenever 510 fw + 255 <= 2^23, i.e. for radius < ~0.077
// (3 of 4 of the reference's draws): such samples are not touched and the later stages read the unblurred image.
struct BlurPar { uint32_t ww, fw; int on; };
__device__ __forceinline__ BlurPar blur_params(float radius) {
    const float sigma2 = radius * radius / 3.0f;
    const float L = (float)sqrt(12.0 * (double)sigma2 + 1.0);
    const float l = (float)floor(((double)L - 1.0) / 2.0);
    float a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2);
    a /= 6 * (sigma2 - (l + 1) * (l + 1));
    const float fr = l + a;
    BlurPar p; p.ww = 1u << 24; p.fw = 0; p.on = 0;
    if (!(fr > 0.f)) return p;
    p.ww = (uint32_t)((float)(1 << 24) / (fr * 2 + 1));
    p.fw = ((1u << 24) - p.ww) / 2;
    p.on = 510u * p.fw + 255u > (1u << 23);
    return p;
}
__device__ __forceinline__ uint32_t box3(uint32_t l, uint32_t c, uint32_t r, uint32_t ww, uint32_t fw) {
    uint32_t o = c & 0xff000000u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t cc = (c >> (8 * k)) & 255u, s = ((l >> (8 * k)) & 255u) + ((r >> (8 * k)) & 255u);
        o |= ((__umul24(cc, ww) + __umul24(s, fw) + (1u << 23)) >> 24) << (8 * k);     // ww < 2^24 whenever the pass is not the identity
    }
    return o;
}
#define BL_HALO 3
#define BL_P (TILE + 2 * BL_HALO)
// One workgroup per 32x32 tile: the tile + a 3-pixel halo goes to LDS, the three horizontal passes shrink the valid
// columns by one each, the three vertical passes the valid rows; positions outside the image are never read (the
// neighbour index is clamped at the image border exactly as PIL clamps it at the line ends).
__global__ __launch_bounds__(256) void gauss_blur_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int W, int H,
                                                         const float* __restrict__ radius, int copy_identity) {
    __shared__ uint32_t buf[2][BL_P * BL_P];
    const int b = blockIdx.y;
    const int tiles_x = W / TILE;
    const int ty0 = (blockIdx.x / tiles_x) * TILE, tx0 = (blockIdx.x % tiles_x) * TILE;
    const BlurPar bp = blur_params(radius[b]);
    const uint32_t* src = (const uint32_t*)in + (size_t)b * W * H;
    uint32_t* dst = (uint32_t*)out + (size_t)b * W * H;
    if (!bp.on) {
        if (copy_identity)
            for (int i = threadIdx.x; i < TILE * TILE; i += 256) {
                const size_t o = (size_t)(ty0 + i / TILE) * W + tx0 + i % TILE;
                dst[o] = src[o];
            }
        return;
    }
    for (int i = threadIdx.x; i < BL_P * BL_P; i += 256) {
        const int r = i / BL_P, c = i - r * BL_P;
        const int gy = min(max(ty0 - BL_HALO + r, 0), H - 1), gx = min(max(tx0 - BL_HALO + c, 0), W - 1);
        buf[0][i] = src[(size_t)gy * W + gx];
    }
    __syncthreads();
    int cur = 0;
#pragma unroll
    for (int p = 1; p <= 3; ++p) {
        const int ncol = BL_P - 2 * p;
        for (int i = threadIdx.x; i < BL_P * ncol; i += 256) {
            const int r = i / ncol, c = p + (i - r * ncol);
            const int x = tx0 - BL_HALO + c;
            const int cl = x == 0 ? c : c - 1, cr = x == W - 1 ? c : c + 1;
            const uint32_t* row = buf[cur] + r * BL_P;
            buf[cur ^ 1][r * BL_P + c] = box3(row[cl], row[c], row[cr], bp.ww, bp.fw);
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int p = 1; p <= 3; ++p) {
        const int nrow = BL_P - 2 * p;
        for (int i = threadIdx.x; i < nrow * TILE; i += 256) {
            const int r = p + i / TILE, c = BL_HALO + i % TILE;
            const int y = ty0 - BL_HALO + r;
            const int ru = y == 0 ? r : r - 1, rd = y == H - 1 ? r : r + 1;
            const uint32_t v = box3(buf[cur][ru * BL_P + c], buf[cur][r * BL_P + c], buf[cur][rd * BL_P + c], bp.ww, bp.fw);
            if (p == 3) dst[(size_t)y * W + tx0 + c - BL_HALO] = v;
            else buf[cur ^ 1][r * BL_P + c] = v;
        }
        __syncthreads();
        cur ^= 1;
    }
}

// kernel 3: luma sum of the image as it is when the contrast op is reached (ops before it applied on the fly)
#define LSUM_STRIDE 16     // u64 slots between two samples' sums in the render / augment workspaces: device-scope atomics serialise per 128-byte line
#define JS_WGS 64          // workgroups per sample (16 / 32 / 64 / 128 / 256 measure 93 / 78 / 66 / 66 / 72 us once the sums below no longer share lines)
__global__ __launch_bounds__(256) void jitter_stats_kernel(const uint8_t* __restrict__ rgbx_plain, int npix,
                                                           const int32_t* __restrict__ order, const float* __restrict__ factor,
                                                           unsigned long long* __restrict__ lsum,
                                                           const uint8_t* __restrict__ rgbx_blur = nullptr,
                                                           const float* __restrict__ radius = nullptr, int ls = 1) {
    HUE_LUT_DECL;
    const int b = blockIdx.y;
    const uint8_t* rgbx = (radius && blur_params(radius[b]).on) ? rgbx_blur : rgbx_plain;     // blurred copy exists only where the blur acts
    const int32_t* ord = order + b * 4; const float* fac = factor + b * 4;
    int kc = 0;
    while (kc < 4 && ord[kc] != 3) ++kc;
    unsigned long long s = 0;
    if (kc < 4) {
        // four pixels (one 16-byte load) per lane and iteration; the tail (npix % 4) one by one
        const uint8_t* img = rgbx + (size_t)b * npix * 4;
        const int nq = npix >> 2;
        const int stride = gridDim.x * 256;
        int i = blockIdx.x * 256 + threadIdx.x;
        uint4 nxt = make_uint4(0u, 0u, 0u, 0u);
        if (i < nq) nxt = *(const uint4*)(img + (size_t)i * 16);
        for (; i < nq; i += stride) {
            const uint4 q4 = nxt;                               // the next iteration's pixels are requested before this one's ~600 instructions
            if (i + stride < nq) nxt = *(const uint4*)(img + (size_t)(i + stride) * 16);
            const uint32_t qq[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint8_t px[3] = {(uint8_t)qq[j], (uint8_t)(qq[j] >> 8), (uint8_t)(qq[j] >> 16)};
                for (int k = 0; k < kc; ++k) jitter_op(lut, ord[k], fac[k], 0, px);
                s += (unsigned long long)luma8(px[0], px[1], px[2]);
            }
        }
        for (int i = nq * 4 + blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
            uint32_t q = *(const uint32_t*)(img + (size_t)i * 4);
            uint8_t px[3] = {(uint8_t)q, (uint8_t)(q >> 8), (uint8_t)(q >> 16)};
            for (int k = 0; k < kc; ++k) jitter_op(lut, ord[k], fac[k], 0, px);
            s += (unsigned long long)luma8(px[0], px[1], px[2]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    // ONE atomic per workgroup: the B sums sit 16 to a 128-byte line and device-scope atomics on a line serialise at ~21 ns each -- with one
    // per wave (4 096 per line at 64 workgroups per sample) the launch was as long as its atomics (86 of 93 us)
    __shared__ unsigned long long wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        if (t) atomicAdd(&lsum[(size_t)b * ls], t);                             // integer: order-independent, exact
    }
}

// kernel 4: nearest-neighbour affine crop + full jitter chain + normalise; writes zero-bordered NHWC4 and/or CHW f32
// U8N (T = bf16): the padded tensor receives the odd integers 2 v - 255 instead of v / 255 - 0.5 (AB_DT_U8N: the stem's exact image plane)
template <typename T, bool U8N = false>
__global__ __launch_bounds__(256) void warp_jitter_kernel(const uint8_t* __restrict__ rgbx_plain, int W, int H,
                                                          const int32_t* __restrict__ order, const float* __restrict__ factor,
                                                          const float* __restrict__ inv_affine,
                                                          const unsigned long long* __restrict__ lsum, int ow, int oh,
                                                          T* __restrict__ out_pad, float* __restrict__ out_chw,
                                                          const uint8_t* __restrict__ rgbx_blur, const float* __restrict__ radius,
                                                          const uint8_t* __restrict__ flip = nullptr, int ls = 1) {
    HUE_LUT_DECL;
    const int b = blockIdx.y;
    const uint8_t* rgbx = (radius && blur_params(radius[b]).on) ? rgbx_blur : rgbx_plain;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ow * oh) return;
    const int y = i / ow, x = i - y * ow;
    const float* inv = inv_affine + b * 6;
    float xin = (inv[0] * ((float)x + 0.5f) + inv[1] * ((float)y + 0.5f)) + inv[2];
    float yin = (inv[3] * ((float)x + 0.5f) + inv[4] * ((float)y + 0.5f)) + inv[5];
    int sx = (int)floorf(xin), sy = (int)floorf(yin);
    float v[3] = {0.f, 0.f, 0.f};
    if (sx >= 0 && sx < W && sy >= 0 && sy < H) {
        // Image.FLIP_LEFT_RIGHT before the blur / jitter / warp (hodata.py:336-337) == the same chain on the unflipped
        // image read at the mirrored column: the blur is symmetric and the jitter is per pixel (+ a global mean)
        if (flip && flip[b]) sx = W - 1 - sx;
        uint32_t q = *(const uint32_t*)(rgbx + (((size_t)b * H + sy) * W + sx) * 4);
        uint8_t px[3] = {(uint8_t)q, (uint8_t)(q >> 8), (uint8_t)(q >> 16)};
        int mean = (int)((double)lsum[(size_t)b * ls] / (double)(W * H) + 0.5);
        for (int k = 0; k < 4; ++k) jitter_op(lut, order[b * 4 + k], factor[b * 4 + k], mean, px);
        v[0] = (float)px[0]; v[1] = (float)px[1]; v[2] = (float)px[2];
    }
    float o[3] = {v[0] / 255.0f - 0.5f, v[1] / 255.0f - 0.5f, v[2] / 255.0f - 0.5f};
    if (out_pad) {
        T* p = out_pad + (((size_t)b * (oh + 6) + (y + 3)) * (ow + 8) + (x + 3)) * 4;
        if constexpr (U8N) { st_f32(p, 2.f * v[0] - 255.f); st_f32(p + 1, 2.f * v[1] - 255.f); st_f32(p + 2, 2.f * v[2] - 255.f); st_f32(p + 3, 0.f); }
        else { st_f32(p, o[0]); st_f32(p + 1, o[1]); st_f32(p + 2, o[2]); st_f32(p + 3, 0.f); }
    }
    if (out_chw) {
        size_t plane = (size_t)oh * ow;
        float* p = out_chw + (size_t)b * 3 * plane + (size_t)y * ow + x;
        p[0] = o[0]; p[plane] = o[1]; p[2 * plane] = o[2];
    }
}

// ================================================================ C ABI
static SceneDev to_dev(const ab_scene* s) {
    SceneDev d;
    d.hand_faces = (const int32_t*)s->hand_faces; d.hand_normals = (const float*)s->hand_normals; d.hand_uv = (const float*)s->hand_uv;
    d.hand_map = (const int32_t*)s->hand_map;
    d.hand_tex = (const uint8_t*)s->hand_tex; d.hts = s->hts; d.obj_verts = (const float*)s->obj_verts;
    d.obj_normals = (const float*)s->obj_normals; d.obj_uv = (const float*)s->obj_uv; d.obj_faces = (const int32_t*)s->obj_faces;
    d.obj_vert_off = (const int32_t*)s->obj_vert_off; d.obj_face_off = (const int32_t*)s->obj_face_off;
    d.obj_tex = (const uint8_t*)s->obj_tex; d.ots = s->ots; d.bg = (const uint8_t*)s->bg; d.bgs = s->bgs;
    d.srgb2lin = (const float*)s->srgb2lin; d.lin2srgb = (const uint8_t*)s->lin2srgb;
    d.fx = s->fx; d.fy = s->fy; d.cx = s->cx; d.cy = s->cy; d.W = s->W; d.H = s->H;
    return d;
}

// the full 4-op chain on an RGBX image, in place semantics of the oracle's ro_color_jitter (X byte -> 255)
__global__ __launch_bounds__(256) void jitter_apply_kernel(const uint8_t* __restrict__ rgbx, int npix,
                                                           const int32_t* __restrict__ order, const float* __restrict__ factor,
                                                           const unsigned long long* __restrict__ lsum, uint8_t* __restrict__ out) {
    HUE_LUT_DECL;
    const int b = blockIdx.y;
    const int mean = (int)((double)lsum[b] / (double)npix + 0.5);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        const size_t o = ((size_t)b * npix + i) * 4;
        uint32_t q = *(const uint32_t*)(rgbx + o);
        uint8_t px[3] = {(uint8_t)q, (uint8_t)(q >> 8), (uint8_t)(q >> 16)};
        for (int k = 0; k < 4; ++k) jitter_op(lut, order[b * 4 + k], factor[b * 4 + k], mean, px);
        *(uint32_t*)(out + o) = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16) | 0xff000000u;
    }
}

__global__ void zero_words_kernel(unsigned* __restrict__ p, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}

extern "C" long ab_render_workspace_bytes(int B, int W, int H, int max_faces) {
    // tri records + rgbx + lsum + sbox + compact (valid, bbox) tails + per-tile bin counts and lists, each 256-byte aligned
    auto al = [](long x) { return (x + 255) / 256 * 256; };
    return al((long)B * max_faces * 48) + al((long)B * W * H * 4) + al((long)B * 8 * LSUM_STRIDE) + al((long)B * 16) + al((long)B * max_faces * 16) +
           al((long)B * (W / TILE) * (H / TILE) * 4) + al((long)B * (W / TILE) * (H / TILE) * BIN_CAP * 4) +
           al((long)B * W * H * 4) +       // the blurred copy of the samples the GaussianBlur acts on
           2 * al((long)B * (W / TILE) * (H / TILE) * 4);      // dispatch order of the tiles (with triangles | background only)
}

extern "C" int ab_render_batch(const ab_scene* scene_host, const void* samples, const float* hand_verts,
                               const int32_t* order, const float* factor, const float* inv_affine,
                               const float* blur_radius, int B, int max_faces, int ow, int oh, int out_dtype,
                               void* out_pad, float* out_chw, void* workspace, void* keys_out, void* rgbx_out,
                               void* stream) {
    if (!scene_host || !samples || !hand_verts || !order || !factor || !inv_affine || !workspace) return AB_EINVAL;
    if (!out_pad && !out_chw) return AB_EINVAL;
    SceneDev sc = to_dev(scene_host);
    if (sc.W % TILE || sc.H % TILE || B < 1 || max_faces < HAND_FACES) return AB_ESHAPE;
    if ((long)(sc.W / TILE) * (sc.H / TILE) * 12 > 64 * 1024) return AB_ESHAPE;      // per-tile LDS counters of the setup kernel
    hipStream_t st = as_stream(stream);
    auto al = [](long x) { return (x + 255) / 256 * 256; };
    char* ws = (char*)workspace;
    TriRec* tri = (TriRec*)ws; ws += al((long)B * max_faces * 48);
    uint8_t* rgbx = rgbx_out ? (uint8_t*)rgbx_out : (uint8_t*)ws; ws += al((long)B * sc.W * sc.H * 4);
    unsigned long long* lsum = (unsigned long long*)ws; ws += al((long)B * 8 * LSUM_STRIDE);      // one 128-byte line per sample
    ws += al((long)B * 16);                                   // (reserved)
    int4* tails = (int4*)ws; ws += al((long)B * max_faces * 16);
    const long ntile = (long)(sc.W / TILE) * (sc.H / TILE);
    int* bin_count = (int*)ws; ws += al(B * ntile * 4);
    int* bin_list = (int*)ws; ws += al(B * ntile * BIN_CAP * 4);
    uint8_t* rgbx_blur = (uint8_t*)ws; ws += al((long)B * sc.W * sc.H * 4);
    int* order_act = (int*)ws; ws += al(B * ntile * 4);
    int* order_bg = (int*)ws;
    int* cursor = (int*)(lsum + (size_t)B * LSUM_STRIDE);     // the two words behind the luma sums (the reserved slot), zeroed with them
    if (ntile > 2048 || B > 32767) return AB_ESHAPE;
    // zeroing by kernel, not hipMemsetAsync: under stream capture the 64 KiB memset node of bin_count faulted on replay
    // (ROCm 7.2), so neither buffer goes through a memset node
    zero_words_kernel<<<(unsigned)((B * 2 * LSUM_STRIDE + 2 + 255) / 256), 256, 0, st>>>((unsigned*)lsum, (long)B * 2 * LSUM_STRIDE + 2);
    zero_words_kernel<<<(unsigned)((B * ntile + 255) / 256), 256, 0, st>>>((unsigned*)bin_count, B * ntile);
    AB_LAUNCH_CHECK();
    raster_setup_kernel<<<dim3((max_faces + 255) / 256, B), 256, (size_t)ntile * 12, st>>>(sc, (const SampleDev*)samples, hand_verts,
                                                                           max_faces, tri, tails, bin_count, bin_list);
    AB_LAUNCH_CHECK();
    tile_order_kernel<<<B, 256, 0, st>>>(bin_count, (int)ntile, order_act, order_bg, cursor);
    AB_LAUNCH_CHECK();
    raster_shade_kernel<<<(unsigned)(ntile * B), RS_THREADS, 0, st>>>(sc, (const SampleDev*)samples, hand_verts, max_faces, tri, tails,
                                                                     bin_count, bin_list, rgbx, (uint64_t*)keys_out, order_act, order_bg, cursor);
    AB_LAUNCH_CHECK();
    int npix = sc.W * sc.H;
    if (blur_radius) {
        gauss_blur_kernel<<<dim3((unsigned)ntile, B), 256, 0, st>>>(rgbx, rgbx_blur, sc.W, sc.H, blur_radius, 0);
        AB_LAUNCH_CHECK();
    }
    hue_tab_ready(st);
    jitter_stats_kernel<<<dim3(JS_WGS, B), 256, 0, st>>>(rgbx, npix, order, factor, lsum, rgbx_blur, blur_radius, LSUM_STRIDE);
    AB_LAUNCH_CHECK();
    dim3 g((ow * oh + 255) / 256, B);
    if (out_dtype == AB_DT_F32)
        warp_jitter_kernel<float><<<g, 256, 0, st>>>(rgbx, sc.W, sc.H, order, factor, inv_affine, lsum, ow, oh, (float*)out_pad, out_chw, rgbx_blur, blur_radius, nullptr, LSUM_STRIDE);
    else if (out_dtype == AB_DT_BF16)
        warp_jitter_kernel<bf16_t><<<g, 256, 0, st>>>(rgbx, sc.W, sc.H, order, factor, inv_affine, lsum, ow, oh, (bf16_t*)out_pad, out_chw, rgbx_blur, blur_radius, nullptr, LSUM_STRIDE);
    else if (out_dtype == AB_DT_U8N)
        warp_jitter_kernel<bf16_t, true><<<g, 256, 0, st>>>(rgbx, sc.W, sc.H, order, factor, inv_affine, lsum, ow, oh, (bf16_t*)out_pad, out_chw, rgbx_blur, blur_radius, nullptr, LSUM_STRIDE);
    else return AB_EINVAL;
    AB_LAUNCH_CHECK();
    return 0;
}

// Colour jitter alone on B RGBX images of npix pixels (the chain ab_render_batch fuses into its crop): out may alias rgbx.
// lsum_ws: B x 8 bytes of device scratch.
extern "C" int ab_color_jitter(const void* rgbx, int B, int npix, const int32_t* order, const float* factor, void* out,
                               void* lsum_ws, void* stream) {
    if (!rgbx || !order || !factor || !out || !lsum_ws || B < 1 || npix < 1) return AB_EINVAL;
    hipStream_t st = as_stream(stream);
    zero_words_kernel<<<(unsigned)((B * 2 + 255) / 256), 256, 0, st>>>((unsigned*)lsum_ws, (long)B * 2);
    hue_tab_ready(st);
    jitter_stats_kernel<<<dim3(JS_WGS, B), 256, 0, st>>>((const uint8_t*)rgbx, npix, order, factor, (unsigned long long*)lsum_ws);
    AB_LAUNCH_CHECK();
    jitter_apply_kernel<<<dim3(256, B), 256, 0, st>>>((const uint8_t*)rgbx, npix, order, factor,
                                                       (const unsigned long long*)lsum_ws, (uint8_t*)out);
    AB_LAUNCH_CHECK();
    return 0;
}

// PIL GaussianBlur alone on B RGBX images (radius float [B] on the device, each < 1.41; X byte copied); out must not alias rgbx.
extern "C" int ab_gaussian_blur(const void* rgbx, int B, int W, int H, const float* radius, void* out, void* stream) {
    if (!rgbx || !radius || !out || rgbx == out || B < 1) return AB_EINVAL;
    if (W < TILE || H < TILE || W % TILE || H % TILE) return AB_ESHAPE;
    gauss_blur_kernel<<<dim3((unsigned)((W / TILE) * (H / TILE)), B), 256, 0, as_stream(stream)>>>((const uint8_t*)rgbx, (uint8_t*)out, W, H, radius, 1);
    AB_LAUNCH_CHECK();
    return 0;
}

// The augmentation chain of a real frame (anakin/datasets/hodata.py:336-337,435-446: optional left-right flip, PIL
// GaussianBlur, colour jitter in drawn order, inverse-affine nearest crop, to_tensor - 0.5) for B decoded RGBX images of
// one size: the same kernels as the tail of ab_render_batch.  workspace: ab_augment_workspace_bytes(B, W, H).
extern "C" long ab_augment_workspace_bytes(int B, int W, int H) {
    auto al = [](long x) { return (x + 255) / 256 * 256; };
    return al((long)B * 8 * LSUM_STRIDE) + al((long)B * W * H * 4);
}
extern "C" int ab_augment_batch(const void* rgbx_in, int B, int W, int H, const int32_t* order, const float* factor,
                                const float* inv_affine, const float* blur_radius, const uint8_t* flip, int ow, int oh,
                                int out_dtype, void* out_pad, float* out_chw, void* workspace, void* stream) {
    if (!rgbx_in || !order || !factor || !inv_affine || !workspace || B < 1 || (!out_pad && !out_chw)) return AB_EINVAL;
    if (blur_radius && (W < TILE || H < TILE || W % TILE || H % TILE)) return AB_ESHAPE;
    hipStream_t st = as_stream(stream);
    auto al = [](long x) { return (x + 255) / 256 * 256; };
    const uint8_t* rgbx = (const uint8_t*)rgbx_in;
    unsigned long long* lsum = (unsigned long long*)workspace;
    uint8_t* rgbx_blur = (uint8_t*)workspace + al((long)B * 8 * LSUM_STRIDE);
    zero_words_kernel<<<(unsigned)((B * 2 * LSUM_STRIDE + 255) / 256), 256, 0, st>>>((unsigned*)lsum, (long)B * 2 * LSUM_STRIDE);
    if (blur_radius) gauss_blur_kernel<<<dim3((unsigned)((W / TILE) * (H / TILE)), B), 256, 0, st>>>(rgbx, rgbx_blur, W, H, blur_radius, 0);
    hue_tab_ready(st);
    jitter_stats_kernel<<<dim3(JS_WGS, B), 256, 0, st>>>(rgbx, W * H, order, factor, lsum, rgbx_blur, blur_radius, LSUM_STRIDE);
    AB_LAUNCH_CHECK();
    dim3 g((ow * oh + 255) / 256, B);
    if (out_dtype == AB_DT_F32)
        warp_jitter_kernel<float><<<g, 256, 0, st>>>(rgbx, W, H, order, factor, inv_affine, lsum, ow, oh, (float*)out_pad, out_chw, rgbx_blur, blur_radius, flip, LSUM_STRIDE);
    else if (out_dtype == AB_DT_BF16)
        warp_jitter_kernel<bf16_t><<<g, 256, 0, st>>>(rgbx, W, H, order, factor, inv_affine, lsum, ow, oh, (bf16_t*)out_pad, out_chw, rgbx_blur, blur_radius, flip, LSUM_STRIDE);
    else if (out_dtype == AB_DT_U8N)
        warp_jitter_kernel<bf16_t, true><<<g, 256, 0, st>>>(rgbx, W, H, order, factor, inv_affine, lsum, ow, oh, (bf16_t*)out_pad, out_chw, rgbx_blur, blur_radius, flip, LSUM_STRIDE);
    else return AB_EINVAL;
    AB_LAUNCH_CHECK();
    return 0;
}
