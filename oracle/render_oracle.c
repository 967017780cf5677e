/* TEST INFRASTRUCTURE ONLY -- CPU oracle of the synthesis (render) half of the hot path: triangle setup, z-buffer
 * rasterisation, deferred shading, background composite, PIL-semantics colour jitter and the nearest-neighbour affine
 * crop.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY UNPINNED against the reference's pixels: the reference rasterises through pyrender==0.1.43 -> PyOpenGL -> EGL ->
 * the vendor GL driver (requirements.txt:112,114; call site anakin/utils/renderer.py:110), which is neither in
 * /root/reference nor runnable here, and the reference holds no golden image.  This file therefore DEFINES the integer
 * rules ("bit-exact" means HIP == this file):
 *   - camera: OpenCV pinhole, K = (fx, fy, cx, cy), points already in the camera frame (renderer.py:57-99 with
 *     PYRENDER_EXTRINSIC, utils/misc.py:87-95); pixel centres at (x+0.5, y+0.5)
 *   - vertices snapped to 1/256 px: xi = floor(x*256 + 0.5) (int32); edge functions in int64; a sample on an edge
 *     (dx,dy) belongs to the triangle iff dy > 0 or (dy == 0 and dx < 0) once the triangle is oriented to positive area
 *   - depth: per-vertex z01 = (1/Z - 1/near)/(1/far - 1/near), near 0.05, far 100 (pyrender IntrinsicsCamera defaults),
 *     quantised to 24 bits; per-pixel depth = floor((w0*z0 + w1*z1 + w2*z2)/(w0+w1+w2)) in int64; GL_LESS, ties -> lower
 *     global face id; key = depth << 32 | face id
 *   - object faces are back-face culled with the camera-space geometric normal; the hand is double sided; triangles
 *     with a vertex at Z <= near are dropped
 *   - shading: nearest texel, sRGB->linear by LUT, then pyrender's published fragment shader (mesh.frag: glTF
 *     metallic-roughness BRDF, see ro_shade) with ambient 0.8 and one point light at the camera as in
 *     renderer.py:72-84,103-104; rest-pose ("stale") hand normals as in anakin/utils/frender_utils.py:36-46,139;
 *     linear->sRGB by 4096-entry LUT.  No MSAA resolve: pyrender's offscreen target is 4x multisampled and cleared to the scene's
 *     bg_color 0.5 (renderer.py:76), and the reference keeps the resolved colour wherever the resolved DEPTH is non-zero
 *     (renderer.py:110-119: np.putmask(color, depth == 0, background)) -- so its silhouette pixels are a blend of shaded
 *     samples and grey where the depth sample was covered and pure background where it was not.  Which sample a
 *     multisample depth blit keeps is implementation-defined in GL, so that fringe cannot be restated from a published
 *     rule; this oracle takes every pixel from its centre sample (interior pixels are unaffected)
 *   - background where no geometry (renderer.py:117-119,125-136): the random crop resized with cv2's INTER_LINEAR
 *     fixed-point arithmetic (restated at bg_pixel; cv2 is absent, so unpinned as well)
 * The GaussianBlur + colour jitter + crop stage restates PIL (anakin/utils/img_augment.py:6-80,
 * rendered_dataset.py:256-270) and IS pinned against the real Pillow in tests/test_render_oracle.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NEAR_INV 20.0f      /* 1 / 0.05 */
#define FAR_INV 0.01f       /* 1 / 100  */
#define ZMAX 16777215.0f
/* pyrender's mesh.frag: glTF metallic-roughness BRDF.  Material factors: the only ones the reference states
 * (frender_utils.py:153-157: metallicFactor 0.2, roughnessFactor 0.8); light colour 0.9 (artiboost_loader.py:194);
 * ambient 0.8 (renderer.py:76).  The point light sits at the camera, so l = v = h and the Fresnel term is its r0. */
#define PBR_METALLIC 0.2f
#define PBR_ROUGHNESS 0.8f
#define PBR_F0 0.04f
#define PBR_LIGHT_COLOR 0.9f
#define PBR_AMBIENT 0.8f
#define PBR_INV_PI 0.31830987f
#define HAND_FACES 1538
#define HAND_VERTS 778

typedef struct {
    const float* hand_verts;   /* [B,778,3] camera frame */
    const int32_t* hand_faces; /* [1538,3] indices into the V_dup render vertices */
    const float* hand_normals; /* [V_dup,3] rest pose */
    const float* hand_uv;      /* [V_dup,2] */
    const int32_t* hand_map;   /* [V_dup] render vertex -> MANO vertex (UV-seam duplicates; renderer.py:17-28,107) or NULL */
    const uint8_t* hand_tex;   /* [nht, hts, hts, 3] */
    int hts;
    const float* obj_verts;    /* [Vtot,3] */
    const float* obj_normals;  /* [Vtot,3] */
    const float* obj_uv;       /* [Vtot,2] */
    const int32_t* obj_faces;  /* [Ftot,3] object-local vertex ids */
    const int32_t* obj_vert_off; /* [nobj+1] */
    const int32_t* obj_face_off; /* [nobj+1] */
    const uint8_t* obj_tex;    /* [nobj, ots, ots, 3] */
    int ots;
    const uint8_t* bg;         /* [nbg, bgs, bgs, 3] */
    int bgs;
    const float* srgb2lin;     /* [256] */
    const uint8_t* lin2srgb;   /* [4096] */
    float fx, fy, cx, cy;
    int W, H;                  /* render size (512) */
} ro_scene;

typedef struct {
    int32_t obj_id, hand_tex_id, bg_id;
    int32_t bg_x0, bg_y0, bg_w, bg_h;      /* crop rectangle inside the background image */
    float light;                            /* intensity in [1,5] */
    float obj_pose[16];                     /* row-major 4x4 */
} ro_sample;

static inline int32_t snap(float x) { return (int32_t)floorf(x * 256.0f + 0.5f); }

static inline uint32_t quant_z(float Z) {
    float inv = 1.0f / Z;
    float z01 = (inv - NEAR_INV) / (FAR_INV - NEAR_INV);
    float q = floorf(z01 * ZMAX + 0.5f);
    if (q < 0.f) q = 0.f;
    if (q > ZMAX) q = ZMAX;
    return (uint32_t)q;
}

/* vertex fetch: camera-frame position of vertex k of global face gid for sample s */
static void face_verts(const ro_scene* sc, const ro_sample* sm, const float* hv, int gid, float P[3][3], int vid[3]) {
    if (gid < HAND_FACES) {
        for (int k = 0; k < 3; ++k) {
            int v = sc->hand_faces[gid * 3 + k];
            const int pv = sc->hand_map ? sc->hand_map[v] : v;     /* position from the MANO vertex, attributes from the render vertex */
            vid[k] = v;
            P[k][0] = hv[pv * 3]; P[k][1] = hv[pv * 3 + 1]; P[k][2] = hv[pv * 3 + 2];
        }
    } else {
        int o = sm->obj_id;
        int f = sc->obj_face_off[o] + (gid - HAND_FACES);
        const float* T = sm->obj_pose;
        for (int k = 0; k < 3; ++k) {
            int v = sc->obj_vert_off[o] + sc->obj_faces[f * 3 + k];
            vid[k] = v;
            const float* p = sc->obj_verts + (size_t)v * 3;
            P[k][0] = (T[0] * p[0] + T[1] * p[1]) + (T[2] * p[2] + T[3]);
            P[k][1] = (T[4] * p[0] + T[5] * p[1]) + (T[6] * p[2] + T[7]);
            P[k][2] = (T[8] * p[0] + T[9] * p[1]) + (T[10] * p[2] + T[11]);
        }
    }
}

typedef struct { int32_t x[3], y[3]; uint32_t z[3]; int valid; } tri_t;

static void setup_tri(const ro_scene* sc, const ro_sample* sm, const float* hv, int gid, tri_t* t) {
    float P[3][3]; int vid[3];
    face_verts(sc, sm, hv, gid, P, vid);
    t->valid = 0;
    for (int k = 0; k < 3; ++k) if (!(P[k][2] > 0.05f)) return;
    if (gid >= HAND_FACES) {   /* back-face cull on the camera-space geometric normal */
        float e1[3] = {P[1][0] - P[0][0], P[1][1] - P[0][1], P[1][2] - P[0][2]};
        float e2[3] = {P[2][0] - P[0][0], P[2][1] - P[0][1], P[2][2] - P[0][2]};
        float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        float d = (n[0] * P[0][0] + n[1] * P[0][1]) + n[2] * P[0][2];
        if (!(d < 0.f)) return;
    }
    for (int k = 0; k < 3; ++k) {
        float x = (sc->fx * P[k][0]) / P[k][2] + sc->cx;
        float y = (sc->fy * P[k][1]) / P[k][2] + sc->cy;
        if (!(fabsf(x) < 1.0e6f) || !(fabsf(y) < 1.0e6f)) return;
        t->x[k] = snap(x); t->y[k] = snap(y); t->z[k] = quant_z(P[k][2]);
    }
    int64_t area = (int64_t)(t->x[1] - t->x[0]) * (t->y[2] - t->y[0]) - (int64_t)(t->y[1] - t->y[0]) * (t->x[2] - t->x[0]);
    if (area == 0) return;
    if (area < 0) {   /* orient to positive area (swap 1 <-> 2) */
        int32_t a = t->x[1]; t->x[1] = t->x[2]; t->x[2] = a;
        a = t->y[1]; t->y[1] = t->y[2]; t->y[2] = a;
        uint32_t b = t->z[1]; t->z[1] = t->z[2]; t->z[2] = b;
        t->valid = 2;  /* swapped */
    } else t->valid = 1;
}

static inline int64_t edge(int32_t ax, int32_t ay, int32_t bx, int32_t by, int32_t px, int32_t py) {
    return (int64_t)(bx - ax) * (py - ay) - (int64_t)(by - ay) * (px - ax);
}
static inline int edge_incl(int32_t ax, int32_t ay, int32_t bx, int32_t by) {
    int32_t dx = bx - ax, dy = by - ay;
    return (dy > 0) || (dy == 0 && dx < 0);
}
/* coverage + depth of sample point (px,py) [1/256 px units]; returns 1 and *key if covered */
static inline int cover(const tri_t* t, int32_t px, int32_t py, int64_t w[3]) {
    w[0] = edge(t->x[1], t->y[1], t->x[2], t->y[2], px, py);
    w[1] = edge(t->x[2], t->y[2], t->x[0], t->y[0], px, py);
    w[2] = edge(t->x[0], t->y[0], t->x[1], t->y[1], px, py);
    if (w[0] < 0 || w[1] < 0 || w[2] < 0) return 0;
    if (w[0] == 0 && !edge_incl(t->x[1], t->y[1], t->x[2], t->y[2])) return 0;
    if (w[1] == 0 && !edge_incl(t->x[2], t->y[2], t->x[0], t->y[0])) return 0;
    if (w[2] == 0 && !edge_incl(t->x[0], t->y[0], t->x[1], t->y[1])) return 0;
    return 1;
}

/* keys: uint64 [H*W], initialised to ~0 (no geometry) */
void ro_rasterize(const ro_scene* sc, const ro_sample* sm, const float* hv, uint64_t* keys) {
    const int W = sc->W, H = sc->H;
    for (size_t i = 0; i < (size_t)W * H; ++i) keys[i] = ~(uint64_t)0;
    int nf = HAND_FACES + (sc->obj_face_off[sm->obj_id + 1] - sc->obj_face_off[sm->obj_id]);
    for (int gid = 0; gid < nf; ++gid) {
        tri_t t;
        setup_tri(sc, sm, hv, gid, &t);
        if (!t.valid) continue;
        int32_t minx = t.x[0], maxx = t.x[0], miny = t.y[0], maxy = t.y[0];
        for (int k = 1; k < 3; ++k) {
            if (t.x[k] < minx) minx = t.x[k]; if (t.x[k] > maxx) maxx = t.x[k];
            if (t.y[k] < miny) miny = t.y[k]; if (t.y[k] > maxy) maxy = t.y[k];
        }
        /* pixel centre c covers iff 256*c+128 within [min,max] */
        int x0 = (minx - 128 + 255) >> 8, x1 = (maxx - 128) >> 8, y0 = (miny - 128 + 255) >> 8, y1 = (maxy - 128) >> 8;
        if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0; if (x1 > W - 1) x1 = W - 1; if (y1 > H - 1) y1 = H - 1;
        for (int y = y0; y <= y1; ++y)
            for (int x = x0; x <= x1; ++x) {
                int64_t w[3];
                if (!cover(&t, x * 256 + 128, y * 256 + 128, w)) continue;
                int64_t sum = w[0] + w[1] + w[2];
                uint64_t z = (uint64_t)((w[0] * (int64_t)t.z[0] + w[1] * (int64_t)t.z[1] + w[2] * (int64_t)t.z[2]) / sum);
                uint64_t key = (z << 32) | (uint32_t)gid;
                if (key < keys[(size_t)y * W + x]) keys[(size_t)y * W + x] = key;
            }
    }
}

/* Background = cv2.resize(crop, (W, H)) with the default INTER_LINEAR (renderer.py:125-136).  PARITY UNPINNED: cv2
 * (opencv-python-headless==4.5.1.48, requirements.txt:84) is not in this image.  Restated from the published algorithm
 * of that version's 8-bit path (modules/imgproc/src/resize.cpp: resizeGeneric_ with HResizeLinear / VResizeLinear,
 * INTER_RESIZE_COEF_BITS = 11): source coordinate f = (float)((d + 0.5) * scale - 0.5) with scale = 1 / ((double)dsize /
 * ssize), s = floor(f), f -= s; s < 0 -> (0, f = 0); s >= ssize - 1 -> (ssize - 1, f = 0); coefficients
 * cvRound((1 - f) * 2048), cvRound(f * 2048) (round half to even); rows: S = p[s] * a0 + p[s + 1] * a1; columns:
 * (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.                                                     */
static inline void lin_coef(int d, int dsize, int ssize, int* s0, int* s1, int* a0, int* a1) {
    const double scale = 1.0 / ((double)dsize / (double)ssize);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    *s0 = s; *s1 = s + 1 < ssize ? s + 1 : ssize - 1;
    *a0 = (int)lrintf((1.f - f) * 2048.f); *a1 = (int)lrintf(f * 2048.f);
}
static inline void bg_pixel(const ro_scene* sc, const ro_sample* sm, int x, int y, uint8_t* rgb) {
    int x0, x1, a0, a1, y0, y1, b0, b1;
    lin_coef(x, sc->W, sm->bg_w, &x0, &x1, &a0, &a1);
    lin_coef(y, sc->H, sm->bg_h, &y0, &y1, &b0, &b1);
    const uint8_t* img = sc->bg + (size_t)sm->bg_id * sc->bgs * sc->bgs * 3;
    const uint8_t* r0 = img + ((size_t)(sm->bg_y0 + y0) * sc->bgs + sm->bg_x0) * 3;
    const uint8_t* r1 = img + ((size_t)(sm->bg_y0 + y1) * sc->bgs + sm->bg_x0) * 3;
    for (int c = 0; c < 3; ++c) {
        int S0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
        int S1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
        rgb[c] = (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
    }
}

/* rgbx: uint8 [H*W*4]; X = 255 where geometry, 0 where background */
void ro_shade(const ro_scene* sc, const ro_sample* sm, const float* hv, const uint64_t* keys, uint8_t* rgbx) {
    const int W = sc->W, H = sc->H;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint64_t key = keys[(size_t)y * W + x];
            uint8_t* o = rgbx + ((size_t)y * W + x) * 4;
            if (key == ~(uint64_t)0) { bg_pixel(sc, sm, x, y, o); o[3] = 0; continue; }
            int gid = (int)(uint32_t)key;
            tri_t t; setup_tri(sc, sm, hv, gid, &t);
            float P[3][3]; int vid[3];
            face_verts(sc, sm, hv, gid, P, vid);
            if (t.valid == 2) {   /* keep attribute order consistent with the oriented triangle */
                int a = vid[1]; vid[1] = vid[2]; vid[2] = a;
                for (int i = 0; i < 3; ++i) { float f = P[1][i]; P[1][i] = P[2][i]; P[2][i] = f; }
            }
            int64_t w[3]; cover(&t, x * 256 + 128, y * 256 + 128, w);
            float ws = (float)(w[0] + w[1] + w[2]);
            float l0 = (float)w[0] / ws, l1 = (float)w[1] / ws, l2 = (float)w[2] / ws;
            float i0 = 1.0f / P[0][2], i1 = 1.0f / P[1][2], i2 = 1.0f / P[2][2];
            float d = (l0 * i0 + l1 * i1) + l2 * i2;
            float m0 = (l0 * i0) / d, m1 = (l1 * i1) / d, m2 = (l2 * i2) / d;
            const float *n0, *n1, *n2, *u0, *u1, *u2; const uint8_t* tex; int ts;
            float nn[3][3];
            if (gid < HAND_FACES) {
                n0 = sc->hand_normals + vid[0] * 3; n1 = sc->hand_normals + vid[1] * 3; n2 = sc->hand_normals + vid[2] * 3;
                u0 = sc->hand_uv + vid[0] * 2; u1 = sc->hand_uv + vid[1] * 2; u2 = sc->hand_uv + vid[2] * 2;
                ts = sc->hts; tex = sc->hand_tex + (size_t)sm->hand_tex_id * ts * ts * 3;
            } else {
                const float* T = sm->obj_pose;
                for (int k = 0; k < 3; ++k) {
                    const float* n = sc->obj_normals + (size_t)vid[k] * 3;
                    nn[k][0] = (T[0] * n[0] + T[1] * n[1]) + T[2] * n[2];
                    nn[k][1] = (T[4] * n[0] + T[5] * n[1]) + T[6] * n[2];
                    nn[k][2] = (T[8] * n[0] + T[9] * n[1]) + T[10] * n[2];
                }
                n0 = nn[0]; n1 = nn[1]; n2 = nn[2];
                u0 = sc->obj_uv + (size_t)vid[0] * 2; u1 = sc->obj_uv + (size_t)vid[1] * 2; u2 = sc->obj_uv + (size_t)vid[2] * 2;
                ts = sc->ots; tex = sc->obj_tex + (size_t)sm->obj_id * ts * ts * 3;
            }
            float n[3], p[3];
            for (int i = 0; i < 3; ++i) {
                n[i] = (m0 * n0[i] + m1 * n1[i]) + m2 * n2[i];
                p[i] = (m0 * P[0][i] + m1 * P[1][i]) + m2 * P[2][i];
            }
            float u = (m0 * u0[0] + m1 * u1[0]) + m2 * u2[0], v = (m0 * u0[1] + m1 * u1[1]) + m2 * u2[1];
            float nl = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]); if (nl < 1e-20f) nl = 1e-20f;
            float d2 = (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2];
            float dl = sqrtf(d2);
            float ndl = -((n[0] * p[0] + n[1] * p[1]) + n[2] * p[2]) / (nl * dl);   /* light at the camera origin */
            if (gid < HAND_FACES) ndl = fabsf(ndl);                                  /* double-sided hand */
            if (ndl < 0.f) ndl = 0.f;
            /* ---- pyrender mesh.frag, one point light at the camera (l = v = h; vh = 1 -> F = r0 = specular colour):
             *   nl = clamp(n.l, 0.001, 1), nv = clamp(|n.v|, 0.001, 1), nh = clamp(n.h, 0, 1)
             *   G = aL * aV, a? = 2 n? / (n? + sqrt(a2 + (1 - a2) n?^2)),  D = a2 / (pi ((nh a2 - nh) nh + 1)^2),  a2 = roughness^4
             *   colour = nl * radiance * ((1 - F) diffuse / pi + F G D / (4 nl nv)) + base * ambient,  radiance = 0.9 I / d^2
             *   diffuse = base (1 - f0)(1 - metallic),  F = mix(f0, base, metallic)                                         */
            const float a2 = (PBR_ROUGHNESS * PBR_ROUGHNESS) * (PBR_ROUGHNESS * PBR_ROUGHNESS);
            float nlc = ndl < 0.001f ? 0.001f : (ndl > 1.f ? 1.f : ndl);
            float nhc = ndl > 1.f ? 1.f : ndl;
            float att = (2.0f * nlc) / (nlc + sqrtf(a2 + (1.0f - a2) * (nlc * nlc)));
            float ff = (nhc * a2 - nhc) * nhc + 1.0f;
            float Dm = a2 / (3.14159274f * (ff * ff));
            float gd4 = ((att * att) * Dm) / ((4.0f * nlc) * nlc);
            float rad = (PBR_LIGHT_COLOR * sm->light) / d2;
            float nlrad = nlc * rad;
            u = u - floorf(u); v = v - floorf(v);
            int tx = (int)(u * (float)ts), ty = (int)(v * (float)ts);
            if (tx > ts - 1) tx = ts - 1; if (ty > ts - 1) ty = ts - 1;
            const uint8_t* texel = tex + ((size_t)ty * ts + tx) * 3;
            for (int c = 0; c < 3; ++c) {
                float base = sc->srgb2lin[texel[c]];
                float F = PBR_F0 * (1.0f - PBR_METALLIC) + PBR_METALLIC * base;
                float dif = ((1.0f - F) * (((1.0f - PBR_F0) * (1.0f - PBR_METALLIC)) * base)) * PBR_INV_PI;
                float lin = base * PBR_AMBIENT + nlrad * (dif + F * gd4);
                if (lin < 0.f) lin = 0.f; if (lin > 1.f) lin = 1.f;
                o[c] = sc->lin2srgb[(int)(lin * 4095.0f + 0.5f)];
            }
            o[3] = 255;
        }
}

/* ------------------------------------------------------------------ PIL-semantics colour jitter (u8 RGB in place)
 * ops: 0 brightness, 1 saturation(Color), 2 hue, 3 contrast  (img_augment.py:25-46); param = factor.
 * ImageEnhance._Enhance.enhance(f) = Image.blend(degenerate, image, f); ImagingBlend (libImaging/Blend.c):
 *   0<=f<=1 : out = (UINT8)(in1 + f*(in2 - in1))            (float math, C cast truncation)
 *   else    : t = in1 + f*(in2 - in1); out = t<=0 ? 0 : t>=255 ? 255 : (UINT8)t                                   */
static inline uint8_t blend8(int in1, int in2, float f) {
    float t = (float)in1 + f * (float)(in2 - in1);
    if (f >= 0.f && f <= 1.f) return (uint8_t)t;
    if (t <= 0.f) return 0;
    if (t >= 255.f) return 255;
    return (uint8_t)t;
}
static inline int luma8(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }  /* Convert.c L24 */

static inline void rgb2hsv8(const uint8_t* in, uint8_t* out) {   /* libImaging/Convert.c rgb2hsv_row (mixed float/double as there) */
    int r = in[0], g = in[1], b = in[2];
    int maxc = r > g ? (r > b ? r : b) : (g > b ? g : b);
    int minc = r < g ? (r < b ? r : b) : (g < b ? g : b);
    uint8_t uh, us, uv = (uint8_t)maxc;
    if (minc == maxc) { uh = 0; us = 0; }
    else {
        float cr = (float)(maxc - minc);
        float s = cr / (float)maxc;
        float rc = ((float)(maxc - r)) / cr, gc = ((float)(maxc - g)) / cr, bc = ((float)(maxc - b)) / cr;
        float h;
        if (r == maxc) h = bc - gc;
        else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
        else h = (float)(4.0 + (double)gc - (double)rc);
        h = (float)fmod(((double)h / 6.0 + 1.0), 1.0);
        int ih = (int)((double)h * 255.0), is = (int)((double)s * 255.0);
        uh = (uint8_t)(ih < 0 ? 0 : ih > 255 ? 255 : ih);
        us = (uint8_t)(is < 0 ? 0 : is > 255 ? 255 : is);
    }
    out[0] = uh; out[1] = us; out[2] = uv;
}
static inline void hsv2rgb8(const uint8_t* in, uint8_t* out) {   /* libImaging/Convert.c hsv2rgb (mixed float/double as there) */
    uint8_t h = in[0], s = in[1], v = in[2];
    if (s == 0) { out[0] = out[1] = out[2] = v; return; }
    int i = (int)floor((double)(float)h * 6.0 / 255.0);
    float f = (float)((double)(float)h * 6.0 / 255.0 - (double)(float)i);
    float fs = (float)((double)(float)s / 255.0);
    int p = (int)round((double)(float)v * (1.0 - (double)fs));
    int q = (int)round((double)(float)v * (1.0 - (double)fs * (double)f));
    int t = (int)round((double)(float)v * (1.0 - (double)fs * (1.0 - (double)f)));
    p = p < 0 ? 0 : p > 255 ? 255 : p; q = q < 0 ? 0 : q > 255 ? 255 : q; t = t < 0 ? 0 : t > 255 ? 255 : t;
    switch (i % 6) {
        case 0: out[0] = v; out[1] = (uint8_t)t; out[2] = (uint8_t)p; break;
        case 1: out[0] = (uint8_t)q; out[1] = v; out[2] = (uint8_t)p; break;
        case 2: out[0] = (uint8_t)p; out[1] = v; out[2] = (uint8_t)t; break;
        case 3: out[0] = (uint8_t)p; out[1] = (uint8_t)q; out[2] = v; break;
        case 4: out[0] = (uint8_t)t; out[1] = (uint8_t)p; out[2] = v; break;
        default: out[0] = v; out[1] = (uint8_t)p; out[2] = (uint8_t)q; break;
    }
}
static inline void jitter_op(int op, float f, int mean_gray, uint8_t* px) {
    if (op == 0) { for (int c = 0; c < 3; ++c) px[c] = blend8(0, px[c], f); }
    else if (op == 1) { int l = luma8(px[0], px[1], px[2]); for (int c = 0; c < 3; ++c) px[c] = blend8(l, px[c], f); }
    else if (op == 2) { uint8_t hsv[3]; rgb2hsv8(px, hsv); hsv[0] = (uint8_t)(hsv[0] + (uint8_t)(int)(f * 255.0f)); hsv2rgb8(hsv, px); }
    else { for (int c = 0; c < 3; ++c) px[c] = blend8(mean_gray, px[c], f); }
}

/* Applies the 4-op chain (order[4] = op ids, factor[4]) to a full RGBX image in place (X untouched); the contrast op
 * uses int(mean(L) + 0.5) of the image as it is when the op is reached (ImageEnhance.Contrast.__init__).          */
void ro_color_jitter(uint8_t* rgbx, int npix, const int32_t* order, const float* factor) {
    for (int k = 0; k < 4; ++k) {
        int op = order[k];
        int mean = 0;
        if (op == 3) {
            uint64_t s = 0;
            for (int i = 0; i < npix; ++i) s += (uint64_t)luma8(rgbx[i * 4], rgbx[i * 4 + 1], rgbx[i * 4 + 2]);
            mean = (int)((double)s / (double)npix + 0.5);
        }
        for (int i = 0; i < npix; ++i) jitter_op(op, factor[k], mean, rgbx + (size_t)i * 4);
    }
}

/* PIL ImageFilter.GaussianBlur(radius) on the RGB bands of an RGBX image, in place (rendered_dataset.py:257-258).
 * Restates libImaging/BoxBlur.c: ImagingGaussianBlur = 3 passes of a box blur of fractional radius
 * _gaussian_blur_radius(radius, 3) along x, then 3 along y; one pass (ImagingLineBoxBlur8/32) is
 *   out[x] = (acc * ww + (in[x - r - 1] + in[x + r + 1]) * fw + 2^23) >> 24,   acc = sum in[x - r .. x + r],
 * indices clamped to the line, r = (int)radius, ww = (UINT32)(2^24 / (radius * 2 + 1)) (float division),
 * fw = (2^24 - (2r + 1) * ww) / 2.  Pinned against the real Pillow in tests/test_render_oracle.py.                   */
float ro_gaussian_box_radius(float radius) {
    const int passes = 3;
    float sigma2 = radius * radius / passes;
    float L = (float)sqrt(12.0 * sigma2 + 1.0);
    float l = (float)floor((L - 1.0) / 2.0);
    float a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2);
    a /= 6 * (sigma2 - (l + 1) * (l + 1));
    return l + a;
}
static void box_line(const uint8_t* in, int n, int stride, int r, uint32_t ww, uint32_t fw, uint8_t* out) {
    for (int x = 0; x < n; ++x) {
        uint32_t acc = 0;
        for (int k = x - r; k <= x + r; ++k) { int kk = k < 0 ? 0 : k > n - 1 ? n - 1 : k; acc += in[(size_t)kk * stride]; }
        int lo = x - r - 1 < 0 ? 0 : x - r - 1, hi = x + r + 1 > n - 1 ? n - 1 : x + r + 1;
        uint32_t bulk = acc * ww + ((uint32_t)in[(size_t)lo * stride] + in[(size_t)hi * stride]) * fw;
        out[x] = (uint8_t)((bulk + (1u << 23)) >> 24);
    }
}
void ro_gaussian_blur(uint8_t* rgbx, int W, int H, float radius) {
    const float fr = ro_gaussian_box_radius(radius);
    if (!(fr > 0.f)) return;                           /* ImagingBoxBlur skips a zero radius */
    const int r = (int)fr;
    const uint32_t ww = (uint32_t)((float)(1 << 24) / (fr * 2 + 1));
    const uint32_t fw = ((1u << 24) - (uint32_t)(r * 2 + 1) * ww) / 2;
    uint8_t* line = (uint8_t*)malloc((size_t)(W > H ? W : H));
    for (int pass = 0; pass < 3; ++pass)
        for (int y = 0; y < H; ++y)
            for (int c = 0; c < 3; ++c) {
                uint8_t* p = rgbx + (size_t)y * W * 4 + c;
                box_line(p, W, 4, r, ww, fw, line);
                for (int x = 0; x < W; ++x) p[(size_t)x * 4] = line[x];
            }
    for (int pass = 0; pass < 3; ++pass)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < 3; ++c) {
                uint8_t* p = rgbx + (size_t)x * 4 + c;
                box_line(p, H, W * 4, r, ww, fw, line);
                for (int y = 0; y < H; ++y) p[(size_t)y * W * 4] = line[y];
            }
    free(line);
}

/* Nearest-neighbour affine crop (PIL Image.transform(AFFINE, NEAREST), img_augment.transform_img): inv = 2x3 matrix
 * mapping output pixel centres to source coordinates; out float [3][oh][ow] = px/255 - 0.5, 0 outside the source. */
void ro_affine_crop(const uint8_t* rgbx, int W, int H, const float* inv, int ow, int oh, float* out_chw) {
    for (int y = 0; y < oh; ++y)
        for (int x = 0; x < ow; ++x) {
            float xin = (inv[0] * ((float)x + 0.5f) + inv[1] * ((float)y + 0.5f)) + inv[2];
            float yin = (inv[3] * ((float)x + 0.5f) + inv[4] * ((float)y + 0.5f)) + inv[5];
            int sx = (int)floorf(xin), sy = (int)floorf(yin);
            for (int c = 0; c < 3; ++c) {
                float v = 0.f;
                if (sx >= 0 && sx < W && sy >= 0 && sy < H) v = (float)rgbx[((size_t)sy * W + sx) * 4 + c];
                out_chw[((size_t)c * oh + y) * ow + x] = v / 255.0f - 0.5f;
            }
        }
}

/* The augmentation chain of a real frame, HOdata.__getitem__ (anakin/datasets/hodata.py:336-337,435-446): optional
 * Image.FLIP_LEFT_RIGHT, GaussianBlur, colour jitter, inverse-affine crop, to_tensor - 0.5; rgbx is modified in place. */
void ro_augment(uint8_t* rgbx, int W, int H, int flip, float blur_radius, const int32_t* order, const float* factor,
                const float* inv, int ow, int oh, float* out_chw) {
    if (flip)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W / 2; ++x) {
                uint32_t* a = (uint32_t*)rgbx + (size_t)y * W + x;
                uint32_t* b = (uint32_t*)rgbx + (size_t)y * W + (W - 1 - x);
                uint32_t t = *a; *a = *b; *b = t;
            }
    ro_gaussian_blur(rgbx, W, H, blur_radius);
    ro_color_jitter(rgbx, W * H, order, factor);
    ro_affine_crop(rgbx, W, H, inv, ow, oh, out_chw);
}

/* Whole synthesis of one batch (used as the CPU baseline): OpenMP over samples when compiled with -fopenmp. */
void ro_render_batch(const ro_scene* sc, const ro_sample* sm, const float* hand_verts, int B, const int32_t* order,
                     const float* factor, const float* inv_affine, const float* blur_radius, int ow, int oh,
                     float* out_bchw, uint8_t* scratch_rgbx, uint64_t* scratch_keys) {
#pragma omp parallel for schedule(dynamic)
    for (int b = 0; b < B; ++b) {
        uint64_t* keys = scratch_keys + (size_t)b * sc->W * sc->H;
        uint8_t* img = scratch_rgbx + (size_t)b * sc->W * sc->H * 4;
        const float* hv = hand_verts + (size_t)b * HAND_VERTS * 3;
        ro_rasterize(sc, sm + b, hv, keys);
        ro_shade(sc, sm + b, hv, keys, img);
        if (blur_radius) ro_gaussian_blur(img, sc->W, sc->H, blur_radius[b]);
        ro_color_jitter(img, sc->W * sc->H, order + b * 4, factor + b * 4);
        ro_affine_crop(img, sc->W, sc->H, inv_affine + b * 6, ow, oh, out_bchw + (size_t)b * 3 * ow * oh);
    }
}
