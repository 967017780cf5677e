// Fused pose assembly + criterion, forward AND backward, one wave per sample.
//
// replaces ~150 tiny torch kernels per step:
//   pose assembly   anakin/models/hybridbaseline.py:49-96 (batch_uvd2xyz utils/transform.py:512-546,
//                   compute_rotation_matrix_from_ortho6d utils/transform.py:578-618, corners = R*can + boxroot)
//   JointsLoss      anakin/criterions/jointloss.py:25-67      (vis-masked MSE, joints + 0.2 corners)
//   HandOrdLoss     anakin/criterions/ordinal.py:144-227      (joint-pair log(1+relu) + part-pair relu, 1+20 views)
//   SceneOrdLoss    anakin/criterions/ordinal.py:262-306      (joint-corner pairs, 1+40 views)
//   Criterion       anakin/criterions/criterion.py:57-67      (lambda-weighted sum)
//   per-sample EPE  anakin/metrics/val_metric.py:84-106       (mm, feeds the CCV re-weighting)
// and their autograd backward down to d(kp3d) and d(box6d).  The random draws (view vectors, pair subsets) are
// inputs, drawn on the host in the reference's RNG order.  Deterministic: no atomics; fixed reduction order.
#include "common.h"

#define PL_NJ 21
#define PL_NC 8
#define PL_MAXPAIR 96
#define PL_MAXVIEW 48

struct PoseLossArgs {
    const float* kp3d;      // [B,22,3] uvd in [0,1)
    const float* box6d;     // [B,box_stride] first 6 valid
    int box_stride;
    const float* root_joint;   // [B,3]
    const float* cam_intr;     // [B,3,3]
    const float* corners_can;  // [B,8,3]
    const float* joints_3d;    // [B,21,3] root-relative targets
    const float* corners_3d;   // [B,8,3]
    const float* joints_vis;   // [B,21]
    const float* corners_vis;  // [B,8]
    const float* hand_views;   // [nvh,3]
    const float* scene_views;  // [nvs,3]
    const int64_t* j0; const int64_t* j1;   // [njp] joint pairs
    const int64_t* p0; const int64_t* p1;   // [npp] part pairs (indices into the 20 parts)
    const int64_t* s0; const int64_t* s1;   // [nsp] (joint, corner) pairs
    int nvh, nvs, njp, npp, nsp;
    int B, center_idx;
    float res_w, res_h, depth_range;
    float lam_joints, lam_corners;          // inside JointsLoss (1.0, 0.2)
    float lam_hand_joint, lam_hand_part, lam_scene;   // inside the ordinal losses (1,1,1)
    float w_jointsloss, w_handord, w_sceneord;        // Criterion LAMBDAS (0.5, 0.2, 0.1); 0 disables a loss
    // SymCornerLoss (symcornerloss.py:49-102): min over the object's symmetry set of the vis-masked corner MSE
    const float* sym_R;        // [nobj][symK][3][3]
    const float* sym_t;        // [nobj][symK][3]  (metres)
    const int64_t* obj_idx;    // [B] 1-based object index
    const float* obj_transf;   // [B][4][4]
    int symK;                  // 0: loss absent
    float lam_sym, w_sym;      // LAMBDA_SYM_CORNERS_3D, Criterion LAMBDA of the loss
    float* sym_loss;           // [1] output (mean over the batch)
    // outputs
    float* joints_abs;      // [B,21,3]
    float* corners_abs;     // [B,8,3]
    float* rotmat;          // [B,3,3]
    float* uvd2d;           // [B,30,3]  (2d_uvd: 21 joints uvd, 8 corners (u,v,0), boxroot uvd)
    float* sample_part;     // [B,8] per-sample loss partial sums + EPE: Lj, Lc, jo, po, so, epe_j_mm, epe_c_mm, -
    float* g_kp3d;          // [B,22,3]   (may be NULL: forward only)
    float* g_box6d;         // [B,6]
};

__device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

__constant__ int c_parents[21] = {0, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 0, 13, 14, 15, 0, 17, 18, 19};

// One workgroup of sixteen waves per sample.  The short assembly phases run on wave 0 (`lane` is out of range on the other
// waves, so every `lane < N` guard excludes them).  The three ordinal losses run side by side on disjoint wave ranges with
// the views of a pair spread over 4 (hand) or 8 (scene) adjacent lanes, whose gradient partials are combined by a fixed
// butterfly; the deterministic gradient gathers split each pair list in sixteen fixed ranges, summed in wave order.  As a
// single wave the kernel was a 75 us chain of dependent LDS / transcendental latencies (~800 cycles per view).
#define PL_WAVES 16
__global__ __launch_bounds__(PL_WAVES * 64) void pose_loss_kernel(PoseLossArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, l64 = tid & 63;
    const int lane = wave == 0 ? tid : 1 << 20;
    __shared__ float P[22][3];        // predicted abs positions (21 joints + boxroot)
    __shared__ float C[8][3];         // predicted abs corners
    __shared__ float mP[21][3], mT[21][3], mC[8][3], mTC[8][3];   // vis-masked pred / target
    __shared__ float T[21][3], TC[8][3];
    __shared__ float part_p[20][3], part_t[20][3];
    __shared__ float vj[21], vc[8];
    __shared__ float R[3][3], xh[3], yh[3], zh[3], zraw[3], avec[3], bvec[3], an, zn;
    __shared__ float hv[PL_MAXVIEW][3], sv[PL_MAXVIEW][3];
    __shared__ float gjp[PL_MAXPAIR][3];                 // gradient wrt (mP_a - mP_b) per joint pair
    __shared__ float gpp[PL_MAXPAIR][3], gpq[PL_MAXPAIR][3];   // gradient wrt part_p, part_q per part pair
    __shared__ float gsp[PL_MAXPAIR][3];                 // gradient wrt (mP_a - mC_c) per scene pair
    __shared__ float gpart[20][3];
    __shared__ float gP[22][3], gC[8][3];
    __shared__ float red[PL_WAVES][5];
    __shared__ float gpart_w[PL_WAVES][20][3], gP_w[PL_WAVES][21][3], gC_w[PL_WAVES][8][3];   // per-wave partial gathers
    // Every global input of the sample is requested in this first phase -- pair tables, view vectors, predictions and
    // targets -- with static (predicated) trip counts so the loads issue back to back and their latencies overlap: the kernel
    // is one wave per sample, and as ~30 dependent load -> wait -> use rounds it was a 75 us latency chain.
    __shared__ uint8_t ij0[PL_MAXPAIR], ij1[PL_MAXPAIR], ip0[PL_MAXPAIR], ip1[PL_MAXPAIR], is0[PL_MAXPAIR], is1[PL_MAXPAIR];
    __shared__ float kp[66], Kc[9], CAN[8][3], J3[63], C3[24], RJ[3];
    // unconditional loads (index clamped, absent tables redirected to a valid address), conditional LDS writes: branches
    // around the loads would put each one in its own basic block with its own s_waitcnt
    if (wave == 0) {
    const void* any = a.kp3d;
    auto ld = [&](auto* p, int i, int n) { p = n > 0 ? p : (decltype(p))any; return p[i < n ? i : 0]; };
    int64_t rj0[2], rj1[2], rp0[2], rp1[2], rs0[2], rs1[2];
    float rhv[3], rsv[3];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = lane + it * 64;
        rj0[it] = ld(a.j0, i, a.njp); rj1[it] = ld(a.j1, i, a.njp);
        rp0[it] = ld(a.p0, i, a.npp); rp1[it] = ld(a.p1, i, a.npp);
        rs0[it] = ld(a.s0, i, a.nsp); rs1[it] = ld(a.s1, i, a.nsp);
    }
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int i = lane + it * 64;
        rhv[it] = ld(a.hand_views, i, a.nvh * 3); rsv[it] = ld(a.scene_views, i, a.nvs * 3);
    }
    const float r_kp = a.kp3d[(long)b * 66 + lane], r_kp2 = a.kp3d[(long)b * 66 + 64 + (lane & 1)];
    const float r_K = a.cam_intr[(long)b * 9 + (lane < 9 ? lane : 0)];
    const int l24 = lane < 24 ? lane : 0;
    const float r_can = a.corners_can[(long)b * 24 + l24], r_c3 = a.corners_3d[(long)b * 24 + l24];
    const float r_j3 = a.joints_3d[(long)b * 63 + (lane < 63 ? lane : 0)];
    const float r_rj = a.root_joint[b * 3 + (lane < 3 ? lane : 0)];
    const float r_vj = a.joints_vis[b * 21 + (lane < 21 ? lane : 0)], r_vc = a.corners_vis[b * 8 + (lane & 7)];
    const float r_box = a.box6d[(long)b * a.box_stride + (lane < 6 ? lane : 0)];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = lane + it * 64;
        if (i < a.njp) { ij0[i] = (uint8_t)rj0[it]; ij1[i] = (uint8_t)rj1[it]; }
        if (i < a.npp) { ip0[i] = (uint8_t)rp0[it]; ip1[i] = (uint8_t)rp1[it]; }
        if (i < a.nsp) { is0[i] = (uint8_t)rs0[it]; is1[i] = (uint8_t)rs1[it]; }
    }
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int i = lane + it * 64;
        if (i < a.nvh * 3) (&hv[0][0])[i] = rhv[it];
        if (i < a.nvs * 3) (&sv[0][0])[i] = rsv[it];
    }
    kp[lane] = r_kp;
    if (lane < 2) kp[64 + lane] = r_kp2;
    if (lane < 9) Kc[lane] = r_K;
    if (lane < 24) { (&CAN[0][0])[lane] = r_can; C3[lane] = r_c3; }
    if (lane < 63) J3[lane] = r_j3;
    if (lane < 3) { RJ[lane] = r_rj; avec[lane] = r_box; }
    else if (lane < 6) bvec[lane - 3] = r_box;
    if (lane < 21) vj[lane] = r_vj;
    if (lane < 8) vc[lane] = r_vc;
    }
    __syncthreads();
    const float fx = Kc[0], fy = Kc[4], cx = Kc[2], cy = Kc[5], rootz = RJ[2];
    // ---- uvd -> xyz (transform.py:512-546)
    if (lane < 22) {
        float u = kp[lane * 3], v = kp[lane * 3 + 1], d = kp[lane * 3 + 2];
        float z = (d - 0.5f) * a.depth_range + rootz;
        P[lane][0] = (u * a.res_w - cx) / fx * z;
        P[lane][1] = (v * a.res_h - cy) / fy * z;
        P[lane][2] = z;
    }
    // ---- ortho6d -> R (transform.py:578-618)
    if (lane == 0) {
        float n = sqrtf(dot3(avec, avec)); n = fmaxf(n, 1e-8f); an = n;
        for (int i = 0; i < 3; ++i) xh[i] = avec[i] / n;
        cross3(xh, bvec, zraw);
        float m = sqrtf(dot3(zraw, zraw)); m = fmaxf(m, 1e-8f); zn = m;
        for (int i = 0; i < 3; ++i) zh[i] = zraw[i] / m;
        cross3(zh, xh, yh);
        for (int i = 0; i < 3; ++i) { R[i][0] = xh[i]; R[i][1] = yh[i]; R[i][2] = zh[i]; }
    }
    __syncthreads();
    if (lane < 24) {
        int c = lane / 3, i = lane % 3;
        const float* can = CAN[c];
        C[c][i] = R[i][0] * can[0] + R[i][1] * can[1] + R[i][2] * can[2] + P[21][i];
    }
    if (lane < 63) {
        int k = lane / 3, i = lane % 3;
        float t = J3[k * 3 + i] + RJ[i];
        T[k][i] = t; mT[k][i] = t * vj[k]; mP[k][i] = P[k][i] * vj[k];
    }
    __syncthreads();
    if (lane < 24) {
        int c = lane / 3, i = lane % 3;
        float t = C3[c * 3 + i] + RJ[i];
        TC[c][i] = t; mTC[c][i] = t * vc[c]; mC[c][i] = C[c][i] * vc[c];
    }
    if (lane < 60) {
        int k = lane / 3 + 1, i = lane % 3;
        part_p[k - 1][i] = mP[k][i] - mP[c_parents[k]][i];
        part_t[k - 1][i] = mT[k][i] - mT[c_parents[k]][i];
    }
    __syncthreads();
    // ---- outputs
    if (lane < 63) a.joints_abs[(long)b * 63 + lane] = P[lane / 3][lane % 3];
    if (lane < 24) a.corners_abs[(long)b * 24 + lane] = C[lane / 3][lane % 3];
    if (lane < 9) a.rotmat[(long)b * 9 + lane] = R[lane / 3][lane % 3];
    if (a.uvd2d) {
        float* o = a.uvd2d + (long)b * 90;
        if (lane < 21) { o[lane * 3] = kp[lane * 3]; o[lane * 3 + 1] = kp[lane * 3 + 1]; o[lane * 3 + 2] = kp[lane * 3 + 2]; }
        if (lane < 8) {
            const float* K = Kc;
            float X = C[lane][0], Y = C[lane][1], Z = C[lane][2];
            float hx = K[0] * X + K[1] * Y + K[2] * Z, hy = K[3] * X + K[4] * Y + K[5] * Z, hz = K[6] * X + K[7] * Y + K[8] * Z;
            o[(21 + lane) * 3] = hx / hz / a.res_w; o[(21 + lane) * 3 + 1] = hy / hz / a.res_h; o[(21 + lane) * 3 + 2] = 0.f;
        }
        if (lane < 3) o[29 * 3 + lane] = kp[21 * 3 + lane];
    }
    // ---- losses: per-lane partial sums   [Lj, Lc, jo, po, so]
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const float B_ = (float)a.B;
    const float nJ = B_ * 63.f, nC = B_ * 24.f;
    const float nJO = B_ * a.njp * a.nvh, nPO = B_ * a.npp * a.nvh, nSO = B_ * a.nsp * a.nvs;
    if (lane < 63) { float d = mP[lane / 3][lane % 3] - mT[lane / 3][lane % 3]; acc[0] = d * d; }
    if (lane < 24) { float d = mC[lane / 3][lane % 3] - mTC[lane / 3][lane % 3]; acc[1] = d * d; }
    // wave ranges of the three losses: 16 joint / part pairs or 8 scene pairs per wave when that fits, else a fixed split
    int wj = (a.njp + 15) / 16, wp = (a.npp + 15) / 16, ws = (a.nsp + 7) / 8;
    if (wj + wp + ws > PL_WAVES) { wj = 5; wp = 4; ws = PL_WAVES - 9; }
    const float wjo = a.w_handord * a.lam_hand_joint / nJO;
    const float wpo = a.w_handord * a.lam_hand_part / nPO;
    const float wso = a.w_sceneord * a.lam_scene / nSO;
    if (wave < wj) {                                    // joint-level ordinal: 4 lanes per pair
        const int sub = tid & 3;
        for (int pi = tid >> 2; pi < a.njp; pi += wj * 16) {
            int i0 = (int)ij0[pi], i1 = (int)ij1[pi];
            float dt[3], dp[3], g[3] = {0.f, 0.f, 0.f};
            for (int i = 0; i < 3; ++i) { dt[i] = mT[i0][i] - mT[i1][i]; dp[i] = mP[i0][i] - mP[i1][i]; }
            for (int v = sub; v < a.nvh; v += 4) {
                float s = sgn(dot3(dt, hv[v])), t = -s * dot3(dp, hv[v]);
                if (t > 0.f) {
                    acc[2] += log1pf(t);
                    float c = wjo * (-s) / (1.f + t);
                    g[0] += c * hv[v][0]; g[1] += c * hv[v][1]; g[2] += c * hv[v][2];
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) { g[i] += __shfl_xor(g[i], 1, 64); g[i] += __shfl_xor(g[i], 2, 64); }
            if (sub == 0) { gjp[pi][0] = g[0]; gjp[pi][1] = g[1]; gjp[pi][2] = g[2]; }
        }
    } else if (wave < wj + wp) {                        // part-level ordinal: 4 lanes per pair
        const int t0 = tid - wj * 64, sub = t0 & 3;
        for (int pi = t0 >> 2; pi < a.npp; pi += wp * 16) {
            int i0 = (int)ip0[pi], i1 = (int)ip1[pi];
            float ct[3], cp[3], gp[3] = {0.f, 0.f, 0.f}, gq[3] = {0.f, 0.f, 0.f};
            cross3(part_t[i0], part_t[i1], ct);
            cross3(part_p[i0], part_p[i1], cp);
            for (int v = sub; v < a.nvh; v += 4) {
                float s = sgn(dot3(ct, hv[v])), t = -s * dot3(cp, hv[v]);
                if (t > 0.f) {
                    acc[3] += t;
                    float c = wpo * (-s);
                    float qn[3], np_[3];
                    cross3(part_p[i1], hv[v], qn);       // d((p x q).n)/dp = q x n
                    cross3(hv[v], part_p[i0], np_);      // d((p x q).n)/dq = n x p
                    for (int i = 0; i < 3; ++i) { gp[i] += c * qn[i]; gq[i] += c * np_[i]; }
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                gp[i] += __shfl_xor(gp[i], 1, 64); gp[i] += __shfl_xor(gp[i], 2, 64);
                gq[i] += __shfl_xor(gq[i], 1, 64); gq[i] += __shfl_xor(gq[i], 2, 64);
            }
            if (sub == 0) for (int i = 0; i < 3; ++i) { gpp[pi][i] = gp[i]; gpq[pi][i] = gq[i]; }
        }
    } else if (wave < wj + wp + ws) {                   // scene ordinal: 8 lanes per pair
        const int t0 = tid - (wj + wp) * 64, sub = t0 & 7;
        for (int pi = t0 >> 3; pi < a.nsp; pi += ws * 8) {
            int i0 = (int)is0[pi], i1 = (int)is1[pi];
            float dt[3], dp[3], g[3] = {0.f, 0.f, 0.f};
            for (int i = 0; i < 3; ++i) { dt[i] = mT[i0][i] - mTC[i1][i]; dp[i] = mP[i0][i] - mC[i1][i]; }
            for (int v = sub; v < a.nvs; v += 8) {
                float s = sgn(dot3(dt, sv[v])), t = -s * dot3(dp, sv[v]);
                if (t > 0.f) {
                    acc[4] += log1pf(t);
                    float c = wso * (-s) / (1.f + t);
                    g[0] += c * sv[v][0]; g[1] += c * sv[v][1]; g[2] += c * sv[v][2];
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) { g[i] += __shfl_xor(g[i], 1, 64); g[i] += __shfl_xor(g[i], 2, 64); g[i] += __shfl_xor(g[i], 4, 64); }
            if (sub == 0) { gsp[pi][0] = g[0]; gsp[pi][1] = g[1]; gsp[pi][2] = g[2]; }
        }
    }
    // ---- SymCornerLoss: lanes scan the symmetry set, wave arg-min (ties -> lowest k), gradient from the winner only
    __shared__ float gsym[8][3];
    if (lane < 24) gsym[lane / 3][lane % 3] = 0.f;
    if (a.symK > 0) {
        const long obj = (long)a.obj_idx[b] - 1;
        const float* Tm = a.obj_transf + (long)b * 16;
        auto sym_gt = [&](int k, int c, float* gt) {           // vis-masked transformed canonical corner c under symmetry k
            const float* Rk = a.sym_R + (obj * a.symK + k) * 9;
            const float* tk = a.sym_t + (obj * a.symK + k) * 3;
            const float* can = CAN[c];
            float sc[3];
            for (int i = 0; i < 3; ++i) sc[i] = (Rk[i * 3] * can[0] + Rk[i * 3 + 1] * can[1] + Rk[i * 3 + 2] * can[2]) + tk[i];
            for (int i = 0; i < 3; ++i) gt[i] = ((Tm[i * 4] * sc[0] + Tm[i * 4 + 1] * sc[1] + Tm[i * 4 + 2] * sc[2]) + Tm[i * 4 + 3]) * vc[c];
        };
        float best = INFINITY; int bk = 0x7fffffff;
        for (int k = lane; k < a.symK; k += 64) {
            float e = 0.f;
#pragma unroll 1
            for (int c = 0; c < 8; ++c) {
                float gt[3]; sym_gt(k, c, gt);
                float per = 0.f;
                for (int i = 0; i < 3; ++i) { float d = gt[i] - mC[c][i]; per += d * d; }
                e += per / 3.f;                                 // .mean(-1) over xyz, then .mean(-1) over corners
            }
            e /= 8.f;
            if (e < best) { best = e; bk = k; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ob = __shfl_xor(best, o, 64); int ok = __shfl_xor(bk, o, 64);
            if (ob < best || (ob == best && ok < bk)) { best = ob; bk = ok; }
        }
        if (lane == 0) a.sample_part[(long)b * 8 + 7] = best;
        if (lane < 8 && a.g_kp3d) {
            float gt[3]; sym_gt(bk, lane, gt);
            const float wS = a.w_sym * a.lam_sym * 2.f / (24.f * B_);
            for (int i = 0; i < 3; ++i) gsym[lane][i] = wS * (mC[lane][i] - gt[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        float v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (l64 == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (lane < 5) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < PL_WAVES; ++w) s += red[w][lane];
        a.sample_part[(long)b * 8 + lane] = s;
    }
    if (lane == 5 || lane == 6) {     // per-sample EPE in mm (unmasked, absolute), val_metric.py:96-104
        float s = 0.f;
        if (lane == 5) { for (int k = 0; k < 21; ++k) { float d[3] = {P[k][0] - T[k][0], P[k][1] - T[k][1], P[k][2] - T[k][2]}; s += sqrtf(dot3(d, d)); } s = s / 21.f * 1000.f; }
        else { for (int k = 0; k < 8; ++k) { float d[3] = {C[k][0] - TC[k][0], C[k][1] - TC[k][1], C[k][2] - TC[k][2]}; s += sqrtf(dot3(d, d)); } s = s / 8.f * 1000.f; }
        a.sample_part[(long)b * 8 + lane] = s;
    }
    if (!a.g_kp3d) return;
    // ---- backward, deterministic gather
    // every wave scans its sixteenth of each pair list, branch-free (selects, so the LDS loads pipeline); partials are summed
    // in wave order
    const int qpp = (a.npp + PL_WAVES - 1) / PL_WAVES, qjp = (a.njp + PL_WAVES - 1) / PL_WAVES, qsp = (a.nsp + PL_WAVES - 1) / PL_WAVES;
    if (l64 < 60) {      // part gradients: gpart[k] = sum over part pairs containing k
        int k = l64 / 3, i = l64 % 3;
        float s = 0.f;
        for (int pi = wave * qpp; pi < min((wave + 1) * qpp, a.npp); ++pi) {
            const float u = gpp[pi][i], w = gpq[pi][i];
            s += ((int)ip0[pi] == k ? u : 0.f) + ((int)ip1[pi] == k ? w : 0.f);
        }
        gpart_w[wave][k][i] = s;
    }
    if (l64 < 63) {
        int k = l64 / 3, i = l64 % 3;
        float s = 0.f;
        for (int pi = wave * qjp; pi < min((wave + 1) * qjp, a.njp); ++pi) {
            const float u = gjp[pi][i];
            s += ((int)ij0[pi] == k ? u : 0.f) - ((int)ij1[pi] == k ? u : 0.f);
        }
        for (int pi = wave * qsp; pi < min((wave + 1) * qsp, a.nsp); ++pi) s += ((int)is0[pi] == k ? gsp[pi][i] : 0.f);
        gP_w[wave][k][i] = s;
    }
    if (l64 < 24) {
        int c = l64 / 3, i = l64 % 3;
        float s = 0.f;
        for (int pi = wave * qsp; pi < min((wave + 1) * qsp, a.nsp); ++pi) s -= ((int)is1[pi] == c ? gsp[pi][i] : 0.f);
        gC_w[wave][c][i] = s;
    }
    __syncthreads();
    if (lane < 60) {
        int k = lane / 3, i = lane % 3;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < PL_WAVES; ++w) s += gpart_w[w][k][i];
        gpart[k][i] = s;
    }
    __syncthreads();
    const float wJ = a.w_jointsloss * a.lam_joints * 2.f / nJ, wC = a.w_jointsloss * a.lam_corners * 2.f / nC;
    if (lane < 63) {      // gradient wrt masked joint mP[k][i], then * vis
        int k = lane / 3, i = lane % 3;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < PL_WAVES; ++w) s += gP_w[w][k][i];
        s += wJ * (mP[k][i] - mT[k][i]);
        if (k >= 1) s += gpart[k - 1][i];
        for (int c = 1; c < 21; ++c) if (c_parents[c] == k) s -= gpart[c - 1][i];
        gP[k][i] = s * vj[k];
    }
    if (lane < 24) {
        int c = lane / 3, i = lane % 3;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < PL_WAVES; ++w) s += gC_w[w][c][i];
        s += wC * (mC[c][i] - mTC[c][i]) + gsym[c][i];
        gC[c][i] = s * vc[c];
    }
    __syncthreads();
    if (lane < 3) {      // boxroot receives every corner gradient
        float s = 0.f;
        for (int c = 0; c < 8; ++c) s += gC[c][lane];
        gP[21][lane] = s;
    }
    __syncthreads();
    if (lane < 22) {     // xyz -> uvd
        float u = kp[lane * 3], v = kp[lane * 3 + 1];
        float z = P[lane][2];
        float gx = gP[lane][0], gy = gP[lane][1], gz = gP[lane][2];
        float* o = a.g_kp3d + (long)b * 66 + lane * 3;
        o[0] = gx * a.res_w / fx * z;
        o[1] = gy * a.res_h / fy * z;
        o[2] = a.depth_range * (gx * (u * a.res_w - cx) / fx + gy * (v * a.res_h - cy) / fy + gz);
    }
    if (lane == 0) {     // R -> 6D
        float GR[3][3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            float s = 0.f;
            for (int c = 0; c < 8; ++c) s += gC[c][i] * CAN[c][j];
            GR[i][j] = s;
        }
        float gx[3] = {GR[0][0], GR[1][0], GR[2][0]}, gy[3] = {GR[0][1], GR[1][1], GR[2][1]}, gz[3] = {GR[0][2], GR[1][2], GR[2][2]};
        float t[3];
        cross3(xh, gy, t); for (int i = 0; i < 3; ++i) gz[i] += t[i];        // y = z x x
        cross3(gy, zh, t); for (int i = 0; i < 3; ++i) gx[i] += t[i];
        float gzr[3]; float dz = dot3(zh, gz);
        for (int i = 0; i < 3; ++i) gzr[i] = (gz[i] - zh[i] * dz) / zn;       // z = zraw / |zraw|
        cross3(bvec, gzr, t); for (int i = 0; i < 3; ++i) gx[i] += t[i];      // zraw = x x b
        float gb[3]; cross3(gzr, xh, gb);
        float dx = dot3(xh, gx), ga[3];
        for (int i = 0; i < 3; ++i) ga[i] = (gx[i] - xh[i] * dx) / an;        // x = a / |a|
        float* o = a.g_box6d + (long)b * 6;
        o[0] = ga[0]; o[1] = ga[1]; o[2] = ga[2]; o[3] = gb[0]; o[4] = gb[1]; o[5] = gb[2];
    }
}

// losses[8]: joints_3d_loss, corners_3d_loss, joint_ord_loss, part_ord_loss, scene_ord_loss, final_loss, mean epe_j, mean epe_c
__global__ __launch_bounds__(64) void pose_loss_finalize(const float* __restrict__ sample_part, PoseLossArgs a, float* __restrict__ losses) {
    const int lane = threadIdx.x;
    // column sums over the batch in double: lane-strided partials (independent loads), then a butterfly in fixed order
    double col[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        double s = 0.0;
        for (int b = lane; b < a.B; b += 64) s += (double)sample_part[(long)b * 8 + c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        col[c] = s;
    }
    if (lane != 0) return;
    const float B_ = (float)a.B;
    float v[7];
    v[0] = (float)(col[0] / (B_ * 63.f));
    v[1] = (float)(col[1] / (B_ * 24.f));
    v[2] = (float)(col[2] / (B_ * a.njp * a.nvh));
    v[3] = (float)(col[3] / (B_ * a.npp * a.nvh));
    v[4] = (float)(col[4] / (B_ * a.nsp * a.nvs));
    v[5] = (float)(col[5] / B_);
    v[6] = (float)(col[6] / B_);
    float symv = 0.f;
    if (a.symK > 0) {
        symv = (float)(col[7] / (double)a.B);
        if (a.sym_loss) a.sym_loss[0] = symv;
    }
    for (int i = 0; i < 5; ++i) losses[i] = v[i];
    losses[6] = v[5]; losses[7] = v[6];
    losses[5] = a.w_jointsloss * (a.lam_joints * v[0] + a.lam_corners * v[1]) +
                a.w_handord * (a.lam_hand_joint * v[2] + a.lam_hand_part * v[3]) +
                a.w_sceneord * (a.lam_scene * v[4]) + a.w_sym * (a.lam_sym * symv);
}

static int pose_loss_impl(const float* kp3d, const float* box6d, int box_stride, const float* root_joint,
                            const float* cam_intr, const float* corners_can, const float* joints_3d,
                            const float* corners_3d, const float* joints_vis, const float* corners_vis,
                            const float* hand_views, int nvh, const int64_t* j0, const int64_t* j1, int njp,
                            const int64_t* p0, const int64_t* p1, int npp, const float* scene_views, int nvs,
                            const int64_t* s0, const int64_t* s1, int nsp, int B, int center_idx, float res_w,
                            float res_h, const float* weights8_host, float* joints_abs, float* corners_abs, float* rotmat,
                            float* uvd2d, float* sample_part, float* losses, float* g_kp3d, float* g_box6d,
                            void* stream, const ab_symcorner* sym) {
    if (!kp3d || !box6d || !root_joint || !cam_intr || !corners_can || !joints_3d || !corners_3d || !joints_vis ||
        !corners_vis || !weights8_host || !joints_abs || !corners_abs || !rotmat || !sample_part || !losses)
        return AB_EINVAL;
    if (njp > PL_MAXPAIR || npp > PL_MAXPAIR || nsp > PL_MAXPAIR || nvh > PL_MAXVIEW || nvs > PL_MAXVIEW || B < 1) return AB_ESHAPE;
    if ((weights8_host[5] != 0.f || weights8_host[6] != 0.f) && (!hand_views || !j0 || !j1 || !p0 || !p1)) return AB_EINVAL;
    if (weights8_host[7] != 0.f && (!scene_views || !s0 || !s1)) return AB_EINVAL;
    PoseLossArgs a = {};
    a.kp3d = kp3d; a.box6d = box6d; a.box_stride = box_stride; a.root_joint = root_joint; a.cam_intr = cam_intr;
    a.corners_can = corners_can; a.joints_3d = joints_3d; a.corners_3d = corners_3d; a.joints_vis = joints_vis;
    a.corners_vis = corners_vis; a.hand_views = hand_views; a.scene_views = scene_views;
    a.j0 = j0; a.j1 = j1; a.p0 = p0; a.p1 = p1; a.s0 = s0; a.s1 = s1;
    a.nvh = weights8_host[5] != 0.f || weights8_host[6] != 0.f ? nvh : 0; a.nvs = weights8_host[7] != 0.f ? nvs : 0;
    a.njp = a.nvh ? njp : 0; a.npp = a.nvh ? npp : 0; a.nsp = a.nvs ? nsp : 0;
    a.B = B; a.center_idx = center_idx; a.res_w = res_w; a.res_h = res_h; a.depth_range = 0.4f;
    // weights8_host = {lam_joints, lam_corners, lam_hand_joint, lam_hand_part, lam_scene, w_JointsLoss, w_HandOrdLoss, w_SceneOrdLoss}
    a.lam_joints = weights8_host[0]; a.lam_corners = weights8_host[1]; a.lam_hand_joint = weights8_host[2]; a.lam_hand_part = weights8_host[3];
    a.lam_scene = weights8_host[4]; a.w_jointsloss = weights8_host[5]; a.w_handord = weights8_host[6]; a.w_sceneord = weights8_host[7];
    a.joints_abs = joints_abs; a.corners_abs = corners_abs; a.rotmat = rotmat; a.uvd2d = uvd2d;
    a.sample_part = sample_part; a.g_kp3d = g_kp3d; a.g_box6d = g_box6d;
    if (sym && sym->K > 0 && sym->weight != 0.f && sym->lambda != 0.f) {
        if (!sym->R || !sym->t || !sym->obj_idx || !sym->obj_transf) return AB_EINVAL;
        a.sym_R = sym->R; a.sym_t = sym->t; a.obj_idx = sym->obj_idx; a.obj_transf = sym->obj_transf;
        a.symK = sym->K; a.lam_sym = sym->lambda; a.w_sym = sym->weight; a.sym_loss = sym->loss_out;
    }
    pose_loss_kernel<<<B, PL_WAVES * 64, 0, as_stream(stream)>>>(a);
    AB_LAUNCH_CHECK();
    // guard the normalisers of disabled losses
    PoseLossArgs f = a;
    if (f.njp == 0) { f.njp = 1; f.npp = 1; f.nvh = 1; }
    if (f.nsp == 0) { f.nsp = 1; f.nvs = 1; }
    pose_loss_finalize<<<1, 64, 0, as_stream(stream)>>>(sample_part, f, losses);
    AB_LAUNCH_CHECK();
    return 0;
}

// ---- M4 alone (eval-mode forwards: no targets, no criterion): hybridbaseline.py:49-96 for one sample per wave -- the same
// arithmetic, in the same order, as the assembly phase of pose_loss_kernel above.
__global__ __launch_bounds__(64) void pose_assemble_kernel(const float* __restrict__ kp3d, const float* __restrict__ box6d, int box_stride,
                                                           const float* __restrict__ root_joint, const float* __restrict__ cam_intr,
                                                           const float* __restrict__ corners_can, int center_idx, float res_w, float res_h,
                                                           float depth_range, float* __restrict__ joints_abs, float* __restrict__ corners_abs,
                                                           float* __restrict__ rotmat, float* __restrict__ uvd2d, float* __restrict__ joints_rel,
                                                           float* __restrict__ corners_rel, float* __restrict__ boxroot) {
    const int b = blockIdx.x, lane = threadIdx.x;
    __shared__ float kp[66], Kc[9], CAN[8][3], RJ[3], avec[3], bvec[3], P[22][3], C[8][3], R[3][3];
    kp[lane] = kp3d[(long)b * 66 + lane];
    if (lane < 2) kp[64 + lane] = kp3d[(long)b * 66 + 64 + lane];
    if (lane < 9) Kc[lane] = cam_intr[(long)b * 9 + lane];
    if (lane < 24) (&CAN[0][0])[lane] = corners_can[(long)b * 24 + lane];
    if (lane < 3) { RJ[lane] = root_joint[b * 3 + lane]; avec[lane] = box6d[(long)b * box_stride + lane]; }
    else if (lane < 6) bvec[lane - 3] = box6d[(long)b * box_stride + lane];
    __syncthreads();
    const float fx = Kc[0], fy = Kc[4], cx = Kc[2], cy = Kc[5], rootz = RJ[2];
    if (lane < 22) {
        float u = kp[lane * 3], v = kp[lane * 3 + 1], d = kp[lane * 3 + 2];
        float z = (d - 0.5f) * depth_range + rootz;
        P[lane][0] = (u * res_w - cx) / fx * z;
        P[lane][1] = (v * res_h - cy) / fy * z;
        P[lane][2] = z;
    }
    if (lane == 0) {
        float xh[3], yh[3], zh[3], zraw[3];
        float n = sqrtf(dot3(avec, avec)); n = fmaxf(n, 1e-8f);
        for (int i = 0; i < 3; ++i) xh[i] = avec[i] / n;
        cross3(xh, bvec, zraw);
        float m = sqrtf(dot3(zraw, zraw)); m = fmaxf(m, 1e-8f);
        for (int i = 0; i < 3; ++i) zh[i] = zraw[i] / m;
        cross3(zh, xh, yh);
        for (int i = 0; i < 3; ++i) { R[i][0] = xh[i]; R[i][1] = yh[i]; R[i][2] = zh[i]; }
    }
    __syncthreads();
    if (lane < 24) {
        int c = lane / 3, i = lane % 3;
        const float* can = CAN[c];
        C[c][i] = R[i][0] * can[0] + R[i][1] * can[1] + R[i][2] * can[2] + P[21][i];
    }
    __syncthreads();
    if (lane < 63) {
        const float v = P[lane / 3][lane % 3];
        joints_abs[(long)b * 63 + lane] = v;
        if (joints_rel) joints_rel[(long)b * 63 + lane] = v - P[center_idx][lane % 3];
    }
    if (lane < 24) {
        const float v = C[lane / 3][lane % 3];
        corners_abs[(long)b * 24 + lane] = v;
        if (corners_rel) corners_rel[(long)b * 24 + lane] = v - P[center_idx][lane % 3];
    }
    if (lane < 9) rotmat[(long)b * 9 + lane] = R[lane / 3][lane % 3];
    if (lane < 3 && boxroot) boxroot[b * 3 + lane] = P[21][lane];
    if (uvd2d) {
        float* o = uvd2d + (long)b * 90;
        if (lane < 21) { o[lane * 3] = kp[lane * 3]; o[lane * 3 + 1] = kp[lane * 3 + 1]; o[lane * 3 + 2] = kp[lane * 3 + 2]; }
        if (lane < 8) {
            const float* K = Kc;
            float X = C[lane][0], Y = C[lane][1], Z = C[lane][2];
            float hx = K[0] * X + K[1] * Y + K[2] * Z, hy = K[3] * X + K[4] * Y + K[5] * Z, hz = K[6] * X + K[7] * Y + K[8] * Z;
            o[(21 + lane) * 3] = hx / hz / res_w; o[(21 + lane) * 3 + 1] = hy / hz / res_h; o[(21 + lane) * 3 + 2] = 0.f;
        }
        if (lane < 3) o[29 * 3 + lane] = kp[21 * 3 + lane];
    }
}

extern "C" int ab_pose_assemble(const float* kp3d, const float* box6d, int box_stride, const float* root_joint, const float* cam_intr,
                                const float* corners_can, int B, int center_idx, float res_w, float res_h, float* joints_abs,
                                float* corners_abs, float* rotmat, float* uvd2d, float* joints_rel, float* corners_rel, float* boxroot,
                                void* stream) {
    if (!kp3d || !box6d || !root_joint || !cam_intr || !corners_can || !joints_abs || !corners_abs || !rotmat) return AB_EINVAL;
    if (B <= 0 || center_idx < 0 || center_idx > 20 || box_stride < 6) return AB_ESHAPE;
    pose_assemble_kernel<<<B, 64, 0, as_stream(stream)>>>(kp3d, box6d, box_stride, root_joint, cam_intr, corners_can, center_idx, res_w, res_h,
                                                          0.4f, joints_abs, corners_abs, rotmat, uvd2d, joints_rel, corners_rel, boxroot);
    AB_LAUNCH_CHECK(); return 0;
}

extern "C" int ab_pose_loss(const float* kp3d, const float* box6d, int box_stride, const float* root_joint,
                            const float* cam_intr, const float* corners_can, const float* joints_3d,
                            const float* corners_3d, const float* joints_vis, const float* corners_vis,
                            const float* hand_views, int nvh, const int64_t* j0, const int64_t* j1, int njp,
                            const int64_t* p0, const int64_t* p1, int npp, const float* scene_views, int nvs,
                            const int64_t* s0, const int64_t* s1, int nsp, int B, int center_idx, float res_w,
                            float res_h, const float* weights8_host, float* joints_abs, float* corners_abs, float* rotmat,
                            float* uvd2d, float* sample_part, float* losses, float* g_kp3d, float* g_box6d,
                            void* stream) {
    return pose_loss_impl(kp3d, box6d, box_stride, root_joint, cam_intr, corners_can, joints_3d, corners_3d, joints_vis,
                          corners_vis, hand_views, nvh, j0, j1, njp, p0, p1, npp, scene_views, nvs, s0, s1, nsp, B, center_idx,
                          res_w, res_h, weights8_host, joints_abs, corners_abs, rotmat, uvd2d, sample_part, losses, g_kp3d,
                          g_box6d, stream, nullptr);
}

// ab_pose_loss + SymCornerLoss (sym may be NULL): the final loss (losses[5]) and the corner gradients include it
extern "C" int ab_pose_loss_sym(const float* kp3d, const float* box6d, int box_stride, const float* root_joint,
                                const float* cam_intr, const float* corners_can, const float* joints_3d,
                                const float* corners_3d, const float* joints_vis, const float* corners_vis,
                                const float* hand_views, int nvh, const int64_t* j0, const int64_t* j1, int njp,
                                const int64_t* p0, const int64_t* p1, int npp, const float* scene_views, int nvs,
                                const int64_t* s0, const int64_t* s1, int nsp, int B, int center_idx, float res_w,
                                float res_h, const float* weights8_host, const ab_symcorner* sym, float* joints_abs,
                                float* corners_abs, float* rotmat, float* uvd2d, float* sample_part, float* losses,
                                float* g_kp3d, float* g_box6d, void* stream) {
    return pose_loss_impl(kp3d, box6d, box_stride, root_joint, cam_intr, corners_can, joints_3d, corners_3d, joints_vis,
                          corners_vis, hand_views, nvh, j0, j1, njp, p0, p1, npp, scene_views, nvs, s0, s1, nsp, B, center_idx,
                          res_w, res_h, weights8_host, joints_abs, corners_abs, rotmat, uvd2d, sample_part, losses, g_kp3d,
                          g_box6d, stream, sym);
}
