// MANO linear-blend skinning, one workgroup per hand.
// replaces manotorch.ManoLayer.forward (un-vendored third party) at its ArtiBoost call sites
//   anakin/artiboost/preprocessor.py:25,62; refiner.py:138,193,216,265; grasp_engine.py:90-95
// following the in-tree statement of the same maths, anakin/postprocess/iknet/manolayer.py:182-276 (center_idx=None,
// flat_hand_mean, axis-angle input): Rodrigues x16 -> shape blend -> joint regression -> pose blend -> 3-level
// kinematic chain -> skinning -> 21 joints (16 + 5 fingertip vertices, reordered).
// The blend tables (posedirs 1.26 MB, shapedirs 93 KB, J_regressor 50 KB, weights 50 KB) are shared by the batch
// and stay L2-resident; per hand the kernel reads 232 B and writes 9.6 KB.
#include "common.h"

#define NV 778
#define NJ 16

__constant__ int c_mano_parents[16] = {-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14};
__constant__ int c_mano_tips[5] = {745, 317, 444, 556, 673};
__constant__ int c_mano_reorder[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};

__global__ __launch_bounds__(256) void mano_lbs_kernel(const float* __restrict__ pose, const float* __restrict__ betas,
                                                       const float* __restrict__ v_template,   // [778,3]
                                                       const float* __restrict__ shapedirs,    // [778,3,10]
                                                       const float* __restrict__ posedirs,     // [778,3,135]
                                                       const float* __restrict__ J_regressor,  // [16,778]
                                                       const float* __restrict__ weights,      // [778,16]
                                                       const float* __restrict__ hands_mean,   // [45]
                                                       float* __restrict__ verts, float* __restrict__ joints,
                                                       float* __restrict__ T_abs) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ float R[NJ][9];
    __shared__ float pmap[135];
    __shared__ float beta[10];
    __shared__ float vs[NV * 3];        // v_shaped, later v_posed
    __shared__ float J[NJ][3];
    __shared__ float G[NJ][12];         // 3x4 global transforms
    __shared__ float G2[NJ][12];        // with rest-joint removed
    __shared__ float part[256][3];
    if (tid < 10) beta[tid] = betas[b * 10 + tid];
    if (tid < NJ) {
        // manolayer.py:162-172 (_batch_rodrigues through a quaternion, +1e-8 inside the norm) and :135-160 (_quat2mat)
        float a[3];
        for (int i = 0; i < 3; ++i) a[i] = pose[b * 48 + tid * 3 + i] + (tid > 0 ? hands_mean[(tid - 1) * 3 + i] : 0.f);
        float e[3] = {a[0] + 1e-8f, a[1] + 1e-8f, a[2] + 1e-8f};
        float n = sqrtf((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
        float h = n * 0.5f, s = sinf(h), w = cosf(h);
        float x = s * (a[0] / n), y = s * (a[1] / n), z = s * (a[2] / n);
        float nq = sqrtf(((w * w + x * x) + y * y) + z * z);
        w /= nq; x /= nq; y /= nq; z /= nq;
        float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z, wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
        float* r = R[tid];
        r[0] = w2 + x2 - y2 - z2; r[1] = 2 * xy - 2 * wz; r[2] = 2 * wy + 2 * xz;
        r[3] = 2 * wz + 2 * xy; r[4] = w2 - x2 + y2 - z2; r[5] = 2 * yz - 2 * wx;
        r[6] = 2 * xz - 2 * wy; r[7] = 2 * wx + 2 * yz; r[8] = w2 - x2 - y2 + z2;
    }
    __syncthreads();
    if (tid < 135) { int j = tid / 9 + 1, k = tid % 9; pmap[tid] = R[j][k] - ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f); }
    // v_shaped = v_template + shapedirs . beta
    for (int i = tid; i < NV * 3; i += 256) {
        float s = v_template[i];
        const float* sd = shapedirs + (size_t)i * 10;
        for (int k = 0; k < 10; ++k) s += sd[k] * beta[k];
        vs[i] = s;
    }
    __syncthreads();
    // J = J_regressor . v_shaped   (16 x 778 x 3): 48 outputs, each reduced by 5 threads
    {
        const int o = tid / 5, l = tid % 5;      // 240 active threads
        float s = 0.f;
        if (o < 48) {
            int j = o / 3, c = o % 3;
            for (int v = l; v < NV; v += 5) s += J_regressor[j * NV + v] * vs[v * 3 + c];
        }
        part[tid][0] = s;
    }
    __syncthreads();
    if (tid < 48) { float s = 0.f; for (int l = 0; l < 5; ++l) s += part[tid * 5 + l][0]; J[tid / 3][tid % 3] = s; }
    __syncthreads();
    // v_posed = v_shaped + posedirs . pose_map
    for (int i = tid; i < NV * 3; i += 256) {
        float s = vs[i];
        const float* pd = posedirs + (size_t)i * 135;
        for (int k = 0; k < 135; ++k) s += pd[k] * pmap[k];
        vs[i] = s;
    }
    // kinematic chain (3 levels below the root; serial per finger, 5 fingers in parallel would also do)
    if (tid == 0) {
        for (int j = 0; j < NJ; ++j) {
            int par = c_mano_parents[j];
            float L[12];
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) L[r * 4 + c] = R[j][r * 3 + c];
                L[r * 4 + 3] = par < 0 ? J[0][r] : (J[j][r] - J[par][r]);
            }
            if (par < 0) { for (int k = 0; k < 12; ++k) G[j][k] = L[k]; }
            else {
                const float* P = G[par];
                for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 4; ++c) {
                        float s = (P[r * 4] * L[c] + P[r * 4 + 1] * L[4 + c]) + P[r * 4 + 2] * L[8 + c];
                        if (c == 3) s += P[r * 4 + 3];
                        G[j][r * 4 + c] = s;
                    }
                }
            }
        }
        for (int j = 0; j < NJ; ++j)
            for (int r = 0; r < 3; ++r) {
                const float* g = G[j];
                float corr = (g[r * 4] * J[j][0] + g[r * 4 + 1] * J[j][1]) + g[r * 4 + 2] * J[j][2];
                G2[j][r * 4] = g[r * 4]; G2[j][r * 4 + 1] = g[r * 4 + 1]; G2[j][r * 4 + 2] = g[r * 4 + 2];
                G2[j][r * 4 + 3] = g[r * 4 + 3] - corr;
            }
    }
    __syncthreads();
    if (T_abs && tid < NJ * 16) {
        int j = tid / 16, k = tid % 16, r = k / 4, c = k % 4;
        T_abs[((size_t)b * NJ + j) * 16 + k] = r < 3 ? G[j][r * 4 + c] : (c == 3 ? 1.f : 0.f);
    }
    // skinning
    float* vo = verts + (size_t)b * NV * 3;
    for (int v = tid; v < NV; v += 256) {
        float T[12];
        for (int k = 0; k < 12; ++k) T[k] = 0.f;
        for (int j = 0; j < NJ; ++j) {
            float w = weights[v * NJ + j];
            if (w != 0.f) for (int k = 0; k < 12; ++k) T[k] += w * G2[j][k];
        }
        float x = vs[v * 3], y = vs[v * 3 + 1], z = vs[v * 3 + 2];
        for (int r = 0; r < 3; ++r) vo[v * 3 + r] = ((T[r * 4] * x + T[r * 4 + 1] * y) + T[r * 4 + 2] * z) + T[r * 4 + 3];
    }
    __syncthreads();
    if (tid < 63) {
        int k = tid / 3, c = tid % 3, src = c_mano_reorder[k];
        float val = src < 16 ? G[src][c * 4 + 3] : vo[c_mano_tips[src - 16] * 3 + c];
        joints[(size_t)b * 63 + tid] = val;
    }
}

extern "C" int ab_mano_lbs(const float* pose, const float* betas, const float* v_template, const float* shapedirs,
                           const float* posedirs, const float* J_regressor, const float* weights,
                           const float* hands_mean, int B, float* verts, float* joints, float* T_abs, void* stream) {
    if (!pose || !betas || !v_template || !shapedirs || !posedirs || !J_regressor || !weights || !hands_mean || !verts || !joints)
        return AB_EINVAL;
    if (B < 1) return AB_ESHAPE;
    mano_lbs_kernel<<<B, 256, 0, as_stream(stream)>>>(pose, betas, v_template, shapedirs, posedirs, J_regressor, weights,
                                                      hands_mean, verts, joints, T_abs);
    AB_LAUNCH_CHECK();
    return 0;
}
