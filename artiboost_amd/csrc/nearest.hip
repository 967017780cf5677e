// Hand -> object nearest-point distances of the grasp refiner:
//   anakin/artiboost/refiner.py:21-83 point2point_signed(hand_verts, verts_object) with y_normals = None, i.e. for every
//   hand vertex the Euclidean distance to its nearest object point, where the nearest neighbour comes from the
//   third-party chamfer_distance CUDA extension (un-pinned git dependency, requirements.txt; absent here) and
//   verts_object = (obj_rot @ resampled_objs[obj_idx]^T)^T (refiner.py:196-199) is a [B, 10000, 3] temporary.
// Here the object points are rotated while they are staged in LDS (the 30 MB temporary is never written); a lane owns
// two hand vertices (packed fp32 math) and the four waves of a workgroup each scan a quarter of the staged points with
// broadcast LDS reads: brute force, 2e9 point pairs per B = 256 call.
// d2 = (dx*dx + dy*dy) + dz*dz in fp32, strict '<' (first minimum wins) -- the same arithmetic as oracle/refiner_oracle.py,
// so the indices are bit-exact against the oracle; the distance is sqrt(d2) of the winner.  Optional per-vertex affine
// (the eval-mode BatchNorm1d(778) that follows, refiner.py:267) and an output row pitch so that the result lands
// directly in the feature matrix of the RefineNet MLP.
#include "common.h"

#define NN_TILE 2048          // object points staged per round: 32 KiB as float4
#define NN_THREADS 256        // 4 waves: each lane owns two hand vertices, each wave scans a quarter of the staged points
#define NN_XPB 128            // hand vertices per workgroup

typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(NN_THREADS) void nearest_dist_kernel(const float* __restrict__ x, const float* __restrict__ ypts,
                                                                  const int64_t* __restrict__ obj_idx, const float* __restrict__ rot,
                                                                  int P1, int P2, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, float* __restrict__ dist, int ld,
                                                                  int32_t* __restrict__ idx_out) {
    __shared__ float4 ys[NN_TILE];
    __shared__ float s_best[4][NN_XPB];
    __shared__ int s_idx[4][NN_XPB];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = blockIdx.x * NN_XPB + lane, i1 = i0 + 64;
    const float* yb = ypts + (size_t)(obj_idx ? obj_idx[b] : b) * P2 * 3;
    float R[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
    if (rot) {
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = rot[b * 9 + k];
    }
    const float* xa = x + ((size_t)b * P1 + min(i0, P1 - 1)) * 3;
    const float* xb = x + ((size_t)b * P1 + min(i1, P1 - 1)) * 3;
    const f32x2 px = {xa[0], xb[0]}, py = {xa[1], xb[1]}, pz = {xa[2], xb[2]};
    f32x2 best = {3.0e38f, 3.0e38f};
    int bi0 = 0, bi1 = 0;
    for (int j0 = 0; j0 < P2; j0 += NN_TILE) {
        const int n = min(NN_TILE, P2 - j0);
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += NN_THREADS) {
            const float* q = yb + (size_t)(j0 + j) * 3;
            const float qx = q[0], qy = q[1], qz = q[2];
            ys[j] = make_float4((R[0] * qx + R[1] * qy) + R[2] * qz, (R[3] * qx + R[4] * qy) + R[5] * qz,
                                (R[6] * qx + R[7] * qy) + R[8] * qz, 0.f);
        }
        __syncthreads();
        const int per = (n + 3) >> 2;                      // this wave's contiguous quarter: index order is kept inside a wave
        const int ja = wave * per, jb = min(n, ja + per);
#pragma unroll 4
        for (int j = ja; j < jb; ++j) {
            const float4 q = ys[j];                         // wave-uniform address: one broadcast read serves 128 point pairs
            const f32x2 dx = px - q.x, dy = py - q.y, dz = pz - q.z;
            const f32x2 d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2.x < best.x) { best.x = d2.x; bi0 = j0 + j; }
            if (d2.y < best.y) { best.y = d2.y; bi1 = j0 + j; }
        }
    }
    s_best[wave][lane] = best.x; s_best[wave][lane + 64] = best.y;
    s_idx[wave][lane] = bi0; s_idx[wave][lane + 64] = bi1;
    __syncthreads();
    if (threadIdx.x < NN_XPB) {
        const int i = blockIdx.x * NN_XPB + threadIdx.x;
        float bb = s_best[0][threadIdx.x];
        int bidx = s_idx[0][threadIdx.x];
        // across the four quarters of every tile: strict '<' in wave order is NOT global index order (quarters interleave
        // over tiles), so ties take the lower index explicitly -- the oracle's first minimum
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float v = s_best[w][threadIdx.x];
            const int vi = s_idx[w][threadIdx.x];
            if (v < bb || (v == bb && vi < bidx)) { bb = v; bidx = vi; }
        }
        if (i < P1) {
            float d = sqrtf(bb);
            if (scale) d = d * scale[i] + shift[i];
            dist[(size_t)b * ld + i] = d;
            if (idx_out) idx_out[(size_t)b * P1 + i] = bidx;
        }
    }
}

extern "C" int ab_nearest_dist(const float* x, const float* ypts, const int64_t* obj_idx, const float* rot, int B, int P1,
                               int P2, const float* scale, const float* shift, float* dist, int ld, int32_t* idx_out,
                               void* stream) {
    if (!x || !ypts || !dist || B < 1 || P1 < 1 || P2 < 1 || ld < P1) return AB_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return AB_EINVAL;
    nearest_dist_kernel<<<dim3((P1 + NN_XPB - 1) / NN_XPB, B), NN_THREADS, 0, as_stream(stream)>>>(
        x, ypts, obj_idx, rot, P1, P2, scale, shift, dist, ld, idx_out);
    AB_LAUNCH_CHECK();
    return 0;
}
