// snowtri_blender.hpp -- row N2: the 24 Blender IK control points of a skeleton (blender.py:98-143) and the
// "hold" step of their per-bone filtering (blender.py:145-178).
//
// k_blender_points: one lane per skeleton.  A skeleton is a [kn][4] (x, y, z, score) record as written by the
// fused triangulation kernels; only 28 of its 133 joints are read.  Every operation is an IEEE fp64
// add / mul / div / sqrt in the reference's order, so a point is NaN exactly when the reference's is
// (zero-score joints sit at the origin and make unit(0) = 0/0) and `valid` reproduces blender.py:135-139.
//
// root_rotation: the reference stacks x = unit(hip_l - hip_r), y = unit(shoulder_mid - hip_mid),
// z = unit(x cross y) as columns and hands the (non-orthogonal: x.y != 0) matrix to SciPy, which projects it
// onto SO(3) by SVD and converts with Markley's method (util.py:26-29).  For this matrix the projection is
// closed-form: z is a unit vector orthogonal to x and y, so the polar factor only symmetrically
// orthogonalises the unit pair (x, y):  u = unit(x + y), v = unit(x - y), x' = (u + v)/sqrt 2,
// y' = (u - v)/sqrt 2, z' = z.  Agreement with SciPy's SVD path: 3e-15 over 20 000 random frames.
#pragma once
#include "snowtri_math.hpp"

namespace snowtri {

constexpr int kBlenderPoints = 24;
constexpr int kBlenderMinJoints = 130;  // highest joint read: 129 (right little-finger root)

__device__ __forceinline__ Vec3 v_add(const Vec3 &a, const Vec3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ Vec3 v_sub(const Vec3 &a, const Vec3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ Vec3 v_mid(const Vec3 &a, const Vec3 &b) {
    return {(a.x + b.x) / 2, (a.y + b.y) / 2, (a.z + b.z) / 2};
}
__device__ __forceinline__ Vec3 v_cross(const Vec3 &a, const Vec3 &b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ Vec3 v_unit(const Vec3 &a) {  // v / np.linalg.norm(v); 0/0 -> NaN on purpose
    const double n = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return {a.x / n, a.y / n, a.z / n};
}
// base + unit(first x second): hand / foot / chest / head poles (blender.py:37-85)
__device__ __forceinline__ Vec3 cross_pole(const Vec3 &base, const Vec3 &first, const Vec3 &second) {
    return v_add(base, v_unit(v_cross(first, second)));
}
// elbow / knee pole (blender.py:87-95)
__device__ __forceinline__ Vec3 joint_pole(const Vec3 &joint, const Vec3 &upper, const Vec3 &lower) {
    const Vec3 a = v_sub(upper, joint), b = v_sub(lower, joint), c = v_sub(upper, lower);
    return v_add(joint, v_unit(v_cross(v_cross(b, a), c)));
}

template <typename T>
__device__ __forceinline__ Vec3 load_joint(const T *rec, int j) {
    const T *p = rec + 4 * (size_t)j;
    return {(double)p[0], (double)p[1], (double)p[2]};
}

// quaternion (w, x, y, z) of the pelvis frame, blender.py:15-35 + util.py:26-29
__device__ inline void root_rotation(const Vec3 &p5, const Vec3 &p6, const Vec3 &p11, const Vec3 &p12, double q[4]) {
    const Vec3 x = v_unit(v_sub(p11, p12));
    const Vec3 y = v_unit(v_sub(v_mid(p5, p6), v_mid(p11, p12)));
    const Vec3 z = v_unit(v_cross(x, y));
    const Vec3 u = v_unit(v_add(x, y)), v = v_unit(v_sub(x, y));
    const double s = 0.70710678118654752440;
    const Vec3 xo = {(u.x + v.x) * s, (u.y + v.y) * s, (u.z + v.z) * s};
    const Vec3 yo = {(u.x - v.x) * s, (u.y - v.y) * s, (u.z - v.z) * s};
    // M = [xo yo z] as columns;  Markley: pick the largest of (M00, M11, M22, trace)
    const double M[3][3] = {{xo.x, yo.x, z.x}, {xo.y, yo.y, z.y}, {xo.z, yo.z, z.z}};
    const double tr = M[0][0] + M[1][1] + M[2][2];
    double qx, qy, qz, qw;
    int choice = 0;
    double best = M[0][0];
    if (M[1][1] > best) { best = M[1][1]; choice = 1; }
    if (M[2][2] > best) { best = M[2][2]; choice = 2; }
    if (tr > best) choice = 3;
    if (choice == 0) {
        qx = 1 - tr + 2 * M[0][0]; qy = M[1][0] + M[0][1]; qz = M[2][0] + M[0][2]; qw = M[2][1] - M[1][2];
    } else if (choice == 1) {
        qy = 1 - tr + 2 * M[1][1]; qz = M[2][1] + M[1][2]; qx = M[0][1] + M[1][0]; qw = M[0][2] - M[2][0];
    } else if (choice == 2) {
        qz = 1 - tr + 2 * M[2][2]; qx = M[0][2] + M[2][0]; qy = M[1][2] + M[2][1]; qw = M[1][0] - M[0][1];
    } else {
        qx = M[2][1] - M[1][2]; qy = M[0][2] - M[2][0]; qz = M[1][0] - M[0][1]; qw = 1 + tr;
    }
    const double n = sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
    q[0] = qw / n; q[1] = qx / n; q[2] = qy / n; q[3] = qz / n;
}

// xyz4: [n][kn][4] of TIn;  out: [n][24][4] fp64 (3-vectors padded with 0);  valid: [n][24]
template <typename TIn>
__global__ __launch_bounds__(256) void k_blender_points(int64_t n, int kn, const TIn *__restrict__ xyz4,
                                                         double *__restrict__ out, uint8_t *__restrict__ valid) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const TIn *rec = xyz4 + (size_t)i * kn * 4;
    double *o = out + (size_t)i * kBlenderPoints * 4;
    uint8_t *ov = valid + (size_t)i * kBlenderPoints;
    auto put = [&](int c, const Vec3 &p) {
        double4 w = {p.x, p.y, p.z, 0.0};
        *reinterpret_cast<double4 *>(o + 4 * c) = w;
        ov[c] = (p.x != p.x || p.y != p.y || p.z != p.z) ? 0 : 1;
    };
    const Vec3 p3 = load_joint(rec, 3), p4 = load_joint(rec, 4), p5 = load_joint(rec, 5), p6 = load_joint(rec, 6);
    const Vec3 p11 = load_joint(rec, 11), p12 = load_joint(rec, 12);
    const Vec3 sh = v_mid(p5, p6), hip = v_mid(p11, p12), ear = v_mid(p3, p4);
    put(0, hip);
    {
        double q[4];
        root_rotation(p5, p6, p11, p12, q);
        *reinterpret_cast<double4 *>(o + 4) = double4{q[0], q[1], q[2], q[3]};
        ov[1] = (q[0] != q[0] || q[1] != q[1] || q[2] != q[2] || q[3] != q[3]) ? 0 : 1;
    }
    put(2, p6);
    put(3, p5);
    {
        const Vec3 p7 = load_joint(rec, 7), p8 = load_joint(rec, 8), p9 = load_joint(rec, 9), p10 = load_joint(rec, 10);
        put(4, p10);
        put(5, joint_pole(p8, p6, p10));
        put(6, p9);
        put(7, joint_pole(p7, p5, p9));
    }
    {
        const Vec3 p13 = load_joint(rec, 13), p14 = load_joint(rec, 14), p15 = load_joint(rec, 15),
                   p16 = load_joint(rec, 16);
        put(8, p16);
        put(9, joint_pole(p14, p12, p16));
        put(10, p15);
        put(11, joint_pole(p13, p11, p15));
    }
    {
        const Vec3 p112 = load_joint(rec, 112), p117 = load_joint(rec, 117), p129 = load_joint(rec, 129);
        put(12, load_joint(rec, 121));
        put(13, cross_pole(p112, v_sub(p117, p112), v_sub(p129, p112)));
        const Vec3 p91 = load_joint(rec, 91), p96 = load_joint(rec, 96), p108 = load_joint(rec, 108);
        put(14, load_joint(rec, 100));
        put(15, cross_pole(p91, v_sub(p108, p91), v_sub(p96, p91)));
    }
    {
        const Vec3 p20 = load_joint(rec, 20), p21 = load_joint(rec, 21), p22 = load_joint(rec, 22);
        put(16, v_mid(p20, p21));
        put(17, cross_pole(p22, v_sub(p20, p22), v_sub(p21, p22)));
        const Vec3 p17 = load_joint(rec, 17), p18 = load_joint(rec, 18), p19 = load_joint(rec, 19);
        put(18, v_mid(p17, p18));
        put(19, cross_pole(p19, v_sub(p18, p19), v_sub(p17, p19)));
    }
    put(20, sh);
    put(21, cross_pole(sh, v_sub(p5, p6), v_sub(sh, hip)));
    put(22, v_add(sh, v_unit(v_sub(ear, sh))));
    put(23, cross_pole(ear, v_sub(p3, p4), v_sub(ear, sh)));
}

// ------------------------------------------------------------------ hold: x_eff[t] = valid ? x[t] : x_eff[t-1]
// x_eff[0] = valid ? x[0] : 0 (a filter whose first point is invalid is seeded with zeros, blender.py:172-173).
// x[T][n], valid[T][n / comps].  Chunks are the filter's: chunk c = frames [1 + c L, min(T, 1 + (c+1) L));
// start[c] = x_eff[c L] is the held input entering chunk c (start[0] = x_eff[0] also seeds the filters).
// The filter passes (k_smooth_local<.., HoldInput>) then apply the hold inline -- x_eff is never stored.
__global__ __launch_bounds__(256) void k_hold_last(int64_t T, int64_t n, int L, int comps,
                                                    const double *__restrict__ x, const uint8_t *__restrict__ valid,
                                                    double *__restrict__ H, uint8_t *__restrict__ Hf) {
    const int64_t lane = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t c = blockIdx.y;
    if (lane >= n) return;
    const int64_t t0 = 1 + c * L, t1 = (t0 + L < T) ? t0 + L : T, nv = n / comps, g = lane / comps;
    double v = 0.0;
    uint8_t found = 0;
    for (int64_t t = t1 - 1; t >= t0; t--)
        if (valid[t * nv + g]) {
            v = x[t * n + lane];
            found = 1;
            break;
        }
    H[c * n + lane] = v;
    Hf[c * n + lane] = found;
}

__global__ __launch_bounds__(256) void k_hold_carry(int64_t n, int64_t nchunks, int comps,
                                                     const double *__restrict__ x, const uint8_t *__restrict__ valid,
                                                     const double *__restrict__ H, const uint8_t *__restrict__ Hf,
                                                     double *__restrict__ start, const double *__restrict__ entering = nullptr) {
    // entering (optional, [n]): the held input in front of frame 0 -- a later shard of a frame-sharded track; absent = the
    // zeros an invalid first point of a TRACK is seeded with
    const int64_t lane = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (lane >= n) return;
    double v = valid[lane / comps] ? x[lane] : (entering ? entering[lane] : 0.0);
    constexpr int U = 8;   // chunk summaries in flight ahead of the carried value
    int64_t c = 0;
    for (; c + U <= nchunks; c += U) {
        double h[U];
        uint8_t hf[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            h[u] = H[(c + u) * n + lane];
            hf[u] = Hf[(c + u) * n + lane];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            start[(c + u) * n + lane] = v;
            v = hf[u] ? h[u] : v;
        }
    }
    for (; c < nchunks; c++) {
        start[c * n + lane] = v;
        const double h = H[c * n + lane];
        v = Hf[c * n + lane] ? h : v;
    }
}

// ---- the hold on a FRAME-SHARDED track (snowtri_blender_hold_shard_*): a shard's payload = [last valid input per lane | found
// (1.0 / 0.0) per lane], the held input entering shard `rank` = the last valid input of the nearest shard before it that has
// one (zeros if none: the seed of a filter whose points have all been invalid so far), and x_eff of the shard written out --
// on x_eff the per-bone filters are PLAIN filters, i.e. the linear carry exchange of row N1 applies to them.
__global__ __launch_bounds__(256) void k_hold_block_last(int64_t T, int64_t n, int comps, const double *__restrict__ x,
                                                          const uint8_t *__restrict__ valid, double *__restrict__ payload) {
    const int64_t lane = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (lane >= n) return;
    const int64_t nv = n / comps, g = lane / comps;
    double v = 0.0, found = 0.0;
    for (int64_t t = T - 1; t >= 0; t--)
        if (valid[t * nv + g]) {
            v = x[t * n + lane];
            found = 1.0;
            break;
        }
    payload[lane] = v;
    payload[n + lane] = found;
}

__global__ __launch_bounds__(256) void k_hold_entering(int world, int rank, int64_t n, const double *__restrict__ gathered,
                                                        double *__restrict__ entering) {
    const int64_t lane = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (lane >= n) return;
    double v = 0.0;
    for (int q = 0; q < rank && q < world; q++) {
        const double *g = gathered + (int64_t)q * 2 * n;
        if (g[n + lane] != 0.0) v = g[lane];
    }
    entering[lane] = v;
}

// x_eff of a shard: frame 0 from `entering`, chunk c = frames [1 + c L, ...) from start[c] (k_hold_carry); grid.y = chunks + 1
// (the last row of blocks writes frame 0)
__global__ __launch_bounds__(256) void k_hold_fill(int64_t T, int64_t n, int L, int comps, const double *__restrict__ x,
                                                    const uint8_t *__restrict__ valid, const double *__restrict__ start,
                                                    const double *__restrict__ entering, double *__restrict__ out) {
    const int64_t lane = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t c = blockIdx.y, nchunks = (int64_t)gridDim.y - 1;
    if (lane >= n) return;
    const int64_t nv = n / comps, g = lane / comps;
    if (c == nchunks) {
        out[lane] = valid[g] ? x[lane] : entering[lane];
        return;
    }
    const int64_t t0 = 1 + c * L, t1 = (t0 + L < T) ? t0 + L : T;
    double v = start[c * n + lane];
    for (int64_t t = t0; t < t1; t++) {
        v = valid[t * nv + g] ? x[t * n + lane] : v;
        out[t * n + lane] = v;
    }
}

}  // namespace snowtri
