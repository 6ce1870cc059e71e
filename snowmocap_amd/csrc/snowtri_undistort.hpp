// snowtri_undistort.hpp -- row N4: keypoint-level lens undistortion, so detections made on RAW frames can feed
// the triangulation (the reference undistorts whole images instead, main.py:52: cv2.undistort(frame, K, D)).
//
// Model: OpenCV's 5-coefficient Brown-Conrady, D = (k1, k2, p1, p2, k3) (camera_group_floor.json:53-61):
//   x_d = x rho + 2 p1 x y + p2 (r2 + 2 x^2),  y_d = y rho + p1 (r2 + 2 y^2) + 2 p2 x y,
//   rho = 1 + k1 r2 + k2 r2^2 + k3 r2^3,  r2 = x^2 + y^2  (normalised coordinates).
// cv2.undistort paints output pixel p_u from input pixel K.distort(K^-1 p_u), so a raw-image keypoint p_d maps
// to p_u = K.undistort(K^-1 p_d): the exact inverse, here by Newton on the 2x2 system (symmetric Jacobian),
// iterated until every lane of the wave has converged (NaN inputs stop at the bound).
// Three to five iterations from x = x_d reach 5e-13 px on the shipped rig (55 px of distortion at the corners);
// OpenCV's own undistortPoints default (5 fixed-point sweeps) stops 0.15 px short of that point.
// One lane per observation, (u, v) rewritten and the score copied: 24 B of traffic for ~300 flop --
// fp64-VALU-bound like the triangulation itself, hence also fusable into its ray construction.
#pragma once
#include "snowtri_kernels.hpp"

namespace snowtri {

constexpr int kLensStride = 16;  // doubles per camera: fx s cx fy cy 1/fx 1/fy k1 k2 p1 p2 k3 (pad)
constexpr int kUndistortIters = 8;   // upper bound; the shipped lenses need 3 (image centre) to 5 (corners)

struct Lens {
    double fx, s, cx, fy, cy, ifx, ify, k1, k2, p1, p2, k3;
};

__device__ __forceinline__ Lens load_lens(const double *__restrict__ L) {
    return Lens{L[0], L[1], L[2], L[3], L[4], L[5], L[6], L[7], L[8], L[9], L[10], L[11]};
}

__device__ __forceinline__ void undistort_pixel(const Lens &q, double u, double v, double &uo, double &vo) {
    const double yd = (v - q.cy) * q.ify;
    const double xd = (u - q.cx - q.s * yd) * q.ifx;
    double x = xd, y = yd;
#pragma unroll 1
    for (int it = 0; it < kUndistortIters; it++) {
        const double r2 = fma(x, x, y * y);
        const double rho = fma(r2, fma(r2, fma(r2, q.k3, q.k2), q.k1), 1.0);
        const double drho = fma(r2, fma(r2, 3.0 * q.k3, 2.0 * q.k2), q.k1);
        const double xy2 = 2.0 * x * y;
        const double f1 = fma(x, rho, fma(q.p1, xy2, q.p2 * fma(2.0 * x, x, r2))) - xd;
        const double f2 = fma(y, rho, fma(q.p2, xy2, q.p1 * fma(2.0 * y, y, r2))) - yd;
        const double a = fma(2.0 * x * x, drho, rho) + fma(2.0 * q.p1, y, 6.0 * q.p2 * x);
        const double b = fma(xy2, drho, 2.0 * fma(q.p1, x, q.p2 * y));
        const double d = fma(2.0 * y * y, drho, rho) + fma(6.0 * q.p1, y, 2.0 * q.p2 * x);
        const double idet = rcp_nr2(fma(a, d, -b * b));
        const double dx = (d * f1 - b * f2) * idet, dy = (a * f2 - b * f1) * idet;
        x -= dx;
        y -= dy;
        // Newton converges quadratically: a step below 1e-8 (normalised units) leaves an error of ~1e-16.
        // Wave-uniform exit: the joints of one detection sit close together and need the same step count.
        if (__all(fmax(fabs(dx), fabs(dy)) < 1e-8)) break;
    }
    uo = fma(q.fx, x, fma(q.s, y, q.cx));
    vo = fma(q.fy, y, q.cy);
}

// kpts [F][C][per_cam][3] (per_cam = Pmax * J observations of one camera in one frame); out may alias in.
template <typename T>
__global__ __launch_bounds__(256) void k_undistort(int64_t n_obs, int C, int per_cam, const double *__restrict__ lens,
                                                    const T *__restrict__ in, T *__restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n_obs) return;
    const int c = (int)((i / per_cam) % C);
    const Lens q = load_lens(lens + (size_t)c * kLensStride);
    const Kp3<T> kp = reinterpret_cast<const Kp3<T> *>(in)[i];
    double uo, vo;
    undistort_pixel(q, (double)kp.u, (double)kp.v, uo, vo);
    Kp3<T> o;
    o.u = (T)uo;
    o.v = (T)vo;
    o.s = kp.s;
    reinterpret_cast<Kp3<T> *>(out)[i] = o;
}

}  // namespace snowtri
