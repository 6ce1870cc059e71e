"""CPU restatement of row N4: keypoint-level lens undistortion (Brown-Conrady, OpenCV's 5-coefficient model
D = (k1, k2, p1, p2, k3) as stored in configs/camera_group_floor.json:53-61).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference undistorts whole images with `cv2.undistort(frame, K, D)`
before detection (main.py:52); OpenCV is not in this image and the reference holds no test or vector at this
boundary, so nothing here is checked against reference outputs.  What is restated is OpenCV's published camera
model (calib3d documentation, `projectPoints` / `undistortPoints`):

    x = (u - cx - s y)/fx,  y = (v - cy)/fy                      normalised coordinates
    r2 = x^2 + y^2,  rho = 1 + k1 r2 + k2 r2^2 + k3 r2^3
    x_d = x rho + 2 p1 x y + p2 (r2 + 2 x^2)
    y_d = y rho + p1 (r2 + 2 y^2) + 2 p2 x y

`cv2.undistort(img, K, D)` paints output pixel p_u from input pixel K.distort(K^-1 p_u); a keypoint detected in
the RAW image at p_d therefore sits at p_u = K.undistort(K^-1 p_d) in the undistorted image -- the exact inverse
of the forward model, computed here by Newton iteration to convergence.  `undistort_opencv5` restates the
fixed-point scheme of cv2.undistortPoints (5 iterations by default), kept only to show both converge to the
same point.  Tests pin the pair by round trip (distort(undistort(p)) = p to 1e-10 px).
"""
from __future__ import annotations

import numpy as np


def _split(K):
    K = np.asarray(K, dtype=np.float64)
    return K[0, 0], K[0, 1], K[0, 2], K[1, 1], K[1, 2]


def distort_normalized(x, y, D):
    k1, k2, p1, p2, k3 = [float(v) for v in np.asarray(D, dtype=np.float64).reshape(-1)[:5]]
    r2 = x * x + y * y
    rho = 1 + r2 * (k1 + r2 * (k2 + r2 * k3))
    xd = x * rho + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rho + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return xd, yd


def distort_pixels(K, D, uv):
    """Undistorted pixel -> where the raw (distorted) image shows it."""
    fx, s, cx, fy, cy = _split(K)
    uv = np.asarray(uv, dtype=np.float64)
    y = (uv[..., 1] - cy) / fy
    x = (uv[..., 0] - cx - s * y) / fx
    xd, yd = distort_normalized(x, y, D)
    return np.stack([fx * xd + s * yd + cx, fy * yd + cy], axis=-1)


def undistort_pixels(K, D, uv, iters=60):
    """Raw-image pixel -> undistorted pixel: Newton on the forward model, to convergence."""
    fx, s, cx, fy, cy = _split(K)
    k1, k2, p1, p2, k3 = [float(v) for v in np.asarray(D, dtype=np.float64).reshape(-1)[:5]]
    uv = np.asarray(uv, dtype=np.float64)
    yd = (uv[..., 1] - cy) / fy
    xd = (uv[..., 0] - cx - s * yd) / fx
    x, y = xd.copy(), yd.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        rho = 1 + r2 * (k1 + r2 * (k2 + r2 * k3))
        drho = k1 + r2 * (2 * k2 + r2 * 3 * k3)
        f1 = x * rho + 2 * p1 * x * y + p2 * (r2 + 2 * x * x) - xd
        f2 = y * rho + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y - yd
        a = rho + 2 * x * x * drho + 2 * p1 * y + 6 * p2 * x
        b = 2 * x * y * drho + 2 * p1 * x + 2 * p2 * y
        d = rho + 2 * y * y * drho + 6 * p1 * y + 2 * p2 * x
        det = a * d - b * b
        dx, dy = (d * f1 - b * f2) / det, (a * f2 - b * f1) / det
        x, y = x - dx, y - dy
        if np.max(np.abs(dx)) < 1e-17 and np.max(np.abs(dy)) < 1e-17:
            break
    return np.stack([fx * x + s * y + cx, fy * y + cy], axis=-1)


def undistort_opencv5(K, D, uv, iters=5):
    """cv2.undistortPoints' fixed-point scheme: x <- (x_d - tangential(x)) / rho(x)."""
    fx, s, cx, fy, cy = _split(K)
    k1, k2, p1, p2, k3 = [float(v) for v in np.asarray(D, dtype=np.float64).reshape(-1)[:5]]
    uv = np.asarray(uv, dtype=np.float64)
    yd = (uv[..., 1] - cy) / fy
    xd = (uv[..., 0] - cx - s * yd) / fx
    x, y = xd.copy(), yd.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        rho = 1 + r2 * (k1 + r2 * (k2 + r2 * k3))
        tx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        ty = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x, y = (xd - tx) / rho, (yd - ty) / rho
    return np.stack([fx * x + s * y + cx, fy * y + cy], axis=-1)
