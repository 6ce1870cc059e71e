"""N-view DLT oracle (NumPy fp64 SVD) for `method = SNOWTRI_DLT`.

TEST INFRASTRUCTURE ONLY (see oracle/snowtri_oracle.c header for the rules).

Parity status: the reference does NOT implement DLT (it triangulates camera PAIRS and
fuses midpoints), so this oracle restates the build's OWN definition and is "parity unpinned" against
the reference in general.  It is anchored to the reference only on the near-exact fixture class
(tests/golden/g2_near_exact.npz), where pairwise-midpoint and DLT agree to ~2e-7 m.

Definition (single detection per camera): for every (frame, joint) the cameras whose confidence is not
below keypoint_score_threshold contribute two rows  u*P[2]-P[0],  v*P[2]-P[1]  of the 2N x 4 matrix A,
with the world->pixel matrix  P = K [R^T | -R^T t]  (R camera->world, t camera centre: SURVEY.md §8a A0).
X_h = right singular vector of A for the smallest singular value, X = X_h[:3] / X_h[3].
Joint score = mean confidence of the contributing cameras; fewer than two cameras -> (0,0,0), score 0.
Person score = mean of the first keypoint_num joint scores; count = 1 per frame.
"""
import numpy as np


def projection_matrices(K, R, t):
    C = K.shape[0]
    P = np.zeros((C, 3, 4))
    for c in range(C):
        Rt = R[c].T
        P[c] = K[c] @ np.concatenate([Rt, -(Rt @ t[c].reshape(3, 1))], axis=1)
    return P


def dlt_batch(K, R, t, kpts, keypoint_score_threshold, keypoint_num):
    """kpts[F,C,1,J,3] -> xyzs[F,1,kn,4] (x,y,z,score), pscore[F,1], count[F]."""
    kpts = np.asarray(kpts)
    F, C, Pm, J, _ = kpts.shape
    assert Pm == 1
    P = projection_matrices(np.asarray(K, float), np.asarray(R, float), np.asarray(t, float).reshape(C, 3))
    kn = keypoint_num
    out = np.zeros((F, 1, kn, 4))
    for f in range(F):
        for j in range(kn):
            rows, sc = [], []
            for c in range(C):
                u, v, s = (np.float64(x) for x in kpts[f, c, 0, j])
                if s < keypoint_score_threshold:
                    continue
                rows.append(u * P[c, 2] - P[c, 0])
                rows.append(v * P[c, 2] - P[c, 1])
                sc.append(s)
            if len(sc) < 2:
                continue
            A = np.array(rows)
            Xh = np.linalg.svd(A)[2][-1]
            out[f, 0, j, :3] = Xh[:3] / Xh[3]
            out[f, 0, j, 3] = np.mean(sc)
    pscore = out[:, :, :, 3].mean(axis=2)
    return out, pscore, np.ones(F, dtype=np.int32)
