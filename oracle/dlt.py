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

Several detections per camera (`dlt_multi_batch`): association is the REFERENCE's -- candidates of
Human_Triangulation in list order (triangulation.py:50-93, through oracle.human_triangulation_frame), greedy
clustering of their centre joints (triangulation.py:107-130, restated below in Python), size filter
(:132-134).  Each surviving cluster is then solved per joint by the DLT above over the DISTINCT (camera, person)
observations its member candidates are made of; the person is dropped if its mean joint score is below
condense_score_tol (:150-152).
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


def _dlt_point(P, obs, kthr):
    """obs: list of (camera, u, v, s).  -> (xyz, score) or None when fewer than two cameras qualify."""
    rows, sc = [], []
    for c, u, v, s in obs:
        if s < kthr:
            continue
        rows.append(u * P[c, 2] - P[c, 0])
        rows.append(v * P[c, 2] - P[c, 1])
        sc.append(s)
    if len(sc) < 2:
        return None
    Xh = np.linalg.svd(np.array(rows))[2][-1]
    return Xh[:3] / Xh[3], float(np.mean(sc))


def greedy_clusters(centres, tol):
    """triangulation.py:107-130: seeds in list order, the last candidate never seeds, distance to the SEED,
    `dist > tol` skips (so NaN absorbs).  -> list of member-index lists (seed first, then list order)."""
    n = len(centres)
    absorbed = []
    clusters = []
    for mc in range(n - 1):
        if mc in absorbed:
            continue
        members = [mc]
        absorbed.append(mc)
        for sc in range(mc + 1, n):
            if sc in absorbed:
                continue
            dist = np.linalg.norm(centres[mc] - centres[sc])
            if dist > tol:
                continue
            members.append(sc)
            absorbed.append(sc)
        clusters.append(members)
    return clusters


def dlt_multi_batch(K, R, t, kpts, n_persons, params, max_out):
    """kpts[F,C,Pmax,J,3], n_persons[F,C], params = oracle.OrcParams -> xyzs[F,max_out,kn,4], pscore, count."""
    from . import oracle as orc
    kpts = np.asarray(kpts)
    F, C, Pmax, J, _ = kpts.shape
    K, R, t = np.asarray(K, float), np.asarray(R, float), np.asarray(t, float).reshape(C, 3)
    P = projection_matrices(K, R, t)
    kn, ci = params.keypoint_num, params.center_point_index
    out = np.zeros((F, max_out, kn, 4))
    pscore = np.zeros((F, max_out))
    count = np.zeros(F, dtype=np.int32)
    for f in range(F):
        npf = [int(v) for v in n_persons[f]]
        res = orc.human_triangulation_frame(K, R, t, kpts[f], n_persons[f], params)
        combos = [(mc, sc, pm, ps) for mc in range(C - 1) for sc in range(mc + 1, C)
                  for pm in range(npf[mc]) for ps in range(npf[sc])]
        kept = [combos[i] for i in res["enum_index"]]
        centres = [p[ci] for p in res["hrnet_triangulate_points"]]
        nout = 0
        for members in greedy_clusters(centres, params.condense_distance_tol):
            if len(members) < params.condense_person_num_tol:
                continue
            rows = sorted({(kept[m][0], kept[m][2]) for m in members} | {(kept[m][1], kept[m][3]) for m in members})
            person = np.zeros((kn, 4))
            for j in range(kn):
                obs = [(c, np.float64(kpts[f, c, p, j, 0]), np.float64(kpts[f, c, p, j, 1]),
                        np.float64(kpts[f, c, p, j, 2])) for c, p in rows]
                sol = _dlt_point(P, obs, params.keypoint_score_threshold)
                if sol is not None:
                    person[j, :3], person[j, 3] = sol
            avg = person[:, 3].mean()
            if avg < params.condense_score_tol:
                continue
            if nout < max_out:
                out[f, nout] = person
                pscore[f, nout] = avg
            nout += 1
        count[f] = nout
    return out, pscore, count
