#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE UNMODIFIED REFERENCE.

Runs only in the build container (it needs /root/reference, which never travels to the GPU
box); the .npz files it writes are plain input/output arrays and are committed.  The reference
is imported with a stub `cv2` module (camera.py:2 imports cv2, but no code on the hot path calls
it -- SURVEY.md §8c); nothing of the reference's source is copied.

    python tests/golden/make_golden.py            # regenerates every fixture (~2 min)

Fixture classes: G1 cfg-1 plumbing, G2 near-exact, G3 multi-person,
G4 edge cases, G5 Skew_Ray_Solver unit vectors, G6 smoothing + Blender control points, G7 the main.py
sequence, G8 a Blender control-point track with invalid (NaN) points through the per-bone filters.

Scenario schema (one prefix per scenario inside an .npz):
  K[C,3,3] R[C,3,3] t[C,3]      rig (R = camera->world, t = camera centre)
  kpts[F,C,Pmax,J,3]            (u, v, score); float32 or float64 = the dtype handed to the reference
  n_persons[F,C]                persons per camera (call order of add_human_2D_points)
  params                        JSON string with the 8 threshold keys
  error[F]                      0 ok, 1 LinAlgError in Human_Triangulation, 2 IndexError in Condense
  cand_n[F] cand_pscore[F,Kmax] every frame;  cand_frames[...] + cand_xyz/cand_kscore for a subset
  cond_n[F] cond_xyz[F,Pout,kn,3] cond_kscore[F,Pout,kn] cond_pscore[F,Pout]
"""
import json
import os
import sys
import tempfile
import types
import warnings

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

REF = "/root/reference"


def import_reference():
    cv2 = types.ModuleType("cv2")
    cv2.TERM_CRITERIA_EPS = 2          # only read by ChessBoard.__init__ (camera.py:14)
    sys.modules.setdefault("cv2", cv2)
    sys.path.insert(0, REF)
    import snowvision as sv            # noqa: E402  (the reference)
    return sv


sv = import_reference()
from snowmocap_amd import synth      # noqa: E402  (our own input generator)

PARAM_KEYS = ["keypoint_score_threshold", "average_score_threshold", "distance_threshold",
              "condense_distance_tol", "condense_person_num_tol", "condense_score_tol",
              "center_point_index", "keypoint_num"]


def ref_camera_group(K, R, t):
    """Build a reference CameraGroup for an arbitrary rig through its own JSON loader."""
    info = {"camera_num": int(K.shape[0]), "camera_group_info": [
        {"cap_id": i, "frame_width": 1280, "frame_height": 720, "K": K[i].tolist(),
         "R": R[i].tolist(), "t": t[i].reshape(3, 1).tolist(), "D": [[0.0] * 5]}
        for i in range(K.shape[0])]}
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as fh:
        json.dump(info, fh)
        path = fh.name
    cg = sv.CameraGroup(camera_group_info_path=path)
    os.unlink(path)
    return cg


def run_reference(K, R, t, kpts, n_persons, params, cand_frames=()):
    """Drive the reference exactly as main.py:50-71,106 does, frame by frame."""
    F, C, Pmax, J, _ = kpts.shape
    kn = params["keypoint_num"]
    cg = ref_camera_group(K, R, t)
    tri_kw = {k: params[k] for k in PARAM_KEYS[:3]}
    con_kw = {k: params[k] for k in PARAM_KEYS[3:]}
    cands, conds, errors = [], [], np.zeros(F, dtype=np.int32)
    for f in range(F):
        cg.clear_2D_points()
        for c in range(C):
            for p in range(int(n_persons[f, c])):
                cg.add_human_2D_points(kpts[f, c, p, :, :2], kpts[f, c, p, :, 2], c)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                tri = sv.Human_Triangulation(cg, **tri_kw)
            except np.linalg.LinAlgError:
                errors[f] = 1
                cands.append(None), conds.append(None)
                continue
            try:
                con = sv.Human_Triangulation_Condense(tri, **con_kw)
            except IndexError:
                errors[f] = 2
                con = None
        cands.append(tri), conds.append(con)
    kmax = max([len(c["hrnet_triangulate_points"]) for c in cands if c] + [1])
    pout = max([len(c["hrnet_triangulate_points"]) for c in conds if c] + [1])
    out = dict(K=K, R=R, t=t.reshape(C, 3), kpts=kpts, n_persons=n_persons.astype(np.int32),
               params=json.dumps({k: params[k] for k in PARAM_KEYS}), error=errors)
    out["cand_n"] = np.array([len(c["hrnet_triangulate_points"]) if c else 0 for c in cands], dtype=np.int32)
    out["cand_pscore"] = np.zeros((F, kmax))
    for f, c in enumerate(cands):
        if c:
            out["cand_pscore"][f, :out["cand_n"][f]] = c["hrnet_triangulate_person_scores"]
    cand_frames = [f for f in cand_frames if f < F]
    out["cand_frames"] = np.array(cand_frames, dtype=np.int32)
    ckmax = max([out["cand_n"][f] for f in cand_frames] + [1])
    out["cand_xyz"] = np.zeros((len(cand_frames), ckmax, J, 3))
    out["cand_kscore"] = np.zeros((len(cand_frames), ckmax, J))
    for i, f in enumerate(cand_frames):
        if cands[f] and out["cand_n"][f]:
            n = out["cand_n"][f]
            out["cand_xyz"][i, :n] = np.stack(cands[f]["hrnet_triangulate_points"])
            out["cand_kscore"][i, :n] = np.stack(cands[f]["hrnet_triangulate_keypoint_scores"])
    out["cond_n"] = np.array([len(c["hrnet_triangulate_points"]) if c else 0 for c in conds], dtype=np.int32)
    kn_eff = max(kn, 0)
    out["cond_xyz"] = np.zeros((F, pout, kn_eff, 3))
    out["cond_kscore"] = np.zeros((F, pout, kn_eff))
    out["cond_pscore"] = np.zeros((F, pout))
    for f, c in enumerate(conds):
        if c and out["cond_n"][f]:
            n = out["cond_n"][f]
            out["cond_xyz"][f, :n] = np.stack(c["hrnet_triangulate_points"])
            out["cond_kscore"][f, :n] = np.stack(c["hrnet_triangulate_keypoint_scores"])
            out["cond_pscore"][f, :n] = c["hrnet_triangulate_person_scores"]
    return out


def save_scenarios(name, scenarios):
    flat = {}
    for sname, sc in scenarios.items():
        for k, v in sc.items():
            flat[f"{sname}/{k}"] = v
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **flat)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB, scenarios: {list(scenarios)}")


# ----------------------------------------------------------------------------------------------
def g1_plumbing():
    """cfg-1: floor rig, 4 cams x 1 person x 133 joints, sigma = 1 px, scores U(2,8), default thresholds."""
    rng = np.random.default_rng(1)
    K, R, t = synth.load_rig_json()
    params = synth.default_thresholds()
    X = synth.make_people(rng, 40, 1)
    kpts, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=1.0, score_range=(2.0, 8.0))
    sc = {"f32": run_reference(K, R, t, kpts, npers, params, cand_frames=range(4))}
    # same class with float64 (non-fp32-representable) inputs: the reference then adds scores in fp64
    X = synth.make_people(rng, 6, 1)
    kpts, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=1.0, score_range=(2.0, 8.0),
                                       dtype=np.float64)
    sc["f64"] = run_reference(K, R, t, kpts, npers, params, cand_frames=range(2))
    save_scenarios("g1_plumbing.npz", sc)


def g2_near_exact():
    """Exact projections rounded to fp32: the class where pairwise-midpoint and DLT agree (2e-7 m)."""
    rng = np.random.default_rng(2)
    K, R, t = synth.load_rig_json()
    params = synth.default_thresholds()
    X = synth.make_people(rng, 8, 1)
    kpts, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.0, score_range=(3.5, 8.0))
    sc = run_reference(K, R, t, kpts, npers, params, cand_frames=range(2))
    sc["X_true"] = X
    save_scenarios("g2_near_exact.npz", {"f32": sc})


def g3_multi_person():
    rng = np.random.default_rng(3)
    # 8-camera ring, 4 persons, permuted per-camera person order, avg_thr=1.0, ctol=0.3
    K, R, t = synth.ring_rig(8)
    params = synth.default_thresholds()
    params.update(average_score_threshold=1.0, condense_distance_tol=0.3)
    X = synth.make_people(rng, 3, 4)
    kpts, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=1.0, score_range=(3.5, 8.0),
                                       permute_persons=True)
    sc = {"ring8x4": run_reference(K, R, t, kpts, npers, params, cand_frames=[0])}
    # ragged: some cameras miss persons (n_persons varies per frame and camera)
    npers2 = npers.copy()
    npers2[0] = [4, 3, 4, 2, 4, 0, 4, 1]
    npers2[1] = [1, 1, 1, 1, 1, 1, 1, 1]
    npers2[2] = [4, 4, 0, 0, 0, 0, 0, 3]
    p2 = dict(params, condense_person_num_tol=2)
    sc["ring8_ragged"] = run_reference(K, R, t, kpts, npers2, p2, cand_frames=[0, 1])
    # 16 cameras x 8 persons with ghosts (tests ragged output + order-dependent clustering), 1 frame
    K, R, t = synth.ring_rig(16)
    X = synth.make_people(rng, 1, 8)
    kpts, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=1.0, score_range=(3.5, 8.0),
                                       permute_persons=True)
    sc["ring16x8_ghosts"] = run_reference(K, R, t, kpts, npers, dict(params), cand_frames=[])
    sc["ring16x8_tol30"] = run_reference(K, R, t, kpts, npers, dict(params, condense_person_num_tol=30),
                                         cand_frames=[])
    save_scenarios("g3_multi_person.npz", sc)


def _exact_rig():
    """3 axis-aligned cameras with K = I: all arithmetic on small integers is exact."""
    C = 3
    K = np.tile(np.eye(3), (C, 1, 1))
    R = np.tile(np.eye(3), (C, 1, 1))
    t = np.array([[0.0, 0.0, 0.0], [2.0, 0.0, 0.0], [0.0, 2.0, 0.0]])
    return K, R, t


def g4_edge_cases():
    rng = np.random.default_rng(4)
    sc = {}
    J = 20
    Kf, Rf, tf = synth.load_rig_json()
    base = dict(synth.default_thresholds(), keypoint_num=J)

    def people(F, P, rig, sigma=1.0, score_range=(3.5, 8.0), dtype=np.float32):
        X = synth.make_people(rng, F, P, J=J)
        return synth.make_keypoints(rng, *rig, X, pixel_sigma=sigma, score_range=score_range, dtype=dtype)

    # n_candidates in {0, 1}: the "last candidate never seeds" quirk -> empty output
    k, n = people(2, 1, (Kf[:2], Rf[:2], tf[:2]))
    sc["one_candidate"] = run_reference(Kf[:2], Rf[:2], tf[:2], k, n, base, cand_frames=[0])
    k, n = people(2, 1, (Kf, Rf, tf))
    n[:] = 0
    n[1, 2] = 1                                             # a single camera sees somebody
    sc["zero_candidates"] = run_reference(Kf, Rf, tf, k, n, base, cand_frames=[0])
    # a joint with every score < kthr (-> (0,0,0), score 0) and the centre joint with score 0
    k, n = people(3, 1, (Kf, Rf, tf))
    k[:, :, :, 5, 2] = 1.0
    k[1:, :, :, 0, 2] = 0.5                                 # centre joint (index 0) below threshold
    sc["zero_joint_and_centre"] = run_reference(Kf, Rf, tf, k, n, base, cand_frames=[0, 1])
    # distance gate: huge pixel noise on one camera -> dist > dthr for its pairs
    k, n = people(3, 1, (Kf, Rf, tf))
    k[:, 3, :, :, :2] += rng.normal(0, 40.0, size=k[:, 3, :, :, :2].shape).astype(np.float32)
    sc["distance_gate"] = run_reference(Kf, Rf, tf, k, n, base, cand_frames=[0])
    # condense filters active, keypoint_num < J, centre index != 0, function-signature-like params
    ring = synth.ring_rig(6)
    k, n = people(3, 3, ring)
    p = dict(base, average_score_threshold=0.5, condense_distance_tol=0.4, condense_person_num_tol=3,
             condense_score_tol=2.0, center_point_index=7, keypoint_num=12, keypoint_score_threshold=4.0)
    sc["filters_active"] = run_reference(*ring, k, n, p, cand_frames=[0])
    p = dict(p, condense_score_tol=1e9)                     # everything dropped by the score filter
    sc["score_tol_drops_all"] = run_reference(*ring, k, n, p, cand_frames=[])
    p = dict(base, condense_person_num_tol=100)             # everything dropped by the size filter
    sc["num_tol_drops_all"] = run_reference(*ring, k, n, p, cand_frames=[])
    p = dict(base, center_point_index=-1, keypoint_num=J)   # Python negative index = last joint
    sc["negative_centre_index"] = run_reference(*ring, k, n, p, cand_frames=[])
    p = dict(base, center_point_index=J + 3)                # IndexError in the reference
    sc["centre_index_oob"] = run_reference(*ring, k, n, p, cand_frames=[])
    p = dict(base, keypoint_num=J + 1)                      # IndexError in the reference
    sc["keypoint_num_oob"] = run_reference(*ring, k, n, p, cand_frames=[])
    # negative confidences: candidate mean < 0 = average_score_threshold -> candidate dropped
    k, n = people(2, 1, (Kf, Rf, tf))
    k[:, 1, :, :, 2] = -9.0
    p = dict(base, keypoint_score_threshold=-100.0)
    sc["negative_scores"] = run_reference(Kf, Rf, tf, k, n, p, cand_frames=[0])
    # exact intersection: dist == 0 -> score = inf -> inf/inf = NaN in the fusion (fp64 inputs)
    Ke, Re, te = _exact_rig()
    Xe = np.array([[1.0, 0.0, 1.0], [0.0, 1.0, 2.0], [1.0, 1.0, 4.0], [-1.0, 2.0, 1.0]])
    uv, _ = synth.project(Ke, Re, te, Xe)                   # [C,4,2] exact in fp64
    k = np.zeros((1, 3, 1, 4, 3))
    k[0, :, 0, :, :2] = uv
    k[0, :, 0, :, 2] = 4.0
    n = np.ones((1, 3), dtype=np.int32)
    p = dict(base, keypoint_num=4, center_point_index=0)
    sc["exact_intersection"] = run_reference(Ke, Re, te, k, n, p, cand_frames=[0])
    # parallel rays: H^T H exactly singular -> the reference raises LinAlgError
    k2 = k.copy()
    k2[0, :, 0, 2, :2] = 0.0                                # every camera looks straight down +z
    sc["parallel_rays"] = run_reference(Ke, Re, te, k2, n, p, cand_frames=[])
    # NaN keypoint: propagates (no exception); NaN centre distances absorb (dist > tol is False)
    k, n = people(2, 2, ring)
    k[0, 2, 1, 3, 0] = np.nan
    k[1, 0, 0, 0, 1] = np.nan                               # centre joint of one detection
    p = dict(base, average_score_threshold=0.5, condense_distance_tol=0.4)
    sc["nan_keypoint"] = run_reference(*ring, k, n, p, cand_frames=[0, 1])
    save_scenarios("g4_edge_cases.npz", sc)


def g5_skew_ray():
    rng = np.random.default_rng(5)
    n = 1000
    hm = rng.normal(size=(n, 3)); hs = rng.normal(size=(n, 3))
    tm = rng.uniform(-4, 4, size=(n, 3)); ts = rng.uniform(-4, 4, size=(n, 3))
    hm[:50] *= 1e-3; hs[50:100] *= 1e3                      # un-normalised rays of any scale
    hs[100:120] = hm[100:120] + 1e-6 * rng.normal(size=(20, 3))   # near-parallel (ill-conditioned)
    dist = np.empty(n); W = np.empty((n, 3))
    for i in range(n):
        dist[i], W[i] = sv.Skew_Ray_Solver(hm[i].reshape(3, 1), hs[i].reshape(3, 1),
                                           tm[i].reshape(3, 1), ts[i].reshape(3, 1))
    path = os.path.join(HERE, "g5_skew_ray.npz")
    np.savez_compressed(path, hm=hm, hs=hs, tm=tm, ts=ts, dist=dist, W=W)
    print("g5_skew_ray.npz", os.path.getsize(path) / 1e6, "MB")


def g6_smooth_blender():
    """Next rows N1/N2: Human_Triangulation_Smooth over a 60-frame track; Blender control points."""
    rng = np.random.default_rng(6)
    T, P, J = 60, 2, 133
    base = synth.make_people(rng, 1, P)[0]                  # [P,J,3]
    track = base[None] + np.cumsum(rng.normal(0, 0.01, size=(T, P, J, 3)), axis=0)
    th = synth.default_thresholds()
    f, z, r, dt = th["smooth_f"], th["smooth_z"], th["smooth_r"], th["smooth_delta_time"]
    prev, smoothed = None, []
    for k in range(T):
        res = {"hrnet_triangulate_points": [track[k, p].copy() for p in range(P)],
               "hrnet_triangulate_keypoint_scores": [np.ones(J) for _ in range(P)],
               "hrnet_triangulate_person_scores": [1.0] * P}
        res = sv.Human_Triangulation_Smooth(res, prev, f=f, z=z, r=r, delta_time=dt)
        prev = res
        smoothed.append(np.array([np.array(p) for p in res["hrnet_triangulate_points"]]))
    smoothed = np.stack(smoothed)
    # Blender control points (blender.py:98-143) for a few persons
    with open(os.path.join(REF, "configs/blender_armature_profile.json")) as fh:
        profile = json.load(fh)
    names = list(profile.keys())
    persons = synth.make_people(rng, 1, 5)[0]
    res = {"hrnet_triangulate_points": [persons[p] for p in range(5)],
           "hrnet_triangulate_keypoint_scores": [np.ones(J) for _ in range(5)]}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        bl = sv.Human_Triangulation_Blender(res, profile)
    ctrl = np.full((5, len(names), 4), np.nan)
    for p in range(5):
        for i, nm in enumerate(names):
            v = bl["blender_armature_control_points"][p][nm]
            ctrl[p, i, :len(v)] = v
    path = os.path.join(HERE, "g6_smooth_blender.npz")
    np.savez_compressed(path, track=track, smoothed=smoothed, f=f, z=z, r=r, dt=dt,
                        blender_persons=persons, blender_ctrl=ctrl, blender_names=np.array(names))
    print("g6_smooth_blender.npz", os.path.getsize(path) / 1e6, "MB")


def g7_pipeline():
    """The whole per-frame sequence of main.py:47-106 (minus video / pose / display) on synthetic detections:
    add points -> triangulate -> condense -> smooth -> Blender points -> Blender smooth -> result list."""
    rng = np.random.default_rng(7)
    K, R, t = synth.load_rig_json()
    th = synth.default_thresholds()
    with open(os.path.join(REF, "configs/blender_armature_profile.json")) as fh:
        arm = json.load(fh)
    with open(os.path.join(REF, "configs/blender_smooth_profile.json")) as fh:
        smo = json.load(fh)
    F = 6
    base = synth.make_people(rng, 1, 1)[0]
    X = base[None] + np.cumsum(rng.normal(0, 0.004, size=(F, 1, 133, 3)), axis=0)
    kpts, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.5, score_range=(2.5, 8.0))
    cg = ref_camera_group(K, R, t)
    prev_tri = prev_bl = None
    frames = []
    for f in range(F):
        for c in range(4):
            cg.add_human_2D_points(kpts[f, c, 0, :, :2], kpts[f, c, 0, :, 2], c)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tri = sv.Human_Triangulation(cg, keypoint_score_threshold=th["keypoint_score_threshold"],
                                         average_score_threshold=th["average_score_threshold"],
                                         distance_threshold=th["distance_threshold"])
            tri = sv.Human_Triangulation_Condense(tri, condense_distance_tol=th["condense_distance_tol"],
                                                  condense_person_num_tol=th["condense_person_num_tol"],
                                                  condense_score_tol=th["condense_score_tol"],
                                                  center_point_index=th["center_point_index"],
                                                  keypoint_num=th["keypoint_num"])
            tri = sv.Human_Triangulation_Smooth(tri, prev_tri, f=th["smooth_f"], z=th["smooth_z"], r=th["smooth_r"],
                                                delta_time=th["smooth_delta_time"])
            prev_tri = tri
            bl = sv.Human_Triangulation_Blender(tri, arm)
            bl = sv.Human_Triangulation_Blender_Smooth(bl, arm, smo, prev_bl, delta_time=th["smooth_delta_time"])
            prev_bl = bl
        frames.append(sv.Human_Triangulation_To_Blender_Result(bl))
        cg.clear_2D_points()
    path = os.path.join(HERE, "g7_pipeline.npz")
    np.savez_compressed(path, K=K, R=R, t=t, kpts=kpts, thresholds=json.dumps(th), armature=json.dumps(arm),
                        smooth=json.dumps(smo), result=json.dumps(frames))
    print("g7_pipeline.npz", os.path.getsize(path) / 1e6, "MB")


def g8_blender_track():
    """Row N2 over a track: Human_Triangulation_Blender + Human_Triangulation_Blender_Smooth (blender.py:98-178)
    frame after frame, with joints knocked out to (0,0,0) (what Condense emits for a zero-score joint) so some
    control points turn NaN -> score 0 -> the filter holds its previous input; includes an invalid FIRST frame."""
    rng = np.random.default_rng(8)
    T, P, J = 80, 2, 133
    with open(os.path.join(REF, "configs/blender_armature_profile.json")) as fh:
        arm = json.load(fh)
    with open(os.path.join(REF, "configs/blender_smooth_profile.json")) as fh:
        smo = json.load(fh)
    names = list(arm.keys())
    base = synth.make_people(rng, 1, P)[0]
    track = base[None] + np.cumsum(rng.normal(0, 0.01, size=(T, P, J, 3)), axis=0)
    # knock-outs: (frames, person, joints)
    for frames, p, joints in ((range(0, 3), 1, (112, 117, 129)),       # right hand, person 1, from frame 0
                              (range(10, 11), 0, (91, 96, 108)),         # single frame
                              (range(20, 29), 0, (19, 17, 18)),          # left foot, a run of 9
                              (range(40, 44), 1, (3, 4)),                # ears: head_ik / head_pole go (hips or shoulders at 0
                              # make the pelvis matrix NaN and the reference raises in SciPy's SVD)
                              (range(60, 62), 0, (7,)), (range(70, 80), 1, (14,))):  # elbow / knee to the end
        for f in frames:
            for j in joints:
                track[f, p, j] = 0.0
    dt = 1 / 30
    raw = np.zeros((T, P, len(names), 4))
    valid = np.zeros((T, P, len(names)), np.uint8)
    smoothed = np.zeros((T, P, len(names), 4))
    prev = None
    for f in range(T):
        res = {"hrnet_triangulate_points": [track[f, p].copy() for p in range(P)],
               "hrnet_triangulate_keypoint_scores": [np.ones(J) for _ in range(P)]}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            bl = sv.Human_Triangulation_Blender(res, arm)
            for p in range(P):
                for i, nm in enumerate(names):
                    v = bl["blender_armature_control_points"][p][nm]
                    raw[f, p, i, :len(v)] = v
                    valid[f, p, i] = bl["blender_armature_control_points_scores"][p][nm]
            sm = sv.Human_Triangulation_Blender_Smooth(bl, arm, smo, prev, delta_time=dt)
        prev = sm
        for p in range(P):
            for i, nm in enumerate(names):
                v = sm["blender_armature_control_points"][p][nm]
                smoothed[f, p, i, :len(v)] = v
    assert valid.min() == 0 and valid[0].min() == 0
    path = os.path.join(HERE, "g8_blender_track.npz")
    np.savez_compressed(path, track=track, raw=raw, valid=valid, smoothed=smoothed, dt=dt,
                        names=np.array(names), fzr=np.array([smo[n] for n in names], dtype=np.float64))
    print("g8_blender_track.npz", os.path.getsize(path) / 1e6, "MB")


def g9_pipeline_multi_person():
    """main.py:47-106 on a BASELINE configs[2]-shaped sequence (ring rig of 8 cameras, 4 persons walking, the configs[2]
    thresholds) whose person count VARIES: a person leaves for two frames (nobody lists it), a frame is empty, opposing
    cameras produce ghost persons in some frames.  The reference matches persons by list index against the filter banks of
    frame 0 (zip, triangulation.py:169-171, blender.py:152-166): pins the batched N1 / N2 with ragged counts."""
    rng = np.random.default_rng(9)
    K, R, t = synth.ring_rig(8)
    th = dict(synth.default_thresholds(), average_score_threshold=1.0, condense_distance_tol=0.3)
    with open(os.path.join(REF, "configs/blender_armature_profile.json")) as fh:
        arm = json.load(fh)
    with open(os.path.join(REF, "configs/blender_smooth_profile.json")) as fh:
        smo = json.load(fh)
    F, P = 14, 4
    base = synth.make_people(rng, 1, P)[0]
    X = base[None] + np.cumsum(rng.normal(0, 0.004, size=(F, P, 133, 3)), axis=0)
    kpts, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=1.0, score_range=(3.5, 8.0))
    npers = npers.copy()
    npers[4:6] = 3              # the last person is listed by nobody in frames 4-5
    npers[8] = 0                # an empty frame
    npers[10, :3] = 3           # three cameras miss the last person: a partly seen person
    kpts[11, :, 1] = kpts[11, :, 3]   # frame 11: person 1 is a second detection of person 3 (a merged / duplicate person)
    cg = ref_camera_group(K, R, t)
    prev_tri = prev_bl = None
    frames, counts, tracked = [], [], []
    for f in range(F):
        for c in range(8):
            for p in range(int(npers[f, c])):
                cg.add_human_2D_points(kpts[f, c, p, :, :2], kpts[f, c, p, :, 2], c)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tri = sv.Human_Triangulation(cg, keypoint_score_threshold=th["keypoint_score_threshold"],
                                         average_score_threshold=th["average_score_threshold"],
                                         distance_threshold=th["distance_threshold"])
            tri = sv.Human_Triangulation_Condense(tri, condense_distance_tol=th["condense_distance_tol"],
                                                  condense_person_num_tol=th["condense_person_num_tol"],
                                                  condense_score_tol=th["condense_score_tol"],
                                                  center_point_index=th["center_point_index"],
                                                  keypoint_num=th["keypoint_num"])
            counts.append(len(tri["hrnet_triangulate_points"]))
            tri = sv.Human_Triangulation_Smooth(tri, prev_tri, f=th["smooth_f"], z=th["smooth_z"], r=th["smooth_r"],
                                                delta_time=th["smooth_delta_time"])
            prev_tri = tri
            tracked.append(len(tri["hrnet_triangulate_points"]))
            bl = sv.Human_Triangulation_Blender(tri, arm)
            bl = sv.Human_Triangulation_Blender_Smooth(bl, arm, smo, prev_bl, delta_time=th["smooth_delta_time"])
            prev_bl = bl
        frames.append(sv.Human_Triangulation_To_Blender_Result(bl))
        cg.clear_2D_points()
    print("g9 persons per frame after condense:", counts, "tracked:", tracked)
    assert len(set(counts)) >= 3 and max(counts) > counts[0] and min(counts) == 0
    path = os.path.join(HERE, "g9_pipeline_multi.npz")
    np.savez_compressed(path, K=K, R=R, t=t, kpts=kpts, n_persons=npers, thresholds=json.dumps(th), armature=json.dumps(arm),
                        smooth=json.dumps(smo), result=json.dumps(frames), counts=np.array(counts), tracked=np.array(tracked))
    print("g9_pipeline_multi.npz", os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9"]
    fns = dict(g1=g1_plumbing, g2=g2_near_exact, g3=g3_multi_person, g4=g4_edge_cases,
               g5=g5_skew_ray, g6=g6_smooth_blender, g7=g7_pipeline, g8=g8_blender_track, g9=g9_pipeline_multi_person)
    for w in which:
        fns[w]()
