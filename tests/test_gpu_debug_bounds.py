"""Device-side bounds checks ("bounds-check under a debug macro"): the kernels compiled with
-DSNOWTRI_DEBUG_BOUNDS (snowmocap_amd/libsnowtri_dbg.so, `make -C snowmocap_amd/csrc debug`, built by
__graft_entry__.build()) check every index they derive -- tile and frame ranges, LDS arena offsets, candidate slots,
descriptor / member-list positions, person fields.  A subprocess binds that library (SNOWTRI_LIB), runs the shapes of
the parity suite through the fused entry -- the fast kernel over several launch shapes, the streaming association on
small and wide rigs with ragged lists, the frames it leaves to k_frame_recompute, float64 outputs, DLT -- checks the
results against the oracle and expects snowtri_debug_faults() == 0."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

DBG = os.path.join(ROOT, "snowmocap_amd", "libsnowtri_dbg.so")

CODE = r'''
import sys, numpy as np
sys.path.insert(0, %(root)r)
from snowmocap_amd import synth, _lib
from snowmocap_amd.batch import BatchTriangulator
from oracle import oracle as orc
assert _lib.LIB_PATH.endswith("libsnowtri_dbg.so")
assert set(_lib.build_info()["variants"]) == {"SNOWTRI_DEBUG_BOUNDS", "SNOWTRI_TEST_KNOBS"}
# the mechanism itself: a kernel in which three lanes violate a check is reported as 3 faults with its code, and cleared
c0 = _lib.Context(device=0)
assert c0.debug_faults()[0] == 0
assert _lib.lib().snowtri_debug_selftest(c0.handle) == _lib.OK
n, first = c0.debug_faults()
assert n == 3 and first >> 32 == 99, (n, first)
assert c0.debug_faults() == (0, 0)
c0.close()
ran = 0
def run(K, R, t, prm, kp, npers, pout, out_dtype=np.float32, method=_lib.PAIRWISE, check=True):
    global ran
    bt = BatchTriangulator(K, R, t, prm, pout_max=pout, out_dtype=out_dtype, method=method)
    out = bt.run_host(kp, npers)
    n, first = bt.ctx.debug_faults()
    assert n == 0, "device-side bounds check failed %%d times; first: code %%d at line %%d (%%s)" %% (n, first >> 32, first & 0xffffffff, bt.ctx.last_kernel_names())
    if check:
        ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), pout)
        assert np.array_equal(out["count"], ref["count"])
        for f in range(len(ref["count"])):
            m = min(int(ref["count"][f]), pout)
            fin = np.isfinite(ref["xyz"][f, :m]) & (np.abs(ref["kscore"][f, :m]) < 1e9)[..., None]
            assert np.abs(out["xyzs"][f, :m, :, :3].astype(np.float64) - ref["xyz"][f, :m])[fin].max(initial=0.0) < 1e-5
    names = bt.ctx.last_kernel_names()
    bt.close()
    ran += 1
    return names
# the fast kernel: one frame ... many tiles per wave
for F in (1, 7, 300, 5000, 30000):
    wl = synth.config_workload(2, F, seed=5)
    K, R, t = wl["rig"]
    run(K, R, t, wl["params"], wl["kpts"], wl["n_persons"], 1, check=F <= 300)
    run(K, R, t, wl["params"], wl["kpts"], wl["n_persons"], 1, out_dtype=np.float64, check=False)
run(K, R, t, wl["params"], wl["kpts"][:2000], wl["n_persons"][:2000], 1, method=_lib.DLT, check=False)
# the streaming association and its fall-backs
PRM = dict(keypoint_score_threshold=3.0, average_score_threshold=0.3, distance_threshold=0.05, condense_distance_tol=0.3,
           condense_person_num_tol=2, condense_score_tol=0.0, center_point_index=0)
rng = np.random.default_rng(7)
for C, P, J, F in ((8, 4, 133, 40), (4, 3, 40, 9), (2, 2, 133, 5), (16, 8, 133, 4), (12, 3, 40, 6), (9, 1, 133, 3), (6, 16, 8, 2)):
    K, R, t = synth.ring_rig(C, radius=5.0)
    X = synth.make_people(rng, F, P, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(2.0, 9.0), permute_persons=True, dtype=np.float32)
    npers = npers.copy()
    npers[0, C - 1] = max(0, P - 1)
    if F > 2:
        npers[2] = 0
    prm = dict(PRM, keypoint_num=J, condense_person_num_tol=1 if C == 2 else 2)
    for pout in (P + 2, 1):
        run(K, R, t, prm, kp, npers, pout)
    run(K, R, t, prm, kp, npers, P + 2, out_dtype=np.float64, check=False)
    run(K, R, t, prm, kp, npers, P + 2, method=_lib.DLT, check=False)
    run(K, R, t, dict(prm, average_score_threshold=0.0, condense_score_tol=1.0), kp, npers, P + 2)   # many kept candidates, the mean-score filter active
wl = synth.config_workload(5, 3)
K, R, t = wl["rig"]
run(K, R, t, wl["params"], wl["kpts"], wl["n_persons"], 32)
# round 5: one detection per camera on 6-8 cameras (the lean kernels on the rolled item, float32 and float64 records, small and
# large launches), the streaming route without its candidate pass (keypoint_num < J, two slots, 12 cameras)
import os
for C in (6, 8):
    K, R, t = synth.ring_rig(C)
    for F in (5, 700, 20000):
        X = synth.make_people(rng, min(F, 200), 1)
        kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=1.0, score_range=(2.0, 8.0))
        reps = (F + len(kp) - 1) // len(kp)
        kp, npers = np.tile(kp, (reps, 1, 1, 1, 1))[:F].copy(), np.tile(npers, (reps, 1))[:F].copy()
        npers[F // 2, 1] = 0
        prm = dict(synth.default_thresholds(), condense_distance_tol=2.0)
        names = run(K, R, t, prm, kp, npers, 1, check=F <= 700)
        assert "k_fused_lean" in names, names
        names = run(K, R, t, prm, kp, npers, 1, out_dtype=np.float64, check=False)
        assert "k_fused_lean" in names, names
for C, kn, pout in ((7, 30, 1), (8, 133, 2), (12, 133, 1)):
    K, R, t = synth.ring_rig(C)
    X = synth.make_people(rng, 30, 1)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=1.0, score_range=(2.0, 8.0))
    npers = npers.copy()
    npers[3, 0] = 0
    prm = dict(synth.default_thresholds(), condense_distance_tol=0.5, keypoint_num=kn, center_point_index=0)
    run(K, R, t, prm, kp, npers, pout)
    run(K, R, t, prm, kp, npers, pout, out_dtype=np.float64, check=False)
# the candidate pass in its 512-thread shape and in its 1 024-thread shape on a small rig (one chunk, no second buffer)
for knobs in ({"SNOWTRI_SUMS_THREADS": "512"}, {"SNOWTRI_SUMS_THREADS": "1024", "SNOWTRI_SUMS_LDS_KB": "160"}):
    os.environ.update(knobs)
    for C, P, J, F in ((8, 4, 133, 20), (4, 8, 57, 6), (6, 3, 33, 6)):
        K, R, t = synth.ring_rig(C, radius=5.0)
        X = synth.make_people(rng, F, P, J=J)
        kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(2.0, 9.0), permute_persons=True, dtype=np.float32)
        npers = npers.copy()
        npers[1, C - 1] = P - 1
        kp[2, 0, 0, 3, 0] = np.nan            # a record that is not finite: the bad-row mask and the exact pass
        names = run(K, R, t, dict(PRM, keypoint_num=J), kp, npers, P + 2, check=False)
        assert "k_candidate_sums<" in names, names
    for k in knobs:
        os.environ.pop(k)
print("debug-bounds ok:", ran, "calls")
'''


def test_parity_workloads_trip_no_device_side_bounds_check():
    assert os.path.exists(DBG), f"{DBG} is missing: `make -C snowmocap_amd/csrc debug` (part of __graft_entry__.build())"
    env = dict(os.environ, SNOWTRI_LIB=DBG)
    p = subprocess.run([sys.executable, "-c", CODE % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "debug-bounds ok" in p.stdout, (p.stdout[-2000:] + p.stderr[-3000:])


def test_production_library_has_no_device_side_checks(api=None):
    import numpy as np
    from snowmocap_amd import _lib
    assert not _lib.LIB_PATH.endswith("_dbg.so")
    ctx = _lib.Context(device=0)
    assert ctx.debug_faults() == (-1, 0)
    assert _lib.lib().snowtri_debug_selftest(ctx.handle) == -1
    ctx.close()
