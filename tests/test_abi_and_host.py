"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header declares,
the host-side mirror behaves like the reference where no kernel is involved, and the product
package never reaches into oracle/."""
import ctypes as ct
import os
import re

import numpy as np
import pytest

from conftest import ROOT, GOLDEN
from snowmocap_amd import _lib, synth
from snowmocap_amd.camera import CameraGroup
from snowmocap_amd.triangulation import SecondOrderDynamic, Human_Triangulation_Smooth, Human_Triangulation_Condense
from snowmocap_amd.util import Check_If_File_Exist, Load_Config_Json


def _header_functions():
    src = open(os.path.join(ROOT, "include", "snowtri.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(snowtri_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = _header_functions()
    assert len(names) >= 15
    assert names == _lib.exported_symbols()          # binding and header agree
    handle = ct.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"libsnowtri.so does not export {n}"
    assert _lib.lib().snowtri_version() == 100
    assert _lib.lib().snowtri_status_string(_lib.ERR_SINGULAR).decode().startswith("singular")
    assert _lib.lib().snowtri_num_candidate_slots(4, 1) == 6
    assert _lib.lib().snowtri_num_candidate_slots(16, 8) == 7680


def test_params_struct_layout_matches_header():
    assert ct.sizeof(_lib.Params) == 6 * 8 + 2 * 4
    p = _lib.make_params()
    assert (p.keypoint_score_threshold, p.distance_threshold, p.center_point_index, p.keypoint_num) == (0.5, 0.05, 18, 30)
    assert p.condense_distance_tol == 0.1     # reference signature defaults, triangulation.py:95-100


@pytest.mark.skipif(_lib.lib().snowtri_device_count() > 0, reason="GPU present")
def test_no_device_is_reported_not_faked():
    K, R, t = synth.load_rig_json()
    with pytest.raises(_lib.SnowtriError) as ei:
        _lib.Context(K, R, t)
    assert ei.value.status == _lib.ERR_NO_DEVICE


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "snowmocap_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in text and "from oracle" not in text, fn


def test_pack_frame_layout_and_dtype():
    cg = CameraGroup(camera_group_info_path=synth.FLOOR_RIG_PATH)
    assert cg.camera_num == 4 and cg.cameras[0].t.shape == (3, 1)
    rng = np.random.default_rng(0)
    a = rng.uniform(0, 1000, (133, 2)).astype(np.float32)
    s = rng.uniform(0, 8, 133).astype(np.float32)
    cg.add_human_2D_points(a, s, 0)
    cg.add_human_2D_points(a + 1, s, 2)
    cg.add_human_2D_points(a + 2, s, 2)
    kpts, n = cg.pack_frame()
    assert kpts.dtype == np.float32 and kpts.shape == (4, 2, 133, 3)
    assert n.tolist() == [1, 0, 2, 0]
    assert np.array_equal(kpts[2, 1, :, :2], a + 2) and np.array_equal(kpts[0, 0, :, 2], s)
    cg.add_human_2D_points(a.astype(np.float64), s, 1)      # one float64 array promotes the frame
    assert cg.pack_frame()[0].dtype == np.float64
    cg.clear_2D_points()
    assert cg.pack_frame()[1].tolist() == [0, 0, 0, 0]
    K, R, t = cg.rig_arrays()
    K0, R0, t0 = synth.load_rig_json()
    assert np.array_equal(K, K0) and np.array_equal(R, R0) and np.array_equal(t, t0)


def test_condense_with_fewer_than_two_candidates_is_empty_without_gpu():
    # triangulation.py:107: range(person_num - 1) is empty -> no kernel is needed, nothing emitted
    res = {"hrnet_triangulate_points": [np.zeros((5, 3))], "hrnet_triangulate_keypoint_scores": [np.ones(5)],
           "hrnet_triangulate_person_scores": [1.0]}
    out = Human_Triangulation_Condense(res, center_point_index=99)
    assert out["hrnet_triangulate_points"] == []


def test_second_order_dynamic_matches_reference_track():
    """N1 host shim against the reference's own smoothed trajectory (fixture G6)."""
    z = np.load(os.path.join(GOLDEN, "g6_smooth_blender.npz"))
    track, want = z["track"], z["smoothed"]
    prev = None
    for k in range(track.shape[0]):
        res = {"hrnet_triangulate_points": [track[k, p] for p in range(track.shape[1])],
               "hrnet_triangulate_keypoint_scores": [None] * track.shape[1],
               "hrnet_triangulate_person_scores": [None] * track.shape[1]}
        res = Human_Triangulation_Smooth(res, prev, f=float(z["f"]), z=float(z["z"]), r=float(z["r"]),
                                         delta_time=float(z["dt"]))
        prev = res
        got = np.array([np.array(p) for p in res["hrnet_triangulate_points"]])
        np.testing.assert_allclose(got, want[k], rtol=0, atol=1e-13)
    sod = SecondOrderDynamic(2.0, 0.75, 0.0, np.zeros(3))
    assert sod.k2 == pytest.approx(1.0 / (4 * np.pi * 2.0) ** 2 * 4)   # 1/(2 pi f)^2


def test_util_helpers(tmp_path):
    p = tmp_path / "out.json"
    assert Check_If_File_Exist(str(p)) == (False, str(p))
    p.write_text("{\"a\": 1}")
    assert Load_Config_Json(str(p)) == {"a": 1}
    exists, alt = Check_If_File_Exist(str(p))
    assert alt.endswith("out_0.json") and exists is False
    (tmp_path / "out_0.json").write_text("{}")
    assert Check_If_File_Exist(str(p))[1].endswith("out_1.json")


def test_blender_smooth_host_protocol():
    """Human_Triangulation_Blender_Smooth keeps the reference's stateful per-frame protocol (blender.py:145-178):
    replay fixture G8's raw control points frame by frame (no GPU involved) and compare with the reference."""
    from snowmocap_amd.blender import (Human_Triangulation_Blender_Smooth, Human_Triangulation_To_Blender_Result,
                                       CONTROL_POINT_NAMES)
    z = np.load(os.path.join(GOLDEN, "g8_blender_track.npz"))
    names = [str(n) for n in z["names"]]
    assert names == list(CONTROL_POINT_NAMES)
    arm = {n: [] for n in names}
    smo = {n: z["fzr"][i].tolist() for i, n in enumerate(names)}
    raw = z["raw"].copy()
    raw[z["valid"] == 0] = np.nan          # the fixture stores padded zeros; invalid points are NaN in the reference
    T, P = raw.shape[:2]
    prev = None
    for f in range(T):
        bl = {"blender_armature_control_points":
              [{n: raw[f, p, i, :4 if n == "root_rotation" else 3].tolist() for i, n in enumerate(names)} for p in range(P)],
              "blender_armature_control_points_scores":
              [{n: int(z["valid"][f, p, i]) for i, n in enumerate(names)} for p in range(P)]}
        sm = Human_Triangulation_Blender_Smooth(bl, arm, smo, prev, delta_time=float(z["dt"]))
        prev = sm
        for p in range(P):
            for i, n in enumerate(names):
                got = np.array(sm["blender_armature_control_points"][p][n])
                want = z["smoothed"][f, p, i, :len(got)]
                if f == 0 and not z["valid"][f, p, i]:
                    assert np.isnan(got).any()
                else:
                    np.testing.assert_allclose(got, want, rtol=0, atol=1e-13, err_msg=f"{f} {n}")
    out = Human_Triangulation_To_Blender_Result(sm)
    assert set(out) == {"armature", "score"} and len(out["armature"]) == P


def test_track_pipeline_result_formatting_is_host_only():
    """TrackPipeline.to_blender_result turns a control-point track into the list the reference dumps
    (blender.py:180-187); pure host code, also fed NumPy arrays.  A NaN pelvis quaternion (valid = 0) is the
    reference's SciPy failure and raises."""
    from snowmocap_amd.pipeline import TrackPipeline
    from snowmocap_amd.blender import CONTROL_POINT_NAMES
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(3, 2, 24, 4))
    val = np.ones((3, 2, 24), np.uint8)
    val[1, 0, 13] = 0
    out = TrackPipeline.to_blender_result(pts, val)
    assert len(out) == 3 and len(out[0]["armature"]) == 2 and list(out[0]["armature"][0]) == list(CONTROL_POINT_NAMES)
    assert len(out[0]["armature"][0]["root_rotation"]) == 4 and len(out[0]["armature"][0]["hand_r_pole"]) == 3
    assert out[1]["score"][0]["hand_r_pole"] == 0 and out[1]["score"][1]["hand_r_pole"] == 1
    np.testing.assert_array_equal(out[2]["armature"][1]["head_ik"], pts[2, 1, 22, :3])
    sub = {"root_position": [], "head_ik": []}
    assert list(TrackPipeline.to_blender_result(pts, val, sub)[0]["armature"][0]) == ["root_position", "head_ik"]
    val[2, 1, 1] = 0
    with pytest.raises(np.linalg.LinAlgError):
        TrackPipeline.to_blender_result(pts, val)


def _build_c_consumer(tmp_path):
    import subprocess
    exe = str(tmp_path / "c_abi_smoke")
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi_smoke.c"), "-o", exe, "-L", lib_dir, "-lsnowtri", "-lm",
                           "-Wl,-rpath," + lib_dir])
    return exe


def test_header_is_plain_c_and_links(tmp_path):
    """include/snowtri.h compiles as C99 and a C program links against libsnowtri.so (no C++ in the ABI)."""
    import subprocess
    exe = _build_c_consumer(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    if _lib.lib().snowtri_device_count() <= 0:
        assert out.returncode == 0 and "no device" in out.stdout


def test_every_entry_rejects_bad_arguments_under_asan(tmp_path):
    """`make asan` (host AddressSanitizer build of the C ABI) + tests/abi_badargs.c: null contexts, impossible sizes,
    bad enum values, context creation failing with SNOWTRI_ERR_NO_DEVICE -- every entry reports, none reads through a
    bad pointer.  (With a GPU the same driver goes on to a real context: tests/test_gpu_parity.py.)"""
    import shutil
    import subprocess
    csrc = os.path.join(ROOT, "snowmocap_amd", "csrc")
    so = os.path.join(csrc, "build", "libsnowtri_asan.so")
    clang = "/opt/rocm/lib/llvm/bin/clang"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which("hipcc")) or not os.path.exists(clang):
        pytest.skip("no ROCm toolchain (hipcc + its clang) on this machine: the ASan build of the C ABI cannot be made")
    subprocess.check_call(["make", "-C", csrc, "-s", "asan"])
    exe = str(tmp_path / "abi_badargs")
    subprocess.check_call([clang, "-std=c99", "-fsanitize=address", "-g", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "abi_badargs.c"), "-o", exe, so, "-Wl,-rpath," + os.path.dirname(so)])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    p = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "0 failure(s)" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
    assert "AddressSanitizer" not in p.stderr, p.stderr[-3000:]


def test_compat_snowvision_package_serves_main_py_imports():
    """compat/snowvision on PYTHONPATH: `from snowvision import *` (main.py:6) yields every name main.py uses
    (main.py:14-106), bound to the snowmocap_amd implementations -- no sys.modules alias, main.py unchanged."""
    import subprocess
    import sys
    code = ("from snowvision import *\n"
            "import snowvision, snowmocap_amd\n"
            "names = ['Load_Config_Json', 'CameraGroup', 'Load_Video', 'Check_If_File_Exist', 'Human_Triangulation',"
            " 'Human_Triangulation_Condense', 'Human_Triangulation_Smooth', 'Human_Triangulation_Blender',"
            " 'Human_Triangulation_Blender_Smooth', 'Human_Triangulation_To_Blender_Result', 'save_blender_result',"
            " 'Draw_Camera_Group', 'Draw_Skeleton']\n"
            "missing = [n for n in names if n not in globals()]\n"
            "assert not missing, missing\n"
            "assert CameraGroup is snowmocap_amd.CameraGroup and snowvision.triangulation is snowmocap_amd.triangulation\n"
            "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "compat"), ROOT]))
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "ok" in p.stdout, p.stdout + p.stderr


def test_shipped_library_carries_no_development_switch():
    """nothing told a user that the loaded .so was built without the development macros.  snowtri_build_info()
    names every build variant a binary carries: the production library none, the debug-bounds library exactly its one."""
    import ctypes as ct
    info = _lib.build_info()
    assert info["arch"] == "gfx950" and info["version"] >= 100
    if not os.environ.get("SNOWTRI_LIB"):
        assert info["variants"] == [], info
    dbg = os.path.join(ROOT, "snowmocap_amd", "libsnowtri_dbg.so")
    if os.path.exists(dbg):
        h = ct.CDLL(dbg)
        h.snowtri_build_info.restype = ct.c_char_p
        assert h.snowtri_build_info().decode().endswith("variants=SNOWTRI_DEBUG_BOUNDS,SNOWTRI_TEST_KNOBS")
    # no wrong-output switch is left in the kernel sources, and trace stamps need the experiments gate
    src = "".join(open(os.path.join(ROOT, "snowmocap_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "snowmocap_amd", "csrc"))
                  if f.endswith((".hpp", ".hip")))
    for gone in ("_NOSOLVE", "_NOFILL", "K1_REPEAT", "LEAN_NOLOOP", "LEAN_NOEPI", "STOP_AFTER_P", "SNOWTRI_MEMTEST", "SNOWTRI_COMPUTETEST"):
        assert ("#ifdef SNOWTRI" + gone not in src) and ("defined(SNOWTRI_" + gone.lstrip("_") not in src) and (gone + "  //" not in src), gone
    assert "!defined(SNOWTRI_DEV_EXPERIMENTS)" in src and "#error" in src


def test_production_library_reads_no_environment():
    """Round-5 review, item 7: eleven environment knobs selected routes in the production library.  They now exist only under
    -DSNOWTRI_TEST_KNOBS (libsnowtri_dbg.so, which the tests that force a route bind through conftest.Knobs): libsnowtri.so does
    not even import getenv, the test build does, and the experiments that were "measured no faster" are gone from the sources."""
    import subprocess
    def imports(path):
        return subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
    prod = os.path.join(ROOT, "snowmocap_amd", "libsnowtri.so")
    assert "getenv" not in imports(prod)
    dbg = os.path.join(ROOT, "snowmocap_amd", "libsnowtri_dbg.so")
    if os.path.exists(dbg):
        assert "getenv" in imports(dbg)
    csrc = os.path.join(ROOT, "snowmocap_amd", "csrc")
    src = "".join(open(os.path.join(csrc, f)).read() for f in os.listdir(csrc) if f.endswith((".hpp", ".hip")))
    assert src.count("getenv(") == 2 and "#ifdef SNOWTRI_TEST_KNOBS" in src       # the knob reader and the split switch, both fenced
    assert "k_candidate_sums_rays" not in src and "SNOWTRI_SUMS_RAYS" not in src
    assert not os.path.exists(os.path.join(csrc, "snowtri_sums_rays.hpp"))
    # the knob names the header documents are the ones the test build reads, and conftest.Knobs knows them all
    from conftest import KNOB_NAMES
    hdr = open(os.path.join(ROOT, "include", "snowtri.h")).read()
    for name in KNOB_NAMES:
        assert name in hdr and f'"{name}"' in src, name
