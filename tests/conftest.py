import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build them once, exactly as
    # __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU; ~1 min)
    so = os.path.join(ROOT, "snowmocap_amd", "libsnowtri.so")
    if not os.path.exists(so) and not os.environ.get("SNOWTRI_LIB"):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "snowmocap_amd", "csrc"), "-s"])


KNOB_NAMES = ("SNOWTRI_GENERAL_MODE", "SNOWTRI_LEAN_MODE", "SNOWTRI_LEAN_COOP", "SNOWTRI_SUMLESS_MODE", "SNOWTRI_HANDOVER_MODE",
              "SNOWTRI_HANDOVER_SEG_FRAMES", "SNOWTRI_SPLIT_SEGMENTS", "SNOWTRI_SUMS_THREADS", "SNOWTRI_SUMS_LDS_KB",
              "SNOWTRI_LEAN_TILES_PER_WAVE", "SNOWTRI_DEBUG")


class Knobs:
    """Route-forcing knobs exist only in the TEST build of the library (snowmocap_amd/libsnowtri_dbg.so: -DSNOWTRI_TEST_KNOBS,
    and device-side bounds checks with it); the production libsnowtri.so reads no environment.  `knobs.set(name, value)` binds
    the package to the test build (snowmocap_amd._lib.use_library) and sets the variable a context reads at creation; when the
    last knob is cleared -- or the test ends -- the production library is bound again.  So a test's DEFAULT-route runs go
    through the product, and only the runs that force a route go through the test build."""

    def __init__(self, monkeypatch):
        self.mp = monkeypatch
        self.prev = None
        self.active = set()

    def set(self, name, value):
        from snowmocap_amd import _lib
        assert name in KNOB_NAMES, name
        if self.prev is None:
            assert os.path.exists(_lib.TEST_LIB_PATH), "build the test library: make -C snowmocap_amd/csrc debug"
            self.prev = _lib.use_library(_lib.TEST_LIB_PATH)
            assert "SNOWTRI_TEST_KNOBS" in _lib.build_info()["variants"]
        self.mp.setenv(name, str(value))
        self.active.add(name)

    def clear(self, *names):
        for n in (names or tuple(self.active)):
            self.mp.delenv(n, raising=False)
            self.active.discard(n)
        if not self.active:
            self.restore()

    def restore(self):
        from snowmocap_amd import _lib
        if self.prev is not None:
            _lib.use_library(self.prev)
            self.prev = None


@pytest.fixture
def knobs(monkeypatch):
    k = Knobs(monkeypatch)
    yield k
    k.clear()
    k.restore()


def load_scenarios(name):
    """tests/golden/<name>.npz -> {scenario: {field: array}} (schema: tests/golden/make_golden.py)."""
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    out = {}
    for key in z.files:
        s, f = key.split("/", 1)
        out.setdefault(s, {})[f] = z[key]
    for sc in out.values():
        sc["params"] = json.loads(str(sc["params"]))
    return out


ALL_SCENARIO_FILES = ["g1_plumbing.npz", "g2_near_exact.npz", "g3_multi_person.npz", "g4_edge_cases.npz"]


def all_scenarios():
    items = []
    for fn in ALL_SCENARIO_FILES:
        for s, sc in load_scenarios(fn).items():
            items.append(pytest.param(sc, id=f"{fn[:2]}-{s}"))
    return items


def assert_scores_close(got, want, rtol=1e-9, what="score", dist_err=5e-14, nterms=1):
    """Scores are ~1/dist: compare relatively; inf/NaN patterns must match; scores above 1e11
    (dist < ~1e-13 m, pure rounding noise in the reference itself) only need to be huge or inf."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    noise = np.abs(want) > 1e11
    assert np.all((np.abs(got[noise]) > 1e10) | np.isnan(got[noise])), what
    g, w = got[~noise], want[~noise]
    assert np.array_equal(np.isnan(g), np.isnan(w)), what + ": NaN pattern"
    fin = ~np.isnan(w)
    g, w = g[fin], w[fin]
    # score = c / (1000 * dist) with c ~ O(1..10): a rounding-level error of ddist (metres, on ~5 m
    # operands) moves the score by 1000 * s^2 * ddist / c.  Budget ddist = 5e-14 m, c >= 1.
    # A mean over J such scores is bounded through mean(s^2) <= J * mean(s)^2: pass nterms=J.
    tol = rtol * np.abs(w) + dist_err * 1000.0 * nterms * w * w
    bad = ~(np.abs(g - w) <= tol)
    assert not bad.any(), f"{what}: {bad.sum()} mismatches, worst |d|/tol = {np.max(np.abs(g - w)[bad] / tol[bad]):.3g}"


def assert_xyz_close(got, want, atol, score_ref=None, what="xyz"):
    """3D joints in metres.  Where the reference itself is NaN (inf/inf in the fusion) ours must be
    NaN or come from a rounding-noise score (see assert_scores_close).  A float32 `got` is additionally allowed one unit in
    the last place of the STORAGE type at the reference's magnitude: with a wide condense_distance_tol the ghost points of
    near-parallel rays merge into clusters whose fused joints lie hundreds of metres out (soak rounds 39 / 43 / 48 / 57:
    70-780 m, float32 spacing 8e-6 ... 6e-5 m there; float64 outputs of the same calls: < 1e-8 m)."""
    ulp32 = 2.0 ** -23 if np.asarray(got).dtype == np.float32 else 0.0
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    fin = np.isfinite(want)
    if score_ref is not None:
        fin &= (np.abs(np.asarray(score_ref)) < 1e11)[..., None] & np.isfinite(np.asarray(score_ref))[..., None]
    assert np.all(np.isfinite(got[fin])), what + ": non-finite where the reference is finite"
    err = np.max(np.abs(got[fin] - want[fin])) if fin.any() else 0.0
    excess = np.max(np.abs(got[fin] - want[fin]) - ulp32 * np.abs(want[fin])) if fin.any() else 0.0
    assert excess <= atol, f"{what}: max |err| = {err:.3e} m > {atol:.1e} (+ one float32 ulp of the value)" if ulp32 else f"{what}: max |err| = {err:.3e} m > {atol:.1e}"
    return err
