"""ctypes binding of libsnowtri.so (the C ABI declared in include/snowtri.h).

This is the binding a SnowMocap maintainer would add: no torch types, plain
pointers and sizes.  There is NO CPU fallback: if the shared library is missing or no MI355X is
visible, calls raise -- the product path never routes through oracle/ or NumPy.
"""
from __future__ import annotations

import ctypes as ct
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SNOWTRI_LIB") or os.path.join(_HERE, "libsnowtri.so")   # override: A/B builds

OK, ERR_BAD_ARG, ERR_BAD_INDEX, ERR_HIP, ERR_SINGULAR, ERR_OVERFLOW, ERR_NO_DEVICE = range(7)
F32, F64 = 0, 1
HOST, DEVICE = 0, 1
PAIRWISE, DLT = 0, 1
FLAG_SINGULAR, FLAG_OVERFLOW, FLAG_FASTPATH = 1, 2, 4
CALL_NO_ZERO_FILL = 1          # snowtri_triangulate_condense_ex: the slots behind out_count[f] are left unwritten
TEST_LIB_PATH = os.path.join(_HERE, "libsnowtri_dbg.so")   # -DSNOWTRI_DEBUG_BOUNDS -DSNOWTRI_TEST_KNOBS (tests only: use_library)


class SnowtriError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = _lib.snowtri_status_string(status).decode() if _lib is not None else str(status)
        detail = _lib.snowtri_last_error().decode() if (_lib is not None and status == ERR_HIP) else ""
        super().__init__(f"{where}: {msg}" + (f" [{detail}]" if detail else ""))


class Params(ct.Structure):
    """snowtri_params: thresholds of Human_Triangulation / _Condense (triangulation.py:50,95-100)."""
    _fields_ = [("keypoint_score_threshold", ct.c_double),
                ("average_score_threshold", ct.c_double),
                ("distance_threshold", ct.c_double),
                ("condense_distance_tol", ct.c_double),
                ("condense_person_num_tol", ct.c_double),
                ("condense_score_tol", ct.c_double),
                ("center_point_index", ct.c_int32),
                ("keypoint_num", ct.c_int32)]


def make_params(keypoint_score_threshold=0.5, average_score_threshold=0.0, distance_threshold=0.05,
                condense_distance_tol=0.1, condense_person_num_tol=0, condense_score_tol=0.0,
                center_point_index=18, keypoint_num=30, **_ignored):
    """Defaults are the reference's function-signature defaults (triangulation.py:50,95-100)."""
    return Params(float(keypoint_score_threshold), float(average_score_threshold),
                  float(distance_threshold), float(condense_distance_tol),
                  float(condense_person_num_tol), float(condense_score_tol),
                  int(center_point_index), int(keypoint_num))


_c_p = ct.c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)  -- one entry per symbol declared in include/snowtri.h
    "snowtri_version": (ct.c_int, []),
    "snowtri_status_string": (ct.c_char_p, [ct.c_int]),
    "snowtri_last_error": (ct.c_char_p, []),
    "snowtri_device_count": (ct.c_int, []),
    "snowtri_build_info": (ct.c_char_p, []),
    "snowtri_ctx_overrides": (ct.c_char_p, [_c_p]),
    "snowtri_ctx_set_overlap": (ct.c_int, [_c_p, ct.c_int]),
    "snowtri_ctx_join": (ct.c_int, [_c_p, _c_p]),
    "snowtri_ctx_set_split": (ct.c_int, [_c_p, ct.c_int]),
    "snowtri_last_stream_counts": (ct.c_int, [_c_p, ct.POINTER(ct.c_int64 * 3)]),
    "snowtri_ctx_stream_probes": (ct.c_int, [_c_p, ct.POINTER(ct.c_int64 * 3)]),
    "snowtri_ctx_create": (ct.c_int, [ct.c_int32, _c_p, _c_p, _c_p, ct.c_int, ct.POINTER(_c_p)]),
    "snowtri_ctx_destroy": (ct.c_int, [_c_p]),
    "snowtri_ctx_num_cameras": (ct.c_int, [_c_p]),
    "snowtri_ctx_ray_matrices": (ct.c_int, [_c_p, _c_p]),
    "snowtri_ctx_synchronize": (ct.c_int, [_c_p]),
    "snowtri_fastmath_probe": (ct.c_int, [_c_p, ct.c_int64, _c_p, _c_p, _c_p, _c_p]),
    "snowtri_fastmath_probe_raw": (ct.c_int, [_c_p, ct.c_int64, _c_p, _c_p, _c_p]),
    "snowtri_calib_stream": (ct.c_int, [_c_p, _c_p, ct.c_int64, _c_p, ct.c_int64, _c_p]),
    "snowtri_rays_from_pixels": (ct.c_int, [_c_p, ct.c_int32, ct.c_int64, _c_p, _c_p]),
    "snowtri_skew_ray_batch": (ct.c_int, [_c_p, ct.c_int64, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p,
                                          ct.POINTER(ct.c_int64)]),
    "snowtri_triangulate": (ct.c_int, [_c_p, ct.c_int64, ct.c_int32, ct.c_int32, _c_p, ct.c_int, _c_p,
                                       ct.POINTER(Params), _c_p, _c_p, _c_p, _c_p, ct.c_int, _c_p]),
    "snowtri_num_candidate_slots": (ct.c_int64, [ct.c_int32, ct.c_int32]),
    "snowtri_condense": (ct.c_int, [_c_p, ct.c_int64, ct.c_int32, ct.c_int32, _c_p, _c_p, _c_p,
                                    ct.POINTER(Params), ct.c_int32, _c_p, _c_p, _c_p, _c_p, _c_p,
                                    ct.c_int, _c_p]),
    "snowtri_candidates_token": (ct.c_int64, [_c_p]),
    "snowtri_condense_resident": (ct.c_int, [_c_p, ct.c_int64, ct.POINTER(Params), ct.c_int32, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "snowtri_triangulate_condense": (ct.c_int, [_c_p, ct.c_int64, ct.c_int32, ct.c_int32, _c_p, ct.c_int,
                                                _c_p, ct.POINTER(Params), ct.c_int, ct.c_int32, _c_p,
                                                _c_p, ct.c_int, _c_p, _c_p, ct.c_int, _c_p]),
    "snowtri_triangulate_condense_ex": (ct.c_int, [_c_p, ct.c_int64, ct.c_int32, ct.c_int32, _c_p, ct.c_int,
                                                   _c_p, ct.POINTER(Params), ct.c_int, ct.c_int32, _c_p,
                                                   _c_p, ct.c_int, _c_p, _c_p, ct.c_int, _c_p, ct.c_uint32]),
    "snowtri_smooth_track": (ct.c_int, [_c_p, ct.c_int64, ct.c_int64, _c_p, ct.c_double, ct.c_double, ct.c_double,
                                        ct.c_double, _c_p, ct.c_int, _c_p]),
    "snowtri_smooth_joint_track": (ct.c_int, [_c_p, ct.c_int64, ct.c_int64, _c_p, ct.c_double, ct.c_double, ct.c_double,
                                              ct.c_double, _c_p, ct.c_int, _c_p]),
    "snowtri_smooth_coeffs": (ct.c_int, [ct.c_double, ct.c_double, ct.c_double, ct.c_double, _c_p]),
    "snowtri_smooth_shard_local": (ct.c_int, [_c_p, ct.c_int64, ct.c_int64, _c_p, ct.c_int, ct.c_double, ct.c_double,
                                              ct.c_double, ct.c_double, _c_p, _c_p, ct.c_int, _c_p]),
    "snowtri_smooth_shard_fix": (ct.c_int, [_c_p, ct.c_int64, ct.c_int64, ct.c_int, _c_p, ct.c_double, ct.c_double,
                                            ct.c_double, ct.c_double, _c_p, ct.c_int, _c_p]),
    "snowtri_smooth_shard_reduce": (ct.c_int, [_c_p, ct.c_int64, ct.c_int64, _c_p, ct.c_int, ct.c_double, ct.c_double, ct.c_double, ct.c_double, _c_p, _c_p]),
    "snowtri_smooth_shard_scan": (ct.c_int, [_c_p, ct.c_int64, ct.c_int64, _c_p, ct.c_int, _c_p, ct.c_double, ct.c_double, ct.c_double, ct.c_double, _c_p, _c_p]),
    "snowtri_blender_smooth_shard_reduce": (ct.c_int, [_c_p, ct.c_int64, ct.c_int64, _c_p, ct.c_int, _c_p, ct.c_double, _c_p, _c_p]),
    "snowtri_blender_smooth_shard_scan": (ct.c_int, [_c_p, ct.c_int64, ct.c_int64, _c_p, ct.c_int, _c_p, _c_p, ct.c_double, _c_p, _c_p]),
    "snowtri_smooth_shard_combine": (ct.c_int, [_c_p, ct.c_int32, ct.c_int32, ct.c_int64, _c_p, ct.c_double, ct.c_double,
                                                ct.c_double, ct.c_double, _c_p, ct.c_int, _c_p]),
    "snowtri_blender_points": (ct.c_int, [_c_p, ct.c_int64, ct.c_int32, _c_p, ct.c_int, _c_p, _c_p, ct.c_int, _c_p]),
    "snowtri_blender_smooth": (ct.c_int, [_c_p, ct.c_int64, ct.c_int64, _c_p, _c_p, _c_p, ct.c_double, _c_p,
                                          ct.c_int, _c_p]),
    "snowtri_blender_hold_shard_last": (ct.c_int, [_c_p, ct.c_int64, ct.c_int64, _c_p, _c_p, _c_p, _c_p]),
    "snowtri_blender_hold_shard_apply": (ct.c_int, [_c_p, ct.c_int32, ct.c_int32, ct.c_int64, ct.c_int64, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "snowtri_blender_smooth_shard_local": (ct.c_int, [_c_p, ct.c_int64, ct.c_int64, _c_p, ct.c_int, _c_p, ct.c_double, _c_p, _c_p, _c_p]),
    "snowtri_blender_smooth_shard_combine": (ct.c_int, [_c_p, ct.c_int32, ct.c_int32, ct.c_int64, _c_p, _c_p, ct.c_double, _c_p, _c_p]),
    "snowtri_blender_smooth_shard_fix": (ct.c_int, [_c_p, ct.c_int64, ct.c_int64, ct.c_int, _c_p, _c_p, ct.c_double, _c_p, _c_p]),
    "snowtri_ctx_set_distortion": (ct.c_int, [_c_p, _c_p]),
    "snowtri_undistort_keypoints": (ct.c_int, [_c_p, ct.c_int64, ct.c_int32, ct.c_int32, _c_p, _c_p, ct.c_int,
                                               ct.c_int, _c_p]),
    "snowtri_last_kernel_ms": (ct.c_int, [_c_p, ct.POINTER(ct.c_float * 2)]),
    "snowtri_set_timing": (ct.c_int, [_c_p, ct.c_int]),
    "snowtri_timing_collect": (ct.c_int, [_c_p, _c_p, ct.c_int32]),
    "snowtri_last_slow_frames": (ct.c_int64, [_c_p]),
    "snowtri_last_handover_persons": (ct.c_int64, [_c_p, _c_p]),
    "snowtri_last_kernel_names": (ct.c_char_p, [_c_p]),
    "snowtri_debug_faults": (ct.c_int64, [_c_p, ct.POINTER(ct.c_uint64)]),
    "snowtri_debug_selftest": (ct.c_int, [_c_p]),
}

_lib = None
_loaded = {}      # path -> bound CDLL handle (a process may hold the production library and the test build side by side)


def exported_symbols():
    return sorted(_SIGNATURES)


def _load(path):
    handle = _loaded.get(path)
    if handle is None:
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  snowmocap_amd has no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as
        # /opt/rocm's).  Loading it FIRST makes the dynamic linker bind libsnowtri.so to that copy,
        # so torch streams / device pointers and our kernels share one runtime.  Loading ours first
        # would leave torch with a second runtime that sees no GPU.
        if os.environ.get("SNOWTRI_NO_TORCH", "0") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        handle = ct.CDLL(path)              # (RTLD_LOCAL: two builds of the library do not see each other's symbols)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _loaded[path] = handle
    return handle


def lib():
    """Load libsnowtri.so (built by __graft_entry__.build() / `make -C snowmocap_amd/csrc`)."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


def use_library(path=None):
    """Bind the package to another build of the library -- the TEST build libsnowtri_dbg.so (device-side bounds checks + the
    route-forcing knobs, which the production library does not contain) -- or back to the production one (path=None).
    Returns the path that was bound before.  Contexts keep the library they were created with (Context.L), so objects made
    under one binding stay valid under the next; everything created afterwards uses the new one.  Tests only
    (tests/conftest.py::knob_lib)."""
    global _lib, LIB_PATH
    prev = LIB_PATH
    LIB_PATH = path or os.environ.get("SNOWTRI_LIB") or os.path.join(_HERE, "libsnowtri.so")
    _lib = _load(LIB_PATH)
    return prev


def build_info():
    """{"version": int, "arch": str, "variants": [build variants of the loaded binary]} (snowtri_build_info)."""
    raw = lib().snowtri_build_info().decode()
    d = dict(kv.split("=", 1) for kv in raw.split(";"))
    return {"version": int(d["version"]), "arch": d["arch"], "variants": [v for v in d.get("variants", "").split(",") if v]}


def check(status, where):
    if status != OK:
        raise SnowtriError(status, where)


def ptr(a):
    """void* of a NumPy array (or None)."""
    if a is None:
        return None
    return ct.c_void_p(a.ctypes.data)


def dtype_code(dt):
    dt = np.dtype(dt)
    if dt == np.float32:
        return F32
    if dt == np.float64:
        return F64
    raise TypeError(f"snowtri supports float32/float64 I/O, not {dt}")


class Context:
    """Owner of a snowtri_ctx (rig constants + device scratch).  Not thread-safe."""

    def __init__(self, K=None, R=None, t=None, device=0):
        L = self.L = lib()                  # the library this context belongs to (use_library may rebind the package later)
        if L.snowtri_device_count() <= 0:
            raise SnowtriError(ERR_NO_DEVICE, "snowtri_ctx_create")
        if K is None:
            C = 0
            Kc = Rc = tc = None
        else:
            Kc = np.ascontiguousarray(K, dtype=np.float64).reshape(-1, 9)
            C = Kc.shape[0]
            Rc = np.ascontiguousarray(R, dtype=np.float64).reshape(C, 9)
            tc = np.ascontiguousarray(t, dtype=np.float64).reshape(C, 3)
        h = ct.c_void_p()
        rc = L.snowtri_ctx_create(C, ptr(Kc), ptr(Rc), ptr(tc), int(device), ct.byref(h))
        if rc == ERR_SINGULAR:
            raise np.linalg.LinAlgError("Singular matrix")     # np.linalg.inv(K), camera.py:242
        check(rc, "snowtri_ctx_create")
        self.handle = h
        self.C = C
        self.device = int(device)

    def close(self):
        if getattr(self, "handle", None):
            self.L.snowtri_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(self.L.snowtri_ctx_synchronize(self.handle), "snowtri_ctx_synchronize")

    def ray_matrices(self):
        M = np.empty((self.C, 9))
        check(self.L.snowtri_ctx_ray_matrices(self.handle, ptr(M)), "snowtri_ctx_ray_matrices")
        return M.reshape(self.C, 3, 3)

    def set_distortion(self, D):
        """Row N4: per-camera lens coefficients D[C, 5] = (k1, k2, p1, p2, k3) (Camera.D)."""
        D2 = np.asarray(D, dtype=np.float64).reshape(self.C, -1)
        if D2.shape[1] > 5:
            if np.any(D2[:, 5:] != 0.0):
                raise ValueError("only the 5-coefficient lens model (k1, k2, p1, p2, k3) is supported")
            D2 = D2[:, :5]
        Dc = np.zeros((self.C, 5), dtype=np.float64)          # OpenCV's shorter forms (4 coefficients) mean k3 = 0
        Dc[:, :D2.shape[1]] = D2
        check(self.L.snowtri_ctx_set_distortion(self.handle, ptr(Dc)), "snowtri_ctx_set_distortion")

    def undistort_keypoints(self, kpts):
        """Raw-image keypoints kpts[F, C, Pmax, J, 3] -> the same array with (u, v) moved to the undistorted
        image (scores untouched); host arrays, float32 or float64."""
        a = np.asarray(kpts)
        if a.dtype != np.float32:
            a = a.astype(np.float64, copy=False)
        a = np.ascontiguousarray(a)
        F, C, Pmax, J, three = a.shape
        assert C == self.C and three == 3
        out = np.empty_like(a)
        check(self.L.snowtri_undistort_keypoints(self.handle, F, Pmax, J, ptr(a), ptr(out), dtype_code(a.dtype), HOST,
                                                None), "snowtri_undistort_keypoints")
        return out

    def set_timing(self, enabled=True, attach=False):
        """attach: single-kernel calls carry the event pair on their dispatch (the kernel's own begin / end) instead of being bracketed."""
        check(self.L.snowtri_set_timing(self.handle, (2 if attach else 1) if enabled else 0), "snowtri_set_timing")

    def last_kernel_ms(self):
        arr = (ct.c_float * 2)()
        check(self.L.snowtri_last_kernel_ms(self.handle, ct.byref(arr)), "snowtri_last_kernel_ms")
        return float(arr[0]), float(arr[1])

    def timing_collect(self, cap=1024):
        """Durations (ms) of the fused calls recorded since the last collect (timing must be enabled)."""
        arr = (ct.c_float * cap)()
        n = self.L.snowtri_timing_collect(self.handle, arr, cap)
        if n < 0:
            raise SnowtriError(ERR_HIP, "snowtri_timing_collect")
        return [float(arr[i]) for i in range(n)]

    def overrides(self):
        """Test knobs (environment) this context was created under: "" when it runs the defaults."""
        return (self.L.snowtri_ctx_overrides(self.handle) or b"").decode()

    def set_overlap(self, n_streams):
        """Overlap mode: device calls of the fused entry rotate over n internal streams (1 = off); join() before reading results."""
        check(self.L.snowtri_ctx_set_overlap(self.handle, int(n_streams)), "snowtri_ctx_set_overlap")

    def set_split(self, segments):
        """Segments of one multi-person call (snowtri_ctx_set_split): 1 = the caller's stream only, >= 2 forced, 0 = the default policy."""
        check(self.L.snowtri_ctx_set_split(self.handle, int(segments)), "snowtri_ctx_set_split")

    def join(self, stream=None):
        """`stream` (a HIP stream handle, default the null stream) waits for every overlapped call issued since the last join."""
        check(self.L.snowtri_ctx_join(self.handle, ct.c_void_p(stream) if stream else None), "snowtri_ctx_join")

    def last_stream_counts(self):
        """(frames past the association's first launch, frames with an exactly re-done candidate sum, frames left to
        k_frame_recompute) of the last multi-person call's last segment; (-1, -1, -1) if it did not take the streaming route."""
        arr = (ct.c_int64 * 3)()
        check(self.L.snowtri_last_stream_counts(self.handle, ct.byref(arr)), "snowtri_last_stream_counts")
        return int(arr[0]), int(arr[1]), int(arr[2])

    def stream_probes(self):
        """(probes run, internal streams discarded, verdict on the stream kept last: 1 = runs beside the others, 0 = no
        candidate did, -1 = no internal stream yet) -- see snowtri_ctx_stream_probes."""
        arr = (ct.c_int64 * 3)()
        check(self.L.snowtri_ctx_stream_probes(self.handle, ct.byref(arr)), "snowtri_ctx_stream_probes")
        return int(arr[0]), int(arr[1]), int(arr[2])

    def last_slow_frames(self):
        return int(self.L.snowtri_last_slow_frames(self.handle))

    def last_kernel_names(self):
        """Template names of the kernels the last fused call launched, in launch order."""
        return (self.L.snowtri_last_kernel_names(self.handle) or b"").decode()

    def debug_faults(self):
        """(violated device-side bounds checks since the last call, code << 32 | line of the first); (-1, 0) unless the
        library is the -DSNOWTRI_DEBUG_BOUNDS build (libsnowtri_dbg.so)."""
        first = ct.c_uint64(0)
        n = int(self.L.snowtri_debug_faults(self.handle, ct.byref(first)))
        return n, int(first.value)

    def last_handover_persons(self):
        """(persons fused as complete-graph clusters, persons fused from member lists) of the last multi-person call,
        (-1, -1) if it did not arm the hand-over."""
        other = ct.c_int64(-1)
        n = int(self.L.snowtri_last_handover_persons(self.handle, ct.byref(other)))
        return n, int(other.value)


_scratch_ctx = {}


def current_device():
    """The HIP device index the calling thread's work should land on: torch's current device when torch is loaded
    and sees a GPU (one process per GPU sets it with torch.cuda.set_device(local_rank)), else SNOWTRI_DEVICE or 0."""
    import sys
    t = sys.modules.get("torch")
    if t is not None:
        try:
            if t.cuda.is_available():
                return int(t.cuda.current_device())
        except Exception:
            pass
    return int(os.environ.get("SNOWTRI_DEVICE", "0"))


def scratch_context(device=None):
    """Rig-less context for entry points that need only device scratch (condense, skew rays, smoothing).
    One per device: a context's scratch and kernels live on the GPU it was created for."""
    dev = current_device() if device is None else int(device)
    key = (LIB_PATH, dev)                   # (a context belongs to the library that made it)
    ctx = _scratch_ctx.get(key)
    if ctx is None:
        ctx = _scratch_ctx[key] = Context(device=dev)
    return ctx
