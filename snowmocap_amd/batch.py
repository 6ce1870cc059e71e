"""Additive batched API over the fused hot-path entry (snowtri_triangulate_condense).

`BatchTriangulator.run_host`  : NumPy in / NumPy out (staged through the context, synchronous).
`BatchTriangulator.run_torch` : CUDA(=HIP) tensors in / out, asynchronous on torch's current stream.
PyTorch is plumbing here (device memory + streams), never arithmetic.
"""
from __future__ import annotations

import numpy as np

from . import _lib

FUSED_CALLS = 0   # fused device calls issued through run_torch by this process (bench.py: which launches of a kernel trace belong to which loop)


class BatchTriangulator:
    def __init__(self, K, R, t, params, pout_max=1, out_dtype=np.float32, device=0, method=_lib.PAIRWISE, D=None, streams=1,
                 zero_fill=True):
        """D (optional, [C, 5] lens coefficients): the keypoints handed to run_* were detected on RAW frames and
        are undistorted on the GPU first (row N4, snowtri_undistort_keypoints).
        streams (1..4): OVERLAP MODE of the library (snowtri_ctx_set_overlap) -- consecutive run_torch calls are issued
        round-robin on that many internal streams, so a plain loop of independent calls (distinct `out` buffers) overlaps the
        tail of one launch with the ramp-up of the next; call join() before reading the results (or queueing work that
        reads them) on the caller's stream.
        zero_fill=False: run_torch leaves the slots behind count[f] as the output buffers hold them (SNOWTRI_CALL_NO_ZERO_FILL:
        the reference returns lists of count[f] persons, the padding is this ABI's; on a multi-person batch with a generous
        pout_max the zeros are most of what a call writes).  Read count[f] before a slot."""
        self.ctx = _lib.Context(K, R, t, device=device)
        self.call_flags = 0 if zero_fill else _lib.CALL_NO_ZERO_FILL
        self.streams = int(streams)
        if self.streams > 1:
            self.ctx.set_overlap(self.streams)
        self.undistort = D is not None
        if self.undistort:
            self.ctx.set_distortion(D)
        self._ukpts = None
        self.C = self.ctx.C
        self.params = params if isinstance(params, _lib.Params) else _lib.make_params(**params)
        self.pout_max = int(pout_max)
        self.out_dtype = np.dtype(out_dtype)
        self.method = method
        self.device = device

    # -- host buffers ---------------------------------------------------------------------------
    def run_host(self, kpts, n_persons=None):
        kpts = np.ascontiguousarray(kpts)
        F, C, Pmax, J, three = kpts.shape
        assert C == self.C and three == 3
        kn = self.params.keypoint_num
        if n_persons is not None:
            n_persons = np.ascontiguousarray(n_persons, dtype=np.int32).reshape(F, C)
        if self.undistort:
            kpts = self.ctx.undistort_keypoints(kpts)
        xyzs = np.empty((F, self.pout_max, max(kn, 0), 4), dtype=self.out_dtype)
        pscore = np.empty((F, self.pout_max), dtype=self.out_dtype)
        count = np.zeros(F, dtype=np.int32)
        flags = np.zeros(F, dtype=np.uint32)
        rc = self.ctx.L.snowtri_triangulate_condense(
            self.ctx.handle, F, Pmax, J, _lib.ptr(kpts), _lib.dtype_code(kpts.dtype), _lib.ptr(n_persons),
            self.params, self.method, self.pout_max, _lib.ptr(xyzs), _lib.ptr(pscore),
            _lib.dtype_code(self.out_dtype), _lib.ptr(count), _lib.ptr(flags), _lib.HOST, None)
        if rc not in (_lib.OK, _lib.ERR_SINGULAR, _lib.ERR_OVERFLOW):
            if rc == _lib.ERR_BAD_INDEX:
                raise IndexError("center_point_index / keypoint_num out of range")
            _lib.check(rc, "snowtri_triangulate_condense")
        return dict(xyzs=xyzs, pscore=pscore, count=count, flags=flags, status=rc)

    # -- device buffers (torch tensors) ------------------------------------------------------------
    def alloc_outputs(self, F, torch_device=None):
        import torch
        dev = torch_device or torch.device("cuda", self.device)
        tdt = torch.float32 if self.out_dtype == np.float32 else torch.float64
        kn = self.params.keypoint_num
        return dict(xyzs=torch.empty((F, self.pout_max, kn, 4), dtype=tdt, device=dev),
                    pscore=torch.empty((F, self.pout_max), dtype=tdt, device=dev),
                    count=torch.empty((F,), dtype=torch.int32, device=dev),
                    flags=torch.empty((F,), dtype=torch.int32, device=dev))

    def run_torch(self, kpts, n_persons=None, out=None, stream=None):
        """kpts: CUDA tensor [F,C,Pmax,J,3] float32/float64 (contiguous).  Launches on `stream`
        (default: torch's current stream) and returns immediately; results are CUDA tensors."""
        import torch
        assert kpts.is_cuda and kpts.is_contiguous()
        F, C, Pmax, J, three = kpts.shape
        assert C == self.C and three == 3
        if kpts.dtype == torch.float32:
            in_code = _lib.F32
        elif kpts.dtype == torch.float64:
            in_code = _lib.F64
        else:
            raise TypeError(f"snowtri supports float32/float64 keypoints, not {kpts.dtype}")
        if out is None:
            out = self.alloc_outputs(F, kpts.device)
        if n_persons is not None:
            assert n_persons.is_cuda and n_persons.dtype == torch.int32 and n_persons.is_contiguous()
        if stream is None:
            stream = torch.cuda.current_stream(kpts.device).cuda_stream
        import ctypes as ct
        if self.undistort:
            if self._ukpts is None or self._ukpts.shape != kpts.shape or self._ukpts.dtype != kpts.dtype \
                    or self._ukpts.device != kpts.device:
                self._ukpts = torch.empty_like(kpts)
            _lib.check(self.ctx.L.snowtri_undistort_keypoints(
                self.ctx.handle, F, Pmax, J, ct.c_void_p(kpts.data_ptr()), ct.c_void_p(self._ukpts.data_ptr()),
                in_code, _lib.DEVICE, ct.c_void_p(stream)), "snowtri_undistort_keypoints")
            kpts = self._ukpts
        global FUSED_CALLS
        FUSED_CALLS += 1
        rc = self.ctx.L.snowtri_triangulate_condense_ex(
            self.ctx.handle, F, Pmax, J, ct.c_void_p(kpts.data_ptr()), in_code,
            ct.c_void_p(n_persons.data_ptr()) if n_persons is not None else None, self.params, self.method,
            self.pout_max, ct.c_void_p(out["xyzs"].data_ptr()), ct.c_void_p(out["pscore"].data_ptr()),
            _lib.dtype_code(self.out_dtype), ct.c_void_p(out["count"].data_ptr()),
            ct.c_void_p(out["flags"].data_ptr()), _lib.DEVICE, ct.c_void_p(stream), self.call_flags)
        if rc == _lib.ERR_BAD_INDEX:
            raise IndexError("center_point_index / keypoint_num out of range")
        _lib.check(rc, "snowtri_triangulate_condense")
        return out

    def join(self, stream=None):
        """Overlap mode: torch's current stream (or `stream`, a HIP stream handle) waits for every call issued since the
        last join.  A no-op with streams=1."""
        if self.streams > 1:
            if stream is None:
                import torch
                stream = torch.cuda.current_stream(torch.device("cuda", self.device)).cuda_stream
            self.ctx.join(stream)

    def close(self):
        self.ctx.close()
