"""ctypes binding of the C ABI in ``include/raft_hip.h`` (libraft_hip.so).

This is the host-side mirror of the seam the engine replaces in the reference --
``create_pileup`` / ``repeat_annotate`` / ``break_reads`` (chop.hpp:366-372) -- for Python
callers (tests, bench.py, the multi-GPU driver).  There is no CPU fallback: if the HIP
library is missing or no gfx950 device is present, construction fails.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from .params import RaftParams

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libraft_hip.so")

OK, ERR_PARAM, ERR_READ_ID, ERR_COORD, ERR_FRAGMENT, ERR_NOMEM, ERR_DEVICE, ERR_STATE, ERR_TOO_LARGE = range(9)

EXPORTS = (
    "raft_hip_abi_version", "raft_hip_strerror", "raft_hip_last_error", "raft_hip_create", "raft_hip_destroy",
    "raft_hip_set_params", "raft_hip_set_stream", "raft_hip_use_own_stream", "raft_hip_get_stream", "raft_hip_run_device", "raft_hip_run_host",
    "raft_hip_finish", "raft_hip_outputs_device", "raft_hip_fetch", "raft_hip_last_timing", "raft_hip_set_tuning",
    "raft_hip_selftest", "raft_hip_fetch_packed", "raft_hip_fetch_packed_w", "raft_hip_run_pipelined", "raft_hip_run_multi",
    "raft_hip_set_output_width", "raft_hip_packed_device", "raft_hip_run_device_grouped", "raft_hip_run_host_grouped",
    "raft_hip_run_multi_grouped", "raft_hip_host_register", "raft_hip_host_unregister", "raft_hip_comm_unique_id",
    "raft_hip_comm_create", "raft_hip_comm_destroy", "raft_hip_exchange", "raft_hip_exchange_local", "raft_hip_warm_up", "raft_hip_reserve",
    "raft_hip_run_device_windows", "raft_hip_run_host_windows", "raft_hip_run_multi_windows",
    "raft_hip_fetch_delta4", "raft_hip_packed_anchor_device", "raft_hip_set_emit_cuts", "raft_hip_device_alloc", "raft_hip_device_free", "raft_hip_group_sides", "raft_hip_presplit_symmetric", "raft_hip_presplit_symmetric_local",
    "raft_hip_trim", "raft_hip_pool_bytes", "raft_hip_run_presplit_local", "raft_hip_set_placement", "raft_hip_placement_trial",
    "raft_hip_set_placement_trial",
)


class _Params(C.Structure):
    _fields_ = [("reso", C.c_int32), ("est_cov", C.c_int32), ("cov_mul", C.c_double),
                ("repeat_length", C.c_int32), ("interval_length", C.c_int32), ("read_length", C.c_int32),
                ("overlap_length", C.c_int32), ("flanking_length", C.c_int32), ("symmetric_mode", C.c_int32)]


class _Summary(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("symmetric", C.c_int32), ("high_cov", C.c_int32),
                ("interval_path", C.c_int32), ("n_segments", C.c_int32),
                ("n_records", C.c_int64), ("n_intervals", C.c_int64), ("n_bins", C.c_int64),
                ("n_repeats", C.c_int64), ("n_cuts", C.c_int64), ("n_fragments", C.c_int64),
                ("total_coverage", C.c_int64), ("total_windows", C.c_int64),
                ("total_repeat_length", C.c_int64), ("total_read_length", C.c_int64),
                ("error_index", C.c_int64), ("n_devices_used", C.c_int32), ("flags", C.c_int32)]


class _HostOutputs(C.Structure):
    _fields_ = [("cov_offset", C.c_void_p), ("cov8", C.c_void_p), ("cov8_cap", C.c_int64),
                ("exc_index", C.c_void_p), ("exc_value", C.c_void_p), ("exc_cap", C.c_int64), ("n_exc", C.c_int64),
                ("rep_offset", C.c_void_p), ("rep_s", C.c_void_p), ("rep_e", C.c_void_p), ("rep_cap", C.c_int64),
                ("frag_offset", C.c_void_p), ("frag_begin", C.c_void_p), ("frag_end", C.c_void_p), ("frag_cap", C.c_int64),
                ("cov_width", C.c_int32), ("reserved", C.c_int32), ("cov_anchor", C.c_void_p), ("anchor_cap", C.c_int64)]


class _Slice(C.Structure):
    _fields_ = [("n_rec", C.c_int64), ("n_runs", C.c_int32), ("rec_offset", C.c_void_p), ("d_qs", C.c_void_p), ("d_qe", C.c_void_p),
                ("d_rec_offset", C.c_void_p)]


class _Records(C.Structure):
    _fields_ = [("n_rec", C.c_int64)] + [(n, C.c_void_p) for n in ("d_qid", "d_qs", "d_qe", "d_tid", "d_ts", "d_te")]


class _Received(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("n_runs", C.c_int32), ("n_rec", C.c_int64), ("d_rec_offset", C.c_void_p), ("d_qs", C.c_void_p),
                ("d_qe", C.c_void_p)]


class _Outputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("cov_offset", "cov", "rep_offset", "rep_s", "rep_e", "cut_offset", "cuts",
                                          "frag_offset", "frag_read", "frag_begin", "frag_end")]


@dataclass
class Summary:
    n_reads: int
    symmetric: int
    high_cov: int
    interval_path: int
    n_segments: int
    n_records: int
    n_intervals: int
    n_bins: int
    n_repeats: int
    n_cuts: int
    n_fragments: int
    total_coverage: int
    total_windows: int
    total_repeat_length: int
    total_read_length: int
    error_index: int
    n_devices_used: int = 0
    flags: int = 0                # RAFT_HIP_SUM_*: bit 0 = the general bucketing handed the pileup kernel window records


class RaftError(RuntimeError):
    def __init__(self, code: int, message: str, index: int = -1):
        super().__init__(f"raft_hip error {code}: {message}" + (f" (index {index})" if index >= 0 else ""))
        self.code = code
        self.index = index


_lib = None


def load_library(path: str | None = None) -> C.CDLL:
    """Loads libraft_hip.so and declares every entry point of include/raft_hip.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("RAFT_HIP_LIB") or _LIB_PATH   # RAFT_HIP_LIB: A/B runs of two builds in one session
    if not os.path.exists(p):
        raise RuntimeError(f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the RAFT hot path)")
    try:
        # PyTorch-ROCm bundles its own libamdhip64.so.7.  It must be the first HIP runtime in the process:
        # loading the system one (our DT_NEEDED) first leaves torch with a second, device-less runtime.
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.raft_hip_abi_version.restype = C.c_int
    lib.raft_hip_strerror.restype = C.c_char_p
    lib.raft_hip_strerror.argtypes = [C.c_int]
    lib.raft_hip_last_error.restype = C.c_char_p
    lib.raft_hip_last_error.argtypes = [vp]
    lib.raft_hip_create.argtypes = [C.c_int, C.POINTER(_Params), C.POINTER(vp)]
    lib.raft_hip_destroy.argtypes = [vp]
    lib.raft_hip_destroy.restype = None
    lib.raft_hip_set_params.argtypes = [vp, C.POINTER(_Params)]
    lib.raft_hip_set_stream.argtypes = [vp, vp]
    lib.raft_hip_use_own_stream.argtypes = [vp]
    lib.raft_hip_get_stream.argtypes = [vp]
    lib.raft_hip_get_stream.restype = vp
    lib.raft_hip_run_device.argtypes = [vp, i32, vp, i64, vp, vp, vp, vp, vp, vp]
    lib.raft_hip_run_host.argtypes = [vp, i32, vp, i64, vp, vp, vp, vp, vp, vp]
    lib.raft_hip_finish.argtypes = [vp, C.POINTER(_Summary)]
    lib.raft_hip_outputs_device.argtypes = [vp, C.POINTER(_Outputs)]
    lib.raft_hip_fetch.argtypes = [vp] + [vp] * 11
    lib.raft_hip_fetch_packed.argtypes = [vp, vp, vp, i64, vp, vp, C.POINTER(i64)] + [vp] * 7
    lib.raft_hip_fetch_packed_w.argtypes = [vp, i32, vp, vp, i64, vp, vp, C.POINTER(i64)] + [vp] * 7
    lib.raft_hip_run_pipelined.argtypes = [vp, i32, vp, i64, vp, vp, vp, vp, vp, vp, i32, C.POINTER(_HostOutputs), C.POINTER(_Summary)]
    lib.raft_hip_run_multi.argtypes = [C.POINTER(vp), i32, i32, vp, i64, vp, vp, vp, vp, vp, vp, i32, C.POINTER(_HostOutputs), C.POINTER(_Summary)]
    lib.raft_hip_run_device_grouped.argtypes = [vp, i32, vp, i64, i32, vp, vp, vp, vp, i64]
    lib.raft_hip_run_presplit_local.argtypes = [C.POINTER(vp), i32, i32, vp, i64, vp, vp, vp, vp, vp, vp, C.POINTER(_HostOutputs), C.POINTER(_Summary)]
    lib.raft_hip_run_host_grouped.argtypes = [vp, i32, vp, i64, i32, vp, vp, vp, i64]
    lib.raft_hip_run_multi_grouped.argtypes = [C.POINTER(vp), i32, i32, vp, i64, i32, vp, vp, vp, i32, C.POINTER(_HostOutputs), C.POINTER(_Summary)]
    lib.raft_hip_fetch_delta4.argtypes = [vp, vp, vp, vp, i64, vp, vp, C.POINTER(i64)] + [vp] * 7
    lib.raft_hip_packed_anchor_device.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    lib.raft_hip_run_device_windows.argtypes = [vp, i32, vp, i64, i32, vp, vp, i64]
    lib.raft_hip_run_host_windows.argtypes = [vp, i32, vp, i64, i32, vp, vp, i64]
    lib.raft_hip_run_multi_windows.argtypes = [C.POINTER(vp), i32, i32, vp, i64, i32, vp, vp, i32, C.POINTER(_HostOutputs), C.POINTER(_Summary)]
    lib.raft_hip_comm_unique_id.argtypes = [vp]
    lib.raft_hip_comm_create.argtypes = [C.c_int, vp, i32, i32, C.POINTER(vp)]
    lib.raft_hip_comm_destroy.argtypes = [vp]
    lib.raft_hip_comm_destroy.restype = None
    lib.raft_hip_exchange.argtypes = [vp, vp, i32, i32, i32, vp, C.POINTER(_Slice), C.POINTER(_Received)]
    lib.raft_hip_exchange_local.argtypes = [C.POINTER(vp), i32, i32, vp, C.POINTER(_Slice), C.POINTER(_Received)]
    lib.raft_hip_warm_up.argtypes = [vp]
    lib.raft_hip_reserve.argtypes = [vp, i32, vp, i64, i32, i32]
    lib.raft_hip_host_register.argtypes = [vp, C.c_uint64]
    lib.raft_hip_host_unregister.argtypes = [vp]
    lib.raft_hip_last_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.raft_hip_set_tuning.argtypes = [vp, i32, i32, i32]
    lib.raft_hip_set_output_width.argtypes = [vp, i32]
    lib.raft_hip_set_emit_cuts.argtypes = [vp, i32]
    lib.raft_hip_device_alloc.argtypes = [vp, i64, C.POINTER(C.c_void_p)]
    lib.raft_hip_device_free.argtypes = [vp, vp]
    lib.raft_hip_trim.argtypes = [C.c_int, i64]
    lib.raft_hip_trim.restype = i64
    lib.raft_hip_pool_bytes.argtypes = [C.c_int]
    lib.raft_hip_pool_bytes.restype = i64
    lib.raft_hip_set_placement.argtypes = [i32]
    lib.raft_hip_set_placement.restype = i32
    lib.raft_hip_placement_trial.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i32)]
    lib.raft_hip_set_placement_trial.argtypes = [vp, i32]
    lib.raft_hip_group_sides.argtypes = [vp, i32, i64, vp, vp, vp, vp, vp, vp, i32, C.POINTER(_Slice)]
    lib.raft_hip_presplit_symmetric.argtypes = [vp, vp, i32, i32, C.POINTER(_Records), C.POINTER(i32)]
    lib.raft_hip_presplit_symmetric_local.argtypes = [C.POINTER(vp), i32, C.POINTER(_Records), C.POINTER(i32)]
    lib.raft_hip_packed_device.argtypes = [vp, C.POINTER(i32), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64)]
    lib.raft_hip_selftest.argtypes = [C.c_int]
    if path is None:
        _lib = lib
    return lib


def _cparams(p: RaftParams) -> _Params:
    return _Params(p.reso, p.est_cov, p.cov_mul, p.repeat_length, p.interval_length, p.read_length,
                   p.overlap_length, p.flanking_length, p.symmetric_mode)


class _DevArray:
    """Zero-copy view of a device buffer for ``torch.as_tensor`` (CUDA array interface v2)."""

    def __init__(self, ptr: int, n: int, typestr: str, owner):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr or 0, False), "version": 2}
        self._owner = owner


class Engine:
    """One context on one MI355X: ``run*`` -> ``finish`` -> ``fetch`` / ``outputs_device``."""

    def __init__(self, params: RaftParams, device: int = 0):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        self.params = params
        cp = _cparams(params)
        rc = self._lib.raft_hip_create(device, C.byref(cp), C.byref(self._ctx))
        if rc != OK:
            self._ctx = C.c_void_p()
            raise RaftError(rc, self._lib.raft_hip_strerror(rc).decode())
        self.device = device
        self._keep = None

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.raft_hip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, index: int = -1):
        if rc != OK:
            msg = self._lib.raft_hip_strerror(rc).decode()
            detail = self._lib.raft_hip_last_error(self._ctx).decode()
            raise RaftError(rc, msg + (f" [{detail}]" if detail else ""), index)

    def set_params(self, params: RaftParams):
        cp = _cparams(params)
        self._check(self._lib.raft_hip_set_params(self._ctx, C.byref(cp)))
        self.params = params

    def set_tuning(self, tile_bins: int = 0, force_bucket_path: bool = False, variant: int = -1):
        self._check(self._lib.raft_hip_set_tuning(self._ctx, tile_bins, int(force_bucket_path), variant))

    def set_placement_trial(self, candidates: int):
        """Opt in to the coverage array's placement trial at the context's first large pass: 2..8 candidate arrays, 0 = off (the default)."""
        self._check(self._lib.raft_hip_set_placement_trial(self._ctx, candidates))

    def warm_up(self):
        """raft_hip_warm_up: the engine's code on the device, the pipeline's lanes -- what a context's first host-to-host job would
        otherwise pay inside its own clock."""
        self._check(self._lib.raft_hip_warm_up(self._ctx))

    def reserve(self, read_len, n_rec_estimate: int, n_ctx: int = 1, cov_width: int = 1):
        """raft_hip_reserve: device buffers and page-locked staging of a coming host-to-host job, from the reads' lengths and an
        estimate of the record count (cov_width: 1, 2 or 8 = four-bit steps)."""
        rl = np.ascontiguousarray(np.asarray(read_len), dtype=np.int32)
        self._check(self._lib.raft_hip_reserve(self._ctx, rl.size, C.c_void_p(rl.ctypes.data if rl.size else 0), int(n_rec_estimate), int(n_ctx), int(cov_width)))

    def placement_trial(self):
        """(first_ms, best_other_ms, kept) of the coverage array's placement trial -- kept: 0 the first placement, 1 a plain block, 2 another
        chunk mapping -- or None when none has run (raft_hip_placement_trial)."""
        a, b, k = C.c_double(), C.c_double(), C.c_int32()
        if self._lib.raft_hip_placement_trial(self._ctx, C.byref(a), C.byref(b), C.byref(k)) != OK:
            return None
        return a.value, b.value, int(k.value)

    def set_output_width(self, width: int):
        """4: cov[] as int32 (default); 1 / 2: later passes write the transfer encoding directly (raft_hip_set_output_width)."""
        self._check(self._lib.raft_hip_set_output_width(self._ctx, width))

    def set_emit_cuts(self, on: bool):
        """True (default): every pass writes the cut points itself; False: the first fetch that asks for them does."""
        self._check(self._lib.raft_hip_set_emit_cuts(self._ctx, 1 if on else 0))

    def device_tensor(self, shape, dtype):
        """A torch tensor over device memory from raft_hip_device_alloc (the engine's placement: shuffled 32 MiB chunks); it
        lives until the context is closed or device_free(tensor) is called."""
        import torch
        shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        n = 1
        for x in shape:
            n *= x
        item = torch.empty(0, dtype=dtype).element_size()
        ptr = C.c_void_p()
        self._check(self._lib.raft_hip_device_alloc(self._ctx, n * item, C.byref(ptr)))
        typestr = {torch.int32: "<i4", torch.int64: "<i8", torch.uint8: "|u1", torch.int16: "<i2"}[dtype]

        class _Mem:
            pass
        m = _Mem()
        m.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr.value or 0), False), "version": 2}
        t = torch.as_tensor(m, device=f"cuda:{self.device}")
        t._raft_ptr = int(ptr.value or 0)
        return t

    def device_copy(self, t):
        """`t` copied into memory from device_tensor."""
        out = self.device_tensor(tuple(t.shape), t.dtype)
        out.copy_(t)
        return out

    def device_free(self, t):
        self._check(self._lib.raft_hip_device_free(self._ctx, C.c_void_p(getattr(t, "_raft_ptr", None) or t.data_ptr())))

    def group_sides(self, n_reads_total: int, qid, qs, qe, tid=None, ts=None, te=None, symmetric: bool = False) -> "Slice":
        """raft_hip_group_sides: the (read, start, end) intervals of a slice's records -- query sides, and target sides of records
        whose two reads differ unless ``symmetric`` -- sorted by read id on the device, as a Slice in grouped form with one run
        (ready for exchange_local / Comm.exchange).  The arrays are the context's (valid until its next group_sides)."""
        import torch
        self.use_torch_stream()
        cols = [qid, qs, qe] + ([] if symmetric else [tid, ts, te])
        for t in cols:
            if t.dtype != torch.int32 or not t.is_cuda or not t.is_contiguous():
                raise TypeError("group_sides needs contiguous int32 CUDA tensors")
        ptr = [C.c_void_p(t.data_ptr() if t.numel() else 0) for t in cols] + ([C.c_void_p(0)] * 3 if symmetric else [])
        out = _Slice()
        self._check(self._lib.raft_hip_group_sides(self._ctx, n_reads_total, int(qid.numel()), *ptr, 1 if symmetric else 0, C.byref(out)))
        n = int(out.n_rec)
        off = np.ctypeslib.as_array(C.cast(out.rec_offset, C.POINTER(C.c_int64)), shape=(1, n_reads_total + 1)).copy()
        dev = f"cuda:{self.device}"

        def view(p):
            return torch.empty(0, dtype=torch.int32, device=dev) if n == 0 else torch.as_tensor(_DevArray(p, n, "<i4", self), device=dev)
        return Slice(off, view(out.d_qs), view(out.d_qe))

    def use_torch_stream(self):
        import torch
        self._check(self._lib.raft_hip_set_stream(self._ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    # -- passes -----------------------------------------------------------------
    def run_device(self, read_len, qid, qs, qe, tid=None, ts=None, te=None):
        """Inputs: int32 torch tensors on this engine's device (kept alive until the next pass); tid / ts / te may be None when
        the params assert symmetric_mode = 1."""
        import torch
        cols = (read_len, qid, qs, qe, tid, ts, te)
        # (a caller that hands over the same tensors again -- a stream of batches through fixed buffers -- is checked once)
        last = getattr(self, "_last_device_call", None)
        if last is not None and all(a is b for a, b in zip(last[0], cols)) and \
                all(t is None or t.data_ptr() == q for t, q in zip(cols, last[2])) and int(qid.numel()) == last[1][2]:
            args = last[1]
        else:
            for t in cols:
                if t is not None and (t.dtype != torch.int32 or not t.is_cuda or not t.is_contiguous()):
                    raise TypeError("run_device needs contiguous int32 CUDA tensors")
            n_rec = int(qid.numel())
            for t in cols[2:]:
                if t is not None and int(t.numel()) != n_rec:
                    raise ValueError("PAF columns differ in length")
            ptr = [C.c_void_p(t.data_ptr() if (t is not None and t.numel()) else 0) for t in cols]
            args = (int(read_len.numel()), ptr[0], n_rec, *ptr[1:])
            self._last_device_call = (cols, args, [None if t is None else t.data_ptr() for t in cols])
        self._keep = cols
        self.use_torch_stream()     # the tensors were produced on torch's current stream: order after it
        self._check(self._lib.raft_hip_run_device(self._ctx, *args))

    def run_device_grouped(self, read_len, rec_offset, qid, qs, qe, n_bins: int = -1):
        """raft_hip_run_device_grouped: ``rec_offset`` int64 CUDA tensor [n_runs, n_reads + 1] (first record of every read in
        every sorted run), ``qid`` may be None (the ids are then rebuilt from the offsets on the device), ``n_bins`` the
        caller's sum of ceil(len / reso) (-1: unknown; >= 0: the pass runs without a host wait).  symmetric_mode must be 1."""
        import torch
        n_reads = int(read_len.numel())
        if rec_offset.dtype != torch.int64 or not rec_offset.is_cuda or not rec_offset.is_contiguous() or rec_offset.dim() != 2 \
                or rec_offset.shape[1] != n_reads + 1:
            raise TypeError("run_device_grouped needs rec_offset as a contiguous int64 CUDA tensor [n_runs, n_reads + 1]")
        cols = (read_len, qs, qe) + (() if qid is None else (qid,))
        for t in cols:
            if t.dtype != torch.int32 or not t.is_cuda or not t.is_contiguous():
                raise TypeError("run_device_grouped needs contiguous int32 CUDA tensors")
        n_rec = int(qs.numel())
        if int(qe.numel()) != n_rec or (qid is not None and int(qid.numel()) != n_rec):
            raise ValueError("PAF columns differ in length")
        self._keep = cols + (rec_offset,)
        self.use_torch_stream()
        P = lambda t: C.c_void_p(t.data_ptr() if (t is not None and t.numel()) else 0)
        self._check(self._lib.raft_hip_run_device_grouped(self._ctx, n_reads, P(read_len), n_rec, int(rec_offset.shape[0]), P(rec_offset),
                                                          P(qid), P(qs), P(qe), int(n_bins)))

    def run_device_windows(self, read_len, rec_offset, win, n_bins: int = -1):
        """raft_hip_run_device_windows: grouped input whose records are ONE word each -- first window | one past the last << 16
        (``hostio.pack_windows``); ``win`` an int32 or uint32-viewed CUDA tensor of 32-bit words.  symmetric_mode must be 1."""
        import torch
        n_reads = int(read_len.numel())
        if rec_offset.dtype != torch.int64 or not rec_offset.is_cuda or not rec_offset.is_contiguous() or rec_offset.dim() != 2 \
                or rec_offset.shape[1] != n_reads + 1:
            raise TypeError("run_device_windows needs rec_offset as a contiguous int64 CUDA tensor [n_runs, n_reads + 1]")
        for t in (read_len, win):
            if t.element_size() != 4 or t.is_floating_point() or not t.is_cuda or not t.is_contiguous():
                raise TypeError("run_device_windows needs contiguous 32-bit integer CUDA tensors")
        self._keep = (read_len, win, rec_offset)
        self.use_torch_stream()
        P = lambda t: C.c_void_p(t.data_ptr() if (t is not None and t.numel()) else 0)
        self._check(self._lib.raft_hip_run_device_windows(self._ctx, n_reads, P(read_len), int(win.numel()), int(rec_offset.shape[0]), P(rec_offset),
                                                          P(win), int(n_bins)))

    def run_host_windows(self, read_len, rec_offset, win, n_bins: int = -1):
        """raft_hip_run_host_windows: numpy arrays; ``win`` uint32 window records."""
        rl = np.ascontiguousarray(np.asarray(read_len), dtype=np.int32)
        w = np.ascontiguousarray(np.asarray(win), dtype=np.uint32)
        off = np.ascontiguousarray(np.asarray(rec_offset), dtype=np.int64)
        if off.ndim != 2 or off.shape[1] != rl.size + 1:
            raise ValueError("run_host_windows: rec_offset must be [n_runs, n_reads + 1]")
        self._keep = (rl, w, off)
        P = lambda x: C.c_void_p(x.ctypes.data if x.size else 0)
        self._check(self._lib.raft_hip_run_host_windows(self._ctx, rl.size, P(rl), w.size, off.shape[0], P(off), P(w), int(n_bins)))

    def run_host_grouped(self, read_len, rec_offset, qs, qe, n_bins: int = -1):
        """raft_hip_run_host_grouped: numpy arrays; ``rec_offset`` int64 [n_runs, n_reads + 1]."""
        rl, a, b = (np.ascontiguousarray(np.asarray(x), dtype=np.int32) for x in (read_len, qs, qe))
        off = np.ascontiguousarray(np.asarray(rec_offset), dtype=np.int64)
        if off.ndim != 2 or off.shape[1] != rl.size + 1 or a.size != b.size:
            raise ValueError("run_host_grouped: rec_offset must be [n_runs, n_reads + 1], qs/qe of equal length")
        self._keep = (rl, a, b, off)
        P = lambda x: C.c_void_p(x.ctypes.data if x.size else 0)
        self._check(self._lib.raft_hip_run_host_grouped(self._ctx, rl.size, P(rl), a.size, off.shape[0], P(off), P(a), P(b), int(n_bins)))

    def run_host(self, read_len, qid, qs, qe, tid=None, ts=None, te=None):
        """tid/ts/te may be None when the params assert symmetric_mode = 1 (they are then neither read nor uploaded)."""
        cols = [None if a is None else np.ascontiguousarray(np.asarray(a), dtype=np.int32) for a in (read_len, qid, qs, qe, tid, ts, te)]
        n_rec = cols[1].size
        for a in cols[2:]:
            if a is not None and a.size != n_rec:
                raise ValueError("PAF columns differ in length")
        self._keep = cols
        ptr = [C.c_void_p(a.ctypes.data if (a is not None and a.size) else 0) for a in cols]
        self._check(self._lib.raft_hip_run_host(self._ctx, cols[0].size, ptr[0], n_rec, *ptr[1:]))

    def finish(self) -> Summary:
        s = _Summary()
        rc = self._lib.raft_hip_finish(self._ctx, C.byref(s))
        summ = Summary(s.n_reads, s.symmetric, s.high_cov, s.interval_path, s.n_segments, s.n_records, s.n_intervals, s.n_bins, s.n_repeats, s.n_cuts,
                       s.n_fragments, s.total_coverage, s.total_windows, s.total_repeat_length, s.total_read_length, s.error_index, s.n_devices_used, s.flags)
        self.summary = summ
        self._check(rc, summ.error_index)
        return summ

    def timing(self) -> tuple[float, float]:
        a, b = C.c_double(), C.c_double()
        self._check(self._lib.raft_hip_last_timing(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def fetch(self, coverage: bool = True, pinned: bool = False, out: dict | None = None) -> dict:
        """Host copies (numpy) of the finished pass, CSR per read.

        pinned: allocate the arrays in page-locked memory (the copies then run at the link's rate instead of the
        pageable path's).  out: arrays of an earlier fetch to reuse when their sizes still fit (a caller that keeps
        one pinned set of buffers pays neither allocation nor page faults per pass)."""
        s = self.summary
        n1 = s.n_reads + 1
        spec = {
            "cov_offset": (n1, np.int64), "cov": (s.n_bins if coverage else 0, np.int32),
            "rep_offset": (n1, np.int64), "rep_s": (s.n_repeats, np.int32), "rep_e": (s.n_repeats, np.int32),
            "cut_offset": (n1, np.int64), "cuts": (s.n_cuts, np.int32),
            "frag_offset": (n1, np.int64), "frag_read": (s.n_fragments, np.int32),
            "frag_begin": (s.n_fragments, np.int32), "frag_end": (s.n_fragments, np.int32),
        }

        def alloc(n, dt):
            if pinned and n:
                import torch
                return torch.empty(int(n), dtype=torch.int64 if dt == np.int64 else torch.int32, pin_memory=True).numpy()
            return np.empty(n, dt)

        res = {}
        for key, (n, dt) in spec.items():
            have = out.get(key) if out else None
            if have is not None and have.dtype == dt and have.size >= n and have.flags["C_CONTIGUOUS"]:
                res[key] = have[:n]
            else:
                res[key] = alloc(n, dt)
        out = res
        order = ("cov_offset", "cov", "rep_offset", "rep_s", "rep_e", "cut_offset", "cuts", "frag_offset",
                 "frag_read", "frag_begin", "frag_end")
        ptr = [C.c_void_p(out[k].ctypes.data if out[k].size else 0) for k in order]
        self._check(self._lib.raft_hip_fetch(self._ctx, *ptr))
        return out

    def host_output_buffers(self, read_len, pinned: bool = True, exc_cap: int = 1 << 20, width: int = 1) -> dict:
        """Caller-owned arrays for ``run_pipelined`` sized by the upper bounds of include/raft_hip.h (page-locked when
        ``pinned``): allocate once, reuse for every pass over inputs of this shape."""
        p = self.params
        rl = np.asarray(read_len, np.int64)
        nb = (rl + p.reso - 1) // p.reso
        minw = max((p.repeat_length + p.reso - 1) // p.reso, 1)
        caps = {"cov8": int(nb.sum()), "rep": (int(nb.sum()) + rl.size) // (minw + 1), "frag": int(rl.sum()) // p.interval_length + 2 * rl.size,
                "exc": int(exc_cap)}
        n1 = rl.size + 1

        def alloc(n, dt):
            n = max(int(n), 1)
            if pinned:
                import torch
                if dt == np.uint16:      # (torch has no uint16 everywhere: page-locked bytes, viewed as uint16)
                    return torch.empty(2 * n, dtype=torch.uint8, pin_memory=True).numpy().view(np.uint16)
                tdt = {np.int64: torch.int64, np.int32: torch.int32, np.uint8: torch.uint8}[dt]
                return torch.empty(n, dtype=tdt, pin_memory=True).numpy()
            return np.empty(n, dt)
        if width == 8:       # delta4: two windows per byte + an anchor per 1024 windows ("cov_nib" / "cov_anchor" name the encoding)
            caps["exc"] = max(caps["exc"], caps["cov8"] // 128)      # (every tile's first window and the large steps: 0.2-0.3 % of a 32x set)
            return {"cov_offset": alloc(n1, np.int64), "cov_nib": alloc((caps["cov8"] + 1) // 2, np.uint8), "cov_anchor": alloc((caps["cov8"] + 1023) // 1024, np.int32),
                    "exc_index": alloc(caps["exc"], np.int64), "exc_value": alloc(caps["exc"], np.int32), "rep_offset": alloc(n1, np.int64),
                    "rep_s": alloc(caps["rep"], np.int32), "rep_e": alloc(caps["rep"], np.int32), "frag_offset": alloc(n1, np.int64),
                    "frag_begin": alloc(caps["frag"], np.int32), "frag_end": alloc(caps["frag"], np.int32)}
        return {"cov_offset": alloc(n1, np.int64), "cov8": alloc(caps["cov8"], np.uint16 if width == 2 else np.uint8),
                "exc_index": alloc(caps["exc"], np.int64),
                "exc_value": alloc(caps["exc"], np.int32), "rep_offset": alloc(n1, np.int64), "rep_s": alloc(caps["rep"], np.int32),
                "rep_e": alloc(caps["rep"], np.int32), "frag_offset": alloc(n1, np.int64), "frag_begin": alloc(caps["frag"], np.int32),
                "frag_end": alloc(caps["frag"], np.int32)}

    def run_pipelined_grouped(self, read_len, rec_offset, qs, qe, n_chunks: int = 0, out: dict | None = None,
                              others: list | None = None):
        """raft_hip_run_multi_grouped: as ``run_pipelined`` with the caller's per-read record offsets (int64
        [n_runs, n_reads + 1]) in place of the query column."""
        rl, a, b = (np.ascontiguousarray(np.asarray(x), dtype=np.int32) for x in (read_len, qs, qe))
        off = np.ascontiguousarray(np.asarray(rec_offset), dtype=np.int64)
        if off.ndim != 2 or off.shape[1] != rl.size + 1 or a.size != b.size:
            raise ValueError("run_pipelined_grouped: rec_offset must be [n_runs, n_reads + 1], qs/qe of equal length")
        if out is None:
            out = self.host_output_buffers(rl, pinned=False)
        ho = self._host_outputs(out)
        P = lambda x: C.c_void_p(x.ctypes.data if x.size else 0)
        s = _Summary()
        ctxs = (C.c_void_p * (1 + len(others or [])))(self._ctx, *[e._ctx for e in (others or [])])
        rc = self._lib.raft_hip_run_multi_grouped(ctxs, len(ctxs), rl.size, P(rl), a.size, off.shape[0], P(off), P(a), P(b), int(n_chunks),
                                                  C.byref(ho), C.byref(s))
        return self._pipelined_result(rc, s, ho, out)

    def run_pipelined_windows(self, read_len, rec_offset, win, n_chunks: int = 0, out: dict | None = None, others: list | None = None):
        """raft_hip_run_multi_windows: as ``run_pipelined_grouped`` with window records (uint32, ``hostio.pack_windows``) in place
        of the two coordinate columns."""
        rl = np.ascontiguousarray(np.asarray(read_len), dtype=np.int32)
        w = np.ascontiguousarray(np.asarray(win), dtype=np.uint32)
        off = np.ascontiguousarray(np.asarray(rec_offset), dtype=np.int64)
        if off.ndim != 2 or off.shape[1] != rl.size + 1:
            raise ValueError("run_pipelined_windows: rec_offset must be [n_runs, n_reads + 1]")
        if out is None:
            out = self.host_output_buffers(rl, pinned=False)
        ho = self._host_outputs(out)
        P = lambda x: C.c_void_p(x.ctypes.data if x.size else 0)
        s = _Summary()
        ctxs = (C.c_void_p * (1 + len(others or [])))(self._ctx, *[e._ctx for e in (others or [])])
        rc = self._lib.raft_hip_run_multi_windows(ctxs, len(ctxs), rl.size, P(rl), w.size, off.shape[0], P(off), P(w), int(n_chunks),
                                                  C.byref(ho), C.byref(s))
        return self._pipelined_result(rc, s, ho, out)

    def _host_outputs(self, out: dict) -> "_HostOutputs":
        ho = _HostOutputs()
        for k in ("cov_offset", "exc_index", "exc_value", "rep_offset", "rep_s", "rep_e", "frag_offset", "frag_begin", "frag_end"):
            setattr(ho, k, out[k].ctypes.data)
        ho.exc_cap, ho.rep_cap, ho.frag_cap = out["exc_index"].size, out["rep_s"].size, out["frag_begin"].size
        if "cov_nib" in out:                                               # delta4 (host_output_buffers(width=8))
            ho.cov8, ho.cov8_cap = out["cov_nib"].ctypes.data, 2 * out["cov_nib"].size
            ho.cov_anchor, ho.anchor_cap = out["cov_anchor"].ctypes.data, out["cov_anchor"].size
            ho.cov_width = 8
        else:
            ho.cov8, ho.cov8_cap = out["cov8"].ctypes.data, out["cov8"].size
            ho.cov_width = 2 if out["cov8"].dtype == np.uint16 else 1      # the buffer's dtype chooses the encoding's width
        return ho

    def _pipelined_result(self, rc, s, ho, out):
        summ = Summary(**{f: int(getattr(s, f)) for f, _ in _Summary._fields_})
        self.summary = summ
        self.last_n_exc = int(ho.n_exc)
        self._check(rc, summ.error_index)
        n1 = summ.n_reads + 1
        cov = ({"cov_nib": out["cov_nib"][:(summ.n_bins + 1) // 2], "cov_anchor": out["cov_anchor"][:(summ.n_bins + 1023) // 1024]} if "cov_nib" in out
               else {"cov8": out["cov8"][:summ.n_bins]})
        res = {"cov_offset": out["cov_offset"][:n1], **cov, "exc_index": out["exc_index"][:ho.n_exc],
               "exc_value": out["exc_value"][:ho.n_exc], "rep_offset": out["rep_offset"][:n1], "rep_s": out["rep_s"][:summ.n_repeats],
               "rep_e": out["rep_e"][:summ.n_repeats], "frag_offset": out["frag_offset"][:n1],
               "frag_begin": out["frag_begin"][:summ.n_fragments], "frag_end": out["frag_end"][:summ.n_fragments]}
        return res, summ

    def run_pipelined(self, read_len, qid, qs, qe, tid=None, ts=None, te=None, n_chunks: int = 0, out: dict | None = None,
                      others: list | None = None):
        """raft_hip_run_pipelined: host columns in, host outputs out, with upload / pass / download of consecutive read
        ranges overlapped.  ``others``: more Engines (other GPUs of the node, or the same one) to share the job with
        (raft_hip_run_multi).  Returns (dict of arrays trimmed to their sizes -- views of ``out`` --, Summary)."""
        cols = [None if a is None else np.ascontiguousarray(np.asarray(a), dtype=np.int32) for a in (read_len, qid, qs, qe, tid, ts, te)]
        n_rec = cols[1].size
        if out is None:
            out = self.host_output_buffers(cols[0], pinned=False)
        ho = self._host_outputs(out)
        ptr = [C.c_void_p(a.ctypes.data if (a is not None and a.size) else 0) for a in cols]
        s = _Summary()
        if others:
            ctxs = (C.c_void_p * (1 + len(others)))(self._ctx, *[e._ctx for e in others])
            rc = self._lib.raft_hip_run_multi(ctxs, 1 + len(others), cols[0].size, ptr[0], n_rec, *ptr[1:], int(n_chunks),
                                              C.byref(ho), C.byref(s))
        else:
            rc = self._lib.raft_hip_run_pipelined(self._ctx, cols[0].size, ptr[0], n_rec, *ptr[1:], int(n_chunks), C.byref(ho), C.byref(s))
        return self._pipelined_result(rc, s, ho, out)

    def run_presplit(self, read_len, qid, qs, qe, tid, ts, te, others: list, out: dict | None = None):
        """raft_hip_run_presplit_local: this Engine is rank 0, ``others`` ranks 1 .. -- the record stream is cut into as many
        contiguous slices, every slice's sides are grouped on its rank's device, ONE exchange routes them to the owners of their
        reads, every rank runs its grouped pass.  Same return as ``run_pipelined``."""
        cols = [np.ascontiguousarray(np.asarray(a), dtype=np.int32) for a in (read_len, qid, qs, qe, tid, ts, te)]
        if out is None:
            out = self.host_output_buffers(cols[0], pinned=False)
        ho = self._host_outputs(out)
        ptr = [C.c_void_p(a.ctypes.data if a.size else 0) for a in cols]
        s = _Summary()
        ctxs = (C.c_void_p * (1 + len(others)))(self._ctx, *[e._ctx for e in others])
        rc = self._lib.raft_hip_run_presplit_local(ctxs, 1 + len(others), cols[0].size, ptr[0], cols[1].size, *ptr[1:], C.byref(ho), C.byref(s))
        return self._pipelined_result(rc, s, ho, out)

    def fetch_packed(self, pinned: bool = False, out: dict | None = None, width: int = 1) -> dict:
        """Host copies with the coverage array in its transfer encoding (raft_hip_fetch_packed_w): ``cov8`` (uint8 per
        window, 255 = see exceptions; with ``width=2`` uint16, 65535), ``exc_index`` / ``exc_value`` (ascending), and the
        repeat / fragment tables.  ``out``: arrays of an earlier call to reuse (pinned buffers kept by the caller; the
        dtype of its ``cov8`` decides the width)."""
        s = self.summary
        n1 = s.n_reads + 1
        if out is not None and out.get("cov8") is not None:
            width = 2 if out["cov8"].dtype == np.uint16 else 1
        cdt = np.uint16 if width == 2 else np.uint8
        n_exc = C.c_int64(0)
        none = [C.c_void_p(0)] * 7
        rc = self._lib.raft_hip_fetch_packed_w(self._ctx, width, None, None, 0, None, None, C.byref(n_exc), *none)
        self._check(rc)
        spec = {"cov_offset": (n1, np.int64), "cov8": (s.n_bins, cdt), "exc_index": (n_exc.value, np.int64),
                "exc_value": (n_exc.value, np.int32), "rep_offset": (n1, np.int64), "rep_s": (s.n_repeats, np.int32),
                "rep_e": (s.n_repeats, np.int32), "frag_offset": (n1, np.int64), "frag_read": (s.n_fragments, np.int32),
                "frag_begin": (s.n_fragments, np.int32), "frag_end": (s.n_fragments, np.int32)}

        def alloc(n, dt):
            if pinned and n:
                import torch
                if dt == np.uint16:
                    return torch.empty(2 * int(n), dtype=torch.uint8, pin_memory=True).numpy().view(np.uint16)
                tdt = {np.int64: torch.int64, np.int32: torch.int32, np.uint8: torch.uint8}[dt]
                return torch.empty(int(n), dtype=tdt, pin_memory=True).numpy()
            return np.empty(n, dt)

        res = {}
        for key, (n, dt) in spec.items():
            have = out.get(key) if out else None
            if have is not None and have.dtype == dt and have.size >= n and have.flags["C_CONTIGUOUS"]:
                res[key] = have[:n]
            else:
                res[key] = alloc(n, dt)
        ptr = {k: C.c_void_p(res[k].ctypes.data if res[k].size else 0) for k in res}
        self._check(self._lib.raft_hip_fetch_packed_w(self._ctx, width, ptr["cov_offset"], ptr["cov8"], n_exc.value, ptr["exc_index"],
                                                      ptr["exc_value"], C.byref(n_exc), ptr["rep_offset"], ptr["rep_s"], ptr["rep_e"],
                                                      ptr["frag_offset"], ptr["frag_read"], ptr["frag_begin"], ptr["frag_end"]))
        return res

    def fetch_delta4(self, pinned: bool = False, out: dict | None = None) -> dict:
        """Host copies with the coverage array in the four-bit step encoding (raft_hip_fetch_delta4): ``cov_nib`` (uint8,
        two windows per byte), ``cov_anchor`` (int32 per 1024 windows), ``exc_index`` / ``exc_value`` (ascending; ABSOLUTE values
        of the escaped windows), and the repeat / fragment tables.  ``hostio.unpack_coverage_d4`` restores the int32 array."""
        s = self.summary
        n1 = s.n_reads + 1
        n_exc = C.c_int64(0)
        none = [C.c_void_p(0)] * 7
        self._check(self._lib.raft_hip_fetch_delta4(self._ctx, None, None, None, 0, None, None, C.byref(n_exc), *none))
        spec = {"cov_offset": (n1, np.int64), "cov_nib": ((s.n_bins + 1) // 2, np.uint8), "cov_anchor": ((s.n_bins + 1023) // 1024, np.int32),
                "exc_index": (n_exc.value, np.int64), "exc_value": (n_exc.value, np.int32), "rep_offset": (n1, np.int64),
                "rep_s": (s.n_repeats, np.int32), "rep_e": (s.n_repeats, np.int32), "frag_offset": (n1, np.int64),
                "frag_read": (s.n_fragments, np.int32), "frag_begin": (s.n_fragments, np.int32), "frag_end": (s.n_fragments, np.int32)}

        def alloc(n, dt):
            if pinned and n:
                import torch
                tdt = {np.int64: torch.int64, np.int32: torch.int32, np.uint8: torch.uint8}[dt]
                return torch.empty(int(n), dtype=tdt, pin_memory=True).numpy()
            return np.empty(n, dt)

        res = {}
        for key, (n, dt) in spec.items():
            have = out.get(key) if out else None
            res[key] = have[:n] if (have is not None and have.dtype == dt and have.size >= n and have.flags["C_CONTIGUOUS"]) else alloc(n, dt)
        ptr = {k: C.c_void_p(res[k].ctypes.data if res[k].size else 0) for k in res}
        self._check(self._lib.raft_hip_fetch_delta4(self._ctx, ptr["cov_offset"], ptr["cov_nib"], ptr["cov_anchor"], n_exc.value, ptr["exc_index"],
                                                    ptr["exc_value"], C.byref(n_exc), ptr["rep_offset"], ptr["rep_s"], ptr["rep_e"],
                                                    ptr["frag_offset"], ptr["frag_read"], ptr["frag_begin"], ptr["frag_end"]))
        return res

    def outputs_device(self) -> dict:
        """Zero-copy torch views of the device-resident outputs (valid until the next pass)."""
        import torch
        o = _Outputs()
        self._check(self._lib.raft_hip_outputs_device(self._ctx, C.byref(o)))
        s = self.summary
        n1 = s.n_reads + 1
        spec = {"cov_offset": (n1, "<i8"), "cov": (s.n_bins, "<i4"), "rep_offset": (n1, "<i8"),
                "rep_s": (s.n_repeats, "<i4"), "rep_e": (s.n_repeats, "<i4"), "cut_offset": (n1, "<i8"),
                "cuts": (s.n_cuts, "<i4"), "frag_offset": (n1, "<i8"), "frag_read": (s.n_fragments, "<i4"),
                "frag_begin": (s.n_fragments, "<i4"), "frag_end": (s.n_fragments, "<i4")}
        res = {}
        for k, (n, ts) in spec.items():
            if n == 0:
                res[k] = torch.empty(0, dtype=torch.int64 if ts == "<i8" else torch.int32, device=f"cuda:{self.device}")
            else:
                res[k] = torch.as_tensor(_DevArray(getattr(o, k), n, ts, self), device=f"cuda:{self.device}")
        return res


    def packed_device(self) -> dict | None:
        """Zero-copy torch views of the encoding the finished pass holds (raft_hip_packed_device), or None when the pass
        wrote int32: ``cov8`` (uint8; for width 2 the uint16 codes as an int16 tensor -- same bits, ``.view(torch.uint16)`` or
        ``& 0xFFFF`` after widening), ``exc_index`` / ``exc_value`` in no particular order."""
        import torch
        w, n = C.c_int32(0), C.c_int64(0)
        codes, ei, ev = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._check(self._lib.raft_hip_packed_device(self._ctx, C.byref(w), C.byref(codes), C.byref(ei), C.byref(ev), C.byref(n)))
        if w.value == 0:
            return None
        dev = f"cuda:{self.device}"
        def view(ptr, count, ts, dt):
            if count == 0:
                return torch.empty(0, dtype=dt, device=dev)
            return torch.as_tensor(_DevArray(ptr.value, count, ts, self), device=dev)
        if w.value == 8:                      # delta4: two windows per byte + block anchors
            an, na = C.c_void_p(), C.c_int64(0)
            self._check(self._lib.raft_hip_packed_anchor_device(self._ctx, C.byref(an), C.byref(na)))
            return {"width": 8, "cov_nib": view(codes, (self.summary.n_bins + 1) // 2, "|u1", torch.uint8),
                    "cov_anchor": view(an, na.value, "<i4", torch.int32),
                    "exc_index": view(ei, n.value, "<i8", torch.int64), "exc_value": view(ev, n.value, "<i4", torch.int32)}
        return {"width": w.value,
                "cov8": view(codes, self.summary.n_bins, "|u1" if w.value == 1 else "<i2", torch.uint8 if w.value == 1 else torch.int16),
                "exc_index": view(ei, n.value, "<i8", torch.int64), "exc_value": view(ev, n.value, "<i4", torch.int32)}


class Slice:
    """One rank's part of a pre-split PAF (raft_hip_slice): the slice's query coordinates on the device and, on the host, its
    grouped form -- int64 [n_runs, n_reads_total + 1], where every read of the whole set begins in every sorted run of the slice
    (raft_amd.hostio.group_offsets on the slice's query column)."""

    def __init__(self, rec_offset, qs, qe=None, device_offsets=False):
        """``qe=None``: ``qs`` holds window records (one int32-viewed word per record, hostio.pack_windows) -- one column travels.
        ``device_offsets``: keep a copy of the offsets on the slice's device (raft_hip_slice::d_rec_offset), so that exchanging the
        same slice again uploads nothing."""
        import torch
        self.off = np.ascontiguousarray(np.asarray(rec_offset), dtype=np.int64)
        if self.off.ndim != 2 or not (1 <= self.off.shape[0] <= 4):
            raise ValueError("Slice: rec_offset must be [n_runs (1..4), n_reads_total + 1]")
        for t in (qs, qe):
            if t is not None and (t.dtype != torch.int32 or not t.is_cuda or not t.is_contiguous()):
                raise TypeError("Slice needs contiguous int32 CUDA tensors")
        self.qs, self.qe = qs, qe
        self.d_off = torch.as_tensor(self.off).to(qs.device) if device_offsets else None

    def c(self) -> "_Slice":
        return _Slice(int(self.qs.numel()), int(self.off.shape[0]), self.off.ctypes.data, self.qs.data_ptr() if self.qs.numel() else 0,
                      self.qe.data_ptr() if (self.qe is not None and self.qe.numel()) else 0,
                      self.d_off.data_ptr() if self.d_off is not None else 0)


def _received_views(eng, r: "_Received") -> dict:
    """Zero-copy torch views of what a context received (valid until its next exchange)."""
    import torch
    dev = f"cuda:{eng.device}"

    def view(ptr, n, ts, dt):
        if n == 0:
            return torch.empty(0, dtype=dt, device=dev)
        return torch.as_tensor(_DevArray(ptr, n, ts, eng), device=dev)
    off = view(r.d_rec_offset, r.n_runs * (r.n_reads + 1), "<i8", torch.int64).reshape(r.n_runs, r.n_reads + 1)
    return {"n_reads": int(r.n_reads), "n_rec": int(r.n_rec), "n_runs": int(r.n_runs), "rec_offset": off,
            "qs": view(r.d_qs, r.n_rec, "<i4", torch.int32),      # (window records when the slices carried them: then "qe" is None)
            "qe": view(r.d_qe, r.n_rec, "<i4", torch.int32) if (r.d_qe or r.n_rec == 0) else None}


def exchange_local(engines, bounds, slices) -> list:
    """raft_hip_exchange_local: one process, one Engine per rank; ``bounds`` = int64 [world + 1] read ranges, ``slices`` one
    Slice per rank (on that rank's device).  Returns, per rank, the grouped input it received (torch views)."""
    import torch
    lib = load_library()
    w = len(engines)
    b = np.ascontiguousarray(np.asarray(bounds), dtype=np.int64)
    torch.cuda.synchronize()
    ctxs = (C.c_void_p * w)(*[e._ctx for e in engines])
    sl = (_Slice * w)(*[s.c() for s in slices])
    out = (_Received * w)()
    rc = lib.raft_hip_exchange_local(ctxs, w, int(slices[0].off.shape[1] - 1), C.c_void_p(b.ctypes.data), sl, out)
    engines[0]._check(rc)
    return [_received_views(e, out[i]) for i, e in enumerate(engines)]


def _records(cols) -> "_Records":
    import torch
    for t in cols:
        if t.dtype != torch.int32 or not t.is_cuda or not t.is_contiguous():
            raise TypeError("record columns must be contiguous int32 CUDA tensors")
    n = int(cols[0].numel())
    return _Records(n, *[t.data_ptr() if n else 0 for t in cols])


def presplit_symmetric_local(engines, slices_cols) -> bool:
    """raft_hip_presplit_symmetric_local: is the pre-split PAF symmetric?  ``slices_cols``: per rank its six device columns."""
    import torch
    lib = load_library()
    w = len(engines)
    torch.cuda.synchronize()
    ctxs = (C.c_void_p * w)(*[e._ctx for e in engines])
    recs = (_Records * w)(*[_records(c) for c in slices_cols])
    flag = C.c_int32(-1)
    engines[0]._check(lib.raft_hip_presplit_symmetric_local(ctxs, w, recs, C.byref(flag)))
    return bool(flag.value)


class Comm:
    """An RCCL communicator for raft_hip_exchange (one process per GPU).  ``unique_id()`` on rank 0, handed to the other ranks
    by the caller (e.g. a torch.distributed broadcast of its 128 bytes), then ``Comm(device, id, rank, world)`` everywhere."""

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        rc = load_library().raft_hip_comm_unique_id(buf)
        if rc != OK:
            raise RaftError(rc, "ncclGetUniqueId (is librccl.so.1 loadable?)")
        return buf.raw

    def __init__(self, device: int, uid: bytes, rank: int, world: int):
        self._lib = load_library()
        self._comm = C.c_void_p()
        self.rank, self.world = rank, world
        rc = self._lib.raft_hip_comm_create(device, C.c_char_p(uid), rank, world, C.byref(self._comm))
        if rc != OK:
            raise RaftError(rc, "ncclCommInitRank")

    def exchange(self, eng, bounds, sl: Slice) -> dict:
        """raft_hip_exchange on the engine's stream; returns the grouped input this rank received (torch views)."""
        b = np.ascontiguousarray(np.asarray(bounds), dtype=np.int64)
        eng.use_torch_stream()
        cs, out = sl.c(), _Received()
        rc = self._lib.raft_hip_exchange(eng._ctx, self._comm, self.rank, self.world, int(sl.off.shape[1] - 1), C.c_void_p(b.ctypes.data),
                                         C.byref(cs), C.byref(out))
        eng._check(rc)
        return _received_views(eng, out)

    def symmetric(self, eng, cols) -> bool:
        """raft_hip_presplit_symmetric: the OR over the ranks of "my slice holds the mirror of record 0"."""
        eng.use_torch_stream()
        import torch
        torch.cuda.current_stream(eng.device).synchronize()
        rec, flag = _records(cols), C.c_int32(-1)
        eng._check(self._lib.raft_hip_presplit_symmetric(eng._ctx, self._comm, self.rank, self.world, C.byref(rec), C.byref(flag)))
        return bool(flag.value)

    def close(self):
        if self._comm.value:
            self._lib.raft_hip_comm_destroy(self._comm)
            self._comm = C.c_void_p()


def selftest(device: int = 0) -> int:
    return int(load_library().raft_hip_selftest(device))


def trim(device: int = 0, keep_bytes: int = 0) -> int:
    """Hands the device's pooled placement chunks beyond keep_bytes back to the driver; returns the bytes released."""
    return int(load_library().raft_hip_trim(device, keep_bytes))


def set_placement(spread: int) -> int:
    """Placement of buffers made from now on: 0 hipMalloc, k >= 1 shuffled chunks with k-fold spread (default 8); returns the old setting."""
    return int(load_library().raft_hip_set_placement(spread))


def pool_bytes(device: int = 0) -> int:
    """Bytes of physical chunks the placement pool of the device holds at the moment (mapped by no buffer)."""
    return int(load_library().raft_hip_pool_bytes(device))
