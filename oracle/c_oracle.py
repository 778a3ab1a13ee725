"""ctypes binding of oracle/oem_oracle.c (TEST INFRASTRUCTURE ONLY).

Each wrapper mirrors one function of the C restatement; see oem_oracle.h for
the reference file:line each follows.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboem_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "oem_oracle.c")
    hdr = os.path.join(_HERE, "oem_oracle.h")
    stale = (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboem_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Store(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint64),
        ("nnz", C.c_uint64),
        ("n_txps", C.c_uint32),
        ("row_ptr", C.c_void_p),
        ("tid", C.c_void_p),
        ("as_prob", C.c_void_p),
        ("cov_prob", C.c_void_p),
    ]


class _RunInfo(C.Structure):
    _fields_ = [
        ("niter", C.c_uint32),
        ("n_passes", C.c_uint32),
        ("converged", C.c_uint32),
        ("rel_diff", C.c_double),
    ]


@dataclass
class RunInfo:
    niter: int
    n_passes: int
    converged: bool
    rel_diff: float


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_do_em.restype = C.c_int
        _lib.oracle_em_par.restype = C.c_int
        _lib.oracle_bootstrap.restype = C.c_int
        _lib.oracle_do_em_aos.restype = C.c_int
        _lib.oracle_em_par_aos.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Store:
    """Holds the CSR arrays (row_ptr u64, tid u32, as_prob f32, cov_prob f64|None)."""

    def __init__(self, row_ptr, tid, as_prob, cov_prob, n_txps):
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        self.tid = np.ascontiguousarray(tid, dtype=np.uint32)
        self.as_prob = np.ascontiguousarray(as_prob, dtype=np.float32)
        self.cov_prob = None if cov_prob is None else np.ascontiguousarray(cov_prob, dtype=np.float64)
        self.n_txps = int(n_txps)
        self.n_reads = len(self.row_ptr) - 1
        self.nnz = len(self.tid)
        assert self.row_ptr[0] == 0 and self.row_ptr[-1] == self.nnz
        self._c = _Store(
            self.n_reads, self.nnz, self.n_txps, _p(self.row_ptr), _p(self.tid), _p(self.as_prob),
            _p(self.cov_prob),
        )

    @property
    def c(self):
        return C.byref(self._c)


def _info(ri: _RunInfo) -> RunInfo:
    return RunInfo(int(ri.niter), int(ri.n_passes), bool(ri.converged), float(ri.rel_diff))


def m_step(store: Store, prev, inds=None, row_w=None):
    prev = np.ascontiguousarray(prev, dtype=np.float64)
    curr = np.zeros(store.n_txps, dtype=np.float64)
    if inds is not None:
        inds = np.ascontiguousarray(inds, dtype=np.uint64)
    if row_w is not None:
        row_w = np.ascontiguousarray(row_w, dtype=np.uint32)
    lib().oracle_m_step(store.c, _p(inds), C.c_uint64(0 if inds is None else len(inds)), _p(row_w),
                        _p(prev), _p(curr))
    return curr


def do_em(store: Store, init=None, max_iter=1000, conv_thresh=1e-3, min_iter_gate=50, inds=None,
          row_w=None):
    """em.rs:144-255 (gate 50 = em::em; gate 1 = em_par's stopping rule)."""
    out = np.zeros(store.n_txps, dtype=np.float64)
    ri = _RunInfo()
    if init is not None:
        init = np.ascontiguousarray(init, dtype=np.float64)
    if inds is not None:
        inds = np.ascontiguousarray(inds, dtype=np.uint64)
    if row_w is not None:
        row_w = np.ascontiguousarray(row_w, dtype=np.uint32)
    rc = lib().oracle_do_em(store.c, _p(init), C.c_uint32(max_iter), C.c_double(conv_thresh),
                            C.c_uint32(min_iter_gate), _p(inds),
                            C.c_uint64(0 if inds is None else len(inds)), _p(row_w), _p(out),
                            C.byref(ri))
    if rc:
        raise MemoryError("oracle_do_em failed")
    return out, _info(ri)


def default_threads() -> int:
    """Threads for the parallel entry points: the CPUs this process may actually use (its affinity mask, capped by the
    cgroup's CPU quota -- a GPU box shows 256 CPUs and grants 16), not os.cpu_count(): 256 threads spinning on
    compare-and-swap adds over 16 CPUs' worth of quota ran the C2 check at 7 iterations/s."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    def read(path):
        with open(path) as f:
            return f.read()

    try:   # cgroup v2: "quota period" or "max period"
        q = read("/sys/fs/cgroup/cpu.max").split()
        if len(q) == 2 and q[0] != "max" and float(q[1]) > 0:
            n = min(n, max(1, int(float(q[0]) / float(q[1]) + 0.5)))
    except (OSError, ValueError):
        try:   # cgroup v1
            quota = float(read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"))
            period = float(read("/sys/fs/cgroup/cpu/cpu.cfs_period_us"))
            if quota > 0 and period > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def em_par(store: Store, init=None, max_iter=1000, conv_thresh=1e-3, min_iter_gate=1, nthreads=None):
    """em.rs:320-447."""
    out = np.zeros(store.n_txps, dtype=np.float64)
    ri = _RunInfo()
    if init is not None:
        init = np.ascontiguousarray(init, dtype=np.float64)
    nthreads = nthreads or default_threads()
    rc = lib().oracle_em_par(store.c, _p(init), C.c_uint32(max_iter), C.c_double(conv_thresh),
                             C.c_uint32(min_iter_gate), C.c_int(nthreads), _p(out), C.byref(ri))
    if rc:
        raise MemoryError("oracle_em_par failed")
    return out, _info(ri)


def get_sample_inds(n: int, seed: int):
    out = np.zeros(n, dtype=np.uint64)
    lib().oracle_get_sample_inds(C.c_uint64(n), C.c_uint64(seed), _p(out))
    return out


def inds_to_weights(inds, n_reads: int):
    inds = np.ascontiguousarray(inds, dtype=np.uint64)
    w = np.zeros(n_reads, dtype=np.uint32)
    lib().oracle_inds_to_weights(_p(inds), C.c_uint64(len(inds)), C.c_uint64(n_reads), _p(w))
    return w


def bootstrap(store: Store, n_boot: int, seed=0, row_w_all=None, init=None, max_iter=1000,
              conv_thresh=1e-3, nthreads=None):
    """em.rs:292-314."""
    out = np.zeros((n_boot, store.n_txps), dtype=np.float64)
    infos = (_RunInfo * max(n_boot, 1))()
    if row_w_all is not None:
        row_w_all = np.ascontiguousarray(row_w_all, dtype=np.uint32)
        assert row_w_all.shape == (n_boot, store.n_reads)
    if init is not None:
        init = np.ascontiguousarray(init, dtype=np.float64)
    nthreads = nthreads or default_threads()
    rc = lib().oracle_bootstrap(store.c, _p(init), C.c_uint32(n_boot), C.c_uint64(seed),
                                _p(row_w_all), C.c_uint32(max_iter), C.c_double(conv_thresh),
                                C.c_int(nthreads), _p(out), infos)
    if rc:
        raise MemoryError("oracle_bootstrap failed")
    return out, [_info(infos[i]) for i in range(n_boot)]


def aux_counts(store: Store):
    u = np.zeros(store.n_txps, dtype=np.uint32)
    t = np.zeros(store.n_txps, dtype=np.uint32)
    lib().oracle_aux_counts(store.c, _p(u), _p(t))
    return u, t


def assignment_probs(store: Store, counts, display_thresh: float):
    """write_function.rs:283-318; -1 marks an alignment that is not printed."""
    counts = np.ascontiguousarray(counts, dtype=np.float64)
    out = np.zeros(store.nnz, dtype=np.float64)
    lib().oracle_assignment_probs(store.c, _p(counts), C.c_double(display_thresh), _p(out))
    return out


ALNINFO_DTYPE = np.dtype(
    [("prob", "<f8"), ("ref_id", "<u4"), ("start", "<u4"), ("end", "<u4"), ("strand", "u1")],
    align=True,
)  # 24 bytes, the expected repr(Rust) size of AlnInfo (oarfish_types.rs:330-337)


def make_aos(store: Store):
    a = np.zeros(store.nnz, dtype=ALNINFO_DTYPE)
    a["ref_id"] = store.tid
    a["end"] = 1000
    return a


def em_aos(store: Store, alns, init=None, max_iter=1000, conv_thresh=1e-3, min_iter_gate=50,
           nthreads=0):
    """Same arithmetic over the reference's 36 B/nnz layout; nthreads>0 => em_par form."""
    assert alns.dtype.itemsize == 24
    out = np.zeros(store.n_txps, dtype=np.float64)
    ri = _RunInfo()
    cov = store.cov_prob if store.cov_prob is not None else np.zeros(store.nnz, dtype=np.float64)
    mc = C.c_int(1 if store.cov_prob is not None else 0)
    if init is not None:
        init = np.ascontiguousarray(init, dtype=np.float64)
    common = (C.c_uint64(store.n_reads), C.c_uint32(store.n_txps), _p(store.row_ptr), _p(alns),
              _p(store.as_prob), _p(cov), mc, _p(init), C.c_uint32(max_iter),
              C.c_double(conv_thresh), C.c_uint32(min_iter_gate))
    if nthreads > 0:
        rc = lib().oracle_em_par_aos(*common, C.c_int(nthreads), _p(out), C.byref(ri))
    else:
        rc = lib().oracle_do_em_aos(*common, _p(out), C.byref(ri))
    if rc:
        raise MemoryError("oracle em_aos failed")
    return out, _info(ri)
