"""Host-side mirror of the reference types the EM path consumes.

Reference (COMBINE-lab/oarfish v0.10.3, src/util/oarfish_types.rs):
  ``InMemoryAlignmentStore``  :548-558, iter :651-656, len :562-564,
                              add_filtered_group :718-738, total_len / num_aligned_reads :741-748
  ``TranscriptInfo``          :431-437 (only ``len``/``lenf`` reach the EM, and only with the KDE)
  ``EMInfo``                  :408-428
  ``AlignmentFilters.model_coverage`` :792 (selects whether cov_prob is used, em.rs:108)

Same names and argument meaning as the reference so the parity tests read like
its own would; the arrays are NumPy, and the HBM-resident form is created lazily
by :meth:`InMemoryAlignmentStore.device_store`.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Iterator, Optional, Sequence, Tuple

import numpy as np

from . import _lib


@dataclass
class AlignmentFilters:
    """Only the field the EM reads (oarfish_types.rs:792)."""
    model_coverage: bool = False


@dataclass
class TranscriptInfo:
    """oarfish_types.rs:431-437.  The EM reads ``lenf`` only under --use-kde."""
    len: int = 1
    total_weight: float = 0.0
    lenf: float = 1.0

    @classmethod
    def with_len(cls, length: int) -> "TranscriptInfo":
        return cls(len=int(length), total_weight=0.0, lenf=float(length))


@dataclass
class RunInfo:
    """What do_em / em_par leave behind besides the counts (oem_run_info)."""
    niter: int
    n_passes: int
    converged: bool
    rel_diff: float


class DeviceStore:
    """RAII wrapper of an ``oem_store*`` (the matrix resident in HBM on one GPU)."""

    def __init__(self, row_ptr, tid, as_prob, cov_prob, n_txps: int, device: int = 0,
                 reorder_rows: int = 0, window_cap: int = 0, layout_build: int = 0, weight_coding: int = 0):
        self._h = C.c_void_p()
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        tid = np.ascontiguousarray(tid, dtype=np.uint32)
        as_prob = np.ascontiguousarray(as_prob, dtype=np.float32)
        cov = None if cov_prob is None else np.ascontiguousarray(cov_prob, dtype=np.float64)
        self.n_reads = len(self.row_ptr) - 1
        self.nnz = len(tid)
        self.n_txps = int(n_txps)
        self.device = int(device)
        if len(as_prob) != self.nnz or (cov is not None and len(cov) != self.nnz):
            raise ValueError("tid / as_prob / cov_prob lengths differ")
        opts = _lib.StoreOptsC()
        opts.reorder_rows = reorder_rows
        opts.window_cap = window_cap      # 0 = chosen from the store; 512 / 2048 force it (oem_store_opts)
        opts.layout_build = layout_build  # 1 = host layout builder (the specification)
        opts.weight_coding = weight_coding  # 1 = never dictionary-code the local weights
        L = _lib.lib()
        self._lib = L                     # the library that owns the handle
        self._check(L.oem_store_create(
            self.row_ptr.ctypes.data, tid.ctypes.data if self.nnz else None,
            as_prob.ctypes.data if self.nnz else None,
            None if cov is None else cov.ctypes.data, self.n_reads, self.nnz, self.n_txps,
            self.device, C.addressof(opts), C.byref(self._h)))

    def _check(self, rc: int) -> None:
        if rc != _lib.OEM_OK:
            msg = self._lib.oem_last_error()
            raise _lib.OemError(rc, msg.decode("utf-8", "replace") if msg else "")

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.oem_store_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        if not self._h.value:
            raise RuntimeError("DeviceStore is closed")
        return self._h

    # -- queries ----------------------------------------------------------
    def bytes(self) -> Tuple[int, int]:
        hbm, alg = C.c_uint64(0), C.c_uint64(0)
        self._check(self._lib.oem_store_bytes(self.handle, C.byref(hbm), C.byref(alg)))
        return int(hbm.value), int(alg.value)

    def set_option(self, option: int, value: int):
        self._check(self._lib.oem_store_set_option(self.handle, option, value))

    # -- compute ----------------------------------------------------------
    def m_step(self, theta, row_w=None) -> np.ndarray:
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        if len(theta) != self.n_txps:
            raise ValueError("theta length != n_txps")
        out = np.zeros(self.n_txps, dtype=np.float64)
        wp = None
        if row_w is not None:
            row_w = np.ascontiguousarray(row_w, dtype=np.uint32)
            if len(row_w) != self.n_reads:
                raise ValueError("row_w length != n_reads")
            wp = row_w.ctypes.data
        self._check(self._lib.oem_m_step(self.handle, theta.ctypes.data, wp, out.ctypes.data))
        return out

    def em_run(self, init=None, max_iter=1000, conv_thresh=1e-3, min_iter_gate=50):
        out = np.zeros(self.n_txps, dtype=np.float64)
        ri = _lib.RunInfoC()
        ip = None
        if init is not None:
            init = np.ascontiguousarray(init, dtype=np.float64)
            if len(init) != self.n_txps:
                raise ValueError("init_abundances length != n_txps")
            ip = init.ctypes.data
        self._check(self._lib.oem_em_run(self.handle, ip, max_iter, conv_thresh, min_iter_gate,
                                         out.ctypes.data, C.byref(ri)))
        return out, RunInfo(ri.niter, ri.n_passes, bool(ri.converged), ri.rel_diff)

    def aux_counts(self):
        """aux_counts.rs:23-50 -> (unique_count u32[T], total_count u32[T])."""
        u = np.zeros(self.n_txps, dtype=np.uint32)
        t = np.zeros(self.n_txps, dtype=np.uint32)
        self._check(self._lib.oem_aux_counts(self.handle, u.ctypes.data, t.ctypes.data))
        return u, t

    def assignment_probs(self, counts, display_thresh: float) -> np.ndarray:
        """write_function.rs:283-318: per-alignment printed probability, -1 where omitted."""
        counts = np.ascontiguousarray(counts, dtype=np.float64)
        if len(counts) != self.n_txps:
            raise ValueError("counts length != n_txps")
        out = np.zeros(self.nnz, dtype=np.float64)
        self._check(self._lib.oem_assignment_probs(self.handle, counts.ctypes.data, display_thresh,
                                                   out.ctypes.data))
        return out

    def bootstrap_weights(self, seed: int, replica: int) -> np.ndarray:
        w = np.zeros(self.n_reads, dtype=np.uint32)
        self._check(self._lib.oem_bootstrap_weights(self.handle, seed, replica, w.ctypes.data))
        return w

    def bootstrap(self, n_boot: int, seed: int = 0, row_w_all=None, init=None, max_iter=1000,
                  conv_thresh=1e-3, first_replica: int = 0):
        """n_boot replicates; replicate k uses the device resample of global replica
        ``first_replica + k`` (a pure function of (seed, replica), so processes holding the same
        store can split one set of replicates)."""
        self.set_option(_lib.OEM_OPT_BOOTSTRAP_FIRST_REPLICA, int(first_replica))
        out = np.zeros((n_boot, self.n_txps), dtype=np.float64)
        infos = (_lib.RunInfoC * max(n_boot, 1))()
        wp = None
        if row_w_all is not None:
            row_w_all = np.ascontiguousarray(row_w_all, dtype=np.uint32)
            if row_w_all.shape != (n_boot, self.n_reads):
                raise ValueError("row_w_all must be n_boot x n_reads")
            wp = row_w_all.ctypes.data
        ip = None
        if init is not None:
            init = np.ascontiguousarray(init, dtype=np.float64)
            if len(init) != self.n_txps:
                raise ValueError("init_abundances length != n_txps")
            ip = init.ctypes.data
        self._check(self._lib.oem_bootstrap(self.handle, n_boot, seed, wp, ip, max_iter, conv_thresh,
                                            out.ctypes.data, C.addressof(infos)))
        return out, [RunInfo(i.niter, i.n_passes, bool(i.converged), i.rel_diff)
                     for i in list(infos)[:n_boot]]

    def time_m_step(self, n_launches: int) -> float:
        ms = C.c_float(0)
        self._check(self._lib.oem_time_m_step(self.handle, n_launches, C.byref(ms)))
        return float(ms.value)

    def info(self, key: int) -> int:
        """oem_store_info: _lib.OEM_INFO_WEIGHT_DICT_ENTRIES / _TILES / _REMOTE_ALIGNMENTS."""
        v = C.c_uint64(0)
        self._check(self._lib.oem_store_info(self.handle, key, C.byref(v)))
        return int(v.value)

    def time_em_iters(self, n_iters: int) -> float:
        ms = C.c_float(0)
        self._check(self._lib.oem_time_em_iters(self.handle, n_iters, C.byref(ms)))
        return float(ms.value)

    def time_bootstrap_passes(self, n_passes: int):
        """(ms per batched bootstrap pass, replicates per pass, algorithmic bytes per batched pass)."""
        ms, slots, nbytes = C.c_float(0), C.c_uint32(0), C.c_uint64(0)
        self._check(self._lib.oem_time_bootstrap_passes(self.handle, n_passes, C.byref(ms), C.byref(slots),
                                                        C.byref(nbytes)))
        return float(ms.value), int(slots.value), int(nbytes.value)

    def time_allreduce(self, n_calls: int) -> float:
        """Collective: microseconds per all-reduce of the n_txps count vector (the exchange by itself)."""
        us = C.c_float(0)
        self._check(self._lib.oem_time_allreduce(self.handle, n_calls, C.byref(us)))
        return float(us.value)

    def attach_comm(self, comm_handle, global_n_reads: int, global_row_offset: int):
        self._check(self._lib.oem_store_attach_comm(self.handle, comm_handle, global_n_reads,
                                                    global_row_offset))


class InMemoryAlignmentStore:
    """oarfish_types.rs:548-558: per-read groups of (ref_id, as_prob f32, cov_prob f64).

    ``alignments`` holds ref_id only (the one AlnInfo field the EM reads,
    em.rs:41,60,103,120); ``boundaries`` is the row-pointer array that is
    private in the reference (:555).
    """

    def __init__(self, filter_opts: Optional[AlignmentFilters] = None):
        self.filter_opts = filter_opts or AlignmentFilters()
        self._chunks_tid = []
        self._chunks_p = []
        self._lens = []
        self.alignments = np.zeros(0, dtype=np.uint32)          # ref_id per alignment
        self.as_probabilities = np.zeros(0, dtype=np.float32)
        self.coverage_probabilities = np.zeros(0, dtype=np.float64)
        self.boundaries = np.zeros(1, dtype=np.uint64)          # oarfish_types.rs:645
        self._dirty = False
        self._dev = {}

    # -- construction -------------------------------------------------------
    @classmethod
    def from_arrays(cls, boundaries, ref_ids, as_probabilities, coverage_probabilities=None,
                    model_coverage: Optional[bool] = None) -> "InMemoryAlignmentStore":
        st = cls(AlignmentFilters(model_coverage=bool(
            coverage_probabilities is not None if model_coverage is None else model_coverage)))
        st.boundaries = np.ascontiguousarray(boundaries, dtype=np.uint64)
        st.alignments = np.ascontiguousarray(ref_ids, dtype=np.uint32)
        st.as_probabilities = np.ascontiguousarray(as_probabilities, dtype=np.float32)
        if coverage_probabilities is None:
            # add_filtered_group fills zeros (oarfish_types.rs:731-732)
            st.coverage_probabilities = np.zeros(len(st.alignments), dtype=np.float64)
        else:
            st.coverage_probabilities = np.ascontiguousarray(coverage_probabilities, dtype=np.float64)
        if st.boundaries[0] != 0 or st.boundaries[-1] != len(st.alignments):
            raise ValueError("boundaries must start at 0 and end at the number of alignments")
        return st

    def add_filtered_group(self, ref_ids: Sequence[int], as_probs: Sequence[float]) -> bool:
        """oarfish_types.rs:718-738: append one read's retained alignments; empty groups are dropped."""
        if len(ref_ids) == 0:
            return False
        if len(ref_ids) != len(as_probs):
            raise ValueError("ref_ids and as_probs differ in length")
        self._chunks_tid.append(np.asarray(ref_ids, dtype=np.uint32))
        self._chunks_p.append(np.asarray(as_probs, dtype=np.float32))
        self._lens.append(len(ref_ids))
        self._dirty = True
        return True

    def _flush(self):
        if not self._dirty:
            return
        self.alignments = np.concatenate([self.alignments] + self._chunks_tid)
        self.as_probabilities = np.concatenate([self.as_probabilities] + self._chunks_p)
        new_b = int(self.boundaries[-1]) + np.cumsum(np.asarray(self._lens, dtype=np.uint64))
        self.boundaries = np.concatenate([self.boundaries, new_b.astype(np.uint64)])
        self.coverage_probabilities = np.concatenate(
            [self.coverage_probabilities,
             np.zeros(len(self.alignments) - len(self.coverage_probabilities), dtype=np.float64)])
        self._chunks_tid, self._chunks_p, self._lens = [], [], []
        self._dirty = False
        self.invalidate_device()

    # -- reference accessors ------------------------------------------------
    def len(self) -> int:                       # oarfish_types.rs:562-564
        self._flush()
        return len(self.boundaries) - 1

    __len__ = len

    def num_aligned_reads(self) -> int:         # :746-748
        return self.len()

    def total_len(self) -> int:                 # :741-743
        self._flush()
        return len(self.alignments)

    def iter(self) -> Iterator[Tuple[np.ndarray, np.ndarray, np.ndarray]]:  # :651-656
        self._flush()
        b = self.boundaries
        for i in range(len(b) - 1):
            s, e = int(b[i]), int(b[i + 1])
            yield self.alignments[s:e], self.as_probabilities[s:e], self.coverage_probabilities[s:e]

    __iter__ = iter

    # -- HBM-resident form ----------------------------------------------------
    def invalidate_device(self):
        for _stamp, d in self._dev.values():
            d.close()
        self._dev = {}

    def device_store(self, n_txps: int, device: int = 0) -> DeviceStore:
        """Upload once, keep resident (the analogue of the store living in RAM across
        em / bootstrap calls, bulk.rs:131-194)."""
        self._flush()
        cov = self.coverage_probabilities if self.filter_opts.model_coverage else None  # em.rs:108
        # The reference mutates the store between calls (normalize_read_probs fills
        # coverage_probabilities, bulk.rs:103-108): the resident copy is keyed by the identity of the
        # arrays it was made from and by model_coverage, and re-made when any of them is replaced.
        # (In-place writes into an array after its upload need an explicit invalidate_device().)
        key = (int(device), int(n_txps))
        # the stamp holds the arrays themselves (compared with `is`): an id() alone can be reused by a
        # later array once the first one is freed, and a stale resident copy would then look current
        stamp = (self.boundaries, self.alignments, self.as_probabilities, cov, bool(self.filter_opts.model_coverage))
        hit = self._dev.get(key)
        if hit is not None and not (len(hit[0]) == len(stamp) and
                                    all(a is b for a, b in zip(hit[0][:4], stamp[:4])) and hit[0][4] == stamp[4]):
            hit[1].close()
            hit = None
        if hit is None:
            hit = (stamp, DeviceStore(self.boundaries, self.alignments, self.as_probabilities, cov, n_txps, device))
            self._dev[key] = hit
        return hit[1]


@dataclass
class EMInfo:
    """oarfish_types.rs:408-428."""
    eq_map: InMemoryAlignmentStore
    txp_info: Sequence[TranscriptInfo]
    max_iter: int = 1000                 # prog_opts.rs:532
    convergence_thresh: float = 1e-3     # prog_opts.rs:536
    init_abundances: Optional[np.ndarray] = None
    kde_model: Optional[object] = None   # hidden --use-kde: not supported (SURVEY.md 8a note 4)
    device: int = 0
    last_run_info: Optional[RunInfo] = field(default=None, compare=False)
