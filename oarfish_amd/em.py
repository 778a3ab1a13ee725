"""Host-side mirror of the reference's EM interface, running on the MI355X engine.

Reference (COMBINE-lab/oarfish v0.10.3, src/em.rs):
  ``em(&EMInfo, nthreads) -> Vec<f64>``            :262-271   (gate niter>50, :212)
  ``em_par(&EMInfo, nthreads) -> Vec<f64>``        :320-447   (gate niter>1,  :399)
  ``bootstrap(&EMInfo, num_boot, nthreads) -> Vec<Vec<f64>>``  :292-314

Same names, argument meaning and results (un-normalised expected read counts,
em.rs:254).  ``nthreads`` is accepted and ignored, as ``em`` itself ignores it
(em.rs:262 ``_nthreads``); the work runs on the GPU that holds the store.
The choice between ``em`` and ``em_par`` is the caller's, as in
bulk.rs:155-159 (``threads > 4``): it only changes the stopping gate.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from .types import EMInfo, RunInfo


def _require_no_kde(em_info: EMInfo):
    if em_info.kde_model is not None:
        raise NotImplementedError(
            "kde_model is not supported: the KDE lives in the un-pinned `kders` crate and is only "
            "reachable through the hidden --use-kde flag (SURVEY.md section 8a note 4)")


def _dev(em_info: EMInfo):
    return em_info.eq_map.device_store(len(em_info.txp_info), em_info.device)


def em(em_info: EMInfo, _nthreads: int = 1) -> np.ndarray:
    """em.rs:262-271: serial-path semantics (stop when rel_diff < thresh and niter > 50)."""
    _require_no_kde(em_info)
    counts, info = _dev(em_info).em_run(em_info.init_abundances, em_info.max_iter,
                                        em_info.convergence_thresh, 50)
    em_info.last_run_info = info
    return counts


def em_par(em_info: EMInfo, nthreads: int = 8) -> np.ndarray:
    """em.rs:320-447: parallel-path semantics (stop when rel_diff < thresh and niter > 1)."""
    _require_no_kde(em_info)
    counts, info = _dev(em_info).em_run(em_info.init_abundances, em_info.max_iter,
                                        em_info.convergence_thresh, 1)
    em_info.last_run_info = info
    return counts


def bootstrap(em_info: EMInfo, num_boot: int, nthreads: int = 1, seed: int = 0,
              row_weights: Optional[np.ndarray] = None) -> List[np.ndarray]:
    """em.rs:292-314.  Returns ``num_boot`` count vectors (``Vec<Vec<f64>>``).

    The reference seeds each replicate from the OS (em.rs:274), so its stream
    is not reproducible; here ``seed`` keys a counter-based device RNG, and
    ``row_weights`` (num_boot x n_reads multiplicities) injects the resamples.
    """
    _require_no_kde(em_info)
    out, _infos = _dev(em_info).bootstrap(num_boot, seed, row_weights, em_info.init_abundances,
                                          em_info.max_iter, em_info.convergence_thresh)
    return [out[b] for b in range(num_boot)]


def em_cells(cell_row_off: Sequence[int], boundaries, ref_ids, as_probabilities,
             coverage_probabilities, n_txps: int, max_iter: int = 1000,
             convergence_thresh: float = 1e-3, device: int = 0):
    """The per-cell contract of single_cell.rs:139-160, batched on the device.

    Every cell is an independent ``em::em(&emi, 1)`` with ``init_abundances: None``.
    Returns (counts[n_cells, n_txps] f64, [RunInfo]); the caller keeps ``v > 0`` as
    (col u32, val f32) triplets (single_cell.rs:155-160).
    """
    cell_row_off = np.ascontiguousarray(cell_row_off, dtype=np.uint64)
    boundaries = np.ascontiguousarray(boundaries, dtype=np.uint64)
    ref_ids = np.ascontiguousarray(ref_ids, dtype=np.uint32)
    as_probabilities = np.ascontiguousarray(as_probabilities, dtype=np.float32)
    cov = None if coverage_probabilities is None else np.ascontiguousarray(
        coverage_probabilities, dtype=np.float64)
    n_cells = len(cell_row_off) - 1
    n_reads = len(boundaries) - 1
    nnz = len(ref_ids)
    out = np.zeros((n_cells, n_txps), dtype=np.float64)
    infos = (_lib.RunInfoC * max(n_cells, 1))()
    _lib.check(_lib.lib().oem_em_run_cells(
        cell_row_off.ctypes.data, n_cells, boundaries.ctypes.data,
        ref_ids.ctypes.data if nnz else None, as_probabilities.ctypes.data if nnz else None,
        None if cov is None else cov.ctypes.data, n_reads, nnz, n_txps, device, max_iter,
        convergence_thresh, out.ctypes.data, C.addressof(infos)))
    return out, [RunInfo(i.niter, i.n_passes, bool(i.converged), i.rel_diff)
                 for i in list(infos)[:n_cells]]


def cells_last_timing():
    """(device milliseconds of the batched EM loops, batched passes launched) of this thread's last
    ``em_cells`` call -- oem_cells_last_timing; bench.py's per-cell roofline."""
    ms, n = C.c_float(0), C.c_uint64(0)
    _lib.check(_lib.lib().oem_cells_last_timing(C.byref(ms), C.byref(n)))
    return float(ms.value), int(n.value)
