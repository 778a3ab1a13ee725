"""Row-sharded multi-GPU EM: one process per GPU, one RCCL all-reduce of the count
vector per E/M pass.

The reference is single-process (its only shared state is the ``Vec<AtomicF64>`` of
em.rs:338-351); sharding by reads is exact because reads are independent given the
abundances, and the only cross-read coupling is the sum into ``curr_counts``
(em.rs:74,129).  Every rank takes the identical stopping decision from the
identical all-reduced vector, so no second collective is needed.

``RowShard`` / ``shard_rows_by_nnz`` / ``em_rowsharded`` are host logic that runs on
any torch.distributed backend (the CPU tests drive them over ``gloo``);
``create_comm`` wires the native library's RCCL communicator, whose unique id is
broadcast with whatever process group the host already has.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import numpy as np

from . import _lib

MIN_READ_THRESH = 1e-5   # constants.rs:1


@dataclass
class RowShard:
    rank: int
    row_begin: int
    row_end: int
    row_ptr: np.ndarray   # local, starts at 0
    tid: np.ndarray
    as_prob: np.ndarray
    cov_prob: Optional[np.ndarray]


def shard_bounds_by_nnz(row_ptr: np.ndarray, world: int) -> List[Tuple[int, int]]:
    """Contiguous row blocks balanced by alignment count, not by read count."""
    row_ptr = np.asarray(row_ptr, dtype=np.uint64)
    n_reads = len(row_ptr) - 1
    nnz = int(row_ptr[-1])
    cuts = [0]
    for r in range(1, world):
        target = nnz * r // world
        c = int(np.searchsorted(row_ptr, target, side="left"))
        cuts.append(min(max(c, cuts[-1]), n_reads))
    cuts.append(n_reads)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def shard_rows_by_nnz(row_ptr, tid, as_prob, cov_prob, rank: int, world: int) -> RowShard:
    b, e = shard_bounds_by_nnz(row_ptr, world)[rank]
    row_ptr = np.asarray(row_ptr, dtype=np.uint64)
    a0, a1 = int(row_ptr[b]), int(row_ptr[e])
    return RowShard(rank, b, e, (row_ptr[b:e + 1] - row_ptr[b]).astype(np.uint64),
                    np.asarray(tid)[a0:a1], np.asarray(as_prob)[a0:a1],
                    None if cov_prob is None else np.asarray(cov_prob)[a0:a1])


def em_rowsharded(local_m_step: Callable[[np.ndarray], np.ndarray],
                  allreduce_sum: Callable[[np.ndarray], np.ndarray], n_txps: int,
                  global_n_reads: int, init=None, max_iter: int = 1000, conv_thresh: float = 1e-3,
                  min_iter_gate: int = 50):
    """The loop of em.rs:144-255 with the E/M pass split into rank-local partial sums and
    one all-reduce.  ``local_m_step(theta) -> partial counts`` over this rank's reads;
    ``allreduce_sum(x) -> sum over ranks``.  Control flow is a function of the reduced
    vector only, hence identical on every rank.  This is the host-driven form (used by
    the CPU/gloo tests and by external loop drivers through ``oem_m_step``); the native
    driver (``oem_em_run`` on a store with an attached communicator) runs the same loop
    on the device stream."""
    if init is not None:
        prev = np.array(init, dtype=np.float64)
    else:
        prev = np.full(n_txps, global_n_reads / n_txps, dtype=np.float64)   # em.rs:165-166
    niter, n_passes, converged, last_rel = 0, 0, False, 0.0
    while niter < max_iter:                                                 # em.rs:181
        curr = allreduce_sum(local_m_step(prev))
        n_passes += 1
        m = prev > MIN_READ_THRESH
        rel = max(0.0, float(np.max((curr[m] - prev[m]) / prev[m]))) if m.any() else 0.0
        prev = curr                                                         # em.rs:204-207
        last_rel = rel
        if rel < conv_thresh and niter > min_iter_gate:                     # em.rs:212 / :399
            converged = True
            break
        niter += 1
    prev = np.where(prev < MIN_READ_THRESH, 0.0, prev)                      # em.rs:238-242
    out = allreduce_sum(local_m_step(prev))                                 # em.rs:245-252
    n_passes += 1
    return out, niter, n_passes, converged, last_rel


def replica_range(n_boot: int, rank: int, world: int) -> Tuple[int, int]:
    """Replica-parallel bootstraps: replicates [b0, b1) of rank ``rank``.  The reference's
    replicates are independent EM runs (em.rs:303-309), so N processes that each hold the WHOLE
    store split them with no collective at all; only the final B x T matrix is gathered."""
    return n_boot * rank // world, n_boot * (rank + 1) // world


def bootstrap_replica_parallel(store, n_boot: int, seed: int, rank: int, world: int, init=None,
                               max_iter: int = 1000, conv_thresh: float = 1e-3, allgather=None):
    """``store``: a DeviceStore over the whole (un-sharded) alignment store, no communicator
    attached.  Runs this rank's replicates; replica b draws the same device resample whichever
    rank runs it.  ``allgather(array[n_local, T]) -> list of per-rank arrays`` (optional) assembles
    the full B x T matrix in replica order; without it (b0, local replicates, infos) is returned."""
    b0, b1 = replica_range(n_boot, rank, world)
    out, infos = store.bootstrap(b1 - b0, seed=seed, init=init, max_iter=max_iter,
                                 conv_thresh=conv_thresh, first_replica=b0)
    if allgather is None:
        return b0, out, infos
    return np.concatenate([np.asarray(x).reshape(-1, out.shape[1]) for x in allgather(out)], axis=0), infos


def cell_bounds_by_nnz(cell_row_off, row_ptr, world: int) -> List[Tuple[int, int]]:
    """Per-cell EM over N GPUs: contiguous blocks of cells balanced by alignment count.  Cells are
    independent problems (single_cell.rs:139-160), so there is no collective, only the final gather."""
    cell_row_off = np.asarray(cell_row_off, dtype=np.uint64)
    row_ptr = np.asarray(row_ptr, dtype=np.uint64)
    n_cells = len(cell_row_off) - 1
    cell_nnz_end = row_ptr[cell_row_off.astype(np.int64)]          # alignments before each cell boundary
    total = int(cell_nnz_end[-1])
    cuts = [0]
    for r in range(1, world):
        c = int(np.searchsorted(cell_nnz_end, total * r // world, side="left"))
        cuts.append(min(max(c, cuts[-1]), n_cells))
    cuts.append(n_cells)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def em_cells_sharded(cell_row_off, row_ptr, tid, as_prob, cov_prob, n_txps: int, rank: int, world: int,
                     max_iter: int = 1000, convergence_thresh: float = 1e-3, device: int = 0):
    """This rank's block of cells through ``em_cells``: returns (c0, c1, counts[c1-c0, T], infos)."""
    from .em import em_cells
    cell_row_off = np.asarray(cell_row_off, dtype=np.uint64)
    row_ptr = np.asarray(row_ptr, dtype=np.uint64)
    c0, c1 = cell_bounds_by_nnz(cell_row_off, row_ptr, world)[rank]
    r0, r1 = int(cell_row_off[c0]), int(cell_row_off[c1])
    a0, a1 = int(row_ptr[r0]), int(row_ptr[r1])
    # a block that starts at cell 0 is passed as views: rebasing the offsets copies 8 bytes per read
    cells_v = cell_row_off[c0:c1 + 1] if r0 == 0 else cell_row_off[c0:c1 + 1] - cell_row_off[c0]
    rows_v = row_ptr[r0:r1 + 1] if a0 == 0 else row_ptr[r0:r1 + 1] - row_ptr[r0]
    out, infos = em_cells(cells_v, rows_v,
                          np.asarray(tid)[a0:a1], np.asarray(as_prob)[a0:a1],
                          None if cov_prob is None else np.asarray(cov_prob)[a0:a1], n_txps,
                          max_iter=max_iter, convergence_thresh=convergence_thresh, device=device)
    return c0, c1, out, infos


class Comm:
    """RAII wrapper of an ``oem_comm*``."""

    def __init__(self, handle):
        self.handle = handle
        self.p2p = False        # the peer-to-peer exchange is connected on every rank
        self.p2p_error = ""

    def set_p2p_max_bytes(self, n: int):
        """Collective: vectors above `n` bytes go to RCCL (0 = always RCCL)."""
        _lib.check(_lib.lib().oem_comm_set_option(self.handle, _lib.OEM_COMM_OPT_P2P_MAX_BYTES, int(n)))

    def set_p2p_shape(self, shape: int):
        """Collective: 0 = by the number of ranks and the vector size (two-phase from three ranks and 512 KB), 1 = one-shot,
        2 = two-phase (reduce-scatter + all-gather over the mapped buffers)."""
        _lib.check(_lib.lib().oem_comm_set_option(self.handle, _lib.OEM_COMM_OPT_P2P_SHAPE, int(shape)))

    def info(self, key: int) -> int:
        """oem_comm_info: _lib.OEM_COMM_INFO_RANKS / _RCCL_RANKS (what ncclCommCount reports) / _P2P_CONNECTED."""
        import ctypes as C
        v = C.c_uint64(0)
        _lib.check(_lib.lib().oem_comm_info(self.handle, int(key), C.byref(v)))
        return int(v.value)

    def close(self):
        if self.handle is not None and self.handle.value:
            _lib.lib().oem_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def _gather_bytes(raw: bytes, rank: int, world: int, device: int) -> bytes:
    """All ranks' `raw` (equal lengths) concatenated in rank order, over the host's process group."""
    if world == 1:
        return raw
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", device) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor(list(raw), dtype=torch.uint8, device=dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return b"".join(bytes(t.cpu().tolist()) for t in parts)


def create_comm(rank: int, world: int, device: int, backend: str = "rccl", p2p_capacity: int = 0,
                p2p_self_check: bool = True, p2p_timeout_ms: int = 0) -> Comm:
    """Create the native communicator of this rank.

    ``backend``: "rccl" (the 128-byte unique id travels over the already-initialised
    torch.distributed process group), "p2p" (no RCCL at all: the one-shot peer-to-peer exchange of
    oem_p2p.hip -- also the only way to run several ranks on ONE device, which RCCL refuses), or
    "both" (RCCL for large vectors, peer to peer for the count vector; ``comm.p2p`` says whether the
    peer-to-peer side came up on every rank -- if it did not, RCCL serves everything).
    ``p2p_capacity``: doubles per exchange buffer (n_txps, or n_txps * 4 to cover a row-sharded
    batched bootstrap).  ``p2p_self_check``: connecting ends with a checked exchange in both shapes against a
    closed-form sum (OEM_COMM_OPT_P2P_SELF_CHECK) -- a rank whose check fails takes the whole communicator to
    RCCL ("both") or raises ("p2p"); ``comm.p2p_error`` then carries the first failing rank's message on every
    rank.  ``p2p_timeout_ms``: bound of one wait inside an exchange kernel (0 = the library's 8 s)."""
    import torch
    import torch.distributed as dist
    if backend not in ("rccl", "p2p", "both"):
        raise ValueError("backend must be 'rccl', 'p2p' or 'both'")
    L = _lib.lib()
    buf = None
    if backend != "p2p":
        buf = (C.c_ubyte * _lib.OEM_UNIQUE_ID_BYTES)()
        if rank == 0:
            _lib.check(L.oem_comm_unique_id(C.addressof(buf)))
        if world > 1:
            bk = dist.get_backend()
            dev = torch.device("cuda", device) if bk == "nccl" else torch.device("cpu")
            t = torch.tensor(list(bytes(buf)), dtype=torch.uint8, device=dev)
            dist.broadcast(t, src=0)
            raw = bytes(t.cpu().tolist())
            buf = (C.c_ubyte * _lib.OEM_UNIQUE_ID_BYTES).from_buffer_copy(raw)
    h = C.c_void_p()
    _lib.check(L.oem_comm_create(None if buf is None else C.addressof(buf), rank, world, device, C.byref(h)))
    comm = Comm(h)
    comm.p2p = False
    if backend in ("p2p", "both"):
        if p2p_capacity <= 0:
            raise ValueError("p2p_capacity (doubles per exchange buffer) is required for the peer-to-peer backend")
        blob = (C.c_ubyte * _lib.OEM_P2P_HANDLE_BYTES)()
        if p2p_timeout_ms:
            _lib.check(L.oem_comm_set_option(h, _lib.OEM_COMM_OPT_P2P_TIMEOUT_MS, int(p2p_timeout_ms)))
        rc = L.oem_comm_p2p_export(h, int(p2p_capacity), C.addressof(blob))
        err = L.oem_last_error().decode("utf-8", "replace") if rc != _lib.OEM_OK else ""
        # every rank must take the same road: connect only if every rank exported
        oks = _gather_bytes(bytes([1 if rc == _lib.OEM_OK else 0]), rank, world, device)
        if all(oks):
            blobs = _gather_bytes(bytes(blob), rank, world, device)
            rc = L.oem_comm_p2p_connect(h, blobs)
            err = L.oem_last_error().decode("utf-8", "replace") if rc != _lib.OEM_OK else ""
            oks = _gather_bytes(bytes([1 if rc == _lib.OEM_OK else 0]), rank, world, device)
            if all(oks) and p2p_self_check and world > 1:
                # every rank has mapped its peers: only now the checked first exchange (a rank that failed to connect
                # would otherwise leave the others waiting in a kernel for the rendezvous bound)
                rc = L.oem_comm_set_option(h, _lib.OEM_COMM_OPT_P2P_SELF_CHECK, 2)
                err = L.oem_last_error().decode("utf-8", "replace") if rc != _lib.OEM_OK else ""
                oks = _gather_bytes(bytes([1 if rc == _lib.OEM_OK else 0]), rank, world, device)
        comm.p2p = bool(all(oks))
        if not comm.p2p:   # the first failing rank's message, on every rank (the parsed bench line shows why RCCL carried the run)
            raw = (f"rank {rank}: {err}" if err else "").encode("utf-8", "replace")[:240].ljust(240, b"\0")
            allerr = _gather_bytes(raw, rank, world, device)
            msgs = [allerr[i * 240:(i + 1) * 240].rstrip(b"\0").decode("utf-8", "replace") for i in range(world)]
            err = next((m for m in msgs if m), err)
        comm.p2p_error = err
        if not comm.p2p:
            if backend == "p2p":
                comm.close()
                raise _lib.OemError(_lib.OEM_ERR_STATE, "peer-to-peer exchange did not come up on every rank: " + (err or "a peer failed"))
            _lib.check(L.oem_comm_set_option(h, _lib.OEM_COMM_OPT_P2P_MAX_BYTES, 0))   # RCCL serves everything
    return comm
