"""world_size-2 `gloo` tests of the row-sharded loop (host logic of oarfish_amd/dist.py).
The rank-local E/M pass is the oracle's m_step here (tests are the only place the
oracle is used); on the GPU the same loop runs natively with RCCL (oem_em_run on a
store with an attached communicator)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oarfish_amd import dist as odist
from oarfish_amd import synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import c_oracle
    st = synth.make_store(20_000, 900, seed=42, threads=1)
    sh = odist.shard_rows_by_nnz(st.row_ptr, st.tid, st.as_prob, None, rank, world)
    o = c_oracle.Store(sh.row_ptr, sh.tid, sh.as_prob, None, st.n_txps)

    def local(theta):
        return c_oracle.m_step(o, theta)

    def allreduce(x):
        t = torch.from_numpy(np.ascontiguousarray(x))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    res = {}
    for gate, thresh, mi in [(50, 1e-3, 1000), (1, 1e-3, 1000), (50, 0.0, 40)]:
        out, niter, npass, conv, rel = odist.em_rowsharded(local, allreduce, st.n_txps, st.n_reads,
                                                           None, mi, thresh, gate)
        res[(gate, thresh, mi)] = (out, niter, npass, conv)
    # bootstrap weights are sharded like the rows
    w = np.bincount(np.random.default_rng(1).integers(0, st.n_reads, st.n_reads), minlength=st.n_reads).astype(np.uint32)
    wl = w[sh.row_begin:sh.row_end]
    outb, nb, *_ = odist.em_rowsharded(lambda th: c_oracle.m_step(o, th, row_w=wl), allreduce, st.n_txps,
                                       st.n_reads, None, 200, 1e-3, 50)
    res["boot"] = (outb, nb)
    q.put((rank, sh.row_begin, sh.row_end, len(sh.tid), res))
    dist.barrier()
    dist.destroy_process_group()


def test_rowsharded_em_world2_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda x: x[0])
    from oracle import c_oracle
    st = synth.make_store(20_000, 900, seed=42, threads=1)
    # shards tile the rows and are balanced by alignment count
    assert got[0][1] == 0 and got[0][2] == got[1][1] and got[1][2] == st.n_reads
    assert abs(got[0][3] - got[1][3]) <= 200 and got[0][3] + got[1][3] == st.nnz
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, st.n_txps)
    for key in [(50, 1e-3, 1000), (1, 1e-3, 1000), (50, 0.0, 40)]:
        gate, thresh, mi = key
        want, wi = c_oracle.do_em(o, max_iter=mi, conv_thresh=thresh, min_iter_gate=gate)
        for r in range(world):
            out, niter, npass, conv = got[r][4][key]
            assert (niter, npass, conv) == (wi.niter, wi.n_passes, wi.converged)
            np.testing.assert_allclose(out, want, rtol=1e-9, atol=1e-9)
        # every rank holds the identical result
        assert np.array_equal(got[0][4][key][0], got[1][4][key][0])
    w = np.bincount(np.random.default_rng(1).integers(0, st.n_reads, st.n_reads), minlength=st.n_reads).astype(np.uint32)
    wantb, wib = c_oracle.do_em(o, row_w=w, max_iter=200, conv_thresh=1e-3)
    for r in range(world):
        np.testing.assert_allclose(got[r][4]["boot"][0], wantb, rtol=1e-9, atol=1e-9)


class _OracleStore:
    """Stand-in for a DeviceStore in the replica-parallel host logic: replica b's resample is a pure
    function of (seed, b), as the device's counter-based draw is."""

    def __init__(self, st):
        from oracle import c_oracle
        self.c, self.st = c_oracle, st
        self.o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, st.n_txps)

    def bootstrap(self, n_boot, seed=0, init=None, max_iter=1000, conv_thresh=1e-3, first_replica=0):
        out, infos = np.zeros((n_boot, self.st.n_txps)), []
        for k in range(n_boot):
            rng = np.random.default_rng([seed, first_replica + k])
            w = np.bincount(rng.integers(0, self.st.n_reads, self.st.n_reads), minlength=self.st.n_reads).astype(np.uint32)
            out[k], info = self.c.do_em(self.o, init=init, max_iter=max_iter, conv_thresh=conv_thresh, row_w=w)
            infos.append(info)
        return out, infos


def _replica_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st = synth.make_store(6_000, 400, seed=8, threads=1)

    def allgather(x):   # ragged: pad to the largest local count
        n = torch.tensor([x.shape[0]]); ns = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(ns, n)
        m = max(int(v) for v in ns)
        buf = torch.zeros((m, x.shape[1]), dtype=torch.float64); buf[:x.shape[0]] = torch.from_numpy(x)
        outs = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(outs, buf)
        return [o[:int(k)].numpy() for o, k in zip(outs, ns)]

    full, _ = odist.bootstrap_replica_parallel(_OracleStore(st), 5, 77, rank, world, max_iter=120, allgather=allgather)
    q.put((rank, full))
    dist.barrier()
    dist.destroy_process_group()


def test_replica_parallel_bootstrap_world2_matches_single_process():
    """Replicates split over 2 ranks with no collective, gathered in replica order = the 5 replicates
    one process computes (uneven split 2 + 3)."""
    assert [odist.replica_range(5, r, 2) for r in range(2)] == [(0, 2), (2, 5)]
    assert [odist.replica_range(3, r, 8) for r in range(8)].count((0, 0)) >= 1   # more ranks than replicates
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replica_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    st = synth.make_store(6_000, 400, seed=8, threads=1)
    want, _ = _OracleStore(st).bootstrap(5, seed=77, max_iter=120)
    for r in range(2):
        np.testing.assert_allclose(got[r], want, rtol=1e-12, atol=1e-12)


def test_cell_bounds_cover_all_cells_and_balance_alignments():
    cell_off, row_ptr, tid, p = synth.make_cells(23, 400, 300, seed=5)
    for world in (1, 2, 3, 8, 40):
        b = odist.cell_bounds_by_nnz(cell_off, row_ptr, world)
        assert b[0][0] == 0 and b[-1][1] == 23 and all(x[1] == y[0] for x, y in zip(b, b[1:]))
        nnz = [int(row_ptr[int(cell_off[c1])] - row_ptr[int(cell_off[c0])]) for c0, c1 in b]
        assert sum(nnz) == len(tid)
        if world <= 8:
            assert max(nnz) <= len(tid) / world + 2 * len(tid) / 23   # within ~2 cells of the ideal share


def test_shard_bounds():
    rp = np.array([0, 10, 11, 12, 13, 14, 24], dtype=np.uint64)
    b = odist.shard_bounds_by_nnz(rp, 2)
    assert b[0][0] == 0 and b[-1][1] == 6 and b[0][1] == b[1][0]
    assert abs(int(rp[b[0][1]]) - 12) <= 2
    for world in (1, 3, 4, 8):
        bb = odist.shard_bounds_by_nnz(rp, world)
        assert len(bb) == world and bb[0][0] == 0 and bb[-1][1] == 6
        assert all(bb[i][1] == bb[i + 1][0] for i in range(world - 1))
    # more ranks than rows: trailing shards are empty, still a tiling
    bb = odist.shard_bounds_by_nnz(np.array([0, 5], dtype=np.uint64), 4)
    assert bb[0][0] == 0 and bb[-1][1] == 1
