#!/usr/bin/env python3
"""One rank of the multi-PROCESS test of the row-sharded loop over the peer-to-peer exchange
(oem_p2p.hip): launched by tests/test_p2p_gpu.py under torch.distributed.run with `gloo` (only the
rendezvous and the gathers of handles / results use it), every rank on cuda:0 -- which RCCL refuses,
and hipIpc memory handles allow.

Every rank owns an nnz-balanced row shard, attaches a communicator WITHOUT RCCL and runs the collective
oem_em_run (both gates), oem_m_step, the row-sharded batched bootstrap and the all-reduce timer.  Rank 0
gathers everything and checks: identical iteration counts and BIT-identical reduced vectors on every rank,
equal to the un-sharded store's run and to the oracle's.
usage: p2p_worker.py <out.json> [full|small (exchange buffers)] [0|1|2 (shape: by rank count, one-shot, two-phase)]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402

from oarfish_amd import dist as odist, synth  # noqa: E402
from oarfish_amd.types import DeviceStore      # noqa: E402

out_path = sys.argv[1]
small_capacity = len(sys.argv) > 2 and sys.argv[2] == "small"   # exchange buffers smaller than the bootstrap's vector: pieces
shape = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
st = synth.make_store(90_000, 6_000, seed=611, threads=2)
T = st.n_txps
sh = odist.shard_rows_by_nnz(st.row_ptr, st.tid, st.as_prob, None, rank, world)
res = {}
err = None
try:
    with DeviceStore(sh.row_ptr, sh.tid, sh.as_prob, None, T, device=0) as d:
        # (many processes time-share the one device: a rank's waiting kernel can sit in front of the kernels it waits
        # for until the hardware scheduler's time slice ends -- a longer bound than the library's 8 s for the large worlds)
        comm = odist.create_comm(rank, world, 0, backend="p2p", p2p_capacity=T if small_capacity else 2 * T * 4,
                                 p2p_timeout_ms=60_000 if world > 4 else 0)
        try:
            comm.set_p2p_shape(shape)
            d.attach_comm(comm.handle, st.n_reads, sh.row_begin)
            theta = np.random.default_rng(3).uniform(0.0, 30.0, T)
            res["m_step"] = d.m_step(theta)
            res["em"], i1 = d.em_run(None, 400, 1e-3, 50)
            res["em_par"], i2 = d.em_run(None, 400, 1e-3, 1)
            res["boots"], bi = d.bootstrap(3, seed=17, max_iter=200)
            res["niter"] = [i1.niter, i2.niter] + [b.niter for b in bi]
            res["allreduce_us"] = d.time_allreduce(50)
            res["iteration_us"] = d.time_em_iters(200) / 200 * 1e3   # tile + fold + publish + exchange-in-rel-diff
        finally:
            comm.close()
except Exception as e:   # every rank must reach the gather
    err = repr(e)
gathered = [None] * world
dist.gather_object((err, res), gathered if rank == 0 else None, dst=0)
ok, report = True, {}
if rank == 0:
    errs = [g[0] for g in gathered if g[0]]
    if errs:
        ok, report = False, {"errors": errs}
    else:
        from oracle import c_oracle
        from tests.common import assert_counts_close
        rs = [g[1] for g in gathered]
        try:
            for r in range(1, world):                                   # the same reduced vectors, bit for bit
                assert rs[r]["niter"] == rs[0]["niter"], (r, rs[r]["niter"], rs[0]["niter"])
                for k in ("m_step", "em", "em_par", "boots"):
                    assert np.array_equal(rs[r][k], rs[0][k]), (r, k)
            o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, T)
            theta = np.random.default_rng(3).uniform(0.0, 30.0, T)
            assert_counts_close(rs[0]["m_step"], c_oracle.m_step(o, theta), st.n_reads, T, 1e-10, "m_step")
            want, wi = c_oracle.do_em(o, max_iter=400, conv_thresh=1e-3)
            wpar, wpi = c_oracle.do_em(o, max_iter=400, conv_thresh=1e-3, min_iter_gate=1)
            with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, T) as full:
                fcnt, finfo = full.em_run(None, 400, 1e-3, 50)
                ws = [full.bootstrap_weights(17, b) for b in range(3)]
            n = rs[0]["niter"]
            assert abs(n[0] - wi.niter) <= 1 and abs(n[1] - wpi.niter) <= 1 and abs(n[0] - finfo.niter) <= 1
            assert_counts_close(rs[0]["em"], want, st.n_reads, T, 1e-4 if n[0] != wi.niter else 1e-9, "em")
            assert_counts_close(rs[0]["em_par"], wpar, st.n_reads, T, 1e-4 if n[1] != wpi.niter else 1e-9, "em_par")
            assert_counts_close(rs[0]["em"], fcnt, st.n_reads, T, 1e-4 if n[0] != finfo.niter else 1e-9, "un-sharded")
            for b in range(3):
                wb, wbi = c_oracle.do_em(o, max_iter=200, conv_thresh=1e-3, row_w=ws[b])
                assert abs(n[2 + b] - wbi.niter) <= 1
                assert_counts_close(rs[0]["boots"][b], wb, st.n_reads, T, 1e-4 if n[2 + b] != wbi.niter else 1e-9,
                                    f"bootstrap {b}")
            with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, T) as full:
                full.time_em_iters(50)
                one = full.time_em_iters(200) / 200 * 1e3
            report = {"world": world, "shape": shape, "niter": n, "n_txps": T, "allreduce_us": [r_["allreduce_us"] for r_ in rs],
                      "sharded_iteration_us": [r_["iteration_us"] for r_ in rs], "unsharded_iteration_us": one}
        except AssertionError as e:
            ok, report = False, {"assertion": repr(e)[:2000]}
    json.dump({"ok": ok, **report}, open(out_path, "w"))
    print("p2p worker:", "OK" if ok else "FAIL", report)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
