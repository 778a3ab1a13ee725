#!/usr/bin/env python3
"""Multi-rank self test of the native row-sharded path: every rank owns a row shard, runs the collective
oem_em_run / oem_bootstrap -- once over the peer-to-peer exchange (oem_p2p.hip) and once over RCCL alone --
and rank 0 compares both with a single-store run.  Launch with
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/dist_selftest.py
(on a 1-GPU box pass --same-device: every rank on cuda:0, peer to peer only -- RCCL refuses that)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# before the HIP runtime loads: RCCL's request, and the dmabuf IPC mode hipIpc memory handles need with this driver
os.environ.setdefault("HSA_NO_SCRATCH_RECLAIM", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from oarfish_amd import synth, dist as odist
from oarfish_amd.types import DeviceStore

same = "--same-device" in sys.argv
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = 0 if same else int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(dev)
dist.init_process_group("gloo" if same else "nccl")
st = synth.make_store(60_000, 20_000, seed=11, threads=2)
sh = odist.shard_rows_by_nnz(st.row_ptr, st.tid, st.as_prob, None, rank, world)
store = DeviceStore(sh.row_ptr, sh.tid, sh.as_prob, None, st.n_txps, device=dev)
comm = odist.create_comm(rank, world, dev, backend="p2p" if same else "both", p2p_capacity=2 * st.n_txps * 4)
store.attach_comm(comm.handle, st.n_reads, sh.row_begin)
cnt, info = store.em_run(None, 200, 1e-3, 1)
boot, binfo = store.bootstrap(2, seed=5, max_iter=80, conv_thresh=1e-3)
w0 = store.bootstrap_weights(5, 0)
us = store.time_allreduce(50)
if not same:   # the same once more with RCCL carrying every exchange
    comm.set_p2p_max_bytes(0)
    cnt_r, info_r = store.em_run(None, 200, 1e-3, 1)
    us_r = store.time_allreduce(50)
    comm.set_p2p_max_bytes(4 << 20)
else:
    cnt_r, info_r, us_r = cnt, info, float("nan")
ok = True
if rank == 0:
    full = DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps, device=dev)
    ref, rinfo = full.em_run(None, 200, 1e-3, 1)
    rb, rbi = full.bootstrap(2, seed=5, max_iter=80, conv_thresh=1e-3)
    wfull = full.bootstrap_weights(5, 0)
    floor = 1e-5 * st.n_reads / st.n_txps
    e1 = np.max(np.abs(cnt - ref) / np.maximum(np.abs(ref), floor))
    e2 = np.max(np.abs(boot - rb) / np.maximum(np.abs(rb), floor))
    e3 = np.max(np.abs(cnt_r - ref) / np.maximum(np.abs(ref), floor))
    ok = e1 < 1e-6 and e2 < 1e-6 and e3 < 1e-6 and abs(info.niter - rinfo.niter) <= 1 and abs(info_r.niter - rinfo.niter) <= 1 \
        and np.array_equal(w0, wfull[sh.row_begin:sh.row_end])
    print(f"exchange: p2p connected={comm.p2p} ({comm.p2p_error}), all-reduce of {st.n_txps} f64: p2p-or-default {us:.1f} us, RCCL {us_r:.1f} us; RCCL-only em rel err {e3:.2e}")
    print(f"dist selftest world={world}: em rel err {e1:.2e} (niter {info.niter} vs {rinfo.niter}), bootstrap rel err {e2:.2e}, weights shard ok={np.array_equal(w0, wfull[sh.row_begin:sh.row_end])} -> {'OK' if ok else 'FAIL'}")
    full.close()
store.close(); comm.close()
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 1)
