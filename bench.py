#!/usr/bin/env python3
"""bench.py -- EM iterations/sec (+ bootstraps/sec, cells/sec) of the MI355X EM engine.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line
on rank 0.  For N>1 it is launched under torch.distributed.run, one rank per GPU.

* workload  : BASELINE.json's metric configuration, "ONT direct-RNA-scale 10M reads x 200k
              txps" (k-bar = 8 => ~80 M alignments), synthetic, seeded.
* a "step"  : one EM loop iteration = E/M pass + (all-reduce) + rel-diff/swap/clear
              (em.rs:181-207) over the whole store, which is resident in HBM before timing.
* value     : iterations/sec of the whole job (strong scaling: the store is row-sharded over
              the N GPUs in contiguous nnz-balanced blocks, one RCCL all-reduce of the count
              vector per iteration).
* roofline  : algorithmic bytes of one E/M pass / HIP-event-timed average duration of the
              E/M kernels, against the 8 TB/s HBM peak.
* cpu_baseline : the C restatement of em_par (oracle/, host cores) on a bounded sample.
* bootstraps   : bootstraps/sec (BASELINE.json's second headline number): resampled EMs to their own
              convergence on the resident store, batches of 4 per pass over the matrix on two streams; with its own
              roofline object (bytes per SURVEY.md section 8d's batched form).  N>1: replica-parallel.
* cells        : per-cell EM (BASELINE configs[4]): a slice of cells x 50 k reads, batched on the device,
              cells/sec end to end from host buffers.  N>1: cells dealt to the ranks, no collective.

``--force-dist`` runs the whole N>1 code path (process group, native RCCL communicator, attach,
barrier / all-reduce timing, replica-parallel bootstraps, dealt cells) with a world of ONE rank, so
that the path an 8-GPU node takes is exercised on a 1-GPU box.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
SETTLE_PASSES = 200         # untimed E/M passes in front of every timing of the C3 store (the power state settles: main())
HBM_ACHIEVABLE_GBS = 6300.0  # the guide's achievable streaming rate (own microbenchmark: 6.4-7.0 TB/s reads)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "tiny"])
    ap.add_argument("--bootstraps", type=int, default=100,
                    help="bootstrap replicates to time per GPU (0 = skip; default 100 = BASELINE configs[2])")
    ap.add_argument("--no-batch-bootstrap", action="store_true", help="one replicate per pass instead of the batched chains")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="skip the rocprofv3 PMC passes that measure the pass's HBM traffic in this run (~40 s)")
    ap.add_argument("--cells-full", type=int, default=None,
                    help="cells of the whole-configs[4] leg on one GPU (default 5000 for c3 at N=1; 0 = skip)")
    ap.add_argument("--cells", type=int, default=None,
                    help="cells of the per-cell leg per GPU (0 = skip; default 625 for c3 = BASELINE configs[4]'s 5 k cells "
                         "over 8 GPUs, 16 otherwise)")
    ap.add_argument("--cell-reads", type=int, default=None, help="reads per cell (default 50000; tiny: 2000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-legs", action="store_true", help="skip the configs[1] leg and the N = 8 shard leg")
    ap.add_argument("--no-numa-bind", action="store_true",
                    help="do not bind the process to the CPUs of the GPU's NUMA node (see bind_to_gpu_numa_node)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-GPU code path even with one rank (1-GPU self test of that path)")
    ap.add_argument("--no-f32-compare", action="store_true",
                    help="skip the extra stores that time the plain f32 weight stream, the coverage column (f64 weights) and "
                         "the long-read score gaps beside the dictionary-coded store (profile collection: only the shipped "
                         "kernels in the trace)")
    ap.add_argument("--same-device", action="store_true",
                    help="every rank on cuda:0 (N processes sharing ONE GPU: RCCL refuses that, so the process group is "
                         "gloo and the count vector travels peer to peer over hipIpc-mapped buffers) -- a self test of the "
                         "N > 1 path on a 1-GPU box; the numbers say nothing about scaling")
    return ap.parse_args(argv)


WORKLOADS = {
    "c3": dict(n_reads=10_000_000, n_txps=200_000, kbar=8.0,
               name="synthetic 10M reads x 200k txps, avg 8 aln/read (BASELINE configs[2]/[3])"),
    "c2": dict(n_reads=1_000_000, n_txps=60_000, kbar=8.0,
               name="synthetic 1M reads x 60k txps, avg 8 aln/read (BASELINE configs[1])"),
    "tiny": dict(n_reads=100_000, n_txps=8_000, kbar=8.0, name="tiny plumbing workload"),
}


def make_full(cfg, world=1):
    """The whole seeded store (a pure function of the configuration), generated by every rank."""
    from oarfish_amd import synth
    threads = max(2, min(32, (os.cpu_count() or 8) // max(world, 1)))
    return synth.make_store(cfg["n_reads"], cfg["n_txps"], cfg["kbar"], threads=threads)


def make_shard(cfg, rank, world, full=None):
    """This rank's row shard: contiguous rows [r0, r1) of the seeded store, blocks balanced by
    ALIGNMENT count (DESIGN.md section 6 / north_star), not by read count.  Returns local arrays
    (row_ptr starts at 0) and the global row range."""
    from oarfish_amd import dist as odist
    st = full if full is not None else make_full(cfg, world)
    if world == 1:
        return st.row_ptr, st.tid, st.as_prob, 0, st.n_reads
    sh = odist.shard_rows_by_nnz(st.row_ptr, st.tid, st.as_prob, None, rank, world)
    return sh.row_ptr, sh.tid, sh.as_prob, sh.row_begin, sh.row_end


def host_cpu_info():
    """What the host actually gives this process: logical CPUs, the affinity mask and the cgroup quota
    (a container may see 256 CPUs and be allowed a handful)."""
    info = dict(host_cpus=os.cpu_count() or 1)
    try:
        info["affinity_cpus"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        info["affinity_cpus"] = None
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            quota = open(path).read().strip()
            if path.endswith("cfs_quota_us"):
                quota += " " + open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            break
        except OSError:
            continue
    info["cgroup_cpu_max"] = quota   # "max 100000" = no quota; "800000 100000" = 8 CPUs' worth
    return info


def usable_cpus(info):
    n = info.get("affinity_cpus") or info["host_cpus"]
    q = (info.get("cgroup_cpu_max") or "").split()
    if len(q) == 2 and q[0] not in ("max", "-1"):
        try:
            n = max(1, min(n, int(int(q[0]) / int(q[1]))))
        except (ValueError, ZeroDivisionError):
            pass
    return n


def cpu_baseline(row_ptr, tid, p, n_txps, seconds):
    """The oracle's em_par restatement (rayon + AtomicF64 analogue) on the host cores, on a
    bounded number of iterations of the same store."""
    from oracle import c_oracle
    hw = host_cpu_info()
    ncpu = usable_cpus(hw)
    s = c_oracle.Store(row_ptr, tid, p, None, n_txps)
    # the CAS-add scatter stops scaling long before all cores: pick the best thread count the way a
    # user would pick -j, on 3 passes each, and keep the whole curve so that the choice can be read
    best_t, per_pass, curve = 1, None, {}
    for nt in sorted({t for t in (1, 2, 4, 8, 16, 32, 64, 128, ncpu) if t <= ncpu}):
        t = time.perf_counter()
        c_oracle.em_par(s, max_iter=2, conv_thresh=0.0, nthreads=nt)  # 3 passes
        pp = (time.perf_counter() - t) / 3
        curve[str(nt)] = round(1.0 / pp, 3)
        if per_pass is None or pp < per_pass:
            best_t, per_pass = nt, pp
    cores = best_t
    iters = int(max(3, min(200, seconds / max(per_pass, 1e-6))))
    t = time.perf_counter()
    c_oracle.em_par(s, max_iter=iters, conv_thresh=0.0, nthreads=cores)
    dt = time.perf_counter() - t
    par = iters / dt  # iters loop iterations (+1 final pass, counted against us)
    # serial em::em semantics, 1 core, fewer iterations
    it1 = max(2, min(iters, int(3.0 / max(per_pass, 1e-6) / 8) + 2))
    t = time.perf_counter()
    c_oracle.do_em(s, max_iter=it1, conv_thresh=0.0)
    ser = it1 / (time.perf_counter() - t)
    # the reference's own layout (24-byte AlnInfo + f32 + f64 columns = 36 B/nnz): the
    # faithful variant, same thread count, a few iterations
    aos = c_oracle.make_aos(s)
    it2 = max(2, min(iters, int(4.0 / max(per_pass, 1e-6)) + 1))
    t = time.perf_counter()
    c_oracle.em_aos(s, aos, max_iter=it2, conv_thresh=0.0, min_iter_gate=1, nthreads=cores)
    par_aos = it2 / (time.perf_counter() - t)
    del aos
    return dict(value=par, unit="EM iterations/s", cores=cores, kind="port",
                sample=f"{iters} iterations of the em_par restatement (oracle/oem_oracle.c, OpenMP "
                       f"row-parallel + CAS f64 add, 8 B/nnz SoA) over the full store, with the best thread "
                       f"count of the scaling curve ({cores} of {ncpu} usable CPUs)",
                serial_1core_value=ser, reference_layout_36B_value=par_aos,
                thread_scaling_iterations_per_s=curve, usable_cpus=ncpu, **hw)


def cpu_boot_baseline(row_ptr, tid, p, n_txps, seconds, mean_passes):
    """BASELINE.md section 3 `cpu_boot` / em.rs:292-314: B resampled EMs, each a SERIAL do_em over its own
    sorted index resample (bootstrap.rs:7-16), one replicate per thread.  A replicate needs ~900 passes of
    ~0.2 s at this size, so the sample is bounded: B = threads replicates run `k` loop iterations each
    (index draw + sort included, as in the reference) and the rate is scaled by the passes a replicate
    needs to converge (`mean_passes`, measured on the device leg; equal to the oracle's by parity)."""
    from oracle import c_oracle
    hw = host_cpu_info()
    ncpu = usable_cpus(hw)
    s = c_oracle.Store(row_ptr, tid, p, None, n_txps)
    curve, best = {}, None
    for nt in sorted({t for t in (8, 16, 32, 64, 128, ncpu) if t <= ncpu}) or [1]:
        t = time.perf_counter()
        c_oracle.bootstrap(s, nt, seed=1, max_iter=1, conv_thresh=0.0, nthreads=nt)   # 2 passes per replicate
        dt = time.perf_counter() - t
        rate = nt * 2 / dt
        curve[str(nt)] = round(rate, 3)
        if best is None or rate > best[1]:
            best = (nt, rate, dt)
    nt = best[0]
    k = int(max(2, min(60, seconds / max(best[2] / 2, 1e-6) - 1)))
    t = time.perf_counter()
    c_oracle.bootstrap(s, nt, seed=2, max_iter=k, conv_thresh=0.0, nthreads=nt)       # k + 1 passes per replicate
    dt = time.perf_counter() - t
    rp_rate = nt * (k + 1) / dt
    mp = mean_passes or 930.0
    return dict(value=rp_rate / mp, unit="bootstraps/s", cores=nt, kind="port",
                replicate_passes_per_s=rp_rate, passes_per_replicate_assumed=mp,
                sample=f"{nt} replicates x {k + 1} passes, one serial resampled do_em per thread (oracle/oem_oracle.c "
                       f"oracle_bootstrap: index draw + sort + EM over the sorted resample), scaled by {mp:.0f} passes per "
                       f"replicate to convergence; best thread count of the curve ({nt} of {ncpu} usable CPUs)",
                thread_scaling_replicate_passes_per_s=curve, usable_cpus=ncpu, **hw)


def kernel_source_sha():
    """Identifies the kernels a traffic file was measured on: sha256 over the sources of the E/M and bootstrap kernels."""
    import hashlib
    h = hashlib.sha256()
    for f in ("oem_tile_kernels.hip", "oem_tile_common.h", "oem_lane_runs.h", "oem_batch_kernels.hip", "oem_layout.h"):
        with open(os.path.join(ROOT, "oarfish_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def hbm_traffic(workload):
    """HBM bytes of one pass from the rocprofv3 PMC passes of this same command (collected by
    scripts/collect_r06.sh; FETCH_SIZE corrected by the calibration recorded beside it).  The counters are not
    collected inside the run, so the file names the kernel sources it was measured on: `stale` says whether they have
    changed since."""
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
        tj = os.path.join(ROOT, "profiles", f"{tag}_{workload}_hbm_traffic.json")
        if os.path.exists(tj):
            j = json.load(open(tj))
            t = j["per_launch_bytes"]
            stale = j.get("kernel_source_sha") != kernel_source_sha()
            if stale:
                print(f"bench.py: warning: {os.path.relpath(tj, ROOT)} was collected on other kernel sources "
                      f"(roofline.traffic is stale)", file=sys.stderr)
            return (sum(t[k]["read"] + t[k]["write"] for k in ("k_em_tile", "k_remote_fold") if k in t),
                    os.path.relpath(tj, ROOT), stale)
    return None, None, None


def live_hbm_traffic(workload, full, cfg, timeout_s=90):
    """HBM bytes of one E/M pass measured IN THIS RUN, on this box: the PMC passes of MI355X_MICROARCH.md's recipe --
    rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE and the FETCH_SIZE calibration stream, each its own pass with
    --kernel-trace only -- over a child process that runs the same pass on the same store (scripts/pass_time.py; the
    store is handed over through the scripts' cache, not generated again).  Returns (bytes per pass, source) or
    (None, reason): the tracked JSON of profiles/ then stands in (hash-stamped, `traffic_stale`)."""
    import shutil
    import subprocess
    if workload not in ("c3", "c2"):
        return None, "no live traffic for this workload"
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    try:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import _ab  # noqa: F401  (the scripts' store cache)
        _ab.put_store(full, cfg["n_reads"], cfg["n_txps"], cfg["kbar"])
        from oarfish_amd import build as _b
        _b.build_microbench()
        out = os.path.join(ROOT, "gpurun_out", "live_traffic")
        shutil.rmtree(out, ignore_errors=True)
        os.makedirs(out, exist_ok=True)
        env = dict(os.environ, TMPDIR="/tmp")
        child = [sys.executable, os.path.join("scripts", "pass_time.py"), workload]
        stream = [os.path.join("scripts", "microbench", "stream")]
        for sub, ctr, tag, cmd in (("cal", "FETCH_SIZE", "cal", stream), ("pf", "FETCH_SIZE", "live", child),
                                   ("pw", "WRITE_SIZE", "live", child)):
            r = subprocess.run(["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d",
                                os.path.join(out, sub), "-o", tag, "--"] + cmd, cwd=ROOT, env=env, capture_output=True,
                               text=True, timeout=timeout_s)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {ctr} failed ({r.returncode}): {r.stderr[-200:]}"
        js = os.path.join(out, "live_hbm_traffic.json")
        r = subprocess.run([sys.executable, os.path.join("scripts", "hbm_traffic_json.py"), out, workload, js, "live"],
                           cwd=ROOT, capture_output=True, text=True, timeout=60)
        if r.returncode != 0:
            return None, "hbm_traffic_json.py failed: " + r.stderr[-200:]
        t = json.load(open(js))["per_launch_bytes"]
        return (sum(t[k]["read"] + t[k]["write"] for k in ("k_em_tile", "k_remote_fold") if k in t),
                "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (own passes, calibrated) over scripts/pass_time.py in this run")
    except Exception as e:  # pragma: no cover  (a profiler that hangs or is missing must not cost the bench line)
        return None, repr(e)


AFFINITY_AT_START = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None


def bind_to_gpu_numa_node(torch, device):
    """Run this process on the CPUs next to its GPU, as `numactl --cpunodebind` in a launcher would: the caller-side
    arrays are pageable host memory, first touched by the thread that fills them, and the boundary copies them over
    PCIe -- from the GPU's own NUMA node at ~35-55 GB/s, from the other socket at a third of that (the per-cell leg of
    one box in three took 0.95-1.0 s instead of 0.70: 2.25 GB up, 0.3 GB back).  Returns what was done, for the line."""
    try:
        pr = torch.cuda.get_device_properties(device)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        base = f"/sys/bus/pci/devices/{bdf}"
        node = int(open(f"{base}/numa_node").read().strip())
        cpus = set()
        for part in open(f"{base}/local_cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if node < 0 or not cpus:
            return dict(bound=False, reason="no NUMA node reported for " + bdf)
        os.sched_setaffinity(0, cpus)
        return dict(bound=True, pci=bdf, numa_node=node, cpus=len(cpus))
    except Exception as e:  # pragma: no cover  (no sysfs, no such attribute: run unbound)
        return dict(bound=False, reason=repr(e))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist_mode = world > 1 or args.force_dist
    if dist_mode:
        # RCCL asks for this with the current HIP runtime ("must be set to avoid performance degradation");
        # it has to be in the environment before the HIP runtime loads
        os.environ.setdefault("HSA_NO_SCRATCH_RECLAIM", "1")
        # hipIpc memory handles (the peer-to-peer exchange of oem_p2p.hip) need the dmabuf IPC mode with this driver;
        # without it hipIpcGetMemHandle fails and the run would silently time RCCL only
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:   # not under torch.distributed.run (--force-dist by hand): any free port
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: oarfish_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    host_binding = dict(bound=False, reason="--no-numa-bind") if args.no_numa_bind else bind_to_gpu_numa_node(torch, local_rank)
    if dist_mode:
        if args.same_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from oarfish_amd import _lib
    from oarfish_amd import build as _b
    if rank == 0:
        _b.build()
    if dist_mode:
        dist.barrier()
    from oarfish_amd.types import DeviceStore
    from oarfish_amd import dist as odist
    from oarfish_amd import synth

    cfg = WORKLOADS[args.workload]
    t_gen = time.perf_counter()
    full = make_full(cfg, world)
    row_ptr, tid, p, r0, r1 = make_shard(cfg, rank, world, full)
    t_gen = time.perf_counter() - t_gen
    # the first store of a process also pays for the HIP runtime (context, first allocation, code objects:
    # ~0.15 s); a small one takes that, so that `upload_s` is what creating the store costs by itself
    t_first = time.perf_counter()
    DeviceStore(row_ptr[:1025], tid[:int(row_ptr[1024])], p[:int(row_ptr[1024])], None, cfg["n_txps"],
                device=local_rank).close()
    t_first = time.perf_counter() - t_first
    t_up = time.perf_counter()
    store = DeviceStore(row_ptr, tid, p, None, cfg["n_txps"], device=local_rank)
    t_up = time.perf_counter() - t_up
    comm = None
    exchange = None
    shard_compute_us = None
    if dist_mode:
        # What this rank's shard costs by itself: the E/M pass of the un-attached shard store (tile kernel + fold; the
        # sweep over the count vector is part of the exchange kernels in a sharded run), HIP-event-timed -- beside the
        # exchange's own time and the sharded iteration below, so that a scaling curve says which of the two did not
        # scale (config.exchange).
        store.time_m_step(20)
        shard_compute_us = store.time_m_step(100) * 1e3
        # RCCL plus the one-shot peer-to-peer exchange for the 1.6 MB count vector (oem_p2p.hip); world 1 under
        # --force-dist: a real one-rank RCCL communicator and a one-rank exchange buffer
        comm = odist.create_comm(rank, world, local_rank, backend="p2p" if args.same_device else "both",
                                 p2p_capacity=cfg["n_txps"])
        store.attach_comm(comm.handle, cfg["n_reads"], r0)
        # (--same-device: no RCCL -- it refuses two ranks per device -- and a gloo group for the agreement)
        exchange = pick_exchange(store, comm, dist, torch, have_rccl=not args.same_device, cpu_group=args.same_device)
        # the three numbers of one rank's iteration side by side, per candidate: the shard's compute alone, the
        # exchange alone (back-to-back all-reduces of the count vector), the sharded iteration as timed
        sc = torch.tensor([shard_compute_us], dtype=torch.float64, device="cpu" if args.same_device else "cuda")
        per_rank = [torch.zeros_like(sc) for _ in range(world)]
        dist.all_gather(per_rank, sc)
        exchange["shard_compute_us"] = max(float(t.item()) for t in per_rank)
        exchange["shard_compute_us_per_rank"] = [round(float(t.item()), 2) for t in per_rank]
        for rec in exchange["candidates"].values():
            if rec.get("ok"):
                rec["exchange_us"] = rec["allreduce_us"]
                rec["shard_compute_us"] = exchange["shard_compute_us"]
                rec["iteration_minus_compute_us"] = rec["iteration_us"] - exchange["shard_compute_us"]
        exchange["ranks"] = comm.info(_lib.OEM_COMM_INFO_RANKS)
        exchange["rccl_ranks_seen"] = comm.info(_lib.OEM_COMM_INFO_RCCL_RANKS)   # ncclCommCount; 0: no RCCL in this run
        exchange["p2p_connected_this_rank"] = bool(comm.info(_lib.OEM_COMM_INFO_P2P_CONNECTED))

    def sync():
        torch.cuda.synchronize()
        if dist_mode:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not dist_mode:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device="cpu" if args.same_device else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    # Dominant kernels first: the HIP-event-timed average launch duration of the E/M pass (1 untimed + 50 timed launches,
    # 7 ms at C3).  It runs BEFORE the timed loop since round 6: the kernel trace of this sequence
    # (profiles/r06_c3_loop_trace.txt) shows no gap between the loop's launches and no slower kernels inside the loop --
    # but k_em_tile's duration follows the power state: 121 -> 136 us over the 3 ms of a 20-iteration loop that starts
    # 1 ms after the device woke up, 119.5 us in the steady state every real run (890 passes = 130 ms) is in.  With the
    # pass timing in front, the loop and `kernel_avg_ms` are measured in the same, settled state; a second pass timing
    # behind the loop is on the line as `kernel_avg_ms_after_loop`.  The slow patch sits 3-10 ms after the wake-up
    # (whatever runs then: with the pass timing first it read 0.152 ms, the one behind the loop 0.142), so 200 untimed
    # passes (29 ms) go first: `config.settle_passes`.
    hbm_bytes, alg_bytes = store.bytes()
    store.time_m_step(SETTLE_PASSES)
    k_ms = store.time_m_step(50)
    # warmup, then exactly K timed iterations
    if args.warmup > 0:
        store.time_em_iters(args.warmup)
    sync()
    t0 = time.perf_counter()
    dev_ms = store.time_em_iters(args.steps)
    sync()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    it_per_s = args.steps / elapsed
    k_ms_after = store.time_m_step(50)
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    traffic, traffic_src, traffic_stale = hbm_traffic(args.workload) if world == 1 else (None, None, None)
    roofline = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=achieved / HBM_PEAK_GBS, traffic=traffic,
                    kernel="k_em_tile + k_remote_fold (one E/M pass)", kernel_avg_ms=k_ms,
                    algorithmic_bytes_per_launch=alg_bytes, traffic_source=traffic_src, traffic_stale=traffic_stale,
                    frac_of_achievable=achieved / HBM_ACHIEVABLE_GBS, achievable_peak=HBM_ACHIEVABLE_GBS,
                    kernel_avg_ms_after_loop=k_ms_after)
    if traffic:
        # the bytes the kernels actually moved (PMC passes of the same command) over the same duration: the rate of
        # the memory system, next to the rate of the algorithmic bytes that `frac` is
        roofline["traffic_gbs"] = traffic / (k_ms * 1e-3) / 1e9
        roofline["frac_traffic"] = roofline["traffic_gbs"] / HBM_PEAK_GBS
    # The store's local weights are dictionary-coded when it has at most 256 distinct ones (as_prob is exp of an
    # integer score gap over a constant; the C3 store has 98): say so, and time the same pass on the plain f32
    # stream beside it (oem_store_opts.weight_coding = 1) so that both figures are on the line.
    def coding_of(n):
        return ("f32" if not n else "fused7" if n <= 128 else "bytes8" if n <= 256 else "words16")

    n_dict = store.info(_lib.OEM_INFO_WEIGHT_DICT_ENTRIES)
    roofline["weight_coding"] = coding_of(n_dict)   # oem_layout_dict.hip: lossless table of the store's distinct f32 weights
    roofline["weight_dict_entries"] = n_dict
    # The same pass on three more stores of the same shape, as numeric keys (the coded figure depends on the store having
    # few distinct weights): the plain f32 stream (weight_coding = 1), the coverage column (f64 weights, 12 B / alignment,
    # the authors' recommended --model-coverage) and long-read score gaps (uniform on [0, 0.05 best], best up to 20 000:
    # ~520 distinct weights -> 16-bit indices).  `frac` of each against its own algorithmic bytes.
    if rank == 0 and world == 1 and not args.no_f32_compare:
        def timed(label, rp, ti, pp, cov, batched=False, **kw):
            try:
                bp = None
                with DeviceStore(rp, ti, pp, cov, cfg["n_txps"], device=local_rank, **kw) as o:
                    o.time_em_iters(5)
                    pk = min(o.time_m_step(50) for _ in range(3))   # (three runs of 50 launches: the side stores are timed once, cold)
                    pit = o.time_em_iters(args.steps) / args.steps
                    _h, ab = o.bytes()
                    nd = o.info(_lib.OEM_INFO_WEIGHT_DICT_ENTRIES)
                    nrem = o.info(_lib.OEM_INFO_REMOTE_ALIGNMENTS)
                    if batched:   # one batched bootstrap pass (4 replicates) over the same store
                        o.time_bootstrap_passes(5)
                        bp = o.time_bootstrap_passes(20)
                roofline["frac_" + label] = ab / (pk * 1e-3) / 1e9 / HBM_PEAK_GBS
                roofline[label] = dict(kernel_avg_ms=pk, device_ms_per_step=pit, algorithmic_bytes_per_launch=ab,
                                       weight_coding=coding_of(nd), weight_dict_entries=nd, remote_alignments=nrem)
                if bp:
                    roofline[label].update(batched_pass_ms=bp[0], replicates_per_batched_pass=bp[1],
                                           us_per_replicate_pass=bp[0] / bp[1] * 1e3)
            except Exception as e:  # pragma: no cover
                roofline[label] = dict(error=repr(e))
        if n_dict:
            timed("f32_stream", row_ptr, tid, p, None, weight_coding=1)
        else:
            roofline["frac_f32_stream"] = roofline["frac"]
        if args.workload != "tiny":
            threads = max(2, min(32, os.cpu_count() or 8))
            cv = synth.make_store(cfg["n_reads"], cfg["n_txps"], cfg["kbar"], coverage=True, threads=threads)
            timed("coverage", cv.row_ptr, cv.tid, cv.as_prob, cv.cov_prob)
            # the same store with the opt-in oem_store_opts.weight_coding = 2: the static weight p * cov rounded once to
            # f32 (<= 6e-8 relative per weight; SURVEY 8a note 2), 8 B per alignment through the f32 kernels
            timed("coverage_f32w", cv.row_ptr, cv.tid, cv.as_prob, cv.cov_prob, weight_coding=2)
            del cv
            ug = synth.make_store(cfg["n_reads"], cfg["n_txps"], cfg["kbar"], threads=threads, gaps="uniform")
            timed("uniform_gaps", ug.row_ptr, ug.tid, ug.as_prob, None)
            del ug
            # What the engine does when a read's far alignments RECUR (multi-mapping reads hit paralogs, not random
            # transcripts): the BASELINE generator's 20 % uniformly random far hits are the worst case of the remote
            # path (a quarter of the pass).  `paralog`: far hits inside families of three genes scattered over the
            # annotation; `paralog_adjacent`: the same families numbered next to each other -- the annotation a
            # co-mapping renumbering at store creation would produce (not built: this prices it).
            for far in ("paralog", "paralog_adjacent"):
                pf = synth.make_store(cfg["n_reads"], cfg["n_txps"], cfg["kbar"], threads=threads, far=far)
                timed(far, pf.row_ptr, pf.tid, pf.as_prob, None, batched=True)
                del pf

    # EM to convergence with the reference's defaults (max_iter 1000, thresh 1e-3), both gates
    sync()
    conv = {}
    for gate, name in ((1, "em_par"), (50, "em")):
        tc = time.perf_counter()
        _cnt, info = store.em_run(None, 1000, 1e-3, gate)
        sync()
        conv[name] = dict(niter=info.niter, n_passes=info.n_passes, converged=info.converged,
                          seconds=time.perf_counter() - tc)

    # bootstraps/sec (each = one resampled EM to convergence, em.rs:273-290)
    boots = None
    if args.bootstraps > 0:
        boots = bootstrap_leg(args, cfg, full, store, dist_mode, rank, world, local_rank, sync, max_over_ranks)

    # per-cell EM (single_cell.rs:139-160), batched on the device
    cells = None
    n_cells = args.cells if args.cells is not None else (625 if args.workload == "c3" else 16)
    if n_cells > 0:
        cells = cells_leg(args, n_cells, rank, world, local_rank, sync, max_over_ranks)

    # BASELINE configs[1] (1 M reads x 60 k transcripts, 1000 EM iterations) and the shard one of eight ranks would own
    # (configs[3]): both are ONE wave of workgroups deep -- the regime where a tile's chain, not the memory system, sets
    # the pass -- and were builder-only figures until round 6
    c2 = shard_n8 = None
    if rank == 0 and world == 1 and args.workload == "c3" and not args.no_side_legs:
        c2 = c2_leg(local_rank, sync)
        shard_n8 = shard_leg(full, cfg, 8, local_rank)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the CPU legs run on every CPU the cgroup grants, not on the GPU's NUMA node alone (the binding above would
        # hand the reference's side fewer cores on a multi-socket box and flatter the ratio)
        if AFFINITY_AT_START:
            try:
                os.sched_setaffinity(0, AFFINITY_AT_START)
            except OSError:
                pass
        cpu = cpu_baseline(row_ptr, tid, p, cfg["n_txps"], args.cpu_seconds)
        if boots and boots.get("value"):
            boots["cpu_baseline"] = cpu_boot_baseline(row_ptr, tid, p, cfg["n_txps"], args.cpu_seconds,
                                                      boots.get("mean_passes"))

    if rank == 0 and world == 1 and not args.no_live_traffic:
        # the pass's HBM traffic measured on THIS box in THIS run (the tracked JSON pairs the builder's box with this
        # box's time); last, so that the child processes run beside nothing that is timed
        lt, lsrc = live_hbm_traffic(args.workload, full, cfg)
        roofline["traffic_tracked"] = roofline.get("traffic")
        roofline["traffic_tracked_source"] = roofline.get("traffic_source")
        if lt:
            roofline["traffic"], roofline["traffic_source"], roofline["traffic_stale"] = lt, lsrc, False
            roofline["traffic_gbs"] = lt / (roofline["kernel_avg_ms"] * 1e-3) / 1e9
            roofline["frac_traffic"] = roofline["traffic_gbs"] / HBM_PEAK_GBS
        else:
            roofline["traffic_live_error"] = lsrc
    if rank == 0:
        out = {
            "metric": "EM iterations/sec",
            "value": it_per_s,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": cfg["name"], "n_reads": cfg["n_reads"], "n_txps": cfg["n_txps"],
                       "nnz_local": int(len(tid)), "rows_local": int(r1 - r0),
                       "parallelism": f"row-shard x{world} (nnz-balanced)" + (" [forced dist path]" if args.force_dist and world == 1 else "")
                                      + (" [all ranks on ONE device: self test, not scaling]" if args.same_device else ""),
                       "gen_s": round(t_gen, 2), "upload_s": round(t_up, 3), "runtime_init_s": round(t_first, 3),
                       "host_binding": host_binding,
                       "device_ms_per_step": dev_ms / args.steps, "settle_passes": SETTLE_PASSES,
                       "order": "settle passes, pass timing (1 + 50 launches), W warm-up iterations, K timed iterations, pass timing again",
                       "exchange": exchange},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "bootstraps": boots,
            "cells": cells,
            "em_to_convergence": conv,
            "c2": c2,
            "shard_n8": shard_n8,
        }
        print(json.dumps(out))
    store.close()
    if comm is not None:
        comm.close()
    if dist_mode:
        dist.destroy_process_group()


def c2_leg(local_rank, sync):
    """BASELINE configs[1]: synthetic 1 M reads x 60 k transcripts (avg 8 alignments per read), one GPU, 1000 EM
    iterations (max_iter 1000, threshold 0: SURVEY 8a note 3) through oem_em_run -- wall time of the whole call, result
    copied out -- plus the HIP-event-timed pass and loop iteration and the pass's fraction of the HBM peak (the store's
    69 MB fit the Infinity Cache: a fraction above 1 would be possible here and is far away)."""
    from oarfish_amd import _lib, synth
    from oarfish_amd.types import DeviceStore
    cfg = WORKLOADS["c2"]
    st = synth.make_store(cfg["n_reads"], cfg["n_txps"], cfg["kbar"], threads=max(2, min(32, os.cpu_count() or 8)))
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps, device=local_rank) as d:
        d.time_m_step(500)                                   # settle (see SETTLE_PASSES)
        pass_ms = min(d.time_m_step(200) for _ in range(3))
        it_ms = min(d.time_em_iters(1000) for _ in range(3)) / 1000
        _h, alg = d.bytes()
        d.em_run(None, 100, 0.0, 50)
        best, info = None, None
        for _ in range(3):
            sync()
            t = time.perf_counter()
            _cnt, info = d.em_run(None, 1000, 0.0, 50)
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        return dict(workload=cfg["name"], value=1000 / best, unit="EM iterations/s", iterations=int(info.niter),
                    n_passes=int(info.n_passes), em_run_seconds=best, kernel_avg_ms=pass_ms, device_ms_per_step=it_ms,
                    algorithmic_bytes_per_launch=alg, frac=alg / (pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    alignments=int(st.tid.size), tiles=d.info(_lib.OEM_INFO_TILES),
                    note="value = 1000 iterations / wall time of one oem_em_run call (best of 3), counts copied out; "
                         "kernel_avg_ms = one E/M pass (HIP events, best of 3 x 200 launches)")


def shard_leg(full, cfg, n, local_rank):
    """What rank 0 of `n` would compute per iteration of configs[3]: its nnz-balanced row shard of the C3 store as an
    un-attached store on this GPU -- E/M pass (tile kernel + fold: what a row shard's iteration has in front of its
    exchange kernels) and the single-device loop iteration, HIP-event-timed.  No exchange: one GPU."""
    from oarfish_amd import dist as odist
    from oarfish_amd.types import DeviceStore
    sh = odist.shard_rows_by_nnz(full.row_ptr, full.tid, full.as_prob, None, 0, n)
    with DeviceStore(sh.row_ptr, sh.tid, sh.as_prob, None, cfg["n_txps"], device=local_rank) as d:
        d.time_m_step(500)
        pass_us = min(d.time_m_step(200) for _ in range(3)) * 1e3
        it_us = min(d.time_em_iters(500) for _ in range(3)) / 500 * 1e3
        return dict(n=n, reads=int(sh.row_end - sh.row_begin), alignments=int(sh.tid.size), pass_us=pass_us,
                    iteration_us=it_us, note="rank 0's row shard as an un-attached store: no exchange in these figures")


def pick_exchange(store, comm, dist, torch, have_rccl=True, cpu_group=False):
    """Which exchange carries the count vector of the timed iterations: RCCL's all-reduce, or the peer-to-peer
    exchange of oem_p2p.hip in its one-shot or its two-phase shape.  Every candidate first has to reproduce
    the reference result on this node (a few iterations, compared with RCCL's -- or, without RCCL, with the
    other shape's -- and agreed over all ranks); the survivors are timed on whole EM iterations (HIP events,
    maximum over ranks) and the fastest is kept.  The all-reduce of the count vector by itself is timed too."""
    dev = "cpu" if cpu_group else "cuda"

    def all_min(x):
        t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item())

    def all_max(x):
        t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    names = {0: "rccl", 1: "p2p one-shot (every rank reads every partial; summed inside the rel-diff kernel)",
             2: "p2p two-phase (rank r sums slice r, all ranks read the reduced slices inside the rel-diff kernel)"}
    out = dict(backend="rccl" if have_rccl else None, p2p_connected=bool(comm.p2p), p2p_error=comm.p2p_error or None,
               candidates={})
    cands = ([0] if have_rccl else []) + ([1, 2] if comm.p2p else [])
    ref = None
    for c in cands:
        rec = {}
        try:
            if c == 0:
                comm.set_p2p_max_bytes(0)                        # RCCL serves everything
            else:
                comm.set_p2p_max_bytes(4 << 20)
                comm.set_p2p_shape(c)
            a, _ = store.em_run(None, 3, 0.0, 50)
            ok = True
            if ref is None:
                ref = a
            else:
                ok = bool(np.allclose(a, ref, rtol=1e-10, atol=1e-9))
            rec["allreduce_us"] = store.time_allreduce(50)
            store.time_em_iters(20)
            rec["iteration_us"] = store.time_em_iters(100) / 100 * 1e3
        except Exception as e:
            ok, rec["error"] = False, repr(e)
        rec["ok"] = all_min(1.0 if ok else 0.0) > 0.5         # every rank, or nobody
        if rec["ok"]:
            rec["iteration_us"] = all_max(rec["iteration_us"])
            rec["allreduce_us"] = all_max(rec["allreduce_us"])
        out["candidates"][names[c].split(" (")[0]] = rec
    good = [c for c in cands if out["candidates"][names[c].split(" (")[0]]["ok"]]
    if not good:
        raise SystemExit("no exchange of the count vector works on this node: " + json.dumps(out))
    best = min(good, key=lambda c: out["candidates"][names[c].split(" (")[0]]["iteration_us"])
    if best == 0:
        comm.set_p2p_max_bytes(0)
    else:
        comm.set_p2p_max_bytes(4 << 20)
        comm.set_p2p_shape(best)
    out["backend"] = names[best]
    out["candidate_ok"] = {k: bool(v.get("ok")) for k, v in out["candidates"].items()}
    if best == 0:   # RCCL carried the timed region: say why the peer-to-peer exchange did not
        out["p2p_not_used_because"] = (("not connected: " + (comm.p2p_error or "a rank failed to export / connect / self-check")) if not comm.p2p
                                       else "slower than RCCL on this node, or its result differed (see candidates)")
    out["allreduce_us"] = out["candidates"][names[best].split(" (")[0]]["allreduce_us"]
    r = out["candidates"].get("rccl")
    out["rccl_allreduce_us"] = r["allreduce_us"] if r and r.get("ok") else None
    return out


def bootstrap_leg(args, cfg, full, store, dist_mode, rank, world, local_rank, sync, max_over_ranks):
    import torch
    from oarfish_amd import _lib
    from oarfish_amd import dist as odist
    from oarfish_amd.types import DeviceStore
    n_total = args.bootstraps * world
    err, tb, passes, roof = None, float("nan"), [], None
    own = None
    try:
        if dist_mode:
            # replica-parallel: replicates are independent EM runs (em.rs:303-309), so every rank holds the
            # WHOLE store (1.5 GB of 288 GB) and runs its own replicates -- no collective until the gather
            own = DeviceStore(full.row_ptr, full.tid, full.as_prob, None, cfg["n_txps"], device=local_rank)
            bstore = own
        else:
            bstore = store
        if args.no_batch_bootstrap:
            bstore.set_option(_lib.OEM_OPT_BATCH_BOOTSTRAP, 0)
        bstore.bootstrap(2, seed=99, max_iter=2)   # untimed: allocates the batch buffers (once per store)
        if rank == 0 and not args.no_batch_bootstrap:
            try:   # dominant kernels of the batched pass, timed with HIP events on the store's stream
                ms, slots, nbytes = bstore.time_bootstrap_passes(10)
                ach = nbytes / (ms * 1e-3) / 1e9
                traffic, tsrc = None, None
                tj = next((q for q in (os.path.join(ROOT, "profiles", f"{tag}_{args.workload}_boot_hbm_traffic.json")
                                      for tag in ("r06", "r05", "r04", "r03", "r02")) if os.path.exists(q)), "")
                tstale = None
                if tj:   # PMC passes of the same kernels (scripts/collect_pmc_cmd.sh on scripts/boot_passes.py)
                    j = json.load(open(tj))
                    traffic, tsrc = j["per_batched_pass_bytes_total"], os.path.relpath(tj, ROOT)
                    tstale = j.get("kernel_source_sha") != kernel_source_sha()
                roof = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                            traffic=traffic, traffic_source=tsrc, traffic_stale=tstale,
                            kernel="k_em_tile_e + k_remote_fold_b + k_reldiff_b (one batched pass)",
                            kernel_avg_ms=ms, replicates_per_launch=slots, algorithmic_bytes_per_launch=nbytes,
                            us_per_replicate_pass=ms / slots * 1e3)
            except Exception as e:   # a store that cannot batch (f64 weights, wide windows)
                roof = dict(error=repr(e))
        sync()
        t0 = time.perf_counter()
        _b0, _out, infos = odist.bootstrap_replica_parallel(bstore, n_total, 1, rank, world)
        torch.cuda.synchronize()
        tb = time.perf_counter() - t0
        passes = [i.n_passes for i in infos]
    except Exception as e:  # pragma: no cover - this leg is optional: the headline line is still printed
        err = repr(e)
    finally:
        if own is not None:
            own.close()
    if dist_mode:
        import torch.distributed as dist
        ok = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device="cpu" if args.same_device else "cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 1.0:
            return dict(value=None, error=err or "a rank failed", mode="replica-parallel")
        tb = max_over_ranks(tb)
    elif err:
        return dict(value=None, error=err)
    mode = (f"replica-parallel over {world} GPU(s), whole store on each, no collective" if dist_mode else "one GPU")
    return dict(value=n_total / tb, unit="bootstraps/s", n=n_total, seconds=tb,
                mean_passes=float(np.mean(passes)) if passes else None, mode=mode,
                replicates_per_pass=1 if args.no_batch_bootstrap else (roof or {}).get("replicates_per_launch"),
                chains=1 if args.no_batch_bootstrap else 2, roofline=roof)


def cells_traffic(n_cells, per_cell, T):
    """HBM bytes of the whole batched loop of the 625-cell slice from the tracked PMC passes
    (scripts/collect_cells_traffic.sh -> profiles/r0N_c5_cells625_hbm_traffic.json), when the leg is that slice."""
    if (n_cells, per_cell, T) != (625, 50_000, 60_000):
        return None, None
    for tag in ("r06", "r05", "r04"):
        q = os.path.join(ROOT, "profiles", f"{tag}_c5_cells625_hbm_traffic.json")
        if os.path.exists(q):
            try:
                j = json.load(open(q))
                v = j.get("loop_total_bytes")
                return (int(v) if v else None), os.path.relpath(q, ROOT)
            except Exception:
                return None, None
    return None, None


def cells_leg(args, n_cells, rank, world, local_rank, sync, max_over_ranks):
    """BASELINE configs[4]: per-cell EM, gate 50, init None, every cell to its own convergence.
    n_cells PER GPU (weak scaling: 5 k cells over 8 GPUs = 625 each); end to end from host buffers
    (validation, upload, layout build on the device, EM loop, read-back)."""
    import torch
    from oarfish_amd import synth
    from oarfish_amd import dist as odist
    per_cell = args.cell_reads or (2_000 if args.workload == "tiny" else 50_000)
    T = 4_000 if args.workload == "tiny" else 60_000
    total = n_cells * world
    threads = max(2, min(32, (os.cpu_count() or 8) // max(world, 1)))
    # every rank generates only its own block of cells (cell c is a pure function of (seed, c))
    c0, c1 = total * rank // world, total * (rank + 1) // world
    tg = time.perf_counter()
    cell_off, row_ptr, tid, p = synth.make_cells(c1 - c0, per_cell, T, first_cell=c0, threads=threads)
    tg = time.perf_counter() - tg
    err, tc, passes, roof, mass, runs = None, float("nan"), [], None, float("nan"), []
    try:
        # Two end-to-end calls, the faster one reported (both on the line as `seconds_runs`): outside the device loop the
        # call is host work -- range checks, 2 GB of pageable uploads, 300 MB read back into fresh pages -- and on a box
        # whose host is busy with other tenants that part alone ranged from 0.11 to 0.36 s (profiles/r06_notes.md)
        from oarfish_amd.em import cells_last_timing
        runs = []
        loop_ms, batched = 0.0, 0
        for _rep in range(2):
            sync()
            t0 = time.perf_counter()
            _c0, _c1, out, infos = odist.em_cells_sharded(cell_off, row_ptr, tid, p, None, T, 0, 1, device=local_rank)
            torch.cuda.synchronize()
            runs.append(time.perf_counter() - t0)
            if runs[-1] <= min(runs):
                # roofline of the batched EM loop: every pass of a cell streams that cell's matrix once (SURVEY.md 8d
                # per problem: nnz*8 + (R+1)*4 + 2*T*8), a cell takes part in n_passes passes, and the loop's duration
                # is HIP-event-timed on the group's stream inside the library (oem_cells_last_timing)
                loop_ms, batched = cells_last_timing()
        tc = min(runs)
        if os.environ.get("OEM_VERBOSE"):
            print(f"[bench] cells leg: em_cells_sharded {[round(r * 1e3, 1) for r in runs]} ms", file=sys.stderr)
        passes = [i.n_passes for i in infos]
        mass = float(np.abs(out.sum(axis=1) - per_cell).max())
        if loop_ms > 0:
            co = np.asarray(cell_off, dtype=np.int64)
            nnz_c = np.asarray(row_ptr, dtype=np.int64)[co[1:]] - np.asarray(row_ptr, dtype=np.int64)[co[:-1]]
            reads_c = co[1:] - co[:-1]
            nbytes = int(np.sum(np.asarray(passes, dtype=np.int64) * (nnz_c * 8 + (reads_c + 1) * 4 + 2 * T * 8)))
            ach = nbytes / (loop_ms * 1e-3) / 1e9
            ctraffic, csrc = cells_traffic(n_cells, per_cell, T)
            roof = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=ctraffic,
                        traffic_source=csrc,
                        kernel="k_em_tile + k_multi_fold_reldiff + per-cell state kernels (all passes of the batched loop)",
                        loop_ms=loop_ms, batched_passes=batched, kernel_avg_ms=loop_ms / max(batched, 1),
                        algorithmic_bytes_total=nbytes,
                        bytes_rule="sum over cells of n_passes * (nnz*8 + (R+1)*4 + 2*T*8): finished cells drop out")
    except Exception as e:  # pragma: no cover
        err = repr(e)
    if world > 1 or args.force_dist:   # every rank must reach the collectives below, failed or not
        import torch.distributed as dist
        ok = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device="cpu" if args.same_device else "cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 1.0:
            return dict(value=None, error=err or "a rank failed")
    elif err:
        return dict(value=None, error=err)
    tc = max_over_ranks(tc)
    # BASELINE configs[4] WHOLE on this one GPU (5 k cells x 50 k reads, 2 G alignments) in ONE oem_em_run_cells call:
    # the generated cells x transcript-id rotations (synth.replicate_cells -- a relabelling is an exact symmetry of the
    # EM, and 5 k cells take minutes to generate in Python), end to end from host buffers like the slice above
    full_leg = None
    n_full = args.cells_full if args.cells_full is not None else (5000 if args.workload == "c3" and n_cells >= 625 else 0)
    if world == 1 and n_full > n_cells and not err:
        try:
            tr = time.perf_counter()
            fco, frp, ftid, fp = synth.replicate_cells((cell_off, row_ptr, tid, p), T, n_full)
            tr = time.perf_counter() - tr
            t0 = time.perf_counter()
            _a, _b, fout, finfos = odist.em_cells_sharded(fco, frp, ftid, fp, None, T, 0, 1, device=local_rank)
            torch.cuda.synchronize()
            tf = time.perf_counter() - t0
            fp_ = np.asarray([i.n_passes for i in finfos])
            # a rotated copy must reproduce its base cell under the rotation (the first copy of every 50th cell)
            worst = 0.0
            for c in range(0, n_cells, 50):
                k = n_cells + c
                if k < n_full:
                    back = np.roll(fout[k], -synth.cell_shift(1, T))
                    worst = max(worst, float(np.max(np.abs(back - fout[c]) / np.maximum(np.abs(fout[c]), 1e-5 * per_cell / T))))
            full_leg = dict(value=n_full / tf, unit="cells/s", n_cells=n_full, seconds=tf, alignments=int(len(ftid)),
                            host_gb=round((fco.nbytes + frp.nbytes + ftid.nbytes + fp.nbytes) / 1e9, 1),
                            mean_passes=float(fp_.mean()), max_passes=int(fp_.max()),
                            worst_mass_error=float(np.abs(fout.sum(axis=1) - per_cell).max()),
                            worst_rel_diff_rotated_copy_vs_base=worst, replicate_s=round(tr, 2),
                            data=f"{n_cells} generated cells x {-(-n_full // n_cells)} transcript-id rotations (synth.replicate_cells)",
                            mode="one oem_em_run_cells call: groups of ~660 cells on two host threads")
            del fco, frp, ftid, fp, fout
        except Exception as e:  # pragma: no cover
            full_leg = dict(value=None, error=repr(e))
    return dict(value=total / tc, unit="cells/s", n_cells=total, reads_per_cell=per_cell, n_txps=T, cells_full=full_leg,
                seconds=tc, seconds_runs=[round(r, 4) for r in runs], mean_passes=float(np.mean(passes)), max_passes=int(np.max(passes)),
                worst_mass_error=mass, gen_s=round(tg, 2), roofline=roof,
                mode=f"{n_cells} cells per GPU, batched on the device, no collective")

if __name__ == "__main__":
    main()
