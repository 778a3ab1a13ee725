"""GPU tests of bench.py itself: the driver's contract line, and the whole multi-GPU code path
(process group, native RCCL communicator, attach, barrier / all-reduce timing, replica-parallel
bootstraps, dealt cells) forced onto a world of one rank -- what an 8-GPU node will execute."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _last_json(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def _check_contract(d, steps, warmup):
    assert d["metric"] == "EM iterations/sec" and d["unit"] == "iterations/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup and d["value"] > 0
    assert d["scaling"] == "strong" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    b = d["bootstraps"]
    assert b["value"] > 0 and b["n"] == 3 and b["roofline"]["achieved"] > 0 and b["roofline"]["replicates_per_launch"] == 4 and b["chains"] == 2
    c = d["cells"]
    assert c["value"] > 0 and c["n_cells"] == 4 and c["worst_mass_error"] < 1e-6 * c["reads_per_cell"]
    for name in ("em", "em_par"):
        assert d["em_to_convergence"][name]["n_passes"] >= 3


@pytest.mark.timeout(900)
def test_bench_contract_line_single_process():
    cmd = [sys.executable, "bench.py", "--workload", "tiny", "--steps", "6", "--warmup", "2", "--bootstraps", "3",
           "--cells", "4", "--cpu-seconds", "1"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _last_json(p.stdout)
    _check_contract(d, 6, 2)
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert "forced" not in d["config"]["parallelism"]


@pytest.mark.timeout(900)
def test_bench_forced_dist_path_under_torchrun():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--workload", "tiny",
           "--steps", "6", "--warmup", "2", "--bootstraps", "3", "--cells", "4", "--no-cpu-baseline", "--force-dist"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=800, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    d = _last_json(p.stdout)
    _check_contract(d, 6, 2)
    assert "forced dist path" in d["config"]["parallelism"]
    assert "replica-parallel" in d["bootstraps"]["mode"] and d["cpu_baseline"] is None


@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_device_over_p2p():
    """The N > 1 bench with N = 2 REAL ranks: two processes under torch.distributed.run, both on cuda:0
    (`--same-device`: gloo process group, nnz-balanced row shards, the count vector exchanged peer to peer over
    hipIpc-mapped buffers inside the rel-diff kernel, replica-parallel bootstraps, dealt cells, max-over-ranks
    timing) -- every collective step of what an 8-GPU node executes, short of RCCL and xGMI."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--workload", "tiny",
           "--steps", "6", "--warmup", "2", "--bootstraps", "3", "--cells", "4", "--no-cpu-baseline", "--same-device"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=800, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["value"] > 0 and d["scaling"] == "strong"
    assert "row-shard x2" in d["config"]["parallelism"] and "ONE device" in d["config"]["parallelism"]
    ex = d["config"]["exchange"]
    assert ex["p2p_connected"] and ex["allreduce_us"] > 0 and ex["backend"].startswith("p2p")
    for shape in ("p2p one-shot", "p2p two-phase"):   # both shapes reproduced each other and were timed
        c = ex["candidates"][shape]
        assert c["ok"] and c["iteration_us"] > 0
        # the curve explains itself: the rank-local compute, the exchange alone and the sharded iteration side by side
        assert c["shard_compute_us"] > 0 and c["exchange_us"] > 0
        assert abs(c["iteration_minus_compute_us"] - (c["iteration_us"] - c["shard_compute_us"])) < 1e-6
    assert abs(ex["shard_compute_us"] - max(ex["shard_compute_us_per_rank"])) < 0.01 and len(ex["shard_compute_us_per_rank"]) == 2
    assert min(ex["shard_compute_us_per_rank"]) > 0
    assert ex["ranks"] == 2 and ex["rccl_ranks_seen"] == 0 and ex["p2p_connected_this_rank"]   # (--same-device: no RCCL)
    assert d["bootstraps"]["n"] == 6 and d["bootstraps"]["value"] > 0 and "replica-parallel over 2" in d["bootstraps"]["mode"]
    assert d["cells"]["n_cells"] == 8 and d["cells"]["worst_mass_error"] < 1e-6 * d["cells"]["reads_per_cell"]
    for name in ("em", "em_par"):   # the sharded loop converges like the un-sharded one (tiny store: a handful of passes)
        assert d["em_to_convergence"][name]["n_passes"] >= 3
    assert d["roofline"] and d["cpu_baseline"] is None


@pytest.mark.timeout(900)
def test_bench_two_ranks_without_the_ipc_variable_in_the_environment():
    """The driver launches bench.py with whatever environment the node has.  With HSA_ENABLE_IPC_MODE_LEGACY absent
    bench.py sets it itself (before the HIP runtime loads), so the first multi-GPU run takes the peer-to-peer
    exchange it was built for instead of silently timing a fall-back; and whatever happens the line appears, with the
    per-candidate outcome in config.exchange."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--workload", "tiny",
           "--steps", "4", "--warmup", "1", "--bootstraps", "0", "--cells", "0", "--no-cpu-baseline", "--same-device"]
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=800, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0
    ex = d["config"]["exchange"]
    assert ex["p2p_connected"], ex                       # the variable was defaulted: hipIpc handles worked
    assert set(ex["candidate_ok"]) == set(ex["candidates"]) and all(ex["candidate_ok"].values())
    assert ex["backend"].startswith("p2p") and "p2p_not_used_because" not in ex


@pytest.mark.timeout(1200)
def test_run_scaling_script_on_one_device():
    """scripts/run_scaling.sh, the N = 1, 2, 4, 8 curve of one node, exercised with all ranks on cuda:0 (SAME_DEVICE=1:
    a self test of the launch lines, the JSON lines and the table, not scaling)."""
    out = os.path.join(ROOT, "gpurun_out", "scaling_selftest")
    env = dict(os.environ, SAME_DEVICE="1")
    p = subprocess.run(["bash", "scripts/run_scaling.sh", out, "--workload", "tiny", "--bootstraps", "0", "--cells", "0",
                        "--no-cpu-baseline", "--no-f32-compare"], cwd=ROOT, capture_output=True, text=True, timeout=1100, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    for n in (1, 2, 4, 8):
        assert f"N={n} rc=0" in p.stdout, p.stdout[-2000:]
        d = _last_json(open(os.path.join(out, f"scale_{n}.json")).read())
        assert d["n_gpus"] == n and d["value"] > 0
        if n > 1:
            assert d["config"]["exchange"]["shard_compute_us"] > 0
    assert "shard compute us" in p.stdout and "candidates" in p.stdout
