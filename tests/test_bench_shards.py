"""CPU tests of bench.py's host logic: the row shards of the ranks must tile the seeded store exactly,
in blocks balanced by alignment count (the multi-GPU bench depends on it), and the command line keeps
the driver's contract."""
import importlib.util
import os

import numpy as np

from oarfish_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_rank_shards_tile_the_store_balanced_by_alignments(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(synth, "CHUNK", 1 << 12)          # many chunks at a small size
    cfg = dict(n_reads=50_000, n_txps=3_000, kbar=8.0)
    full = synth.make_store(cfg["n_reads"], cfg["n_txps"], cfg["kbar"], threads=2)
    for world in (1, 2, 3, 8):
        rows, nnz, sizes = 0, 0, []
        for rank in range(world):
            rp, tid, p, r0, r1 = bench.make_shard(cfg, rank, world)
            assert r0 == rows and rp[0] == 0 and len(rp) - 1 == r1 - r0
            a0, a1 = int(full.row_ptr[r0]), int(full.row_ptr[r1])
            assert np.array_equal(rp, full.row_ptr[r0:r1 + 1] - full.row_ptr[r0])
            assert np.array_equal(tid, full.tid[a0:a1]) and np.array_equal(p, full.as_prob[a0:a1])
            rows, nnz = r1, nnz + len(tid)
            sizes.append(len(tid))
        assert rows == cfg["n_reads"] and nnz == full.nnz
        # balanced by alignments: every block within one read's alignments of the mean
        assert max(sizes) - min(sizes) <= 2 * 100, sizes


def test_cells_are_a_pure_function_of_seed_and_cell_index():
    a = synth.make_cells(5, 300, 200, seed=3)
    b0 = synth.make_cells(2, 300, 200, seed=3, first_cell=0, threads=2)
    b1 = synth.make_cells(3, 300, 200, seed=3, first_cell=2, threads=2)
    n0 = int(b0[1][-1])
    assert np.array_equal(a[2], np.concatenate([b0[2], b1[2]])) and np.array_equal(a[3], np.concatenate([b0[3], b1[3]]))
    assert np.array_equal(a[1], np.concatenate([b0[1], b1[1][1:] + np.uint64(n0)]))


def test_command_line_contract():
    bench = _bench()
    a = bench.parse([])
    assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0 and a.bootstraps >= 16 and not a.force_dist
    a = bench.parse(["--gpus", "8", "--steps", "20", "--warmup", "3", "--force-dist"])
    assert (a.gpus, a.steps, a.warmup, a.force_dist) == (8, 20, 3, True)
