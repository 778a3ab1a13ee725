"""CPU test of bench.py's shard builder: the row shards every rank generates for itself must tile
the seeded store exactly (the multi-GPU bench depends on it)."""
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


def test_rank_shards_tile_the_store(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(synth, "CHUNK", 1 << 12)          # many chunks at a small size
    cfg = dict(n_reads=50_000, n_txps=3_000, kbar=8.0)
    full = synth.make_store(cfg["n_reads"], cfg["n_txps"], cfg["kbar"], threads=2)
    for world in (2, 3, 8):
        rows, nnz = 0, 0
        for rank in range(world):
            rp, tid, p, r0, r1 = bench.make_shard(cfg, rank, world)
            assert r0 == rows and rp[0] == 0 and len(rp) - 1 == r1 - r0
            a0, a1 = int(full.row_ptr[r0]), int(full.row_ptr[r1])
            assert np.array_equal(rp, full.row_ptr[r0:r1 + 1] - full.row_ptr[r0])
            assert np.array_equal(tid, full.tid[a0:a1]) and np.array_equal(p, full.as_prob[a0:a1])
            rows, nnz = r1, nnz + len(tid)
        assert rows == cfg["n_reads"] and nnz == full.nnz
