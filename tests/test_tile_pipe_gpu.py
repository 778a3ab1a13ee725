"""GPU tests of the pipelined tile walk (oem_tile_pipe.hip: persistent workgroups that request tile k + 1 while they
fold tile k).  Same semantics as the one-workgroup-per-tile kernel (em.rs:87-133).  It measured slower than that
kernel (profiles/r05_notes.md) and is not shipped: only the test-only library takes it, through OEM_TILE_PIPE=1
(whenever the store qualifies) and OEM_PIPE_SLOTS=n (n workgroups walk ALL tiles)."""
import numpy as np
import pytest

from oarfish_amd import _lib, synth
from oarfish_amd.types import DeviceStore
from oracle import c_oracle
from tests.common import assert_counts_close

pytestmark = pytest.mark.gpu


def _weights_with(n_distinct, size, rng):
    vals = ((1.0 + np.arange(n_distinct)) / (n_distinct + 3.0)).astype(np.float32)
    p = vals[rng.integers(0, n_distinct, size=size)]
    p[:n_distinct] = vals
    return p


def _store(kind, seed):
    """(row_ptr, tid, as_prob, n_txps) of the shapes the tile kernels are tested on."""
    rng = np.random.default_rng(seed)
    if kind == "dense":        # many reads per transcript: hot anchors, 8 window copies
        st = synth.make_store(150_000, 2_000, seed=seed)
    elif kind == "sparse":     # wide windows, few reads per tile window entry
        st = synth.make_store(60_000, 40_000, seed=seed)
    elif kind == "long":       # reads with 20+ local alignments: the reload loops behind the register sets
        st = synth.make_store(30_000, 3_000, kbar=24.0, seed=seed)
    elif kind == "remote":     # mostly-remote: more records than a thread's six register slots (overflow path)
        n, T = 40_000, 30_000
        k = rng.integers(6, 14, n)
        row_ptr = np.concatenate([[0], np.cumsum(k)]).astype(np.uint64)
        tid = np.empty(int(row_ptr[-1]), np.uint32)
        for i in range(n):     # distinct transcripts per read, all over the annotation
            tid[int(row_ptr[i]):int(row_ptr[i + 1])] = rng.choice(T, int(k[i]), replace=False)
        p = np.exp(-rng.integers(0, 30, len(tid)) / 5.0).astype(np.float32)
        return row_ptr, tid, p, T
    elif kind == "tiny":       # one tile, three slices
        st = synth.make_store(150, 40, seed=seed)
    elif kind == "ragged":     # a last tile with a few reads, single-alignment reads, one very long read
        st = synth.make_store(20_000 + 7, 900, seed=seed)
    else:
        raise ValueError(kind)
    return st.row_ptr, st.tid, st.as_prob, st.n_txps


CODINGS = {  # name -> (distinct weights to impose or None, oem_store_opts.weight_coding)
    "fused": (None, 0), "bytes": (200, 0), "words": (600, 0), "f32": (None, 1),
}


@pytest.mark.parametrize("kind", ["dense", "sparse", "long", "remote", "tiny", "ragged"])
@pytest.mark.parametrize("coding", list(CODINGS))
def test_pipelined_pass_matches_the_oracle_and_the_tile_kernel(kind, coding, monkeypatch):
    row_ptr, tid, p, T = _store(kind, seed=11)
    n_distinct, wc = CODINGS[coding]
    if n_distinct:
        p = _weights_with(n_distinct, len(tid), np.random.default_rng(n_distinct))
    n_reads = len(row_ptr) - 1
    o = c_oracle.Store(row_ptr, tid, p, None, T)
    theta = np.random.default_rng(3).lognormal(0, 1.5, T)
    want_m = c_oracle.m_step(o, theta)
    want, wi = c_oracle.do_em(o, max_iter=120, conv_thresh=1e-3)
    monkeypatch.setenv("OEM_TILE_PIPE", "0")
    with _lib.testing(), DeviceStore(row_ptr, tid, p, None, T, weight_coding=wc) as d:
        n_tiles = d.info(_lib.OEM_INFO_TILES)
        n_dict = d.info(_lib.OEM_INFO_WEIGHT_DICT_ENTRIES)
        base_m = d.m_step(theta)
    if coding == "f32":
        assert n_dict == 0
    elif n_distinct:
        assert n_dict == n_distinct + 1
    monkeypatch.setenv("OEM_TILE_PIPE", "1")
    for slots in (1, 3, 4096):   # one workgroup walks every tile; three; one tile per workgroup (no pipeline at all)
        monkeypatch.setenv("OEM_PIPE_SLOTS", str(slots))
        with _lib.testing(), DeviceStore(row_ptr, tid, p, None, T, weight_coding=wc) as d:
            assert d.info(_lib.OEM_INFO_TILES) == n_tiles
            m = d.m_step(theta)
            m2 = d.m_step(theta)          # a second pass over what the first left in the queue
            cnt, info = d.em_run(None, 120, 1e-3, 50)
        what = f"{kind}/{coding}, {n_tiles} tiles on {slots} slots"
        assert_counts_close(m, want_m, n_reads, T, 1e-11, "m_step vs oracle, " + what)
        assert_counts_close(m2, want_m, n_reads, T, 1e-11, "second m_step vs oracle, " + what)
        assert_counts_close(m, base_m, n_reads, T, 1e-12, "pipelined vs one workgroup per tile, " + what)
        assert info.niter == wi.niter, what
        assert_counts_close(cnt, want, n_reads, T, 1e-9, "em vs oracle, " + what)


def test_pipelined_pass_with_read_multiplicities(monkeypatch):
    """The one-per-pass bootstrap path hands the tile kernel per-read multiplicities (em.rs:273-290 in multiplicity
    form): injected resamples against the oracle, through the pipeline."""
    st = synth.make_store(90_000, 5_000, seed=5)
    T = st.n_txps
    rng = np.random.default_rng(8)
    row_w = rng.multinomial(st.n_reads, np.full(st.n_reads, 1.0 / st.n_reads), size=2).astype(np.uint32)
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, T)
    monkeypatch.setenv("OEM_TILE_PIPE", "1")
    monkeypatch.setenv("OEM_PIPE_SLOTS", "5")
    with _lib.testing(), DeviceStore(st.row_ptr, st.tid, st.as_prob, None, T) as d:
        d.set_option(_lib.OEM_OPT_BATCH_BOOTSTRAP, 0)   # one replicate per pass: k_em_tile_p with row weights
        got, infos = d.bootstrap(2, seed=1, row_w_all=row_w, max_iter=80, conv_thresh=1e-3)
    for b in range(2):
        want, wi = c_oracle.do_em(o, max_iter=80, conv_thresh=1e-3, row_w=row_w[b])
        assert infos[b].niter == wi.niter
        assert_counts_close(got[b], want, st.n_reads, T, 1e-9, f"replicate {b}")


def test_pipelined_pass_on_a_store_of_several_tiles_per_resident_slot(monkeypatch):
    """The regime the pipeline is built for: ~4 200 tiles on the MI355X's 1024 resident slots (no OEM_PIPE_SLOTS),
    every workgroup walking four or five tiles.  One pass against the oracle, a short run against it, mass."""
    st = synth.make_store(4_300_000, 90_000, seed=17, threads=16)
    T = st.n_txps
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, T)
    theta = np.random.default_rng(4).lognormal(0, 1.5, T)
    want_m = c_oracle.m_step(o, theta)
    monkeypatch.setenv("OEM_TILE_PIPE", "1")
    monkeypatch.delenv("OEM_PIPE_SLOTS", raising=False)
    with _lib.testing(), DeviceStore(st.row_ptr, st.tid, st.as_prob, None, T) as d:
        n_tiles = d.info(_lib.OEM_INFO_TILES)
        assert n_tiles >= 4096, n_tiles
        m = d.m_step(theta)
        cnt, info = d.em_run(None, 12, 0.0, 1)
    assert_counts_close(m, want_m, st.n_reads, T, 1e-11, "m_step")
    want, _ = c_oracle.do_em(o, max_iter=12, conv_thresh=0.0, min_iter_gate=1)
    assert_counts_close(cnt, want, st.n_reads, T, 1e-9, "12 iterations")
    assert abs(cnt.sum() - st.n_reads) < 1e-9 * st.n_reads
