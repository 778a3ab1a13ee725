"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the
committed golden fixtures.  Tolerance: BASELINE.json north_star -- abundances
within 1e-4 relative of the reference algorithm (f64 arithmetic; the only
difference allowed is floating-point summation order, so the observed error is
~1e-12 and most checks are far tighter than 1e-4)."""
import os

import numpy as np
import pytest

import oarfish_amd
from oarfish_amd import _lib, synth
from oarfish_amd.types import DeviceStore, InMemoryAlignmentStore
from oracle import c_oracle
from tests.common import assert_counts_close, golden_names, load_golden

pytestmark = pytest.mark.gpu

RTOL = 1e-4  # north_star tolerance


def _dev(g, **kw):
    return DeviceStore(g["row_ptr"], g["tid"], g["as_prob"], g["cov_prob"], g["n_txps"], **kw)


def _orc(g):
    return c_oracle.Store(g["row_ptr"], g["tid"], g["as_prob"], g["cov_prob"], g["n_txps"])


def test_extension_is_loaded_and_device_present():
    from oarfish_amd import _lib
    assert _lib.device_count() >= 1
    assert _lib.lib().oem_abi_version() == 2


@pytest.mark.parametrize("name", golden_names())
def test_golden_fixtures(name):
    g = load_golden(name)
    R, T = len(g["row_ptr"]) - 1, g["n_txps"]
    with _dev(g) as d:
        for run in g["runs"]:
            cnt, info = d.em_run(g["init"], run["max_iter"], run["conv_thresh"], run["gate"])
            what = f"{name} {run['max_iter']}/{run['conv_thresh']}/{run['gate']}"
            assert abs(info.niter - run["niter"]) <= 1, what   # knife-edge rel<thresh may flip
            if info.niter == run["niter"]:
                assert info.n_passes == run["n_passes"] and info.converged == run["converged"], what
                assert_counts_close(cnt, run["counts"], R, T, 1e-9, what)
                assert abs(info.rel_diff - run["rel_diff"]) <= 1e-6 * max(1.0, abs(run["rel_diff"])), what
            else:
                assert_counts_close(cnt, run["counts"], R, T, RTOL, what)
        if "closed_form" in g:
            cnt, _ = d.em_run(g["init"], g["runs"][-1]["max_iter"], g["runs"][-1]["conv_thresh"], 50)
            np.testing.assert_allclose(cnt, g["closed_form"], rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("coverage", [False, True])
def test_m_step_matches_oracle(coverage):
    st = synth.make_store(50_000, 3_000, seed=77, coverage=coverage)
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps)
    rng = np.random.default_rng(1)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps) as d:
        for trial in range(3):
            theta = rng.lognormal(0, 2, size=st.n_txps)
            theta[rng.random(st.n_txps) < 0.1] = 0.0
            want = c_oracle.m_step(o, theta)
            got = d.m_step(theta)
            assert_counts_close(got, want, st.n_reads, st.n_txps, 1e-10, f"m_step trial {trial}")
        w = rng.poisson(1.0, size=st.n_reads).astype(np.uint32)
        theta = np.full(st.n_txps, st.n_reads / st.n_txps)
        assert_counts_close(d.m_step(theta, w), c_oracle.m_step(o, theta, row_w=w), st.n_reads,
                            st.n_txps, 1e-10, "weighted m_step")


@pytest.mark.parametrize("gate,thresh,max_iter", [(50, 1e-3, 1000), (1, 1e-3, 1000), (50, 0.0, 100)])
def test_em_matches_oracle_medium(gate, thresh, max_iter):
    st = synth.make_store(200_000, 12_000, seed=78)
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, st.n_txps)
    want, wi = c_oracle.do_em(o, max_iter=max_iter, conv_thresh=thresh, min_iter_gate=gate)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
        got, gi = d.em_run(None, max_iter, thresh, gate)
    assert abs(gi.niter - wi.niter) <= 1
    assert_counts_close(got, want, st.n_reads, st.n_txps, RTOL if gi.niter != wi.niter else 1e-8, "em")
    assert abs(got.sum() - st.n_reads) < 1e-6 * st.n_reads


def test_reference_interface_em_em_par_bootstrap():
    """The host mirror: EMInfo + em / em_par / bootstrap as bulk.rs:131-194 drives them."""
    st = synth.make_store(60_000, 4_000, seed=79, coverage=True)
    store = InMemoryAlignmentStore.from_arrays(st.row_ptr, st.tid, st.as_prob, st.cov_prob, model_coverage=True)
    txps = [oarfish_amd.TranscriptInfo.with_len(1000)] * st.n_txps
    emi = oarfish_amd.EMInfo(eq_map=store, txp_info=txps, max_iter=1000, convergence_thresh=1e-3)
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps)
    c_ser = oarfish_amd.em(emi, 3)
    w_ser, i_ser = c_oracle.do_em(o, max_iter=1000, conv_thresh=1e-3, min_iter_gate=50)
    assert abs(emi.last_run_info.niter - i_ser.niter) <= 1
    assert_counts_close(c_ser, w_ser, st.n_reads, st.n_txps, RTOL, "em")
    c_par = oarfish_amd.em_par(emi, 8)
    w_par, i_par = c_oracle.do_em(o, max_iter=1000, conv_thresh=1e-3, min_iter_gate=1)
    assert abs(emi.last_run_info.niter - i_par.niter) <= 1
    assert_counts_close(c_par, w_par, st.n_reads, st.n_txps, RTOL, "em_par")
    # model_coverage off => cov_prob column ignored (em.rs:108)
    store2 = InMemoryAlignmentStore.from_arrays(st.row_ptr, st.tid, st.as_prob, st.cov_prob, model_coverage=False)
    emi2 = oarfish_amd.EMInfo(eq_map=store2, txp_info=txps, max_iter=60, convergence_thresh=0.0)
    o2 = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, st.n_txps)
    assert_counts_close(oarfish_amd.em(emi2, 1), c_oracle.do_em(o2, max_iter=60, conv_thresh=0.0)[0],
                        st.n_reads, st.n_txps, 1e-8, "no coverage")
    # bootstrap with injected resamples
    rng = np.random.default_rng(3)
    W = np.stack([np.bincount(rng.integers(0, st.n_reads, st.n_reads), minlength=st.n_reads) for _ in range(3)]).astype(np.uint32)
    emi.max_iter = 200
    got = oarfish_amd.bootstrap(emi, 3, 1, row_weights=W)
    want, infos = c_oracle.bootstrap(o, 3, row_w_all=W, max_iter=200, conv_thresh=1e-3)
    for b in range(3):
        assert_counts_close(got[b], want[b], st.n_reads, st.n_txps, RTOL, f"bootstrap {b}")


def test_bootstrap_inject_golden():
    g = load_golden("bootstrap_inject")
    with _dev(g) as d:
        out, infos = d.bootstrap(g["row_w"].shape[0], row_w_all=g["row_w"], max_iter=int(g["boot_params"][0]),
                                 conv_thresh=float(g["boot_params"][1]))
    for b in range(out.shape[0]):
        assert_counts_close(out[b], g["boot_counts"][b], 600, 40, 1e-6, f"replicate {b}")


def test_batched_bootstrap_matches_oracle_per_replicate():
    """Replicates share passes over the matrix in rolling batches of 4 slots, two such chains side by side
    on their own streams drawing replicates from one counter (a finished slot is handed the next replicate,
    the tail runs with idle slots); every replicate must still stop at its own iteration and match its own
    serial EM."""
    st = synth.make_store(40_000, 2_500, seed=81)
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, st.n_txps)
    rng = np.random.default_rng(8)
    n_boot = 21  # 2 chains x 4 slots, refilled twice; the tail leaves slots idle
    W = np.stack([np.bincount(rng.integers(0, st.n_reads, st.n_reads), minlength=st.n_reads)
                  for _ in range(n_boot)]).astype(np.uint32)
    W[5] = 1      # the un-resampled store: must equal the point estimate
    from oarfish_amd import _lib
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
        d.set_option(_lib.OEM_OPT_BATCH_BOOTSTRAP, 1)
        out, infos = d.bootstrap(n_boot, row_w_all=W, max_iter=300, conv_thresh=1e-3)
        point, pinfo = d.em_run(None, 300, 1e-3, 50)
        # a multiplicity that does not fit a byte forces the one-replicate-per-pass path
        W2 = W[:4].copy()
        W2[2, :] = 0
        W2[2, :5] = 300
        out2, infos2 = d.bootstrap(4, row_w_all=W2, max_iter=120, conv_thresh=1e-3)
        # single iteration budget: every replicate leaves through max_iter
        out3, infos3 = d.bootstrap(4, row_w_all=W[:4], max_iter=1, conv_thresh=1e-3)
    niters = set()
    for b in range(n_boot):
        want, wi = c_oracle.do_em(o, row_w=W[b], max_iter=300, conv_thresh=1e-3)
        assert abs(infos[b].niter - wi.niter) <= 1, (b, infos[b], wi)
        assert infos[b].converged == wi.converged
        if infos[b].niter == wi.niter:
            assert infos[b].n_passes == wi.n_passes
        assert_counts_close(out[b], want, st.n_reads, st.n_txps, RTOL if infos[b].niter != wi.niter else 1e-8,
                            f"replicate {b}")
        niters.add(wi.niter)
    assert len(niters) > 1, "replicates should stop at different iterations in this test"
    assert abs(infos[5].niter - pinfo.niter) <= 1
    assert_counts_close(out[5], point, st.n_reads, st.n_txps, RTOL, "identity resample")
    for b in range(4):
        want, wi = c_oracle.do_em(o, row_w=W2[b], max_iter=120, conv_thresh=1e-3)
        assert_counts_close(out2[b], want, st.n_reads, st.n_txps, RTOL, f"fallback replicate {b}")
        want, wi = c_oracle.do_em(o, row_w=W[b], max_iter=1, conv_thresh=1e-3)
        assert infos3[b].niter == 1 and infos3[b].n_passes == 2
        assert_counts_close(out3[b], want, st.n_reads, st.n_txps, 1e-9, f"max_iter=1 replicate {b}")


def test_device_multinomial_weights():
    """bootstrap.rs:7-16: Multinomial(n; 1/n): sum n, mean 1, var 1-1/n, P(0)=e^-1,
    replicas independent, stream reproducible."""
    st = synth.make_store(200_000, 2_000, seed=80)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
        w0 = d.bootstrap_weights(11, 0)
        w0b = d.bootstrap_weights(11, 0)
        w1 = d.bootstrap_weights(11, 1)
        w2 = d.bootstrap_weights(12, 0)
    n = st.n_reads
    assert np.array_equal(w0, w0b) and not np.array_equal(w0, w1) and not np.array_equal(w0, w2)
    for w in (w0, w1, w2):
        assert int(w.sum()) == n
        assert abs(w.var() - (1 - 1 / n)) < 0.02
        assert abs((w == 0).mean() - np.exp(-1)) < 0.01
        assert abs((w == 1).mean() - np.exp(-1)) < 0.01
        assert abs((w == 2).mean() - np.exp(-1) / 2) < 0.01
    assert abs(np.corrcoef(w0, w1)[0, 1]) < 0.02
    # a device-drawn bootstrap equals the oracle run on the same drawn weights
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, st.n_txps)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
        out, infos = d.bootstrap(2, seed=11, max_iter=120, conv_thresh=1e-3)
    for b, w in enumerate((w0, w1)):
        want, wi = c_oracle.do_em(o, row_w=w, max_iter=120, conv_thresh=1e-3)
        assert abs(infos[b].niter - wi.niter) <= 1
        assert_counts_close(out[b], want, n, st.n_txps, RTOL, f"device bootstrap {b}")


def test_cells_match_per_cell_oracle():
    """single_cell.rs:139-160: each cell is an independent em::em with init None."""
    n_cells, T = 6, 500
    cell_off, row_ptr, tid, p = synth.make_cells(n_cells, 3_000, T, seed=9)
    out, infos = oarfish_amd.em_cells(cell_off, row_ptr, tid, p, None, T, max_iter=300, convergence_thresh=1e-3)
    for c in range(n_cells):
        r0, r1 = int(cell_off[c]), int(cell_off[c + 1])
        a0, a1 = int(row_ptr[r0]), int(row_ptr[r1])
        o = c_oracle.Store(row_ptr[r0:r1 + 1] - row_ptr[r0], tid[a0:a1], p[a0:a1], None, T)
        want, wi = c_oracle.do_em(o, max_iter=300, conv_thresh=1e-3, min_iter_gate=50)
        assert abs(infos[c].niter - wi.niter) <= 1
        assert_counts_close(out[c], want, r1 - r0, T, RTOL, f"cell {c}")


@pytest.mark.parametrize("serial", [False, True])
def test_cells_with_nan_coverage_drop_the_read_on_every_path(serial, monkeypatch):
    """single_cell.rs:132-137 with model_coverage: a zero-span alignment leaves a NaN in the read's
    coverage column (normalize_probability.rs:58), the reference's `denom > 1e-30` test then fails and the
    read contributes nothing (em.rs:115).  The batched per-cell path builds its stores without going through
    oem_store_create, the cell-by-cell fallback goes through it: both must drop exactly those reads."""
    from oarfish_amd import _lib
    n_cells, T = 5, 400
    cell_off, row_ptr, tid, p = synth.make_cells(n_cells, 2_500, T, seed=21)
    rng = np.random.default_rng(5)
    cov = rng.uniform(0.05, 1.0, len(tid))
    n_reads = len(row_ptr) - 1
    bad = rng.choice(n_reads, 40, replace=False)
    for r in bad:   # one NaN somewhere in the read, as normalize_read_probs leaves it
        a0, a1 = int(row_ptr[r]), int(row_ptr[r + 1])
        cov[a0 + int(rng.integers(0, a1 - a0))] = np.nan
    if serial:   # the knob that forces the cell-by-cell fallback exists only in the test-only library
        monkeypatch.setenv("OEM_SERIAL_CELLS", "1")
        with _lib.testing():
            out, infos = oarfish_amd.em_cells(cell_off, row_ptr, tid, p, cov, T, max_iter=300, convergence_thresh=1e-3)
    else:
        out, infos = oarfish_amd.em_cells(cell_off, row_ptr, tid, p, cov, T, max_iter=300, convergence_thresh=1e-3)
    assert np.all(np.isfinite(out))
    for c in range(n_cells):
        r0, r1 = int(cell_off[c]), int(cell_off[c + 1])
        a0, a1 = int(row_ptr[r0]), int(row_ptr[r1])
        o = c_oracle.Store(row_ptr[r0:r1 + 1] - row_ptr[r0], tid[a0:a1], p[a0:a1], cov[a0:a1], T)
        want, wi = c_oracle.do_em(o, max_iter=300, conv_thresh=1e-3, min_iter_gate=50)
        dropped = int(((bad >= r0) & (bad < r1)).sum())
        assert np.all(np.isfinite(want)) and abs(want.sum() - (r1 - r0 - dropped)) < 1e-9 * (r1 - r0)
        assert abs(out[c].sum() - (r1 - r0 - dropped)) < 1e-9 * (r1 - r0)
        assert abs(infos[c].niter - wi.niter) <= 1
        assert_counts_close(out[c], want, r1 - r0, T, RTOL if infos[c].niter != wi.niter else 1e-8, f"cell {c}")


def _layout_hash(row_ptr, tid, p, cov, T, problem_size=0, win_cap=0, host_build=False):
    """Hashes of the resident tiled arrays (hook of the test-only library)."""
    import ctypes as C
    from oarfish_amd import _lib
    import os
    prev = os.environ.get("OEM_KEEP_UNPACKED")
    os.environ["OEM_KEEP_UNPACKED"] = "1"   # keep the builders' remote streams next to their slim form: both are hashed
    try:
        with _lib.testing():
            return _layout_hash_impl(row_ptr, tid, p, cov, T, problem_size, win_cap, host_build)
    finally:
        if prev is None:
            os.environ.pop("OEM_KEEP_UNPACKED", None)
        else:
            os.environ["OEM_KEEP_UNPACKED"] = prev


def _layout_hash_impl(row_ptr, tid, p, cov, T, problem_size, win_cap, host_build):
    import ctypes as C
    from oarfish_amd import _lib
    opts = _lib.StoreOptsC()
    opts.problem_size = problem_size
    opts.reorder_rows = 2 if problem_size else 0
    opts.window_cap = int(win_cap)
    opts.layout_build = 1 if host_build else 0
    h = C.c_void_p()
    row_ptr = np.ascontiguousarray(row_ptr, np.uint64); tid = np.ascontiguousarray(tid, np.uint32)
    p = np.ascontiguousarray(p, np.float32)
    cov = None if cov is None else np.ascontiguousarray(cov, np.float64)
    _lib.check(_lib.lib().oem_store_create(row_ptr.ctypes.data, tid.ctypes.data, p.ctypes.data,
                                           None if cov is None else cov.ctypes.data, len(row_ptr) - 1, len(tid), T, 0,
                                           C.byref(opts), C.byref(h)))
    out = (C.c_uint64 * 18)()
    fn = _lib.lib().oem_debug_layout_hash
    try:
        _lib.check(fn(h, C.addressof(out), 18))
    finally:
        _lib.lib().oem_store_destroy(h)
    return list(out)


LAYOUT_FIELDS = ["n_tiles", "n_rows", "n_local", "n_remote", "tiles", "perm", "codes", "w", "r_tid", "r_w", "r_row",
                 "r_slot", "q_dst", "bucket_base", "built_on_device", "slot_table", "packed_records", "packed"]


@pytest.mark.parametrize("win_cap", [512, 2048])
@pytest.mark.parametrize("case", ["medium", "coverage", "sparse_wide", "tiny", "empty_rows_dups", "cells", "c2"])
def test_device_built_layout_equals_host_built_layout(case, win_cap):
    """oem_layout_device.hip against its specification (oem_layout.cpp): every array of the tiled layout,
    element for element (64-bit hashes of the resident arrays), over dense / sparse / ragged stores, the
    f64 coverage weights and the per-cell problem boundaries."""
    ps, cov = 0, None
    if case == "medium":
        st = synth.make_store(300_000, 20_000, seed=501); rp, tid, p, T = st.row_ptr, st.tid, st.as_prob, st.n_txps
    elif case == "coverage":
        st = synth.make_store(120_000, 9_000, seed=502, coverage=True)
        rp, tid, p, cov, T = st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps
    elif case == "sparse_wide":   # few reads per transcript: window-limited tiles, many buckets
        st = synth.make_store(60_000, 150_000, seed=503); rp, tid, p, T = st.row_ptr, st.tid, st.as_prob, st.n_txps
    elif case == "tiny":
        st = synth.make_store(70, 40, seed=504, threads=1); rp, tid, p, T = st.row_ptr, st.tid, st.as_prob, st.n_txps
    elif case == "empty_rows_dups":
        rng = np.random.default_rng(505)
        T, lens = 5_000, rng.integers(0, 12, size=40_000)
        lens[rng.random(len(lens)) < 0.1] = 0
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        base = np.repeat(rng.integers(0, T, size=len(lens)), lens)
        tid = ((base + rng.integers(0, 6, size=int(rp[-1]))) % T).astype(np.uint32)      # repeats inside reads
        p = np.exp(-rng.integers(0, 30, size=len(tid)) / 5.0).astype(np.float32)
    elif case == "cells":
        T = 3_000
        cell_off, rp, tid, p = synth.make_cells(9, 2_500, T, seed=506)
        tid = (tid.astype(np.uint64) + np.repeat(np.repeat(np.arange(9), np.diff(cell_off).astype(np.int64)),
                                                  np.diff(rp).astype(np.int64)[:]) * T).astype(np.uint32)
        ps, T = T, 9 * T
    else:
        st = synth.make_store(1_000_000, 60_000, seed=synth.BASE_SEED); rp, tid, p, T = st.row_ptr, st.tid, st.as_prob, st.n_txps
    want = _layout_hash(rp, tid, p, cov, T, ps, win_cap, host_build=True)
    got = _layout_hash(rp, tid, p, cov, T, ps, win_cap)
    assert want[14] == 0 and got[14] == 1, "the two builders were not the ones asked for"
    diff = [f for f, a, b in zip(LAYOUT_FIELDS, got, want) if a != b and f != "built_on_device"]
    assert not diff, f"{case}: device-built layout differs from the host-built one in {diff} ({got[:4]} vs {want[:4]})"
    assert want[0] > 0


@pytest.mark.parametrize("seed", range(12))
def test_device_built_layout_equals_host_built_layout_random_shapes(seed):
    """Same equality over randomly shaped stores: read lengths up to 120, 1..300 k transcripts, repeats,
    empty reads, optional coverage weights, optional per-problem boundaries."""
    rng = np.random.default_rng(7000 + seed)
    R = int(rng.choice([1, 9, 700, 20_000, 150_000]))
    T = int(rng.choice([1, 5, 64, 3_000, 40_000, 300_000]))
    maxk = int(rng.choice([1, 4, 15, 120]))
    lens = rng.integers(0, maxk + 1, size=R)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    nnz = int(rp[-1])
    ps = 0
    if seed % 3 == 2 and T >= 64:                       # per-problem layout: alignments stay inside their problem
        ps = max(T // int(rng.integers(2, 9)), 1)
        T = (T // ps) * ps
        prob = np.repeat(np.sort(rng.integers(0, T // ps, size=R)), lens)
        tid = (prob * ps + (np.repeat(rng.integers(0, ps, size=R), lens) + rng.integers(0, min(ps, 9), size=nnz)) % ps).astype(np.uint32)
    else:
        spread = int(rng.choice([1, 8, 200, max(T, 1)]))
        tid = ((np.repeat(rng.integers(0, T, size=R), lens) + rng.integers(0, spread, size=nnz)) % T).astype(np.uint32)
    p = np.exp(-rng.integers(0, 40, size=nnz) / 5.0).astype(np.float32)
    cov = rng.uniform(1e-3, 1.0, size=nnz) if seed % 4 == 0 else None
    win_cap = 0                                          # chosen from the density
    if seed % 2:
        win_cap = 2048 if seed % 4 == 1 else 512
    want = _layout_hash(rp, tid, p, cov, T, ps, win_cap, host_build=True)
    got = _layout_hash(rp, tid, p, cov, T, ps, win_cap)
    what = f"seed {seed}: R={R} T={T} maxk={maxk} ps={ps} cov={cov is not None}"
    assert want[14] == 0 and got[14] == 1, what
    diff = [f for f, a, b in zip(LAYOUT_FIELDS, got, want) if a != b and f != "built_on_device"]
    assert not diff, f"{what}: differs in {diff} ({got[:4]} vs {want[:4]})"


def test_cells_sharded_over_ranks_equal_one_run():
    """Per-cell EM over N GPUs = blocks of cells, no collective: the blocks of 3 ranks (run here one
    after another on the one GPU) concatenate to the single run."""
    from oarfish_amd import dist as odist
    T = 700
    cell_off, row_ptr, tid, p = synth.make_cells(11, 900, T, seed=21)
    full, finfo = oarfish_amd.em_cells(cell_off, row_ptr, tid, p, None, T, max_iter=300)
    parts, seen = [], []
    for rank in range(3):
        c0, c1, out, infos = odist.em_cells_sharded(cell_off, row_ptr, tid, p, None, T, rank, 3, max_iter=300)
        parts.append(out); seen += list(range(c0, c1))
        assert [i.niter for i in infos] == [i.niter for i in finfo[c0:c1]]
    assert seen == list(range(11))
    np.testing.assert_allclose(np.concatenate(parts, axis=0), full, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("group_nnz", [None, 30_000])
def test_cells_ragged_batch(group_nnz, monkeypatch):
    """Cells of very different sizes (incl. an empty one and a one-read one), with the coverage column:
    every cell stops at its own iteration.  With a small group bound the experiment is cut into several
    batched groups (and single-cell groups take the cell-by-cell path): results must not depend on it."""
    import contextlib
    from oarfish_amd import _lib
    ctx = contextlib.nullcontext()
    if group_nnz:   # the knob only exists in the test-only library; the product ignores the environment
        monkeypatch.setenv("OEM_CELLS_GROUP_NNZ", str(group_nnz))
        ctx = _lib.testing()
    T = 900
    rng = np.random.default_rng(12)
    sizes = [4000, 0, 1, 700, 2500, 60]
    rps, tids, ps, covs = [np.zeros(1, np.uint64)], [], [], []
    cell_off = np.zeros(len(sizes) + 1, dtype=np.uint64)
    base = 0
    for c, n in enumerate(sizes):
        if n:
            st = synth.make_store(n, T, seed=100 + c, coverage=True, threads=1)
            rps.append(st.row_ptr[1:] + np.uint64(base)); tids.append(st.tid); ps.append(st.as_prob); covs.append(st.cov_prob)
            base += st.nnz
        cell_off[c + 1] = cell_off[c] + np.uint64(n)
    row_ptr, tid, p, cov = np.concatenate(rps), np.concatenate(tids), np.concatenate(ps), np.concatenate(covs)
    with ctx:
        out, infos = oarfish_amd.em_cells(cell_off, row_ptr, tid, p, cov, T, max_iter=400, convergence_thresh=1e-3)
    niters = []
    for c, n in enumerate(sizes):
        r0, r1 = int(cell_off[c]), int(cell_off[c + 1])
        a0, a1 = int(row_ptr[r0]), int(row_ptr[r1])
        o = c_oracle.Store(row_ptr[r0:r1 + 1] - row_ptr[r0], tid[a0:a1], p[a0:a1], cov[a0:a1], T)
        want, wi = c_oracle.do_em(o, max_iter=400, conv_thresh=1e-3, min_iter_gate=50)
        assert abs(infos[c].niter - wi.niter) <= 1, (c, infos[c], wi)
        assert_counts_close(out[c], want, max(n, 1), T, RTOL if infos[c].niter != wi.niter else 1e-8, f"cell {c}")
        niters.append(wi.niter)
    assert len(set(niters)) > 2
    assert np.all(out[1] == 0.0)


@pytest.mark.parametrize("coverage", [False, True])
def test_aux_counts_and_assignment_probs(coverage):
    """The two steps right after the EM on the same resident store:
    aux_counts.rs:23-50 (integers: bit-exact) and the E-step of write_out_prob
    (write_function.rs:283-318; f64, same operation order: 1e-12)."""
    st = synth.make_store(40_000, 3_000, seed=91, coverage=coverage)
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps) as d:
        u, t = d.aux_counts()
        wu, wt = c_oracle.aux_counts(o)
        assert np.array_equal(u, wu) and np.array_equal(t, wt)
        counts, _ = d.em_run(None, 120, 1e-3, 50)
        for thresh in (0.0, 1e-3, 0.2):
            got = d.assignment_probs(counts, thresh)
            want = c_oracle.assignment_probs(o, counts, thresh)
            assert np.array_equal(got < 0, want < 0), f"kept sets differ at thresh {thresh}"
            k = want >= 0
            np.testing.assert_allclose(got[k], want[k], rtol=1e-12, atol=0)
            s = np.add.reduceat(np.where(k, got, 0.0), st.row_ptr[:-1].astype(np.int64))
            kept_rows = np.add.reduceat(k.astype(np.int64), st.row_ptr[:-1].astype(np.int64)) > 0
            np.testing.assert_allclose(s[kept_rows], 1.0, rtol=1e-12)   # renormalised per read
    # a read none of whose transcripts has mass: denom = 0 -> nothing printed (NaN >= thresh is false)
    rp = np.array([0, 2, 3], dtype=np.uint64)
    with DeviceStore(rp, np.array([0, 1, 2], np.uint32), np.array([1.0, 0.5, 1.0], np.float32), None, 3) as d:
        got = d.assignment_probs(np.array([0.0, 0.0, 4.0]), 1e-3)
    assert np.all(got[:2] == -1.0) and got[2] == 1.0


@pytest.mark.parametrize("threads", [3, 8])
def test_bulk_tail_writes_reference_files(tmp_path, threads):
    """bulk.rs:131-207 from the built store onwards: em/em_par by thread count, aux counts, the
    `.quant` / `.ambig_info.tsv` / `.infreps.pq` / `.prob` files; every number against the oracle."""
    import pyarrow.parquet as pq
    from oarfish_amd.bulk import BulkArgs, perform_inference_and_write_output
    st = synth.make_store(30_000, 2_000, seed=77)
    store = InMemoryAlignmentStore.from_arrays(st.row_ptr, st.tid, st.as_prob)
    names = [f"ENST{i:08d}.1" for i in range(st.n_txps)]
    lens = (500 + np.arange(st.n_txps) % 3000).tolist()
    rnames = [f"read/{i}" for i in range(st.n_reads)]
    out = str(tmp_path / "res" / "sample")
    args = BulkArgs(output=out, threads=threads, num_bootstraps=2, write_assignment_probs=True,
                    display_thresh=1e-4, seed=5)
    counts = perform_inference_and_write_output(store, names, lens, args, read_names=rnames)
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, st.n_txps)
    want, wi = (c_oracle.em_par if threads > 4 else c_oracle.do_em)(o, max_iter=1000, conv_thresh=1e-3)
    assert_counts_close(counts, want, st.n_reads, st.n_txps)
    # .quant: header + one line per transcript, counts printed Rust-style and parsing back exactly
    q = open(out + ".quant").read().split("\n")
    assert q[0] == "tname\tlen\tnum_reads" and len(q) == st.n_txps + 2
    qc = np.array([float(l.split("\t")[2]) for l in q[1:-1]])
    assert np.array_equal(qc, counts) and q[1].split("\t")[:2] == [names[0], str(lens[0])]
    wu, wt = c_oracle.aux_counts(o)
    amb = np.loadtxt(out + ".ambig_info.tsv", skiprows=1, dtype=np.int64)
    assert np.array_equal(amb[:, 0], wu) and np.array_equal(amb[:, 2], wt) and np.array_equal(amb[:, 1], wt - wu)
    breps = pq.read_table(out + ".infreps.pq").to_pandas().to_numpy().T
    assert breps.shape == (2, st.n_txps)
    np.testing.assert_allclose(breps.sum(axis=1), st.n_reads, rtol=1e-9)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
        for b in range(2):                                     # same device resample -> oracle EM
            w = d.bootstrap_weights(5, b)
            wb, _ = c_oracle.do_em(o, max_iter=1000, conv_thresh=1e-3, row_w=w)
            assert_counts_close(breps[b], wb, st.n_reads, st.n_txps)
    wp = c_oracle.assignment_probs(o, counts, 1e-4)
    lines = open(out + ".prob").read().split("\n")
    assert lines[0] == f"{st.n_txps}\t{st.n_reads}" and lines[1:st.n_txps + 1] == names
    for r in (0, 1, 17, st.n_reads - 1):
        f = lines[st.n_txps + 1 + r].split("\t")
        b, e = int(st.row_ptr[r]), int(st.row_ptr[r + 1])
        keep = wp[b:e] >= 0
        n = int(f[1])
        assert f[0] == rnames[r] and n == keep.sum()
        assert [int(x) for x in f[2:2 + n]] == st.tid[b:e][keep].tolist()
        np.testing.assert_allclose([float(x) for x in f[2 + n:]], wp[b:e][keep], atol=0.5001e-4)


def test_ragged_store_with_repeated_transcripts():
    """Reads that hit the same transcript twice, empty reads, zero weights, a 100-alignment read."""
    rng = np.random.default_rng(77)
    T, R = 5000, 3000
    lens = rng.integers(0, 9, size=R)
    lens[5] = 100
    rp = np.zeros(R + 1, dtype=np.uint64)
    np.cumsum(lens, out=rp[1:])
    nnz = int(rp[-1])
    centre = np.repeat(rng.integers(0, T, size=R), lens)
    tid = np.where(rng.random(nnz) < 0.8, (centre + rng.integers(-3, 4, size=nnz)) % T, rng.integers(0, T, size=nnz)).astype(np.uint32)
    p = np.exp(-rng.integers(0, 30, size=nnz).astype(np.float32) / np.float32(5)).astype(np.float32)
    p[rng.random(nnz) < 0.05] = 0.0
    o = c_oracle.Store(rp, tid, p, None, T)
    want, wi = c_oracle.do_em(o, max_iter=300, conv_thresh=1e-3, min_iter_gate=1)
    with DeviceStore(rp, tid, p, None, T) as d:
        got, gi = d.em_run(None, 300, 1e-3, 1)
        theta = rng.lognormal(0, 2, size=T)
        assert_counts_close(d.m_step(theta), c_oracle.m_step(o, theta), R, T, 1e-10, "m_step")
    assert abs(gi.niter - wi.niter) <= 1
    assert_counts_close(got, want, R, T, RTOL if gi.niter != wi.niter else 1e-8, "ragged")


def test_two_stores_from_two_threads():
    """Calls on distinct handles are thread-safe (single_cell.rs:96-150 calls em::em from N workers)."""
    import threading
    stores = [synth.make_store(30_000, 2_000, seed=200 + i) for i in range(3)]
    res = [None] * 3

    def work(i):
        st = stores[i]
        with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
            res[i] = [d.em_run(None, 150, 1e-3, 50)[0] for _ in range(3)]

    th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i, st in enumerate(stores):
        o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, st.n_txps)
        want, _ = c_oracle.do_em(o, max_iter=150, conv_thresh=1e-3)
        for r in res[i]:
            assert_counts_close(r, want, st.n_reads, st.n_txps, RTOL, f"thread {i}")


def test_row_shard_semantics_single_rank():
    """What a rank does with its shard, checked on one GPU with a 1-rank communicator: the uniform
    init uses the GLOBAL read count (em.rs:154,165), and the shard's bootstrap multiplicities are the
    slice of the global resample it owns (same counter-based stream on every rank)."""
    from oarfish_amd import dist as odist
    st = synth.make_store(80_000, 5_000, seed=300)
    world = 3
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as full:
        w_full = full.bootstrap_weights(42, 1)
        partial_sum = np.zeros(st.n_txps)
        theta0 = np.full(st.n_txps, st.n_reads / st.n_txps)
        want_step = full.m_step(theta0)
    comm = odist.create_comm(0, 1, 0)
    try:
        for rank in range(world):
            sh = odist.shard_rows_by_nnz(st.row_ptr, st.tid, st.as_prob, None, rank, world)
            with DeviceStore(sh.row_ptr, sh.tid, sh.as_prob, None, st.n_txps) as d:
                d.attach_comm(comm.handle, st.n_reads, sh.row_begin)
                w = d.bootstrap_weights(42, 1)
                assert np.array_equal(w, w_full[sh.row_begin:sh.row_end])
                partial_sum += d.m_step(theta0)                  # rank-local partial counts
                # one iteration from the uniform init = one pass from theta = R_global / T
                one, info = d.em_run(None, 1, 0.0, 50)
                o = c_oracle.Store(sh.row_ptr, sh.tid, sh.as_prob, None, st.n_txps)
                ref, _ = c_oracle.do_em(o, init=theta0, max_iter=1, conv_thresh=0.0)
                assert_counts_close(one, ref, st.n_reads, st.n_txps, 1e-9, f"shard {rank}")
        assert_counts_close(partial_sum, want_step, st.n_reads, st.n_txps, 1e-10, "sum of shard partials")
    finally:
        comm.close()


def test_bootstrap_first_replica_splits_a_replicate_set():
    """OEM_OPT_BOOTSTRAP_FIRST_REPLICA: replicates [b0, b1) computed by another process (replica-parallel
    multi-GPU bootstraps) are the same replicates one process computes."""
    from oarfish_amd import dist as odist
    st = synth.make_store(50_000, 3_000, seed=410)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
        allb, _ = d.bootstrap(5, seed=31, max_iter=300)
        parts = []
        for rank in range(3):
            b0, out, infos = odist.bootstrap_replica_parallel(d, 5, 31, rank, 3, max_iter=300)
            assert (b0, b0 + len(out)) == odist.replica_range(5, rank, 3)
            parts.append(out)
        again, _ = d.bootstrap(2, seed=31, max_iter=300)            # the option is per call
    got = np.concatenate(parts, axis=0)
    for b in range(5):   # batching pairs replicates differently in the two runs: fp order only
        assert_counts_close(got[b], allb[b], st.n_reads, st.n_txps, RTOL, f"replica {b}")
    assert_counts_close(again[1], allb[1], st.n_reads, st.n_txps, RTOL, "first_replica resets to 0")


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_native_sharded_loop_with_several_shards_on_one_gpu(world):
    """The native row-sharded loop (oem_em_run / oem_bootstrap on stores with an attached communicator)
    with `world` real shards.  RCCL refuses two ranks per device, so the ranks are threads of this process
    joined by the process-local communicator of the TEST-ONLY library (same call sites as RCCL: one sum of the count
    vector per pass).  Results must equal the un-sharded store's and the oracle's, with identical iteration
    counts on every rank."""
    from oarfish_amd import _lib
    with _lib.testing():
        _sharded_loop_body(world)


def _sharded_loop_body(world):
    import ctypes as C
    import threading
    from oarfish_amd import _lib, dist as odist
    st = synth.make_store(90_000, 6_000, seed=611)
    handles = (C.c_void_p * world)()
    _lib.check(_lib.lib().oem_debug_local_comm_create(world, 0, C.addressof(handles)))
    res, errs = [None] * world, []

    def rank_main(rank):
        try:
            sh = odist.shard_rows_by_nnz(st.row_ptr, st.tid, st.as_prob, None, rank, world)
            with DeviceStore(sh.row_ptr, sh.tid, sh.as_prob, None, st.n_txps) as d:
                d.attach_comm(C.c_void_p(handles[rank]), st.n_reads, sh.row_begin)
                cnt, info = d.em_run(None, 400, 1e-3, 50)
                par, pinfo = d.em_run(None, 400, 1e-3, 1)
                boots, binfo = d.bootstrap(3, seed=17, max_iter=200)
                res[rank] = (cnt, info, par, pinfo, boots, binfo)
        except Exception as e:  # pragma: no cover
            errs.append((rank, repr(e)))

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=240)
    assert not errs and all(r is not None for r in res), errs
    for r in range(world):
        _lib.lib().oem_comm_destroy(C.c_void_p(handles[r]))
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, st.n_txps)
    want, wi = c_oracle.do_em(o, max_iter=400, conv_thresh=1e-3)
    wpar, wpi = c_oracle.do_em(o, max_iter=400, conv_thresh=1e-3, min_iter_gate=1)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as full:
        wb = [c_oracle.do_em(o, max_iter=200, conv_thresh=1e-3, row_w=full.bootstrap_weights(17, b)) for b in range(3)]
    for r in range(world):
        cnt, info, par, pinfo, boots, binfo = res[r]
        assert info.niter == res[0][1].niter and pinfo.niter == res[0][3].niter          # same decision on every rank
        assert np.array_equal(cnt, res[0][0]) and np.array_equal(boots, res[0][4])        # same reduced vector
        assert abs(info.niter - wi.niter) <= 1 and abs(pinfo.niter - wpi.niter) <= 1
        assert_counts_close(cnt, want, st.n_reads, st.n_txps, RTOL if info.niter != wi.niter else 1e-9, f"rank {r} em")
        assert_counts_close(par, wpar, st.n_reads, st.n_txps, RTOL if pinfo.niter != wpi.niter else 1e-9, f"rank {r} em_par")
        for b in range(3):
            assert abs(binfo[b].niter - wb[b][1].niter) <= 1
            assert_counts_close(boots[b], wb[b][0], st.n_reads, st.n_txps,
                                RTOL if binfo[b].niter != wb[b][1].niter else 1e-9, f"rank {r} bootstrap {b}")


def test_rccl_entry_points_with_one_rank_communicator():
    """The dlopen'ed RCCL entry points (ncclGetUniqueId / ncclCommInitRank / ncclAllReduce f64 sum on
    the store's stream / ncclCommDestroy) exercised for real with a ONE-rank communicator -- all a
    1-GPU box allows; with it attached every pass runs the all-reduce and results must not move."""
    from oarfish_amd import dist as odist
    st = synth.make_store(60_000, 4_000, seed=301)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
        base, bi = d.em_run(None, 300, 1e-3, 1)
        w = np.stack([d.bootstrap_weights(9, b) for b in range(2)])
        bbase, _ = d.bootstrap(2, seed=9, row_w_all=w, max_iter=200)
        comm = odist.create_comm(0, 1, 0)                         # real RCCL communicator of size 1
        try:
            d.attach_comm(comm.handle, st.n_reads, 0)
            got, gi = d.em_run(None, 300, 1e-3, 1)
            assert gi.niter == bi.niter
            np.testing.assert_allclose(got, base, rtol=1e-9, atol=1e-9)
            bgot, _ = d.bootstrap(2, seed=9, row_w_all=w, max_iter=200)
            np.testing.assert_allclose(bgot, bbase, rtol=1e-9, atol=1e-9)
        finally:
            comm.close()


def test_edge_cases():
    # empty store: every count 0
    with DeviceStore(np.zeros(1, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.float32), None, 4) as d:
        cnt, info = d.em_run(None, 10, 1e-3, 50)
        assert np.all(cnt == 0.0) and info.n_passes == 11
    # reads but no alignment at all: a tiled layout without tiles -- no launch could carry a deferred decision, the
    # classic loop's sweep returns the reference's zeros (em.rs: every denom is an empty sum)
    with DeviceStore(np.zeros(6, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.float32), None, 4) as d:
        o0 = c_oracle.Store(np.zeros(6, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.float32), None, 4)
        for m in (0, 1, 2, 3, 10):
            cnt, info = d.em_run(None, m, 1e-3, 1)
            wo, wi = c_oracle.do_em(o0, max_iter=m, conv_thresh=1e-3, min_iter_gate=1)
            assert np.all(cnt == 0.0) and np.all(wo == 0.0)
            assert (info.niter, info.n_passes, info.converged) == (wi.niter, wi.n_passes, wi.converged), m
    # empty rows are tolerated and contribute nothing
    rp = np.array([0, 0, 2, 2, 3], dtype=np.uint64)
    tid = np.array([0, 1, 1], dtype=np.uint32)
    p = np.array([1.0, 0.5, 1.0], dtype=np.float32)
    o = c_oracle.Store(rp, tid, p, None, 3)
    with DeviceStore(rp, tid, p, None, 3) as d:
        cnt, _ = d.em_run(None, 30, 0.0, 50)
    np.testing.assert_allclose(cnt, c_oracle.do_em(o, max_iter=30, conv_thresh=0.0)[0], rtol=1e-12)
    # denom <= 1e-30 drops the read (em.rs:115)
    rp = np.array([0, 1, 2], dtype=np.uint64)
    with DeviceStore(rp, np.array([0, 1], np.uint32), np.array([1.0, 1e-38], np.float32), None, 2) as d:
        cnt, _ = d.em_run(None, 5, 0.0, 50)
    assert cnt[0] == 1.0 and cnt[1] == 0.0
    # max_iter = 0
    g = load_golden("random_a")
    with _dev(g) as d:
        cnt, info = d.em_run(None, 0, 1e-3, 50)
        assert info.niter == 0 and info.n_passes == 1
        assert_counts_close(cnt, g["runs"][0]["counts"], 1500, 200, 1e-10, "max_iter 0")


def test_store_lifecycle_returns_device_memory():
    """Create / run / destroy cycles (point estimate, batched bootstrap, per-cell batch, a failed create)
    must hand all HBM back: the scratch of the device layout builder included."""
    import torch
    st = synth.make_store(120_000, 9_000, seed=77)
    cell_off, crp, ctid, cp = synth.make_cells(6, 3_000, 800, seed=3)

    def cycle():
        with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
            d.em_run(None, 20, 0.0, 50)
            d.bootstrap(3, seed=1, max_iter=20)
            d.aux_counts()
        oarfish_amd.em_cells(cell_off, crp, ctid, cp, None, 800, max_iter=20)
        bad = st.tid.copy(); bad[5] = st.n_txps + 3
        with pytest.raises(oarfish_amd.OemError):
            DeviceStore(st.row_ptr, bad, st.as_prob, None, st.n_txps)

    cycle()                                       # warm up allocator pools / code objects
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(8):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, f"device memory shrank by {(free0 - free1) >> 20} MiB over 8 cycles"


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_random_shapes_match_oracle(seed):
    """Randomly shaped stores through the whole device path (device-built layout, tile + fold kernels,
    loop state) against the oracle: read lengths 0..120 (beyond --best-n), tiny to wide transcript spaces,
    repeated transcripts inside a read, zero and denormal weights, optional coverage column, random init,
    both gates, point estimate + one injected bootstrap resample."""
    rng = np.random.default_rng(9000 + seed)
    R = int(rng.choice([1, 7, 300, 5_000, 40_000]))
    T = int(rng.choice([1, 3, 40, 2_000, 70_000, 300_000]))
    maxk = int(rng.choice([1, 3, 12, 120]))
    lens = rng.integers(0, maxk + 1, size=R)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    nnz = int(rp[-1])
    base = np.repeat(rng.integers(0, T, size=R), lens)
    spread = int(rng.choice([1, 8, 200, max(T, 1)]))
    tid = ((base + rng.integers(0, spread, size=nnz)) % T).astype(np.uint32)
    p = np.exp(-rng.integers(0, 60, size=nnz) / 5.0).astype(np.float32)
    p[rng.random(nnz) < 0.03] = 0.0
    p[rng.random(nnz) < 0.01] = np.float32(1e-42)                         # denormal
    cov = rng.uniform(1e-3, 1.0, size=nnz) if seed % 3 == 0 else None
    init = rng.gamma(1.0, R / max(T, 1) + 0.1, size=T) if seed % 4 == 1 else None
    gate = 1 if seed % 2 else 50
    o = c_oracle.Store(rp, tid, p, cov, T)
    want, wi = c_oracle.do_em(o, init=init, max_iter=80, conv_thresh=1e-3, min_iter_gate=gate)
    w = np.bincount(rng.integers(0, max(R, 1), size=R), minlength=R).astype(np.uint32)[:R]
    wantb, wbi = c_oracle.do_em(o, max_iter=60, conv_thresh=1e-3, row_w=w)
    with DeviceStore(rp, tid, p, cov, T) as d:
        got, gi = d.em_run(init, 80, 1e-3, gate)
        gotb, gbi = d.bootstrap(1, row_w_all=w[None, :], max_iter=60)
    what = f"seed {seed}: R={R} T={T} maxk={maxk} spread={spread} cov={cov is not None} gate={gate}"
    assert abs(gi.niter - wi.niter) <= 1, what
    assert_counts_close(got, want, R, T, RTOL if gi.niter != wi.niter else 1e-9, what)
    assert abs(gbi[0].niter - wbi.niter) <= 1, what
    assert_counts_close(gotb[0], wantb, R, T, RTOL if gbi[0].niter != wbi.niter else 1e-9, what + " (bootstrap)")


def test_read_with_more_than_255_window_alignments_takes_the_csr_path():
    """A read with 300 alignments inside one window cannot be tiled (slice widths are bytes): both layout
    builders decline, the store runs on the caller-order CSR kernel, and results still match the oracle."""
    rng = np.random.default_rng(77)
    T = 2_000
    lens = np.concatenate([[300], rng.integers(1, 9, size=4_000)])
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    tid = np.concatenate([np.arange(100, 400), (np.repeat(rng.integers(0, T, size=4_000), lens[1:]) +
                                                 rng.integers(0, 5, size=int(lens[1:].sum()))) % T]).astype(np.uint32)
    p = np.exp(-rng.integers(0, 30, size=len(tid)) / 5.0).astype(np.float32)
    o = c_oracle.Store(rp, tid, p, None, T)
    want, wi = c_oracle.do_em(o, max_iter=150, conv_thresh=1e-3)
    import ctypes as C
    from oarfish_amd import _lib
    with _lib.testing(), DeviceStore(rp, tid, p, None, T) as d:     # (the layout-hash hook lives in the test-only library)
        out = (C.c_uint64 * 15)()
        assert _lib.lib().oem_debug_layout_hash(d.handle, C.addressof(out), 15) == _lib.OEM_ERR_STATE  # no tiled layout
    with DeviceStore(rp, tid, p, None, T) as d:
        got, gi = d.em_run(None, 150, 1e-3, 50)
    assert abs(gi.niter - wi.niter) <= 1
    assert_counts_close(got, want, len(lens), T, RTOL if gi.niter != wi.niter else 1e-9, "CSR path")


@pytest.mark.parametrize("tag", ["C", "I", "O"])
def test_config0_sirv_shaped_store(tag):
    """BASELINE configs[0] on the device: SIRV annotation (69 / 44 / 100 transcripts), bulk mode,
    100 EM iterations, with and without the coverage column."""
    for coverage in (False, True):
        st = synth.make_sirv_store(tag, 20_000, coverage=coverage)
        o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps)
        want, wi = c_oracle.do_em(o, max_iter=100, conv_thresh=0.0)
        with DeviceStore(st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps) as d:
            got, gi = d.em_run(None, 100, 0.0, 50)
            par, pi = d.em_run(None, 1000, 1e-3, 1)
        assert gi.niter == wi.niter == 100 and gi.n_passes == 101
        assert_counts_close(got, want, st.n_reads, st.n_txps, 1e-9, f"SIRV {tag} cov={coverage}")
        wpar, wpi = c_oracle.do_em(o, max_iter=1000, conv_thresh=1e-3, min_iter_gate=1)
        assert abs(pi.niter - wpi.niter) <= 1
        assert_counts_close(par, wpar, st.n_reads, st.n_txps, RTOL, f"SIRV {tag} em_par cov={coverage}")


def test_full_size_c3_properties():
    """BASELINE configs[2] (10 M reads x 200 k transcripts, ~80 M alignments) on one GPU:
    size-independent properties, plus parity with the multi-threaded oracle over a short fixed
    number of iterations (the oracle needs ~0.15 s per pass at this size)."""
    st = synth.make_config("c3")
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, st.n_txps)
    u, t = c_oracle.aux_counts(o)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
        du, dt = d.aux_counts()
        assert np.array_equal(du, u) and np.array_equal(dt, t)                # integer work: bit-exact
        cnt, info = d.em_run(None, 12, 0.0, 50)
        assert info.niter == 12 and info.n_passes == 13
        assert abs(cnt.sum() - st.n_reads) < 1e-7 * st.n_reads                # mass conservation
        assert np.all(cnt >= u - 1e-6) and np.all(cnt <= t + 1e-6)            # unique <= count <= total
        want, wi = c_oracle.em_par(o, max_iter=12, conv_thresh=0.0, min_iter_gate=50)
        assert_counts_close(cnt, want, st.n_reads, st.n_txps, 1e-8, "c3, 12 iterations")
        full, finfo = d.em_run(None, 1000, 1e-3, 1)                           # em_par semantics to convergence
        assert abs(full.sum() - st.n_reads) < 1e-7 * st.n_reads
        assert np.all(full >= u - 1e-6) and np.all(full <= t + 1e-6)
        w = d.bootstrap_weights(3, 0)
        assert int(w.sum()) == st.n_reads and abs(w.var() - 1.0) < 0.01
        # The batched bootstrap at full size (k_em_tile_e: 4 slots per pass, one epoch; two chains): a short fixed
        # number of iterations of a drawn resample and of the identity resample against the oracle, ...
        W = np.stack([w, np.ones(st.n_reads, dtype=np.uint32), d.bootstrap_weights(3, 2)])
        bout, binfo = d.bootstrap(3, row_w_all=W, max_iter=8, conv_thresh=0.0)
        for b in (0, 1):
            wantb, _ = c_oracle.do_em(o, row_w=W[b], max_iter=8, conv_thresh=0.0)   # the serial oracle, ~0.2 s per pass
            assert binfo[b].niter == 8 and binfo[b].n_passes == 9
            assert_counts_close(bout[b], wantb, st.n_reads, st.n_txps, 1e-8, f"c3 batched bootstrap, replicate {b}")
        # ... and to convergence: the identity resample through the batch kernels must walk the point
        # estimate's trajectory (gate 50) and stop where it stops
        point, pinfo = d.em_run(None, 1000, 1e-3, 50)
        ident, iinfo = d.bootstrap(2, row_w_all=W[1:3], max_iter=1000, conv_thresh=1e-3)
        assert abs(iinfo[0].niter - pinfo.niter) <= 1
        assert_counts_close(ident[0], point, st.n_reads, st.n_txps, RTOL if iinfo[0].niter != pinfo.niter else 1e-8,
                            "identity resample through the batch path")
        assert abs(ident[1].sum() - st.n_reads) < 1e-7 * st.n_reads           # a resample keeps the read count


@pytest.mark.timeout(1200)
def test_full_size_c3_to_convergence_matches_oracle_under_both_gates():
    """BASELINE configs[2] as stated: the 10 M x 200 k store, the reference's defaults (max_iter 1000,
    conv_thresh 1e-3), EM TO CONVERGENCE, under the gate of em::em (niter > 50, em.rs:212) and of em::em_par
    (niter > 1, em.rs:399), every transcript against the serial oracle (do_em, ~0.15 s per pass: minutes).
    One oracle run serves both gates: it runs with gate 1, and when it stops at niter > 51 the condition
    `rel_diff < thresh` was false at every earlier iteration, so the gate-50 loop walks the identical
    trajectory and stops at the same iteration (the stopping rule differs in nothing else)."""
    st = synth.make_config("c3")
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, st.n_txps)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
        got = {gate: d.em_run(None, 1000, 1e-3, gate) for gate in (50, 1)}
    want, wi = c_oracle.do_em(o, max_iter=1000, conv_thresh=1e-3, min_iter_gate=1)
    assert wi.converged and wi.niter > 51, wi      # else the oracle must be run once per gate
    for gate, (cnt, info) in got.items():
        assert abs(info.niter - wi.niter) <= 1, (gate, info, wi)
        assert info.converged and info.n_passes == info.niter + 2
        assert_counts_close(cnt, want, st.n_reads, st.n_txps, RTOL if info.niter != wi.niter else 1e-8,
                            f"c3 to convergence, gate {gate}")
        assert abs(cnt.sum() - st.n_reads) < 1e-7 * st.n_reads


@pytest.mark.timeout(1500)
def test_full_size_c3_device_drawn_replicate_to_its_own_convergence_matches_oracle():
    """BASELINE configs[2]'s "+ bootstraps" leg as the bench runs it: replicates whose resamples the DEVICE draws
    (Philox, `get_sample_inds` in multiplicity form, bootstrap.rs:7-16), each run through the batched path to ITS OWN
    stop with the reference's defaults (em.rs:273-290: do_em over the resampled reads, gate 50, max_iter 1000 --
    which a C3 replicate may well run into: the point estimate needs 890 iterations).  Replicate 0 against the serial
    oracle on the same multiplicities -- every transcript, iteration count +- 1 (~1000 passes of ~0.2 s).  Beside it
    slots that stop elsewhere: the same replicates under a looser threshold stop at iterations of their own, and every
    one of them is checked against the one-replicate-per-pass path (k_em_tile with per-read multiplicities), which
    shares no kernel with the batched one."""
    st = synth.make_config("c3")
    T = st.n_txps
    seed = 20260929
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, T) as d:
        w0 = d.bootstrap_weights(seed, 0)                   # what the batched path draws for replicate 0
        assert int(w0.sum()) == st.n_reads and int((w0 == 0).sum()) > 0.3 * st.n_reads
        bout, binfo = d.bootstrap(3, seed=seed, max_iter=1000, conv_thresh=1e-3)          # nothing injected
        lout, linfo = d.bootstrap(5, seed=seed, max_iter=1000, conv_thresh=1e-2)          # five slots over 2 x 4
        d.set_option(_lib.OEM_OPT_BATCH_BOOTSTRAP, 0)
        sout, sinfo = d.bootstrap(5, seed=seed, max_iter=1000, conv_thresh=1e-2)
        s1out, s1info = d.bootstrap(1, seed=seed, max_iter=1000, conv_thresh=1e-3, first_replica=1)
    for out_, info_ in ((bout, binfo), (lout, linfo)):
        for b in range(len(info_)):
            assert info_[b].niter > 51 and info_[b].n_passes == info_[b].niter + (2 if info_[b].converged else 1), info_[b]
            assert abs(out_[b].sum() - st.n_reads) < 1e-7 * st.n_reads      # a resample keeps the read count
    assert all(i.converged for i in linfo) and len({i.niter for i in linfo}) > 1, [i.niter for i in linfo]
    for k in range(5):   # batched slot vs the same replicate alone on the point-estimate kernels
        assert linfo[k].niter == sinfo[k].niter, (k, linfo[k], sinfo[k])
        assert_counts_close(lout[k], sout[k], st.n_reads, T, 1e-8, f"replicate {k} at 1e-2: batched vs one per pass")
    assert binfo[1].niter == s1info[0].niter
    assert_counts_close(bout[1], s1out[0], st.n_reads, T, 1e-8, "replicate 1 at the defaults: batched vs one per pass")
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, T)
    want, wi = c_oracle.do_em(o, row_w=w0, max_iter=1000, conv_thresh=1e-3, min_iter_gate=50)
    assert abs(binfo[0].niter - wi.niter) <= 1, (binfo[0], wi)
    assert binfo[0].converged == wi.converged
    assert_counts_close(bout[0], want, st.n_reads, T, RTOL if binfo[0].niter != wi.niter else 1e-8,
                        "c3 device-drawn replicate 0 to its own stop")


@pytest.mark.timeout(900)
def test_full_size_c3_coverage_store_matches_oracle():
    """The C3-sized store WITH the coverage column (--model-coverage, the authors' recommended mode: f64 weights
    w = (f64)as_prob * cov_prob, em.rs:107-111; 12 bytes per alignment, the f64 instantiations of both tile kernels):
    12 iterations against the multi-threaded oracle on every transcript, then to convergence with the invariants --
    and two batched-bootstrap replicates over the same f64 store against the serial oracle."""
    st = synth.make_store(10_000_000, 200_000, 8.0, coverage=True, threads=min(32, os.cpu_count() or 8))
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps)
    u, t = c_oracle.aux_counts(o)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps) as d:
        assert d.info(_lib.OEM_INFO_WEIGHT_DICT_ENTRIES) == 0        # f64 products: the plain stream
        cnt, info = d.em_run(None, 12, 0.0, 50)
        assert info.niter == 12 and info.n_passes == 13
        want, wi = c_oracle.em_par(o, max_iter=12, conv_thresh=0.0, min_iter_gate=50)
        assert_counts_close(cnt, want, st.n_reads, st.n_txps, 1e-8, "c3 with coverage, 12 iterations")
        assert abs(cnt.sum() - st.n_reads) < 1e-7 * st.n_reads       # mass conservation
        full, finfo = d.em_run(None, 1000, 1e-3, 50)                 # the reference's defaults (this store runs into max_iter)
        assert finfo.n_passes == finfo.niter + (2 if finfo.converged else 1) and finfo.niter > 51
        assert abs(full.sum() - st.n_reads) < 1e-7 * st.n_reads
        assert np.all(full >= u - 1e-6) and np.all(full <= t + 1e-6)  # unique <= count <= total
        W = np.stack([d.bootstrap_weights(5, 0), np.ones(st.n_reads, dtype=np.uint32)])
        bout, binfo = d.bootstrap(2, row_w_all=W, max_iter=6, conv_thresh=0.0)
        wantb, _ = c_oracle.do_em(o, row_w=W[0], max_iter=6, conv_thresh=0.0)
        assert binfo[0].niter == 6 and binfo[0].n_passes == 7
        assert_counts_close(bout[0], wantb, st.n_reads, st.n_txps, 1e-8, "c3 with coverage, batched bootstrap replicate")
        assert abs(bout[1].sum() - st.n_reads) < 1e-7 * st.n_reads
    # opt-in weight_coding = 2: the static weight (p as f64) * cov rounded once to f32 (<= 6e-8 relative per weight),
    # the store an f32 store (8 B per alignment, the kernels of `roofline.frac_coverage_f32w`) -- against the SAME
    # strict-f64 oracle at 1e-6, and to convergence against the f64 store's run
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps, weight_coding=2) as d2:
        _h, alg = d2.bytes()
        assert alg == st.tid.size * 8 + (st.n_reads + 1) * 4 + 2 * st.n_txps * 8
        cnt2, info2 = d2.em_run(None, 12, 0.0, 50)
        assert info2.niter == 12 and info2.n_passes == 13
        assert_counts_close(cnt2, want, st.n_reads, st.n_txps, 1e-6, "c3 with coverage, weights rounded to f32, 12 iterations")
        assert abs(cnt2.sum() - st.n_reads) < 1e-7 * st.n_reads
        full2, finfo2 = d2.em_run(None, 1000, 1e-3, 50)
        assert abs(int(finfo2.niter) - int(finfo.niter)) <= 1 and finfo2.converged == finfo.converged
        assert_counts_close(full2, full, st.n_reads, st.n_txps, RTOL, "c3 with coverage, weights rounded to f32, to the stop")


def test_coverage_weights_rounded_to_f32_are_exactly_that():
    """oem_store_opts.weight_coding = 2 changes ONE thing: the stored static weight is f32((p as f64) * cov).  The run
    equals the oracle's on a store whose as_prob IS that rounded product (1e-10: only summation order differs), stays
    within the 1e-4 bar of the strict-f64 oracle to its stopping point, and without a coverage column the option is coding 0 (the lossless table)."""
    st = synth.make_store(120_000, 5_000, 6.0, seed=29, coverage=True)
    T = st.n_txps
    w32 = (st.as_prob.astype(np.float64) * st.cov_prob).astype(np.float32)
    o_rounded = c_oracle.Store(st.row_ptr, st.tid, w32, None, T)
    o_strict = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, st.cov_prob, T)
    for gate, thresh, m in ((50, 1e-3, 1000), (1, 0.0, 40)):
        with DeviceStore(st.row_ptr, st.tid, st.as_prob, st.cov_prob, T, weight_coding=2) as d:
            assert d.info(_lib.OEM_INFO_WEIGHT_DICT_ENTRIES) == 0
            cnt, info = d.em_run(None, m, thresh, gate)
            W = d.bootstrap_weights(3, 0)
            bout, _ = d.bootstrap(1, row_w_all=W[None, :], max_iter=8, conv_thresh=0.0)
        want, wi = c_oracle.do_em(o_rounded, max_iter=m, conv_thresh=thresh, min_iter_gate=gate)
        assert (info.niter, info.converged) == (wi.niter, wi.converged)
        assert_counts_close(cnt, want, st.n_reads, T, 1e-10, "rounded weights vs the oracle on the rounded weights")
        strict, si = c_oracle.do_em(o_strict, max_iter=m, conv_thresh=thresh, min_iter_gate=gate)
        if si.niter == info.niter:
            # (the 6e-8 of a weight grows along the trajectory: 1.4e-5 at this store's stopping point, observed)
            assert_counts_close(cnt, strict, st.n_reads, T, RTOL, "rounded weights vs the strict-f64 oracle")
        wantb, _ = c_oracle.do_em(o_rounded, row_w=W, max_iter=8, conv_thresh=0.0)
        assert_counts_close(bout[0], wantb, st.n_reads, T, 1e-10, "rounded weights, injected replicate")
    plain = synth.make_store(50_000, 2_000, 6.0, seed=30)
    with DeviceStore(plain.row_ptr, plain.tid, plain.as_prob, None, plain.n_txps, weight_coding=2) as d:
        assert d.info(_lib.OEM_INFO_WEIGHT_DICT_ENTRIES) > 0     # no coverage column: coding 0


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind", ["uniform_gaps_words16", "f32_stream"])
def test_full_size_c3_other_weight_codings_match_oracle(kind):
    """The C3-sized store through the two weight codings the default store does not take: long-read score gaps (514
    distinct weights: 16-bit indices, 4 KiB table in LDS) and the plain f32 stream (weight_coding = 1) -- the kernels
    behind `roofline.frac_uniform_gaps` and `roofline.frac_f32_stream`.  12 iterations against the multi-threaded oracle
    on every transcript, mass conservation, and a batched-bootstrap replicate against the serial oracle."""
    threads = min(32, os.cpu_count() or 8)
    if kind == "f32_stream":
        st, coding = synth.make_config("c3"), 1
    else:
        st, coding = synth.make_store(10_000_000, 200_000, 8.0, threads=threads, gaps="uniform"), 0
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, st.n_txps)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps, weight_coding=coding) as d:
        n = d.info(_lib.OEM_INFO_WEIGHT_DICT_ENTRIES)
        assert (n == 0) if coding else (256 < n <= 1024), n
        cnt, info = d.em_run(None, 12, 0.0, 1)
        assert info.niter == 12 and info.n_passes == 13
        want, wi = c_oracle.em_par(o, max_iter=12, conv_thresh=0.0, min_iter_gate=1)
        assert_counts_close(cnt, want, st.n_reads, st.n_txps, 1e-8, f"c3 {kind}, 12 iterations")
        assert abs(cnt.sum() - st.n_reads) < 1e-7 * st.n_reads
        W = np.stack([d.bootstrap_weights(7, 0), np.ones(st.n_reads, dtype=np.uint32)])
        bout, binfo = d.bootstrap(2, row_w_all=W, max_iter=6, conv_thresh=0.0)
        wantb, _ = c_oracle.do_em(o, row_w=W[0], max_iter=6, conv_thresh=0.0)
        assert binfo[0].niter == 6 and binfo[0].n_passes == 7
        assert_counts_close(bout[0], wantb, st.n_reads, st.n_txps, 1e-8, f"c3 {kind}, batched bootstrap replicate")
        assert abs(bout[1].sum() - st.n_reads) < 1e-7 * st.n_reads


@pytest.mark.timeout(900)
def test_c5_slice_of_one_gpu_properties():
    """BASELINE configs[4] at the size ONE GPU sees when 5 k cells are dealt to 8: 625 cells x 50 k reads (31 M reads,
    250 M alignments, one batched store: wide windows, fused fold, live-tile compaction).  No oracle at this size: per
    cell mass conservation and unique <= count <= total (aux counts over the cell's own rows), cells stop at their own
    iterations, and a slice of the batch equals the same cells run as a batch of their own."""
    n_cells, per_cell, T = 625, 50_000, 60_000
    cell_off, row_ptr, tid, p = synth.make_cells(n_cells, per_cell, T, seed=37, threads=min(32, os.cpu_count() or 4))
    out, infos = oarfish_amd.em_cells(cell_off, row_ptr, tid, p, None, T, max_iter=1000, convergence_thresh=1e-3)
    assert out.shape == (n_cells, T)
    assert np.abs(out.sum(axis=1) - per_cell).max() < 1e-6 * per_cell
    passes = np.array([i.n_passes for i in infos])
    assert passes.min() >= 53 and passes.max() <= 1001 and len(set(passes.tolist())) > 20
    for c in (0, 311, 624):   # unique <= count <= total, from the cell's own alignments
        r0, r1 = int(cell_off[c]), int(cell_off[c + 1])
        a0, a1 = int(row_ptr[r0]), int(row_ptr[r1])
        lens = (row_ptr[r0 + 1:r1 + 1] - row_ptr[r0:r1]).astype(np.int64)
        tot = np.bincount(tid[a0:a1], minlength=T)
        uniq = np.bincount(tid[a0:a1][np.repeat(lens == 1, lens)], minlength=T)
        assert np.all(out[c] >= uniq - 1e-6) and np.all(out[c] <= tot + 1e-6), c
    # the first 8 cells as a batch of their own: the same answers (a cell's run does not depend on its batch)
    r8 = int(cell_off[8]); a8 = int(row_ptr[r8])
    out8, infos8 = oarfish_amd.em_cells(cell_off[:9], row_ptr[:r8 + 1], tid[:a8], p[:a8], None, T, max_iter=1000,
                                        convergence_thresh=1e-3)
    for c in range(8):
        assert infos8[c].niter == infos[c].niter
        assert_counts_close(out8[c], out[c], per_cell, T, 1e-9, f"cell {c}: batch of 8 vs batch of 625")


@pytest.mark.timeout(1800)
def test_c5_whole_5000_cells_in_one_call():
    """BASELINE configs[4] WHOLE on one GPU: 5 000 cells x 50 k reads (250 M reads, 2 G alignments, 21 GB of host
    arrays) in ONE oem_em_run_cells call (eight groups of ~660 cells on two host threads, single_cell.rs:139-160 per
    cell).  The cells are 625 generated ones x 8 transcript-id rotations (synth.replicate_cells: a relabelling is an
    exact symmetry of the EM, and generating 5 k cells in Python would take the driver's tier ten minutes), so besides
    per-cell mass conservation, unique <= count <= total and "a batch of 8 = the same cells in the batch of 5 000",
    every rotated copy must reproduce its base cell's counts under the rotation -- through other ids, tiles and
    windows, in another group of the call."""
    n_cells, n_base, per_cell, T = 5_000, 625, 50_000, 60_000
    base = synth.make_cells(n_base, per_cell, T, seed=37, threads=min(32, os.cpu_count() or 4))
    cell_off, row_ptr, tid, p = synth.replicate_cells(base, T, n_cells)
    assert len(cell_off) == n_cells + 1 and int(cell_off[-1]) == n_cells * per_cell
    out, infos = oarfish_amd.em_cells(cell_off, row_ptr, tid, p, None, T, max_iter=1000, convergence_thresh=1e-3)
    assert out.shape == (n_cells, T)
    assert np.abs(out.sum(axis=1) - per_cell).max() < 1e-6 * per_cell         # mass conservation, every cell
    passes = np.array([i.n_passes for i in infos])
    assert passes.min() >= 53 and passes.max() <= 1001 and len(set(passes.tolist())) > 20
    for c in (0, 2_499, 4_999):   # unique <= count <= total, from the cell's own alignments
        r0, r1 = int(cell_off[c]), int(cell_off[c + 1])
        a0, a1 = int(row_ptr[r0]), int(row_ptr[r1])
        lens = (row_ptr[r0 + 1:r1 + 1] - row_ptr[r0:r1]).astype(np.int64)
        tot = np.bincount(tid[a0:a1], minlength=T)
        uniq = np.bincount(tid[a0:a1][np.repeat(lens == 1, lens)], minlength=T)
        assert np.all(out[c] >= uniq - 1e-6) and np.all(out[c] <= tot + 1e-6), c
    # a rotated copy = its base cell under the rotation (copies of one cell sit in different groups of the call)
    worst = 0
    for c in list(range(0, n_base, 7)) + [n_base - 1]:
        for r in range(1, n_cells // n_base):
            k = r * n_base + c
            back = np.roll(out[k], -synth.cell_shift(r, T))
            dn = abs(infos[k].niter - infos[c].niter)
            worst = max(worst, dn)
            assert dn <= 1, (c, r, infos[k], infos[c])
            assert_counts_close(back, out[c], per_cell, T, RTOL if dn else 1e-8, f"cell {c}, rotation {r}")
    # the first 8 cells as a batch of their own: the same answers (a cell's run does not depend on its batch)
    r8 = int(cell_off[8]); a8 = int(row_ptr[r8])
    out8, infos8 = oarfish_amd.em_cells(cell_off[:9], row_ptr[:r8 + 1], tid[:a8], p[:a8], None, T, max_iter=1000,
                                        convergence_thresh=1e-3)
    for c in range(8):
        assert infos8[c].niter == infos[c].niter
        assert_counts_close(out8[c], out[c], per_cell, T, 1e-9, f"cell {c}: batch of 8 vs batch of 5000")


@pytest.mark.parametrize("name", ["c2"])
def test_full_size_properties(name):
    """BASELINE configs[1] (1M reads x 60k txps): size-independent properties + fixed-iteration
    parity against the (multi-threaded) oracle."""
    st = synth.make_config(name)
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, st.n_txps)
    u, t = c_oracle.aux_counts(o)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
        # BASELINE configs[1]: 1000 EM iterations, no early exit, tolerance 1e-4 against the CPU
        cnt, info = d.em_run(None, 1000, 0.0, 50)
        assert info.niter == 1000 and info.n_passes == 1001 and not info.converged
        assert abs(cnt.sum() - st.n_reads) < 1e-7 * st.n_reads       # mass conservation
        assert np.all(cnt >= u - 1e-6) and np.all(cnt <= t + 1e-6)   # unique <= count <= total
        want, wi = c_oracle.em_par(o, max_iter=1000, conv_thresh=0.0, min_iter_gate=50)
        assert wi.niter == 1000
        assert_counts_close(cnt, want, st.n_reads, st.n_txps, RTOL, "c2 1000 iterations")
        # idempotence of the resident store: a second run gives the same answer
        cnt2, _ = d.em_run(None, 1000, 0.0, 50)
        assert_counts_close(cnt2, cnt, st.n_reads, st.n_txps, 1e-9, "rerun")
        # linearity of one pass in the row weights: E(w1) + E(w2) = E(w1 + w2)
        rng = np.random.default_rng(2)
        w1 = rng.integers(0, 3, st.n_reads).astype(np.uint32)
        w2 = rng.integers(0, 3, st.n_reads).astype(np.uint32)
        theta = cnt + 1e-3
        a, b, c = d.m_step(theta, w1), d.m_step(theta, w2), d.m_step(theta, w1 + w2)
        assert_counts_close(a + b, c, st.n_reads, st.n_txps, 1e-9, "linearity")


# ---------------------------------------------------------------------------------------------------
# The wide-window instantiation of the tile kernel (k_em_tile<..., kWinWide>: window cap 2048, one
# count-window copy) is what every large sparse store and every per-cell batch runs.  Forced here
# through oem_store_opts.window_cap on stores of every shape, against the oracle.
# ---------------------------------------------------------------------------------------------------
WIDE = 2048


@pytest.mark.parametrize("coverage", [False, True])
def test_wide_window_m_step_matches_oracle(coverage):
    st = synth.make_store(50_000, 30_000, seed=177, coverage=coverage)      # sparse: the wide cap's home ground
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps)
    rng = np.random.default_rng(11)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps, window_cap=WIDE) as d:
        for trial in range(3):
            theta = rng.lognormal(0, 2, size=st.n_txps)
            theta[rng.random(st.n_txps) < 0.1] = 0.0
            assert_counts_close(d.m_step(theta), c_oracle.m_step(o, theta), st.n_reads, st.n_txps, 1e-10,
                                f"wide m_step trial {trial}")
        w = rng.poisson(1.0, size=st.n_reads).astype(np.uint32)
        theta = np.full(st.n_txps, st.n_reads / st.n_txps)
        assert_counts_close(d.m_step(theta, w), c_oracle.m_step(o, theta, row_w=w), st.n_reads, st.n_txps, 1e-10,
                            "wide weighted m_step")


@pytest.mark.parametrize("shape", ["sparse", "dense", "coverage"])
@pytest.mark.parametrize("gate", [50, 1])
def test_wide_window_em_matches_oracle(shape, gate):
    if shape == "sparse":
        st = synth.make_store(120_000, 100_000, seed=178)
    elif shape == "dense":
        st = synth.make_store(200_000, 12_000, seed=78)
    else:
        st = synth.make_store(80_000, 50_000, seed=179, coverage=True)
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps)
    want, wi = c_oracle.do_em(o, max_iter=1000, conv_thresh=1e-3, min_iter_gate=gate)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps, window_cap=WIDE) as d:
        got, gi = d.em_run(None, 1000, 1e-3, gate)
        narrow = DeviceStore(st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps, window_cap=512)
        try:
            ncnt, ni = narrow.em_run(None, 1000, 1e-3, gate)
        finally:
            narrow.close()
    assert abs(gi.niter - wi.niter) <= 1
    assert_counts_close(got, want, st.n_reads, st.n_txps, RTOL if gi.niter != wi.niter else 1e-8, f"wide em {shape}")
    if ni.niter == gi.niter:   # the two window caps are two layouts of the same sums
        assert_counts_close(got, ncnt, st.n_reads, st.n_txps, 1e-8, "wide vs narrow")
    assert abs(got.sum() - st.n_reads) < 1e-6 * st.n_reads


def test_wide_window_injected_bootstrap_matches_oracle():
    st = synth.make_store(60_000, 45_000, seed=180)
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, st.n_txps)
    rng = np.random.default_rng(5)
    W = np.stack([np.bincount(rng.integers(0, st.n_reads, st.n_reads), minlength=st.n_reads)
                  for _ in range(3)]).astype(np.uint32)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps, window_cap=WIDE) as d:
        out, infos = d.bootstrap(3, row_w_all=W, max_iter=300, conv_thresh=1e-3)
    for b in range(3):
        want, wi = c_oracle.do_em(o, row_w=W[b], max_iter=300, conv_thresh=1e-3)
        assert abs(infos[b].niter - wi.niter) <= 1
        assert_counts_close(out[b], want, st.n_reads, st.n_txps, RTOL if infos[b].niter != wi.niter else 1e-8,
                            f"wide bootstrap {b}")


@pytest.mark.parametrize("seed", range(8))
def test_wide_window_fuzz_random_shapes_match_oracle(seed):
    rng = np.random.default_rng(9100 + seed)
    R = int(rng.choice([1, 33, 900, 25_000, 90_000]))
    T = int(rng.choice([1, 7, 400, 5_000, 120_000]))
    maxk = int(rng.choice([1, 5, 16, 60]))
    lens = rng.integers(0, maxk + 1, size=R)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    nnz = int(rp[-1])
    spread = int(rng.choice([1, 8, 300, max(T, 1)]))
    tid = ((np.repeat(rng.integers(0, T, size=R), lens) + rng.integers(0, spread, size=nnz)) % T).astype(np.uint32)
    p = np.exp(-rng.integers(0, 40, size=nnz) / 5.0).astype(np.float32)
    cov = rng.uniform(1e-3, 1.0, size=nnz) if seed % 3 == 0 else None
    o = c_oracle.Store(rp, tid, p, cov, T)
    want, wi = c_oracle.do_em(o, max_iter=200, conv_thresh=1e-3)
    with DeviceStore(rp, tid, p, cov, T, window_cap=WIDE) as d:
        got, gi = d.em_run(None, 200, 1e-3, 50)
    what = f"wide fuzz seed {seed}: R={R} T={T} maxk={maxk} spread={spread} cov={cov is not None}"
    assert abs(gi.niter - wi.niter) <= 1, what
    assert_counts_close(got, want, max(R, 1), T, RTOL if gi.niter != wi.niter else 1e-8, what)


@pytest.mark.timeout(1500)
def test_c5_slice_per_cell_batch_matches_per_cell_oracle():
    """BASELINE configs[4] at size: a slice of the single-cell workload -- 64 cells x 50 k reads each,
    T = 60 k (3.2 M reads over 3.84 M virtual transcripts => the store picks the wide window cap by
    itself and the fused fold finishes each pass) -- every cell against its own serial oracle run
    (single_cell.rs:139-160: em::em per cell, init None, gate 50; em.rs:212)."""
    from concurrent.futures import ThreadPoolExecutor
    n_cells, per_cell, T = 64, 50_000, 60_000
    cell_off, row_ptr, tid, p = synth.make_cells(n_cells, per_cell, T, seed=31)
    out, infos = oarfish_amd.em_cells(cell_off, row_ptr, tid, p, None, T, max_iter=1000, convergence_thresh=1e-3)

    def oracle_cell(c):
        r0, r1 = int(cell_off[c]), int(cell_off[c + 1])
        a0, a1 = int(row_ptr[r0]), int(row_ptr[r1])
        o = c_oracle.Store(row_ptr[r0:r1 + 1] - row_ptr[r0], tid[a0:a1], p[a0:a1], None, T)
        return c_oracle.do_em(o, max_iter=1000, conv_thresh=1e-3, min_iter_gate=50)

    import os
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 4)) as ex:   # ctypes releases the GIL
        wants = list(ex.map(oracle_cell, range(n_cells)))
    niters = set()
    for c in range(n_cells):
        want, wi = wants[c]
        assert abs(infos[c].niter - wi.niter) <= 1, (c, infos[c], wi)
        assert_counts_close(out[c], want, per_cell, T, RTOL if infos[c].niter != wi.niter else 1e-8, f"cell {c}")
        assert abs(out[c].sum() - per_cell) < 1e-6 * per_cell
        niters.add(wi.niter)
    assert len(niters) > 4      # the cells really stop at different iterations


def test_stopping_rule_kernel_under_stress():
    """k_reldiff_swap_clear elects the workgroup that takes the stopping decision (em.rs:194-218) with
    two device-scope atomics per workgroup (running maximum, then a ticket) ordered by a counted wait
    instead of a fence.  Hook of the test-only library: > 10^5 launches over grids of 1 .. 64
    workgroups, every launch with a planted, exactly representable maximum at a pseudo-random place;
    the decision workgroup's view of the maximum must equal it bit for bit every single time."""
    import ctypes as C
    from oarfish_amd import _lib
    L = _lib.testing_lib()
    total = 0
    for T, n in ((1, 2_000), (900, 10_000), (5_000, 20_000), (33_000, 30_000), (70_000, 30_000), (200_000, 20_000),
                 (1_500_000, 3_000)):
        out = np.zeros(n)
        rc = L.oem_test_reldiff_stress(T, n, 1234 + T, 0, out.ctypes.data)
        assert rc == 0, L.oem_last_error()
        want = 0.75 + np.arange(n) / 1048576.0
        bad = np.nonzero(out != want)[0]
        assert len(bad) == 0, f"T={T}: {len(bad)} of {n} launches saw a stale maximum, first at launch {bad[:5]}: {out[bad[:5]]}"
        total += n
    assert total > 100_000


@pytest.mark.timeout(300)
def test_sharded_bootstrap_routes_every_replicate_the_same_way_on_all_ranks():
    """Row shards decide per replicate between the 8-slot batch and the one-per-pass path (a multiplicity
    >= 256 does not fit the batch's byte).  The decision selects which collectives run, so it must be taken
    together: here the oversized multiplicity sits in ONE shard only, and a second replicate makes one
    shard's whole block of reads vanish (multiplicity 0).  Two real shards as threads over the test-only
    library's process-local communicator; results against the un-sharded store and the oracle."""
    import ctypes as C
    import threading
    from oarfish_amd import _lib, dist as odist
    st = synth.make_store(50_000, 3_000, seed=733)
    rng = np.random.default_rng(2)
    n_boot, world = 10, 2
    W = np.stack([np.bincount(rng.integers(0, st.n_reads, st.n_reads), minlength=st.n_reads)
                  for _ in range(n_boot)]).astype(np.uint32)
    W[3, :] = 0
    W[3, 7] = 700                      # rank 0's shard overflows the byte, rank 1's does not
    W[6, st.n_reads // 2:] = 0         # (roughly) rank 1's reads all drop out of replicate 6
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, st.n_txps)
    with _lib.testing():
        handles = (C.c_void_p * world)()
        _lib.check(_lib.lib().oem_debug_local_comm_create(world, 0, C.addressof(handles)))
        res, errs = [None] * world, []

        def rank_main(rank):
            try:
                sh = odist.shard_rows_by_nnz(st.row_ptr, st.tid, st.as_prob, None, rank, world)
                with DeviceStore(sh.row_ptr, sh.tid, sh.as_prob, None, st.n_txps) as d:
                    d.attach_comm(C.c_void_p(handles[rank]), st.n_reads, sh.row_begin)
                    res[rank] = d.bootstrap(n_boot, row_w_all=np.ascontiguousarray(W[:, sh.row_begin:sh.row_end]),
                                            max_iter=150, conv_thresh=1e-3)
            except Exception as e:  # pragma: no cover
                errs.append((rank, repr(e)))

        th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=200)
        assert not errs and all(r is not None for r in res), errs
        for r in range(world):
            _lib.lib().oem_comm_destroy(C.c_void_p(handles[r]))
    assert np.array_equal(res[0][0], res[1][0])                     # every rank holds the same reduced counts
    for b in range(n_boot):
        want, wi = c_oracle.do_em(o, row_w=W[b], max_iter=150, conv_thresh=1e-3)
        gi = res[0][1][b]
        assert gi.niter == res[1][1][b].niter and abs(gi.niter - wi.niter) <= 1, (b, gi, wi)
        assert_counts_close(res[0][0][b], want, st.n_reads, st.n_txps, RTOL if gi.niter != wi.niter else 1e-8,
                            f"sharded replicate {b}")


# ---------------------------------------------------------------------------------------------------
# Size-independent properties of the E/M pass and the EM through the C ABI (the oracle is pinned by
# constructed vectors only, so these hold the device path to the algebra of em.rs:87-133 directly).
# ---------------------------------------------------------------------------------------------------
def _random_store(seed, R=30_000, T=2_500, maxk=12):
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, maxk + 1, size=R)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    nnz = int(rp[-1])
    tid = ((np.repeat(rng.integers(0, T, size=R), lens) + rng.integers(0, 40, size=nnz)) % T).astype(np.uint32)
    far = rng.random(nnz) < 0.15
    tid[far] = rng.integers(0, T, size=int(far.sum()))
    p = np.exp(-rng.integers(0, 30, size=nnz) / 5.0).astype(np.float32)
    return rp, tid, p, lens, rng


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_properties_of_one_pass(seed):
    rp, tid, p, lens, rng = _random_store(seed)
    R, T = len(lens), 2_500
    theta = rng.lognormal(0, 2, size=T)
    theta[rng.random(T) < 0.2] = 0.0
    with DeviceStore(rp, tid, p, None, T) as d:
        base = d.m_step(theta)
        # (1) an abundance of zero stays zero; the mass that is left is the reads with a positive denominator
        assert np.all(base[theta == 0.0] == 0.0)
        live = np.add.reduceat((theta[tid] * p.astype(np.float64)), rp[:-1].astype(np.int64)) > 1e-30
        assert abs(base.sum() - live.sum()) < 1e-9 * R
        # (2) linearity in the read multiplicities, and multiplicity 2 == the read stored twice
        w = rng.integers(0, 4, size=R).astype(np.uint32)
        assert_counts_close(d.m_step(theta, w) + d.m_step(theta, (3 - w).astype(np.uint32)), 3.0 * base, R, T, 1e-10, "linearity")
    # (3) the order of the reads does not matter (em.rs:97 sums over reads)
    perm = rng.permutation(R)
    starts = rp[:-1].astype(np.int64)[perm]
    idx = np.concatenate([np.arange(s, s + l) for s, l in zip(starts, lens[perm])])
    rp2 = np.concatenate([[0], np.cumsum(lens[perm])]).astype(np.uint64)
    with DeviceStore(rp2, tid[idx], p[idx], None, T) as d2:
        assert_counts_close(d2.m_step(theta), base, R, T, 1e-10, "row permutation")
    # (4) scaling a read's weights by a power of two changes nothing: x / denom is invariant (exactly, in f64)
    scale = np.repeat(2.0 ** rng.integers(-3, 1, size=R), lens).astype(np.float32)
    with DeviceStore(rp, tid, (p * scale).astype(np.float32), None, T) as d3:
        assert_counts_close(d3.m_step(theta), base, R, T, 1e-12, "row scaling")
    # (5) the store twice over == twice the counts
    rp4 = np.concatenate([rp, rp[1:] + rp[-1]]).astype(np.uint64)
    with DeviceStore(rp4, np.concatenate([tid, tid]), np.concatenate([p, p]), None, T) as d4:
        assert_counts_close(d4.m_step(theta), 2.0 * base, R, T, 1e-10, "duplicated store")


def test_em_fixed_point_is_idempotent_and_scale_free():
    """Running the EM from its own converged output moves nothing beyond the threshold, and the result does
    not depend on the scale of the initial abundances' common factor beyond fp (the E-step normalises)."""
    rp, tid, p, lens, rng = _random_store(11, R=40_000, T=1_500)
    R, T = len(lens), 1_500
    with DeviceStore(rp, tid, p, None, T) as d:
        a, ai = d.em_run(None, 2000, 1e-6, 50)
        b, bi = d.em_run(a, 2000, 1e-6, 50)
        assert ai.converged and bi.converged and bi.niter <= 52        # already at the fixed point: the gate only
        assert_counts_close(b, a, R, T, 1e-2, "restart from the fixed point")  # 52 more passes at rel-diff < 1e-6 each
        c, ci = d.em_run(a * 8.0, 2000, 1e-6, 50)                      # theta scaled by 8: same posteriors
        assert_counts_close(c, b, R, T, 1e-9, "scaled restart")
        assert abs(a.sum() - R) < 1e-7 * R


def test_hot_transcripts_do_not_break_the_layout():
    """Every read on the same three transcripts (64 lanes adding into one LDS entry) and a store with a
    single transcript: correctness at the worst case of the window atomics."""
    rng = np.random.default_rng(5)
    R = 200_000
    lens = rng.integers(1, 4, size=R)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    tid = np.concatenate([np.arange(l) for l in lens]).astype(np.uint32)
    p = np.exp(-rng.integers(0, 10, size=int(rp[-1])) / 5.0).astype(np.float32)
    o = c_oracle.Store(rp, tid, p, None, 3)
    want, wi = c_oracle.do_em(o, max_iter=200, conv_thresh=1e-3)
    with DeviceStore(rp, tid, p, None, 3) as d:
        got, gi = d.em_run(None, 200, 1e-3, 50)
        boot, binfo = d.bootstrap(9, seed=4, max_iter=100)
        assert np.all(np.abs(boot.sum(axis=1) - R) < 1e-6 * R)
    assert abs(gi.niter - wi.niter) <= 1
    assert_counts_close(got, want, R, 3, RTOL if gi.niter != wi.niter else 1e-9, "three hot transcripts")
    with DeviceStore(np.arange(1001, dtype=np.uint64), np.zeros(1000, np.uint32), np.ones(1000, np.float32), None, 1) as d:
        cnt, _ = d.em_run(None, 60, 1e-3, 50)
        assert cnt[0] == 1000.0


@pytest.mark.parametrize("coverage", [False, True])
def test_store_wider_than_the_packed_transcript_field(coverage):
    """A remote record packs (transcript - problem base) into 22 bits next to the read index
    (oem_layout_pack.hip); a store with more than 2^22 transcripts keeps (transcript u32, read u16) and only
    loses the slot stream -- the other instantiation of every tile kernel.  Point estimate, m-step with
    multiplicities, batched and one-per-pass bootstraps against the oracle, f32 and f64 weights."""
    from oarfish_amd import _lib
    T = (1 << 22) + 12_345
    st = synth.make_store(60_000, T, seed=909, coverage=coverage)
    assert st.tid.max() >= (1 << 22)
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, st.cov_prob, T)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, st.cov_prob, T) as d:
        rng = np.random.default_rng(4)
        theta = rng.uniform(0.0, 3.0, T)
        w = rng.integers(0, 4, st.n_reads).astype(np.uint32)
        assert_counts_close(d.m_step(theta, w), c_oracle.m_step(o, theta, row_w=w), st.n_reads, T, 1e-10, "m-step")
        got, gi = d.em_run(None, 80, 1e-3, 50)
        want, wi = c_oracle.do_em(o, max_iter=80, conv_thresh=1e-3)
        assert abs(gi.niter - wi.niter) <= 1
        assert_counts_close(got, want, st.n_reads, T, RTOL if gi.niter != wi.niter else 1e-9, "wide store")
        W = np.stack([d.bootstrap_weights(3, b) for b in range(3)])
        for batch in (1, 0):
            d.set_option(_lib.OEM_OPT_BATCH_BOOTSTRAP, batch)
            boots, binfo = d.bootstrap(3, row_w_all=W, max_iter=70)
            for b in range(3):
                wb, wbi = c_oracle.do_em(o, max_iter=70, conv_thresh=1e-3, row_w=W[b])
                assert abs(binfo[b].niter - wbi.niter) <= 1
                assert_counts_close(boots[b], wb, st.n_reads, T, RTOL if binfo[b].niter != wbi.niter else 1e-9,
                                    f"wide store, bootstrap {b}, batch={batch}")


@pytest.mark.parametrize("seed", range(10))
def test_batched_bootstrap_fuzz_random_shapes_match_oracle(seed):
    """The batch kernels (k_em_tile_e / k_remote_fold_b / k_reldiff_b, two chains) over randomly shaped stores:
    reads with up to 120 alignments (the register-resident eight per read overflow into the reload loops),
    tiles with more remote alignments than the 3 x 512 kept in registers (the queue-parking path), empty reads,
    1 .. 200 k transcripts, multiplicities up to 255 and all-zero resamples -- every replicate against the
    oracle's serial EM with the same multiplicities."""
    rng = np.random.default_rng(4200 + seed)
    R = int(rng.choice([1, 40, 3_000, 30_000, 80_000]))
    T = int(rng.choice([1, 6, 300, 4_000, 200_000]))
    maxk = int(rng.choice([1, 6, 20, 120]))
    lens = rng.integers(0, maxk + 1, size=R)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    nnz = int(rp[-1])
    spread = int(rng.choice([1, 10, 400, max(T, 1)]))
    tid = ((np.repeat(rng.integers(0, T, size=R), lens) + rng.integers(0, spread, size=nnz)) % T).astype(np.uint32)
    p = np.exp(-rng.integers(0, 40, size=nnz) / 5.0).astype(np.float32)
    cov = rng.uniform(1e-3, 1.0, size=nnz) if seed % 3 == 0 else None      # f64 weights: the coverage model's column
    n_boot = 11
    W = rng.poisson(1.0, size=(n_boot, R)).astype(np.uint32)
    W[2] = 0                                            # a resample that drew nothing (degenerate, but legal input)
    if R > 3:
        W[4, :3] = 255                                  # the largest multiplicity the batch takes
        W[7, 1] = 256                                   # one more: this replicate goes to the one-per-pass path
    o = c_oracle.Store(rp, tid, p, cov, T)
    with DeviceStore(rp, tid, p, cov, T) as d:
        out, infos = d.bootstrap(n_boot, row_w_all=W, max_iter=120, conv_thresh=1e-3)
    what = f"seed {seed}: R={R} T={T} maxk={maxk} spread={spread} cov={cov is not None}"
    for b in range(n_boot):
        want, wi = c_oracle.do_em(o, row_w=W[b], max_iter=120, conv_thresh=1e-3)
        assert abs(infos[b].niter - wi.niter) <= 1, (what, b, infos[b], wi)
        assert_counts_close(out[b], want, max(R, 1), T, RTOL if infos[b].niter != wi.niter else 1e-8, f"{what} replicate {b}")


def test_batched_bootstrap_with_mostly_remote_alignments():
    """40 alignments per read scattered over 100 k transcripts: ~40 k remote alignments per tile, far beyond
    the 1536 a workgroup of k_em_tile_e keeps in registers -- the bulk of them takes the path that parks
    theta * w in the queue between the two remote phases, for every slot of the batch."""
    rng = np.random.default_rng(77)
    R, T, k = 20_000, 100_000, 40
    rp = (np.arange(R + 1) * k).astype(np.uint64)
    tid = rng.integers(0, T, size=R * k).astype(np.uint32)
    tid = np.sort(tid.reshape(R, k), axis=1)
    tid += np.arange(k, dtype=np.uint32)[None, :] * 0          # (duplicates inside a read are legal input)
    tid = tid.reshape(-1)
    p = np.exp(-rng.integers(0, 25, size=R * k) / 5.0).astype(np.float32)
    W = rng.poisson(1.0, size=(6, R)).astype(np.uint32)
    o = c_oracle.Store(rp, tid, p, None, T)
    with DeviceStore(rp, tid, p, None, T) as d:
        out, infos = d.bootstrap(6, row_w_all=W, max_iter=80, conv_thresh=1e-3)
        point, pinfo = d.em_run(None, 80, 1e-3, 50)
    want, wi = c_oracle.do_em(o, max_iter=80, conv_thresh=1e-3)
    assert_counts_close(point, want, R, T, RTOL if pinfo.niter != wi.niter else 1e-8, "point estimate")
    for b in range(6):
        want, wi = c_oracle.do_em(o, row_w=W[b], max_iter=80, conv_thresh=1e-3)
        assert abs(infos[b].niter - wi.niter) <= 1
        assert_counts_close(out[b], want, R, T, RTOL if infos[b].niter != wi.niter else 1e-8, f"replicate {b}")


def test_host_store_uploads_again_when_the_reference_would_see_new_data():
    """The reference mutates the store between calls (normalize_read_probs fills coverage_probabilities,
    bulk.rs:103-108; model_coverage selects whether em reads them, em.rs:108).  The host mirror keeps one
    resident copy per store and must re-make it when either changes, and reject a bootstrap init vector of
    the wrong length before it reaches the device."""
    st = synth.make_store(30_000, 2_000, seed=91, coverage=True)
    store = InMemoryAlignmentStore.from_arrays(st.row_ptr, st.tid, st.as_prob, None, model_coverage=False)
    txps = [oarfish_amd.TranscriptInfo.with_len(1000)] * st.n_txps
    emi = oarfish_amd.EMInfo(eq_map=store, txp_info=txps, max_iter=80, convergence_thresh=0.0)
    plain = oarfish_amd.em(emi, 1)
    store.coverage_probabilities = st.cov_prob            # what normalize_read_probs assigns (:71)
    store.filter_opts.model_coverage = True
    with_cov = oarfish_amd.em(emi, 1)
    o_plain = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, st.n_txps)
    o_cov = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps)
    assert_counts_close(plain, c_oracle.do_em(o_plain, max_iter=80, conv_thresh=0.0)[0], st.n_reads, st.n_txps, 1e-8, "plain")
    assert_counts_close(with_cov, c_oracle.do_em(o_cov, max_iter=80, conv_thresh=0.0)[0], st.n_reads, st.n_txps, 1e-8, "coverage")
    assert np.max(np.abs(with_cov - plain)) > 1e-3        # the column really changed the answer
    store.filter_opts.model_coverage = False
    assert_counts_close(oarfish_amd.em(emi, 1), plain, st.n_reads, st.n_txps, 1e-9, "switched back")
    with pytest.raises(ValueError):
        store.device_store(st.n_txps).bootstrap(2, init=np.ones(st.n_txps - 1))
    store.invalidate_device()


@pytest.mark.parametrize("coverage", [False, True])
@pytest.mark.parametrize("tile_rows", [64, 256, 1024])
def test_reads_per_tile_do_not_change_the_answer(tile_rows, coverage):
    """Small stores are cut into 256-read tiles (oem_layout.h: tile_rows_for), large ones into 1024-read tiles --
    one slice per wavefront against four, with late slices and the hand-over of register sets.  The same 90 k-read
    store laid out with 64, 256 and 1024 reads per tile (OEM_TILE_ROWS, test-only library): one pass with
    multiplicities, the EM run, batched and row-weighted bootstraps against the oracle; the two builders agree."""
    import os
    from oarfish_amd import _lib
    st = synth.make_store(90_000, 7_000, seed=611, coverage=coverage)
    T = st.n_txps
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, st.cov_prob, T)
    rng = np.random.default_rng(tile_rows)
    theta = rng.lognormal(0, 1.5, T)
    w = rng.poisson(1.0, st.n_reads).astype(np.uint32)
    want_m = c_oracle.m_step(o, theta, row_w=w)
    want, wi = c_oracle.do_em(o, max_iter=120, conv_thresh=1e-3)
    want_w, _ = c_oracle.do_em(o, max_iter=50, conv_thresh=0.0, row_w=w)
    prev = os.environ.get("OEM_TILE_ROWS")
    os.environ["OEM_TILE_ROWS"] = str(tile_rows)
    try:
        with _lib.testing(), DeviceStore(st.row_ptr, st.tid, st.as_prob, st.cov_prob, T) as d:
            n_tiles = d.info(_lib.OEM_INFO_TILES)
            assert n_tiles >= (st.n_reads + tile_rows - 1) // tile_rows
            assert (n_tiles <= 2 * (st.n_reads // tile_rows + 1)) or tile_rows == 1024
            assert_counts_close(d.m_step(theta, w), want_m, st.n_reads, T, 1e-10, f"m-step, {tile_rows} reads per tile")
            got, gi = d.em_run(None, 120, 1e-3, 50)
            assert gi.niter == wi.niter
            assert_counts_close(got, want, st.n_reads, T, 1e-9, f"em, {tile_rows} reads per tile")
            got_w, _ = d.bootstrap(5, row_w_all=np.tile(w, (5, 1)), max_iter=50, conv_thresh=0.0)
            for b in range(5):
                assert_counts_close(got_w[b], want_w, st.n_reads, T, 1e-9, f"resampled {b}, {tile_rows} reads per tile")
        h_host = _layout_hash(st.row_ptr, st.tid, st.as_prob, st.cov_prob, T, host_build=True)
        h_dev = _layout_hash(st.row_ptr, st.tid, st.as_prob, st.cov_prob, T)
        diff = [f for f, a, b in zip(LAYOUT_FIELDS, h_dev, h_host) if a != b and f != "built_on_device"]
        assert not diff and h_host[0] == n_tiles, (diff, h_host[0], n_tiles)
    finally:
        if prev is None:
            os.environ.pop("OEM_TILE_ROWS", None)
        else:
            os.environ["OEM_TILE_ROWS"] = prev


@pytest.mark.parametrize("compact", [1, 0])
def test_sparse_cells_keep_only_the_transcripts_that_occur(compact, monkeypatch):
    """Per-cell transcript compaction (oem_api.hip: k_cells_mark / k_cells_rank): the batched store gives a cell the
    transcripts that occur in it, renumbered by rank, every cell as many ids as the fullest one; the results are
    expanded to [cell][transcript] on the way out.  Cells that express disjoint 2-20 % slices of a 9 000-transcript
    annotation, one cell without reads, one cell that touches everything: every cell against its own oracle run
    (the transcripts a cell does not name are exactly 0), with the compaction and with the plain c * T + t ids."""
    from oarfish_amd import _lib
    T = 9_000
    rng = np.random.default_rng(77)
    parts = []
    fracs = [0.02, 0.2, 0.05, None, 0.1, 1.0, 0.03]
    for c, f in enumerate(fracs):
        if f is None:                                    # a barcode without reads
            parts.append((np.zeros(1, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.float32)))
            continue
        n_sub = max(int(T * f), 8)
        sub = np.sort(rng.choice(T, n_sub, replace=False)).astype(np.uint32)
        st = synth.make_store(2_000 + 500 * c, n_sub, seed=700 + c, threads=1)
        parts.append((st.row_ptr, sub[st.tid], st.as_prob))
    cell_off = np.zeros(len(parts) + 1, np.uint64)
    rps, base = [np.zeros(1, np.uint64)], 0
    for c, (rp, _t, _p) in enumerate(parts):
        rps.append(rp[1:] + np.uint64(base))
        base += int(rp[-1])
        cell_off[c + 1] = cell_off[c] + np.uint64(len(rp) - 1)
    row_ptr = np.concatenate(rps)
    tid = np.concatenate([t for _r, t, _p in parts])
    p = np.concatenate([q for _r, _t, q in parts])
    monkeypatch.setenv("OEM_CELLS_COMPACT_TXPS", str(compact))
    with _lib.testing():
        out, infos = oarfish_amd.em_cells(cell_off, row_ptr, tid, p, None, T, max_iter=200, convergence_thresh=1e-3)
    assert out.shape == (len(parts), T)
    for c, (rp, t, q) in enumerate(parts):
        n = len(rp) - 1
        if n == 0:
            assert not out[c].any()
            continue
        o = c_oracle.Store(rp, t, q, None, T)
        want, wi = c_oracle.do_em(o, max_iter=200, conv_thresh=1e-3, min_iter_gate=50)
        absent = np.ones(T, bool)
        absent[t] = False
        assert not out[c][absent].any() and not want[absent].any()
        assert abs(infos[c].niter - wi.niter) <= 1
        assert_counts_close(out[c], want, n, T, RTOL if infos[c].niter != wi.niter else 1e-8, f"cell {c}")


@pytest.mark.parametrize("T,n_cells,reads,kmax,hot", [(1, 3, 200, 1, False), (3, 2, 500, 3, False), (70, 5, 1000, 6, False),
                                                      (5000, 4, 3000, 8, True), (4097, 3, 2000, 5, False),
                                                      (20000, 40, 300, 4, False)])
def test_cells_of_odd_shapes_through_the_compacted_store(T, n_cells, reads, kmax, hot):
    """One transcript, three transcripts, every read of a cell on the same two transcripts of 5 000, a transcript
    count one past a bucket, forty small cells: the ranked transcript ids of the batched store (a cell keeps what
    occurs in it) against the per-cell oracle."""
    rng = np.random.default_rng(T * 31 + n_cells)
    pool = min(T, 2) if hot else T
    lens = np.minimum(rng.integers(1, kmax + 1, size=n_cells * reads), pool)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    tid = np.concatenate([np.sort(rng.choice(pool, size=int(l), replace=False)) for l in lens]).astype(np.uint32)
    p = np.exp(-rng.integers(0, 20, size=len(tid)) / 5.0).astype(np.float32)
    cell_off = (np.arange(n_cells + 1) * reads).astype(np.uint64)
    out, infos = oarfish_amd.em_cells(cell_off, rp, tid, p, None, T, max_iter=100, convergence_thresh=1e-3)
    for c in range(n_cells):
        r0, r1 = int(cell_off[c]), int(cell_off[c + 1])
        a0, a1 = int(rp[r0]), int(rp[r1])
        o = c_oracle.Store(rp[r0:r1 + 1] - rp[r0], tid[a0:a1], p[a0:a1], None, T)
        want, wi = c_oracle.do_em(o, max_iter=100, conv_thresh=1e-3, min_iter_gate=50)
        assert abs(infos[c].niter - wi.niter) <= 1
        assert_counts_close(out[c], want, reads, T, RTOL if infos[c].niter != wi.niter else 1e-8, f"T={T} cell {c}")


@pytest.mark.parametrize("far", ["paralog", "paralog_adjacent"])
def test_recurring_far_alignments_match_oracle(far):
    """Far alignments that RECUR (a read's hits outside its gene go to the other genes of its paralog family, not
    anywhere): the queue ranges of the fold kernels are then runs of equal destinations, which a wavefront sums across
    its lanes before it touches the LDS (oem_lane_runs.h) -- in k_remote_fold, k_remote_fold_b (batched bootstrap) and
    k_multi_fold_reldiff (per-cell batch).  Hot families: 300 k reads over 3 000 transcripts."""
    st = synth.make_store(300_000, 3_000, seed=23, far=far)
    T = st.n_txps
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, T)
    theta = np.random.default_rng(6).lognormal(0, 1.5, T)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, T) as d:
        n_remote = d.info(_lib.OEM_INFO_REMOTE_ALIGNMENTS)
        assert (n_remote == 0) == (far == "paralog_adjacent") or T < 600, n_remote
        m = d.m_step(theta)
        cnt, info = d.em_run(None, 200, 1e-3, 50)
        W = np.stack([d.bootstrap_weights(9, 0), d.bootstrap_weights(9, 1), np.ones(st.n_reads, dtype=np.uint32)])
        bout, binfo = d.bootstrap(3, row_w_all=W, max_iter=60, conv_thresh=1e-3)
    assert_counts_close(m, c_oracle.m_step(o, theta), st.n_reads, T, 1e-10, f"m_step, {far}")
    want, wi = c_oracle.do_em(o, max_iter=200, conv_thresh=1e-3)
    assert info.niter == wi.niter
    assert_counts_close(cnt, want, st.n_reads, T, 1e-9, f"em, {far}")
    for b in range(3):
        wb, wbi = c_oracle.do_em(o, row_w=W[b], max_iter=60, conv_thresh=1e-3)
        assert binfo[b].niter == wbi.niter
        assert_counts_close(bout[b], wb, st.n_reads, T, 1e-9, f"batched bootstrap {b}, {far}")
    # the per-cell batch: 6 cells of the same shape
    cells = [synth.make_store(40_000, 3_000, seed=100 + c, far=far) for c in range(6)]
    cell_off = np.concatenate([[0], np.cumsum([c.n_reads for c in cells])]).astype(np.uint64)
    row_ptr = np.concatenate([[0]] + [c.row_ptr[1:] + sum(x.nnz for x in cells[:i]) for i, c in enumerate(cells)]).astype(np.uint64)
    tid = np.concatenate([c.tid for c in cells]); p = np.concatenate([c.as_prob for c in cells])
    out, infos = oarfish_amd.em_cells(cell_off, row_ptr, tid, p, None, T, max_iter=120, convergence_thresh=1e-3)
    for c, cs in enumerate(cells):
        wc, wci = c_oracle.do_em(c_oracle.Store(cs.row_ptr, cs.tid, cs.as_prob, None, T), max_iter=120, conv_thresh=1e-3)
        assert infos[c].niter == wci.niter
        assert_counts_close(out[c], wc, cs.n_reads, T, 1e-9, f"cell {c}, {far}")


@pytest.mark.parametrize("knob", ["OEM_TEST_FAIL_RANK_ALLOC", "OEM_TEST_FAIL_FULL_ALLOC"])
def test_cells_survive_a_failed_allocation_of_the_compaction_buffers(knob, monkeypatch):
    """The per-cell transcript compaction needs a rank table (cells x transcripts u32) and, on the way out, a buffer for
    the expanded results.  Without room for the first the batch keeps every cell's full id range; without room for the
    second the host expands the compact results (test-only library: the knobs make the allocations fail)."""
    n_cells, T = 7, 900
    cell_off, row_ptr, tid, p = synth.make_cells(n_cells, 3_000, T, seed=31, expressed_frac=0.2)
    want, winfos = oarfish_amd.em_cells(cell_off, row_ptr, tid, p, None, T, max_iter=200, convergence_thresh=1e-3)
    monkeypatch.setenv(knob, "1")
    with _lib.testing():
        got, infos = oarfish_amd.em_cells(cell_off, row_ptr, tid, p, None, T, max_iter=200, convergence_thresh=1e-3)
    for c in range(n_cells):
        assert infos[c].niter == winfos[c].niter
        assert_counts_close(got[c], want[c], int(cell_off[c + 1] - cell_off[c]), T, 1e-9, f"cell {c}, {knob}")
    o = c_oracle.Store(row_ptr[:int(cell_off[1]) + 1], tid[:int(row_ptr[int(cell_off[1])])], p[:int(row_ptr[int(cell_off[1])])], None, T)
    w0, _ = c_oracle.do_em(o, max_iter=200, conv_thresh=1e-3)
    assert_counts_close(got[0], w0, int(cell_off[1]), T, 1e-9, "cell 0 vs the oracle")


@pytest.mark.parametrize("far", ["uniform", "paralog_adjacent"])
def test_stopping_rule_one_pass_behind_stops_where_the_reference_stops(far, monkeypatch):
    """run_em_deferred (oem_em_driver.hip): the rel-diff of iteration i rides on pass i + 1 (three count vectors rotate,
    the fold's first workgroup -- or, for a store without remote alignments, a one-wavefront kernel -- applies
    em.rs:212-218), so a run is the reference's iterations plus one speculative pass.  Iteration counts, rel_diff and
    counts against the oracle and against the classic loop (sweep kernel after every pass) for both gates, caps that
    bite at every small max_iter, a caller's init vector, a threshold nothing meets."""
    st = synth.make_store(150_000, 6_000, seed=43, far=far)
    T = st.n_txps
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, T)
    init = np.random.default_rng(3).lognormal(0, 1.0, T) * st.n_reads / T
    cases = [(None, 1000, 1e-3, 50), (None, 1000, 1e-3, 1), (None, 1000, 1e-2, 1), (init, 300, 1e-3, 50), (None, 40, 0.0, 50)]
    cases += [(None, m, 1e-3, 1) for m in (0, 1, 2, 3, 4, 5)] + [(None, 52, 1e-1, 50), (None, 53, 1e-1, 50)]
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, T) as d:
        assert (d.info(_lib.OEM_INFO_REMOTE_ALIGNMENTS) == 0) == (far == "paralog_adjacent")
        got = [d.em_run(i, m, th, g) for i, m, th, g in cases]
        again = d.em_run(None, 1000, 1e-3, 50)          # the vectors rotate between runs: a second run from scratch
    monkeypatch.setenv("OEM_DEFERRED_RELDIFF", "0")
    with _lib.testing(), DeviceStore(st.row_ptr, st.tid, st.as_prob, None, T) as d:
        classic = [d.em_run(i, m, th, g) for i, m, th, g in cases]
    for (i, m, th, g), (cnt, info), (ccnt, cinfo) in zip(cases, got, classic):
        want, wi = c_oracle.do_em(o, init=i, max_iter=m, conv_thresh=th, min_iter_gate=g)
        what = f"max_iter {m}, thresh {th}, gate {g}, init {'yes' if i is not None else 'no'}, {far}"
        assert (info.niter, info.n_passes, info.converged) == (wi.niter, wi.n_passes, wi.converged), what
        assert (cinfo.niter, cinfo.n_passes, cinfo.converged) == (wi.niter, wi.n_passes, wi.converged), what
        assert abs(info.rel_diff - wi.rel_diff) <= 1e-9 * max(abs(wi.rel_diff), 1e-12) + 1e-15, what
        assert_counts_close(cnt, want, st.n_reads, T, 1e-9, what)
        assert_counts_close(cnt, ccnt, st.n_reads, T, 1e-10, "one pass behind vs the classic loop, " + what)
    assert again[1].niter == got[0][1].niter
    assert_counts_close(again[0], got[0][0], st.n_reads, T, 1e-12, "second run")


def test_last_iteration_decided_by_the_sweep_on_a_wide_annotation():
    """k_deferred_sweep (oem_tile_kernels.hip) caps its grid at 256 workgroups x 1024 transcripts per trip: an annotation
    of 700 k transcripts takes several trips per workgroup.  Runs that end at max_iter -- the sweep decides, converged or
    not -- against the oracle: iteration count, convergence flag, rel_diff, counts."""
    st = synth.make_store(60_000, 700_000, 4.0, seed=77)
    T = st.n_txps
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, T)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, T) as d:
        for m, th, g in ((1, 1e-3, 1), (2, 0.5, 1), (7, 0.0, 50), (52, 1.0, 50), (60, 1e-3, 50)):
            cnt, info = d.em_run(None, m, th, g)
            want, wi = c_oracle.do_em(o, max_iter=m, conv_thresh=th, min_iter_gate=g)
            what = f"max_iter {m}, thresh {th}, gate {g}"
            assert (info.niter, info.n_passes, info.converged) == (wi.niter, wi.n_passes, wi.converged), what
            assert abs(info.rel_diff - wi.rel_diff) <= 1e-9 * max(abs(wi.rel_diff), 1e-12) + 1e-15, what
            assert_counts_close(cnt, want, st.n_reads, T, 1e-9, what)
