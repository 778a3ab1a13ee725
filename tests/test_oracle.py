"""CPU tests pinning the oracle (oracle/oem_oracle.c, oracle/oracle_np.py).

The reference has no EM tests or golden vectors, so the
oracle is pinned against closed forms, invariants, the committed fixtures and
the two restatements against each other.
"""
import numpy as np
import pytest

from oracle import c_oracle, oracle_np
from tests.common import golden_names, load_golden


def _store(g):
    return c_oracle.Store(g["row_ptr"], g["tid"], g["as_prob"], g["cov_prob"], g["n_txps"])


@pytest.mark.parametrize("name", golden_names())
def test_c_oracle_matches_golden(name):
    g = load_golden(name)
    s = _store(g)
    for run in g["runs"]:
        cnt, info = c_oracle.do_em(s, init=g["init"], max_iter=run["max_iter"],
                                   conv_thresh=run["conv_thresh"], min_iter_gate=run["gate"])
        assert info.niter == run["niter"] and info.n_passes == run["n_passes"]
        assert info.converged == run["converged"]
        np.testing.assert_allclose(cnt, run["counts"], rtol=1e-9, atol=1e-9)
        assert abs(info.rel_diff - run["rel_diff"]) <= 1e-9 * max(1.0, abs(run["rel_diff"]))


@pytest.mark.parametrize("name", ["closed_form_two_txps", "unique_only", "single_read"])
def test_closed_forms(name):
    g = load_golden(name)
    for run in g["runs"]:
        np.testing.assert_allclose(run["counts"], g["closed_form"], rtol=1e-8, atol=1e-12)


def test_numpy_restatement_independent_agreement():
    rng = np.random.default_rng(99)
    from scripts.make_golden import random_store
    for cov in (False, True):
        row_ptr, tid, p, c = random_store(rng, 700, 50, coverage=cov)
        s = c_oracle.Store(row_ptr, tid, p, c, 50)
        for mi, ct, gate in [(30, 0.0, 50), (1000, 1e-3, 1), (1000, 1e-3, 50)]:
            a, info = c_oracle.do_em(s, max_iter=mi, conv_thresh=ct, min_iter_gate=gate)
            b, niter, npass, conv, rel = oracle_np.do_em(row_ptr, tid, p, c, 50, max_iter=mi,
                                                         conv_thresh=ct, min_iter_gate=gate)
            assert (info.niter, info.n_passes, info.converged) == (niter, npass, conv)
            np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-10)


def test_invariants():
    """SURVEY.md section 8c (2)."""
    rng = np.random.default_rng(5)
    from scripts.make_golden import random_store
    row_ptr, tid, p, _ = random_store(rng, 900, 70)
    s = c_oracle.Store(row_ptr, tid, p, None, 70)
    cnt, _ = c_oracle.do_em(s, max_iter=200, conv_thresh=1e-3)
    # mass conservation: every read has denom > 1e-30 here
    assert abs(cnt.sum() - 900) < 1e-8
    # unique <= count <= total (docs/index.md:325; aux_counts.rs)
    u, t = c_oracle.aux_counts(s)
    assert np.all(cnt >= u - 1e-9) and np.all(cnt <= t + 1e-9)
    # theta_t = 0 is absorbing
    init = np.full(70, 900 / 70.0)
    init[::3] = 0.0
    c0, _ = c_oracle.do_em(s, init=init, max_iter=50, conv_thresh=0.0)
    assert np.all(c0[::3] == 0.0)
    # scaling a read's weights by a constant changes nothing (up to rounding)
    p2 = p.copy()
    lens = np.diff(row_ptr.astype(np.int64))
    p2 *= np.repeat(np.where(np.arange(900) % 2 == 0, 0.5, 1.0), lens).astype(np.float32)
    s2 = c_oracle.Store(row_ptr, tid, p2, None, 70)
    c2, _ = c_oracle.do_em(s2, max_iter=40, conv_thresh=0.0)
    c1, _ = c_oracle.do_em(s, max_iter=40, conv_thresh=0.0)
    np.testing.assert_allclose(c2, c1, rtol=1e-9, atol=1e-12)
    # permuting reads changes nothing beyond summation order
    perm = rng.permutation(900)
    starts = row_ptr[:-1].astype(np.int64)
    idx = np.concatenate([np.arange(starts[i], starts[i] + lens[i]) for i in perm])
    rp = np.zeros(901, dtype=np.uint64)
    rp[1:] = np.cumsum(lens[perm])
    s3 = c_oracle.Store(rp, tid[idx], p[idx], None, 70)
    c3, i3 = c_oracle.do_em(s3, max_iter=40, conv_thresh=0.0)
    np.testing.assert_allclose(c3, c1, rtol=1e-9, atol=1e-12)


def test_denominator_threshold_drops_read():
    """em.rs:115: reads with denom <= 1e-30 contribute nothing."""
    row_ptr = np.array([0, 1, 2], dtype=np.uint64)
    tid = np.array([0, 1], dtype=np.uint32)
    p = np.array([1.0, 1e-38], dtype=np.float32)  # second read: theta*w ~ 1e-38 < 1e-30
    s = c_oracle.Store(row_ptr, tid, p, None, 2)
    cnt, _ = c_oracle.do_em(s, max_iter=5, conv_thresh=0.0)
    assert cnt[0] == 1.0 and cnt[1] == 0.0


def test_gate_and_iteration_accounting():
    g = load_golden("gate_cases")
    par, ser = g["runs"][0], g["runs"][1]
    assert par["converged"] and ser["converged"]
    assert par["niter"] < 51 and ser["niter"] == 51  # first niter > 50 is 51 (em.rs:212,218)
    assert ser["n_passes"] == 53                     # 52 loop passes + the final one
    # max_iter = 0: loop body never runs, one final pass (em.rs:181,245)
    s = _store(g)
    cnt, info = c_oracle.do_em(s, max_iter=0)
    assert info.niter == 0 and info.n_passes == 1 and not info.converged


def test_em_par_matches_serial():
    g = load_golden("random_a")
    s = _store(g)
    a, ia = c_oracle.do_em(s, max_iter=1000, conv_thresh=1e-3, min_iter_gate=1)
    b, ib = c_oracle.em_par(s, max_iter=1000, conv_thresh=1e-3, min_iter_gate=1, nthreads=4)
    assert ia.niter == ib.niter
    np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9)
    aos = c_oracle.make_aos(s)
    c, ic = c_oracle.em_aos(s, aos, max_iter=1000, conv_thresh=1e-3, min_iter_gate=1)
    assert ic.niter == ia.niter
    np.testing.assert_allclose(a, c, rtol=1e-12, atol=1e-12)
    d, idd = c_oracle.em_aos(s, aos, max_iter=1000, conv_thresh=1e-3, min_iter_gate=1, nthreads=4)
    np.testing.assert_allclose(a, d, rtol=1e-9, atol=1e-9)


def test_bootstrap_index_and_weight_forms_agree():
    g = load_golden("bootstrap_inject")
    s = _store(g)
    W = g["row_w"]
    for b in range(W.shape[0]):
        inds = np.repeat(np.arange(s.n_reads, dtype=np.uint64), W[b])
        a, _ = c_oracle.do_em(s, inds=inds, max_iter=1000, conv_thresh=1e-3)
        np.testing.assert_allclose(a, g["boot_counts"][b], rtol=1e-9, atol=1e-9)
    out, infos = c_oracle.bootstrap(s, W.shape[0], row_w_all=W, nthreads=2)
    np.testing.assert_allclose(out, g["boot_counts"], rtol=1e-9, atol=1e-9)


def test_sample_inds_distribution():
    """bootstrap.rs:7-16: n draws from Uniform[0,n), sorted."""
    n = 20000
    inds = c_oracle.get_sample_inds(n, 123)
    assert len(inds) == n and np.all(np.diff(inds.astype(np.int64)) >= 0)
    assert inds.min() >= 0 and inds.max() < n
    w = c_oracle.inds_to_weights(inds, n)
    assert w.sum() == n
    # multinomial(n; 1/n): mean 1, var 1 - 1/n, P(0) ~ e^-1
    assert abs(w.var() - 1.0) < 0.05
    assert abs((w == 0).mean() - np.exp(-1)) < 0.02
    assert not np.array_equal(inds, c_oracle.get_sample_inds(n, 124))
    assert np.array_equal(inds, c_oracle.get_sample_inds(n, 123))


def test_assignment_probs_closed_form():
    """write_function.rs:283-318 on a hand-checkable read."""
    rp = np.array([0, 3, 4], dtype=np.uint64)
    tid = np.array([0, 1, 2, 1], dtype=np.uint32)
    p = np.array([1.0, 1.0, 0.5, 1.0], dtype=np.float32)
    s = c_oracle.Store(rp, tid, p, None, 3)
    counts = np.array([6.0, 3.0, 2.0])          # read 0: 6, 3, 1 -> 0.6, 0.3, 0.1
    out = c_oracle.assignment_probs(s, counts, 0.2)
    np.testing.assert_allclose(out, [0.6 / 0.9, 0.3 / 0.9, -1.0, 1.0], rtol=1e-15)
    out0 = c_oracle.assignment_probs(s, counts, 0.0)
    np.testing.assert_allclose(out0, [0.6, 0.3, 0.1, 1.0], rtol=1e-15)


@pytest.mark.parametrize("tag,T", [("C", 69), ("I", 44), ("O", 100)])
def test_config0_sirv_shaped_store_cpu(tag, T):
    """BASELINE configs[0]: SIRV annotation, bulk mode, 100 EM iterations on the CPU path (plumbing):
    C restatement == NumPy restatement; mass conserved; unique <= count <= total."""
    from oarfish_amd import synth
    st = synth.make_sirv_store(tag, 20_000)
    assert st.n_txps == T and st.n_reads == 20_000 and np.diff(st.row_ptr.astype(np.int64)).min() >= 1
    s = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, T)
    a, ia = c_oracle.do_em(s, max_iter=100, conv_thresh=0.0)
    b, niter, npass, conv, rel = oracle_np.do_em(st.row_ptr, st.tid, st.as_prob, None, T, max_iter=100, conv_thresh=0.0)
    assert ia.niter == niter == 100 and ia.n_passes == npass == 101 and not conv
    np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-10)
    assert abs(a.sum() - st.n_reads) < 1e-8
    u, t = c_oracle.aux_counts(s)
    assert np.all(a >= u - 1e-9) and np.all(a <= t + 1e-9)
