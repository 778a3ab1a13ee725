"""The store builder: AlignmentFilters::filter + add_filtered_group
(src/util/oarfish_types.rs:955-1130, :718-738) as oem_builder_* in the C ABI, against a pure-Python
restatement.  Integer work and f32 probabilities: bit-exact.  Host-only; the last test uploads the
built store and checks the EM on it."""
import ctypes as C

import numpy as np
import pytest

from oarfish_amd import _lib
from oracle import filter_py as fp


def _builder(F: fp.Filters, txp_len):
    L = _lib.lib()
    fc = _lib.FiltersC(F.five_prime_clip, F.three_prime_clip, F.score_threshold, F.min_aligned_fraction,
                       F.min_aligned_len, F.which_strand, F.score_prob_denom, 0)
    tl = np.ascontiguousarray(txp_len, dtype=np.uint64)
    h = C.c_void_p()
    _lib.check(L.oem_builder_create(C.addressof(fc), tl.ctypes.data, len(tl), C.byref(h)))
    return h


def _add(h, group):
    L = _lib.lib()
    arr = (_lib.AlnRecordC * max(len(group), 1))()
    for i, x in enumerate(group):
        flags = (_lib.REC_UNMAPPED if x.unmapped else 0) | (_lib.REC_REVERSE if x.reverse else 0) | \
                (_lib.REC_SUPPLEMENTARY if x.supp else 0) | (_lib.REC_HAS_SCORE if x.score is not None else 0)
        arr[i] = _lib.AlnRecordC(x.ref_id, x.aln_start, x.aln_end, x.aln_span, x.score if x.score is not None else 0,
                                 x.seq_len if x.seq_len is not None else -1, flags, 0)
    kept = C.c_uint32(0)
    _lib.check(L.oem_builder_add_group(h, C.addressof(arr), len(group), C.byref(kept)))
    return kept.value


def _export(h):
    L = _lib.lib()
    R, nnz = C.c_uint64(0), C.c_uint64(0)
    _lib.check(L.oem_builder_dims(h, C.byref(R), C.byref(nnz)))
    rp = np.zeros(R.value + 1, dtype=np.uint64)
    tid = np.zeros(nnz.value, dtype=np.uint32)
    p = np.zeros(nnz.value, dtype=np.float32)
    s = np.zeros(nnz.value, dtype=np.uint32)
    e = np.zeros(nnz.value, dtype=np.uint32)
    sd = np.zeros(nnz.value, dtype=np.uint8)
    _lib.check(L.oem_builder_export(h, rp.ctypes.data, tid.ctypes.data, p.ctypes.data, s.ctypes.data,
                                    e.ctypes.data, sd.ctypes.data))
    dt = _lib.DiscardTableC()
    _lib.check(L.oem_builder_discard_table(h, C.addressof(dt)))
    return rp, tid, p, s, e, sd, {n: getattr(dt, n) for n, _ in _lib.DiscardTableC._fields_}


def test_hand_checked_read():
    """Best score 1000, D = 5, threshold 0.95: scores 1000 / 990 / 960 / 940 -> p = 1, e^-2, e^-8, dropped."""
    F = fp.Filters()
    h = _builder(F, [2000] * 4)
    g = [fp.Rec(0, 10, 1500, 1400, 1000, 1500), fp.Rec(1, 10, 1500, 1400, 990), fp.Rec(2, 5, 1400, 1400, 960),
         fp.Rec(3, 5, 1400, 1400, 940)]
    assert _add(h, g) == 3
    rp, tid, p, s, e, sd, dt = _export(h)
    assert list(rp) == [0, 3] and list(tid) == [0, 1, 2]
    np.testing.assert_allclose(p, [1.0, np.exp(-2.0), np.exp(-8.0)], rtol=3e-7)
    assert dt["discard_score"] == 1 and dt["valid_best_aln"] == 1
    # a read whose best alignment covers too little of it is dropped whole (:1084-1089)
    assert _add(h, [fp.Rec(0, 10, 900, 300, 500, 1500)]) == 0
    # unmapped only -> no_mapping; mapped but non-positive best score -> no_valid_aln
    assert _add(h, [fp.Rec(0, 0, 0, 0, None, 100, unmapped=True)]) == 0
    assert _add(h, [fp.Rec(0, 10, 900, 800, 0, 900)]) == 0
    rp, tid, p, s, e, sd, dt = _export(h)
    assert list(rp) == [0, 3] and dt["discard_aln_frac"] == 1 and dt["no_mapping"] == 1 and dt["no_valid_aln"] == 1
    _lib.lib().oem_builder_destroy(h)


def _builder_matches_python_restatement(seed):
    rng = np.random.default_rng(seed)
    T = 50
    txp_len = rng.integers(300, 4000, size=T)
    F = fp.Filters(five_prime_clip=int(rng.choice([2 ** 32 - 1, 400])), three_prime_clip=int(rng.choice([2 ** 62, 600])),
                   score_threshold=float(rng.choice([0.95, 0.9])), min_aligned_fraction=float(rng.choice([0.5, 0.7])),
                   min_aligned_len=int(rng.choice([50, 200])), which_strand=int(rng.integers(0, 3)),
                   score_prob_denom=float(rng.choice([5.0, 2.5])))
    h = _builder(F, txp_len)
    ref = fp.Store()
    for _ in range(800):
        n = int(rng.integers(0, 7))
        read_len = int(rng.integers(200, 3000))
        best = int(rng.integers(-5, 3000))
        g = []
        for j in range(n):
            t = int(rng.integers(0, T))
            span = int(rng.integers(20, read_len + 1))
            start = int(rng.integers(0, max(1, int(txp_len[t]) - 10)))
            sc = None if rng.random() < 0.03 else int(best - rng.integers(0, max(1, abs(best) // 8 + 2)))
            g.append(fp.Rec(t, start, start + span, span, sc, read_len if (j == 0 or rng.random() < 0.5) else None,
                            unmapped=rng.random() < 0.05, reverse=rng.random() < 0.3, supp=rng.random() < 0.05))
        assert _add(h, g) == fp.add_group(ref, F, txp_len, g)
    rp, tid, p, s, e, sd, dt = _export(h)
    assert list(rp) == ref.row_ptr and list(tid) == ref.tid and list(s) == ref.start and list(e) == ref.end
    assert list(sd) == ref.strand and dt == ref.dt
    assert np.array_equal(p.view(np.uint32), np.asarray(ref.as_prob, dtype=np.float32).view(np.uint32))  # bit-exact f32
    assert len(rp) - 1 > 20 and dt["discard_score"] + dt["discard_3p"] + dt["discard_5p"] > 0 and dt["no_valid_aln"] > 0
    _lib.lib().oem_builder_destroy(h)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_builder_matches_python_restatement(seed):
    _builder_matches_python_restatement(seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_builder_matches_python_restatement_on_the_gpu_box(seed):
    """The same bit-exact f32 / discard-table comparison of oem_builder_add_group with
    oracle/filter_py.add_group (oarfish_types.rs:955-1130), run by the driver's GPU tier as well:
    host code, but it is the library build the GPU box loads that is being checked."""
    _builder_matches_python_restatement(seed)


@pytest.mark.gpu
def test_built_store_runs_the_em():
    from oracle import c_oracle
    rng = np.random.default_rng(9)
    T = 300
    txp_len = rng.integers(800, 4000, size=T)
    F = fp.Filters()
    h = _builder(F, txp_len)
    for _ in range(5000):
        t0 = int(rng.integers(0, T))
        read_len = 1000
        g = [fp.Rec(int((t0 + j) % T), 5, 905, 900, 1800 - int(rng.integers(0, 60)) * (j > 0), read_len) for j in range(int(rng.integers(1, 5)))]
        _add(h, g)
    rp, tid, p, *_ = _export(h)
    hs = C.c_void_p()
    _lib.check(_lib.lib().oem_builder_store_create(h, None, 0, None, C.byref(hs)))
    out = np.zeros(T)
    ri = _lib.RunInfoC()
    _lib.check(_lib.lib().oem_em_run(hs, None, 200, 1e-3, 50, out.ctypes.data, C.byref(ri)))
    _lib.lib().oem_store_destroy(hs)
    _lib.lib().oem_builder_destroy(h)
    want, wi = c_oracle.do_em(c_oracle.Store(rp, tid, p, None, T), max_iter=200, conv_thresh=1e-3)
    assert abs(ri.niter - wi.niter) <= 1
    np.testing.assert_allclose(out, want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("seed,bin_width,growth", [(5, 100, 2.0), (6, 50, 1.0), (7, 200, 3.5)])
def test_coverage_model_matches_python_restatement(seed, bin_width, growth):
    """add_interval binning + logistic bin probabilities + per-read
    normalisation (oarfish_types.rs:496-538, logistic_probability.rs, normalize_probability.rs).
    f64 with the same operation order and the C library's exp: equal to 1e-15."""
    rng = np.random.default_rng(seed)
    T = 40
    txp_len = rng.integers(600, 5000, size=T)
    F = fp.Filters()
    h = _builder(F, txp_len)
    ref = fp.Store()
    for _ in range(1500):
        t0 = int(rng.integers(0, T))
        g = []
        for j in range(int(rng.integers(1, 5))):
            t = int((t0 + j) % T)
            L = int(txp_len[t])
            start = int(rng.integers(0, max(1, L - 120)))
            end = int(rng.integers(start + 100, L + 1)) if start + 100 <= L else L
            span = end - start
            g.append(fp.Rec(t, start, end, span, 2000 - int(rng.integers(0, 40)) * (j > 0), 1000 if span >= 600 else max(span, 1)))
        assert _add(h, g) == fp.add_group(ref, F, txp_len, g)
    rp, tid, p, s, e, sd, dt = _export(h)
    assert len(tid) > 1000
    cov = np.zeros(len(tid), dtype=np.float64)
    _lib.check(_lib.lib().oem_builder_coverage_probs(h, bin_width, growth, cov.ctypes.data))
    want = np.asarray(fp.coverage_probs(ref, txp_len, bin_width, growth))
    np.testing.assert_allclose(cov, want, rtol=1e-15, atol=0)
    sums = np.add.reduceat(cov, rp[:-1].astype(np.int64))
    np.testing.assert_allclose(sums, 1.0, rtol=1e-12)            # normalised per read
    assert cov.min() > 0 and len(np.unique(np.round(cov, 6))) > 50
    # bin width 0 is rejected as in the reference (unimplemented!)
    assert _lib.lib().oem_builder_coverage_probs(h, 0, growth, cov.ctypes.data) == _lib.OEM_ERR_ARG
    _lib.lib().oem_builder_destroy(h)


@pytest.mark.parametrize("seed,bin_width", [(15, 100), (16, 50), (17, 250)])
def test_binomial_coverage_model_matches_python_restatement(seed, bin_width):
    """The single-cell coverage model (binomial_continuous_prob, binomial_probability.rs:7-224, hook
    single_cell.rs:132-137): same bins and read normalisation, binomial bin probabilities with the
    reference's f32/f64 mix.  Same operation order, libm lgamma here, CPython's own Lanczos lgamma there (as statrs has its own): 1e-9."""
    rng = np.random.default_rng(seed)
    T = 30
    txp_len = rng.integers(400, 4000, size=T)
    F = fp.Filters()
    h = _builder(F, txp_len)
    ref = fp.Store()
    for _ in range(1200):
        t0 = int(rng.integers(0, T))
        g = []
        for j in range(int(rng.integers(1, 4))):
            t = int((t0 + j) % T)
            L = int(txp_len[t])
            start = int(rng.integers(0, max(1, L - 120)))
            end = int(rng.integers(start + 100, L + 1)) if start + 100 <= L else L
            span = end - start
            g.append(fp.Rec(t, start, end, span, 2000 - int(rng.integers(0, 40)) * (j > 0), 1000 if span >= 600 else max(span, 1)))
        assert _add(h, g) == fp.add_group(ref, F, txp_len, g)
    rp, tid, p, s, e, sd, dt = _export(h)
    cov = np.zeros(len(tid), dtype=np.float64)
    _lib.check(_lib.lib().oem_builder_coverage_probs_binomial(h, bin_width, cov.ctypes.data))
    want = np.asarray(fp.coverage_probs(ref, txp_len, bin_width, 0.0, model="binomial"))
    np.testing.assert_allclose(cov, want, rtol=1e-9, atol=1e-300)
    sums = np.add.reduceat(cov, rp[:-1].astype(np.int64))
    live = sums > 0
    np.testing.assert_allclose(sums[live], 1.0, rtol=1e-12)
    assert live.mean() > 0.9 and len(np.unique(np.round(cov, 6))) > 50
    assert _lib.lib().oem_builder_coverage_probs_binomial(h, 0, cov.ctypes.data) == _lib.OEM_ERR_ARG
    _lib.lib().oem_builder_destroy(h)


def test_binomial_probability_against_scipy():
    """Pins the restatement of binomial_probability.rs to an independent evaluation: after the 709
    rescale, bin i gets Binomial(n = sum of rescaled counts, p_i) pmf at its rescaled count
    (gamma-function form), normalised over the bins."""
    from scipy.special import gammaln
    rng = np.random.default_rng(3)
    cnt = [float(np.float32(x)) for x in rng.uniform(0.2, 40.0, size=17)]
    length = [100.0] * 16 + [37.0]
    rate = sum(c / l for c, l in zip(cnt, length))
    got = np.asarray(fp.binomial_probability(cnt, length, rate))
    c = np.asarray(cnt); l = np.asarray(length)
    p = c / (l * rate)
    m32 = (c * 709.0 / c.max()).astype(np.float32)                  # the reference holds these as f32
    n32 = np.float32(0.0)
    for v in m32:
        n32 = np.float32(n32 + v)
    m, n, rest = m32.astype(np.float64), float(n32), (n32 - m32).astype(np.float64)
    logpmf = gammaln(n + 1) - gammaln(m + 1) - gammaln(rest + 1) + m * np.log(p) + rest * np.log1p(-p)
    want = np.exp(logpmf); want /= want.sum()
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-300)
    # and the f32 roundings move the result by well under the 1e-4 budget of the abundances
    m, n = c * 709.0 / c.max(), (c * 709.0 / c.max()).sum()
    exact = np.exp(gammaln(n + 1) - gammaln(m + 1) - gammaln(n - m + 1) + m * np.log(p) + (n - m) * np.log1p(-p))
    np.testing.assert_allclose(got, exact / exact.sum(), rtol=5e-3)
    assert fp.binomial_probability([0.0, 0.0], [100.0, 100.0], 0.0) == [0.0, 0.0]


def _synthetic_coordinates(rng, n_reads, T):
    txp_len = rng.integers(400, 6000, size=T).astype(np.uint64)
    lens = rng.integers(1, 9, size=n_reads)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    hot = (rng.random(n_reads) < 0.3) * 0 + (rng.random(n_reads) >= 0.3) * rng.integers(0, T, size=n_reads)  # 30 % of reads on transcript 0
    tid = ((np.repeat(hot, lens) + np.concatenate([np.arange(k) for k in lens])) % T).astype(np.uint32)
    L = txp_len[tid].astype(np.int64)
    start = (rng.random(len(tid)) * np.maximum(L - 350, 1)).astype(np.int64)
    end = np.minimum(start + rng.integers(100, 3000, size=len(tid)), L)
    return txp_len, rp, tid, start.astype(np.uint32), end.astype(np.uint32)


@pytest.mark.gpu
@pytest.mark.parametrize("model,bin_width,growth", [(0, 100, 2.0), (0, 37, 0.7), (1, 100, 0.0), (1, 250, 0.0)])
def test_device_coverage_model_matches_host(model, bin_width, growth):
    """oem_coverage_probs_device against the host functions (same arithmetic; the bins are summed with
    atomics, so agreement is to the last bits, loosened where a bin count sits on an f32 rounding
    boundary), through a builder for the host side and raw arrays for the device side."""
    rng = np.random.default_rng(40 + model)
    T = 300
    txp_len, rp, tid, start, end = _synthetic_coordinates(rng, 60_000, T)
    nnz = len(tid)
    got = np.zeros(nnz)
    _lib.check(_lib.lib().oem_coverage_probs_device(rp.ctypes.data, tid.ctypes.data, start.ctypes.data, end.ctypes.data,
                                                    txp_len.ctypes.data, len(rp) - 1, nnz, T, bin_width, model, growth, 0,
                                                    got.ctypes.data))
    # host side: the same store through the Python restatement's data model is slow at this size; use the
    # C++ host functions via a builder filled without filtering
    F = fp.Filters()
    F.min_aligned_len, F.min_aligned_fraction, F.score_threshold = 0, 0.0, 0.0
    h = _builder(F, txp_len)
    recs = (_lib.AlnRecordC * 16)()
    kept = C.c_uint32(0)
    for r in range(len(rp) - 1):
        b, e = int(rp[r]), int(rp[r + 1])
        for q, j in enumerate(range(b, e)):
            recs[q].ref_id, recs[q].aln_start, recs[q].aln_end = int(tid[j]), int(start[j]), int(end[j])
            recs[q].aln_span, recs[q].score, recs[q].seq_len = int(end[j] - start[j]), 1000, int(end[j] - start[j])
            recs[q].flags = _lib.REC_HAS_SCORE
        _lib.check(_lib.lib().oem_builder_add_group(h, recs, e - b, C.byref(kept)))
        assert kept.value == e - b
    want = np.zeros(nnz)
    if model == 0:
        _lib.check(_lib.lib().oem_builder_coverage_probs(h, bin_width, growth, want.ctypes.data))
    else:
        _lib.check(_lib.lib().oem_builder_coverage_probs_binomial(h, bin_width, want.ctypes.data))
    again = np.zeros(nnz)
    _lib.check(_lib.lib().oem_builder_coverage_probs_device(h, bin_width, model, growth, 0, again.ctypes.data))
    _lib.lib().oem_builder_destroy(h)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-300)
    # typical agreement: rounding only (the binomial model exponentiates log-gammas of magnitude ~700)
    assert np.median(np.abs(got - want) / np.maximum(want, 1e-300)) < (1e-12 if model == 0 else 1e-10)
    np.testing.assert_allclose(again, got, rtol=2e-6, atol=1e-300)
    assert _lib.lib().oem_coverage_probs_device(rp.ctypes.data, tid.ctypes.data, start.ctypes.data, end.ctypes.data,
                                                txp_len.ctypes.data, len(rp) - 1, nnz, T, 0, model, growth, 0,
                                                got.ctypes.data) == _lib.OEM_ERR_ARG


@pytest.mark.gpu
def test_records_to_abundances_with_coverage_model():
    """End to end on the device: alignment records -> filters -> as_prob, coverage model -> cov_prob,
    upload, EM (the --model-coverage flow of bulk.rs:103-108,131-159)."""
    from oracle import c_oracle
    rng = np.random.default_rng(15)
    T = 120
    txp_len = rng.integers(900, 4000, size=T)
    F = fp.Filters()
    h = _builder(F, txp_len)
    for _ in range(8000):
        t0 = int(rng.integers(0, T))
        g = []
        for j in range(int(rng.integers(1, 4))):
            t = int((t0 + j) % T)
            L = int(txp_len[t])
            start = int(rng.integers(0, 60))
            end = int(rng.integers(L - 60, L + 1))
            g.append(fp.Rec(t, start, end, end - start, 1500 - int(rng.integers(0, 50)) * (j > 0), end - start))
        _add(h, g)
    rp, tid, p, *_ = _export(h)
    cov = np.zeros(len(tid))
    _lib.check(_lib.lib().oem_builder_coverage_probs(h, 100, 2.0, cov.ctypes.data))
    hs = C.c_void_p()
    _lib.check(_lib.lib().oem_builder_store_create(h, cov.ctypes.data, 0, None, C.byref(hs)))
    out = np.zeros(T)
    ri = _lib.RunInfoC()
    _lib.check(_lib.lib().oem_em_run(hs, None, 1000, 1e-3, 1, out.ctypes.data, C.byref(ri)))
    _lib.lib().oem_store_destroy(hs)
    _lib.lib().oem_builder_destroy(h)
    want, wi = c_oracle.do_em(c_oracle.Store(rp, tid, p, cov, T), max_iter=1000, conv_thresh=1e-3, min_iter_gate=1)
    assert abs(ri.niter - wi.niter) <= 1
    np.testing.assert_allclose(out, want, rtol=1e-4, atol=1e-6)
    assert abs(out.sum() - (len(rp) - 1)) < 1e-6 * len(rp)


@pytest.mark.gpu
@pytest.mark.parametrize("model,seed,bin_width,growth", [("logistic", 25, 100, 2.0), ("logistic", 26, 40, 0.8),
                                                          ("binomial", 27, 100, 0.0), ("binomial", 28, 230, 0.0)])
def test_device_coverage_model_matches_the_oracle(model, seed, bin_width, growth):
    """oem_coverage_probs_device (both coverage models, oem_coverage_device.hip) directly against the
    ORACLE (oracle/filter_py.coverage_probs: add_interval binning oarfish_types.rs:496-538, logistic_prob
    logistic_probability.rs:41-79, binomial_continuous_prob binomial_probability.rs:170-224,
    normalize_read_probs normalize_probability.rs:5-74) -- not against the library's own host functions.
    The device sums the bins with f64 atomics: equal to rounding, and to ~1e-7 where a bin count sits on
    an f32 rounding boundary (the reference truncates the counts to f32, oarfish_types.rs:478)."""
    rng = np.random.default_rng(seed)
    T = 35
    txp_len = rng.integers(500, 4500, size=T)
    F = fp.Filters()
    h = _builder(F, txp_len)
    ref = fp.Store()
    for _ in range(1400):
        t0 = int(rng.integers(0, T))
        g = []
        for j in range(int(rng.integers(1, 5))):
            t = int((t0 + j) % T)
            L = int(txp_len[t])
            start = int(rng.integers(0, max(1, L - 120)))
            end = int(rng.integers(start + 100, L + 1)) if start + 100 <= L else L
            span = end - start
            g.append(fp.Rec(t, start, end, span, 2000 - int(rng.integers(0, 40)) * (j > 0), 1000 if span >= 600 else max(span, 1)))
        assert _add(h, g) == fp.add_group(ref, F, txp_len, g)
    rp, tid, p, s, e, sd, dt = _export(h)
    nnz = len(tid)
    assert nnz > 1000
    want = np.asarray(fp.coverage_probs(ref, txp_len, bin_width, growth, model=model))
    m = 0 if model == "logistic" else 1
    # through the builder handle ...
    got = np.zeros(nnz)
    _lib.check(_lib.lib().oem_builder_coverage_probs_device(h, bin_width, m, growth, 0, got.ctypes.data))
    _lib.lib().oem_builder_destroy(h)
    # ... and from raw arrays (the exported CSR + coordinates)
    raw = np.zeros(nnz)
    tl = np.ascontiguousarray(txp_len, dtype=np.uint64)
    s32, e32 = np.ascontiguousarray(s, dtype=np.uint32), np.ascontiguousarray(e, dtype=np.uint32)
    rp64, tid32 = np.ascontiguousarray(rp, dtype=np.uint64), np.ascontiguousarray(tid, dtype=np.uint32)
    _lib.check(_lib.lib().oem_coverage_probs_device(rp64.ctypes.data, tid32.ctypes.data, s32.ctypes.data, e32.ctypes.data,
                                                    tl.ctypes.data, len(rp64) - 1, nnz, T, bin_width, m, growth, 0,
                                                    raw.ctypes.data))
    for what, out in (("builder handle", got), ("raw arrays", raw)):
        np.testing.assert_allclose(out, want, rtol=2e-6, atol=1e-300, err_msg=what)
        assert np.median(np.abs(out - want) / np.maximum(want, 1e-300)) < (1e-12 if m == 0 else 1e-9), what
    sums = np.add.reduceat(got, rp64[:-1].astype(np.int64))
    live = sums > 0
    np.testing.assert_allclose(sums[live], 1.0, rtol=1e-12)      # normalised per read
    assert live.mean() > 0.9


def _store_with_a_zero_span_alignment():
    """A read whose second alignment has start == end: total_weight = 0 and cov_prob = 0 in
    normalize_read_probs, so its expected value is 0/0 = NaN (normalize_probability.rs:58) -- which the
    reference does NOT treat as an error (only a non-finite cov_prob panics, :49-57)."""
    rng = np.random.default_rng(61)
    T = 12
    txp_len = rng.integers(900, 3000, size=T)
    F = fp.Filters()
    F.min_aligned_len, F.min_aligned_fraction = 0, 0.0
    h = _builder(F, txp_len)
    ref = fp.Store()
    for r in range(300):
        t0 = int(rng.integers(0, T))
        g = [fp.Rec(t0, 10, 700, 690, 1500, 700)]
        if r % 50 == 7:
            g.append(fp.Rec(int((t0 + 1) % T), 333, 333, 0, 1490, None))          # zero span, same bin
        else:
            g.append(fp.Rec(int((t0 + 1) % T), 20, 650, 630, 1480, None))
        assert _add(h, g) == fp.add_group(ref, F, txp_len, g) == 2
    return h, ref, txp_len, T


def test_zero_span_alignment_gives_nan_coverage_as_in_the_reference():
    h, ref, txp_len, T = _store_with_a_zero_span_alignment()
    rp, tid, p, s, e, sd, dt = _export(h)
    cov = np.zeros(len(tid))
    _lib.check(_lib.lib().oem_builder_coverage_probs(h, 100, 2.0, cov.ctypes.data))     # OEM_OK, not OEM_ERR_STATE
    _lib.lib().oem_builder_destroy(h)
    want = np.asarray(fp.coverage_probs(ref, txp_len, 100, 2.0))
    assert np.isnan(want).sum() == 6 and np.array_equal(np.isnan(cov), np.isnan(want))   # that alignment only; the row stays un-normalised (:62)
    ok = ~np.isnan(want)
    np.testing.assert_allclose(cov[ok], want[ok], rtol=1e-15)


@pytest.mark.gpu
def test_nan_coverage_reads_are_dropped_by_the_em_as_in_the_reference():
    """em.rs:115: a NaN denominator fails `denom > 1e-30`, the read contributes nothing.  Device coverage
    model -> NaN column -> store -> EM, against the oracle on the same column."""
    from oracle import c_oracle
    h, ref, txp_len, T = _store_with_a_zero_span_alignment()
    rp, tid, p, s, e, sd, dt = _export(h)
    cov = np.zeros(len(tid))
    _lib.check(_lib.lib().oem_builder_coverage_probs_device(h, 100, 0, 2.0, 0, cov.ctypes.data))
    want_cov = np.asarray(fp.coverage_probs(ref, txp_len, 100, 2.0))
    assert np.array_equal(np.isnan(cov), np.isnan(want_cov)) and np.isnan(cov).sum() == 6
    hs = C.c_void_p()
    _lib.check(_lib.lib().oem_builder_store_create(h, cov.ctypes.data, 0, None, C.byref(hs)))
    out = np.zeros(T)
    ri = _lib.RunInfoC()
    _lib.check(_lib.lib().oem_em_run(hs, None, 300, 1e-3, 50, out.ctypes.data, C.byref(ri)))
    _lib.lib().oem_store_destroy(hs)
    _lib.lib().oem_builder_destroy(h)
    want, wi = c_oracle.do_em(c_oracle.Store(rp, tid, p, cov, T), max_iter=300, conv_thresh=1e-3)
    assert np.all(np.isfinite(out)) and np.all(np.isfinite(want))
    assert abs(out.sum() - (300 - 6)) < 1e-9 * 300 and abs(want.sum() - (300 - 6)) < 1e-9 * 300   # six reads dropped
    assert abs(ri.niter - wi.niter) <= 1
    np.testing.assert_allclose(out, want, rtol=1e-8, atol=1e-9)
