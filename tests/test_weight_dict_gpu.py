"""GPU tests of the dictionary-coded local weights (oem_layout_dict.hip): as_prob = exp((score - best) / D) with
integer scores (oarfish_types.rs:1100-1114) takes few distinct values; up to 128 of them are coded into the spare
bits of the window codes, up to 256 as one-byte indices, up to 1024 as 16-bit indices into a table of the f32
values -- lossless -- and anything else keeps the f32 stream."""
import numpy as np
import pytest

from oarfish_amd import _lib, synth
from oarfish_amd.types import DeviceStore
from oracle import c_oracle
from tests.common import assert_counts_close

pytestmark = pytest.mark.gpu


def _weights_with(n_distinct, size, rng):
    vals = ((1.0 + np.arange(n_distinct)) / (n_distinct + 3.0)).astype(np.float32)
    assert len(np.unique(vals)) == n_distinct
    p = vals[rng.integers(0, n_distinct, size=size)]
    p[:n_distinct] = vals            # every value occurs
    return p


def test_coded_and_plain_weights_give_the_same_answer_and_the_oracles():
    st = synth.make_store(120_000, 9_000, seed=41)
    T = st.n_txps
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, T)
    theta = np.random.default_rng(2).lognormal(0, 1.5, T)
    want_m = c_oracle.m_step(o, theta)
    want, wi = c_oracle.do_em(o, max_iter=300, conv_thresh=1e-3)
    res = {}
    for coding in (0, 1):
        with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, T, weight_coding=coding) as d:
            n = d.info(_lib.OEM_INFO_WEIGHT_DICT_ENTRIES)
            assert (n > 0) == (coding == 0) and n <= 1024
            assert d.info(_lib.OEM_INFO_TILES) > 0 and 0 < d.info(_lib.OEM_INFO_REMOTE_ALIGNMENTS) < st.nnz
            m = d.m_step(theta)
            cnt, info = d.em_run(None, 300, 1e-3, 50)
            boots, _ = d.bootstrap(2, seed=4, max_iter=40)
            res[coding] = (m, cnt, info.niter, boots)
        assert_counts_close(m, want_m, st.n_reads, T, 1e-11, f"m_step, coding {coding}")
        assert info.niter == wi.niter
        assert_counts_close(cnt, want, st.n_reads, T, 1e-9, f"em, coding {coding}")
    # the table holds the caller's f32 values bit for bit: the two layouts differ by summation order only
    assert_counts_close(res[0][0], res[1][0], st.n_reads, T, 1e-12, "coded vs plain, one pass")
    assert res[0][2] == res[1][2]
    assert_counts_close(res[0][3][0], res[1][3][0], st.n_reads, T, 1e-9, "coded vs plain, bootstrap replicate")


@pytest.mark.parametrize("n_distinct,coded", [(3, True), (127, True), (128, True), (255, True), (256, True), (600, True),
                                              (1023, True), (1024, False), (5000, False)])
def test_the_table_holds_at_most_1024_values_including_the_zero_of_the_padding(n_distinct, coded):
    """127 | 128 distinct values + the padding's 0.0: fused index | byte indices; 255 | 256: bytes | 16-bit indices;
    1023 | 1024: 16-bit indices | the f32 stream."""
    st = synth.make_store(40_000, 3_000, seed=7)
    rng = np.random.default_rng(n_distinct)
    p = _weights_with(n_distinct, st.nnz, rng)
    o = c_oracle.Store(st.row_ptr, st.tid, p, None, st.n_txps)
    want, _ = c_oracle.do_em(o, max_iter=40, conv_thresh=0.0)
    with DeviceStore(st.row_ptr, st.tid, p, None, st.n_txps) as d:
        n = d.info(_lib.OEM_INFO_WEIGHT_DICT_ENTRIES)
        assert (n > 0) == coded, n
        if coded:
            assert n == n_distinct + 1     # + the 0.0 of the SELL padding, index 0
        got, _ = d.em_run(None, 40, 0.0, 50)
    assert_counts_close(got, want, st.n_reads, st.n_txps, 1e-9, f"{n_distinct} distinct weights")


def test_continuous_and_coverage_weights_keep_their_streams():
    st = synth.make_store(50_000, 4_000, seed=9, coverage=True)
    p = np.random.default_rng(5).uniform(1e-3, 1.0, st.nnz).astype(np.float32)
    for cov in (None, st.cov_prob):
        o = c_oracle.Store(st.row_ptr, st.tid, p if cov is None else st.as_prob, cov, st.n_txps)
        want, _ = c_oracle.do_em(o, max_iter=30, conv_thresh=0.0)
        with DeviceStore(st.row_ptr, st.tid, p if cov is None else st.as_prob, cov, st.n_txps) as d:
            assert d.info(_lib.OEM_INFO_WEIGHT_DICT_ENTRIES) == 0   # > 1024 distinct f32 values / f64 products
            got, _ = d.em_run(None, 30, 0.0, 50)
        assert_counts_close(got, want, st.n_reads, st.n_txps, 1e-9, "plain stream")


@pytest.mark.parametrize("n_values", [30, 200, 600])
def test_long_reads_take_the_reload_path_of_the_coded_weights(n_values):
    """Reads with more than 16 alignments inside one window: the coded weights of alignments 16.. are reloaded
    by both passes of the fold, rows and slices of every width up to 60 -- with 30 distinct weights (index fused
    into the window codes), with 200 (index bytes in their own stream) and with 600 (16-bit indices)."""
    rng = np.random.default_rng(12)
    R, T = 6_000, 900
    k = rng.integers(1, 61, size=R)
    k[:8] = 60
    rp = np.zeros(R + 1, dtype=np.uint64)
    rp[1:] = np.cumsum(k)
    base = rng.integers(0, T - 200, size=R)
    tid = np.concatenate([np.sort(rng.choice(200, size=int(kk), replace=False)) + b for kk, b in zip(k, base)]).astype(np.uint32)
    p = np.exp(-rng.integers(0, n_values, size=len(tid)) / 40.0).astype(np.float32)
    assert len(np.unique(p)) == n_values
    o = c_oracle.Store(rp, tid, p, None, T)
    want, wi = c_oracle.do_em(o, max_iter=60, conv_thresh=0.0)
    W = rng.poisson(1.0, size=R).astype(np.uint32)
    want_w, _ = c_oracle.do_em(o, max_iter=60, conv_thresh=0.0, row_w=W)
    for coding in (0, 1):
        with DeviceStore(rp, tid, p, None, T, weight_coding=coding) as d:
            assert (d.info(_lib.OEM_INFO_WEIGHT_DICT_ENTRIES) > 0) == (coding == 0)
            got, _ = d.em_run(None, 60, 0.0, 50)
            got_w, _ = d.bootstrap(1, row_w_all=W[None, :], max_iter=60, conv_thresh=0.0)
        assert_counts_close(got, want, R, T, 1e-9, f"long reads, coding {coding}")
        assert_counts_close(got_w[0], want_w, R, T, 1e-9, f"long reads resampled, coding {coding}")


def test_unknown_info_key_is_an_argument_error():
    st = synth.make_store(2_000, 300, seed=1)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
        with pytest.raises(Exception):
            d.info(99)


def test_a_negative_zero_weight_keeps_the_f32_stream():
    """-0.0 (or any pattern with the sign bit) is not ordered like its bits: such a store is not coded, and runs."""
    st = synth.make_store(8_000, 600, seed=3)
    p = st.as_prob.copy()
    p[5] = np.float32(-0.0)
    o = c_oracle.Store(st.row_ptr, st.tid, p, None, st.n_txps)
    want, _ = c_oracle.do_em(o, max_iter=20, conv_thresh=0.0)
    with DeviceStore(st.row_ptr, st.tid, p, None, st.n_txps) as d:
        assert d.info(_lib.OEM_INFO_WEIGHT_DICT_ENTRIES) == 0
        got, _ = d.em_run(None, 20, 0.0, 50)
    assert_counts_close(got, want, st.n_reads, st.n_txps, 1e-9, "store with a -0.0 weight")


def test_uniform_score_gaps_take_the_16_bit_indices():
    """The long-read variant of the generator (deficits uniform on [0, 0.05 best], best score up to 20 000:
    what oarfish_types.rs:1107-1118 admits) has a few hundred distinct weights: 16-bit indices, same answer as the
    f32 stream and the oracle -- point estimate and a resampled run."""
    st = synth.make_store(150_000, 9_000, seed=43, gaps="uniform")
    T = st.n_txps
    n_vals = len(np.unique(st.as_prob))
    assert 256 < n_vals <= 1023, n_vals
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, None, T)
    theta = np.random.default_rng(3).lognormal(0, 1.5, T)
    want_m = c_oracle.m_step(o, theta)
    want, wi = c_oracle.do_em(o, max_iter=200, conv_thresh=1e-3)
    W = np.random.default_rng(4).poisson(1.0, size=st.n_reads).astype(np.uint32)
    want_w, _ = c_oracle.do_em(o, max_iter=60, conv_thresh=0.0, row_w=W)
    ms = []
    for coding in (0, 1):
        with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, T, weight_coding=coding) as d:
            n = d.info(_lib.OEM_INFO_WEIGHT_DICT_ENTRIES)
            assert n == (n_vals + (0 if (st.as_prob == 0).any() else 1) if coding == 0 else 0), n
            m = d.m_step(theta)
            cnt, info = d.em_run(None, 200, 1e-3, 50)
            got_w, _ = d.bootstrap(1, row_w_all=W[None, :], max_iter=60, conv_thresh=0.0)
        ms.append(m)
        assert_counts_close(m, want_m, st.n_reads, T, 1e-11, f"m_step, coding {coding}")
        assert info.niter == wi.niter
        assert_counts_close(cnt, want, st.n_reads, T, 1e-9, f"em, coding {coding}")
        assert_counts_close(got_w[0], want_w, st.n_reads, T, 1e-9, f"resampled, coding {coding}")
    assert_counts_close(ms[0], ms[1], st.n_reads, T, 1e-12, "16-bit indices vs the f32 stream, one pass")
