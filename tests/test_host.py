"""CPU tests of the host-side mirror of the reference interface and the generator."""
import numpy as np
import pytest

import oarfish_amd
from oarfish_amd import synth
from oarfish_amd.types import InMemoryAlignmentStore


def test_store_mirror_semantics():
    st = InMemoryAlignmentStore()
    assert st.len() == 0 and st.total_len() == 0 and list(st.boundaries) == [0]  # oarfish_types.rs:645
    assert st.add_filtered_group([], []) is False                                 # :724,735-737
    assert st.add_filtered_group([3, 1], [1.0, 0.5]) is True
    assert st.add_filtered_group([2], [0.25]) is True
    assert st.len() == 2 and st.num_aligned_reads() == 2 and st.total_len() == 3
    rows = list(st.iter())
    assert [list(r[0]) for r in rows] == [[3, 1], [2]]
    assert rows[0][1].dtype == np.float32 and rows[0][2].dtype == np.float64
    assert np.all(st.coverage_probabilities == 0.0)                               # :731-732
    with pytest.raises(ValueError):
        st.add_filtered_group([1, 2], [1.0])
    with pytest.raises(ValueError):
        InMemoryAlignmentStore.from_arrays([1, 2], [0], [1.0])


def test_kde_is_rejected():
    st = InMemoryAlignmentStore.from_arrays([0, 1], [0], [1.0])
    emi = oarfish_amd.EMInfo(eq_map=st, txp_info=[oarfish_amd.TranscriptInfo()], kde_model=object())
    with pytest.raises(NotImplementedError):
        oarfish_amd.em(emi, 1)


def test_synth_is_deterministic_and_well_formed():
    a = synth.make_store(30_000, 2_000, seed=5, threads=1)
    b = synth.make_store(30_000, 2_000, seed=5, threads=4)
    assert np.array_equal(a.row_ptr, b.row_ptr) and np.array_equal(a.tid, b.tid)
    assert np.array_equal(a.as_prob, b.as_prob)
    c = synth.make_store(30_000, 2_000, seed=6)
    assert not np.array_equal(a.tid[:1000], c.tid[:1000])
    lens = np.diff(a.row_ptr.astype(np.int64))
    assert lens.min() >= 1 and lens.max() <= 100 and 7.5 < lens.mean() < 8.5
    assert a.tid.max() < 2_000 and a.as_prob.dtype == np.float32
    assert np.all(a.as_prob > 0) and np.all(a.as_prob <= 1.0)
    # targets are distinct within a read
    row = np.repeat(np.arange(a.n_reads), lens)
    key = row.astype(np.int64) * 2_000 + a.tid
    assert len(np.unique(key)) == len(key)
    # every read has exactly one best alignment with p = 1 (d = 0) unless ties
    assert np.all(np.maximum.reduceat(a.as_prob, a.row_ptr[:-1].astype(np.int64)) == 1.0)
    cv = synth.make_store(5_000, 500, seed=5, coverage=True)
    s = np.add.reduceat(cv.cov_prob, cv.row_ptr[:-1].astype(np.int64))
    np.testing.assert_allclose(s, 1.0, rtol=1e-12)


def test_long_read_score_gaps_of_the_generator():
    """synth.make_store(gaps="uniform"): deficits uniform on [0, 0.05 best] for best scores up to 20 000 -- what
    oarfish_types.rs:1107-1118 admits -- give a few hundred distinct f32 weights (exp(-gap / 5) reaches 0 near a gap
    of 520), more than the byte-coded weight table takes and within the 16-bit one; the default generator stays
    under 128.  Seeded: the same store twice."""
    import numpy as np
    from oarfish_amd import synth
    a = synth.make_store(60_000, 4_000, seed=5, gaps="uniform")
    b = synth.make_store(60_000, 4_000, seed=5, gaps="uniform")
    assert np.array_equal(a.tid, b.tid) and np.array_equal(a.as_prob, b.as_prob) and np.array_equal(a.row_ptr, b.row_ptr)
    n = len(np.unique(a.as_prob))
    assert 256 < n <= 1023, n
    assert a.as_prob.max() == 1.0 and a.as_prob.min() >= 0.0
    first = a.row_ptr[:-1].astype(np.int64)
    assert np.all(np.maximum.reduceat(a.as_prob, first) == 1.0)      # every read keeps its best alignment at weight 1
    g = synth.make_store(60_000, 4_000, seed=5)
    assert len(np.unique(g.as_prob)) <= 128
    import pytest
    with pytest.raises(ValueError):
        synth.make_store(100, 10, gaps="bimodal")
