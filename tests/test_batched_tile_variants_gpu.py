"""GPU tests of the variants of the batched bootstrap's tile kernel (k_em_tile_e, oem_batch_kernels.hip):
the weight table of a store with <= 128 distinct weights read from the spare bits of the window codes (kFused, what a
default store takes) against the f32 weight stream (oem_store_opts.weight_coding = 1).  Same semantics: em.rs:273-290 over resampled reads, per-replicate stopping."""
import numpy as np
import pytest

from oarfish_amd import _lib
from oarfish_amd.types import DeviceStore
from oracle import c_oracle
from tests.common import assert_counts_close, tile_test_store as _store

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["dense", "sparse", "long", "remote", "ragged"])
@pytest.mark.parametrize("variant", ["fused", "f32"])
def test_batched_replicates_match_the_oracle_on_every_kernel_variant(kind, variant, monkeypatch):
    row_ptr, tid, p, T = _store(kind, seed=23)
    n_reads = len(row_ptr) - 1
    rng = np.random.default_rng(5)
    n_rep = 5   # more replicates than slots: a slot is handed its next replicate while the others run on
    row_w = rng.multinomial(n_reads, np.full(n_reads, 1.0 / n_reads), size=n_rep).astype(np.uint32)
    o = c_oracle.Store(row_ptr, tid, p, None, T)
    wc = 1 if variant.startswith("f32") else 0
    with _lib.testing(), DeviceStore(row_ptr, tid, p, None, T, weight_coding=wc) as d:
        n_dict = d.info(_lib.OEM_INFO_WEIGHT_DICT_ENTRIES)
        got, infos = d.bootstrap(n_rep, seed=1, row_w_all=row_w, max_iter=60, conv_thresh=1e-3)
    assert (n_dict == 0) == (wc == 1), (variant, n_dict)
    if wc == 0:
        assert 0 < n_dict <= 128, n_dict   # the generator's weights: a few dozen distinct values -> the fused coding
    for b in range(n_rep):
        want, wi = c_oracle.do_em(o, max_iter=60, conv_thresh=1e-3, row_w=row_w[b])
        assert infos[b].niter == wi.niter, (kind, variant, b)
        assert_counts_close(got[b], want, n_reads, T, 1e-9, f"{kind}/{variant} replicate {b}")


def test_fused_and_streamed_weights_draw_and_estimate_the_same_replicates():
    """Device-drawn resamples (Philox, the same seed): the two codings are the same f32 values, so the replicates stop
    at the same iteration and agree to rounding -- through the product library."""
    row_ptr, tid, p, T = _store("dense", seed=31)
    n_reads = len(row_ptr) - 1
    res = []
    for wc in (0, 1):
        with DeviceStore(row_ptr, tid, p, None, T, weight_coding=wc) as d:
            res.append(d.bootstrap(6, seed=9, max_iter=80, conv_thresh=1e-3))
    for b in range(6):
        assert res[0][1][b].niter == res[1][1][b].niter
        assert_counts_close(res[0][0][b], res[1][0][b], n_reads, T, 1e-10, f"replicate {b}")
