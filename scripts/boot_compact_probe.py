#!/usr/bin/env python3
"""What a bootstrap replicate would cost on its OWN store: the reads its resample did not draw (1/e of them) dropped,
the others laid out afresh (tiles, dictionary-coded weights) with their multiplicities as row weights, and the
point-estimate kernels run over that.  Compared with the same replicate on the full store (multiplicity 0 rows kept)
and with the batched pass.  usage: boot_compact_probe.py [c3|c2] [iters]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _ab  # noqa: E401,E402,F401
from oarfish_amd import synth, _lib
from oarfish_amd.types import DeviceStore

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
st = synth.make_config(wl)
T = st.n_txps
with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, T) as d:
    W = d.bootstrap_weights(7, 0)
    d.bootstrap(1, row_w_all=W[None, :], max_iter=5, conv_thresh=0.0)
    t = time.perf_counter()
    full, info = d.bootstrap(1, row_w_all=W[None, :], max_iter=iters, conv_thresh=0.0)
    dt_full = time.perf_counter() - t
    print(f"full store, one replicate per pass: {dt_full / (iters + 1) * 1e3:.4f} ms per pass ({iters + 1} passes, H2D of the weights included)")
    ms, slots, _ = d.time_bootstrap_passes(10)
    print(f"batched pass: {ms:.4f} ms for {slots} = {ms / slots:.4f} ms per replicate-pass")
keep = W > 0
lens = (st.row_ptr[1:] - st.row_ptr[:-1]).astype(np.int64)
t0 = time.perf_counter()
rp = np.zeros(int(keep.sum()) + 1, dtype=np.uint64)
np.cumsum(lens[keep], out=rp[1:])
amask = np.repeat(keep, lens)
tid = st.tid[amask]
p = st.as_prob[amask]
Wc = W[keep]
print(f"kept {keep.sum()} of {len(keep)} reads ({keep.mean():.3f}), {len(tid)} of {st.nnz} alignments; host compaction {time.perf_counter() - t0:.2f} s")
t0 = time.perf_counter()
with DeviceStore(rp, tid, p, None, T) as c:
    print(f"store creation (upload + device layout + dictionary) {time.perf_counter() - t0:.3f} s, dict {c.info(_lib.OEM_INFO_WEIGHT_DICT_ENTRIES)}")
    # theta-init uses the store's own read count here (R' instead of R): timing only
    c.bootstrap(1, row_w_all=Wc[None, :], max_iter=5, conv_thresh=0.0)
    t = time.perf_counter()
    got, info = c.bootstrap(1, row_w_all=Wc[None, :], max_iter=iters, conv_thresh=0.0)
    dt = time.perf_counter() - t
    print(f"compacted store, one replicate per pass: {dt / (iters + 1) * 1e3:.4f} ms per pass")
    pm = c.time_m_step(50)
    print(f"compacted store, unweighted pass {pm:.4f} ms")
