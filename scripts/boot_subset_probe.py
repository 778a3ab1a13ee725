#!/usr/bin/env python3
"""What the reads a resample leaves out are worth (task 3).  random_sampling_iter visits only the
sampled rows (oarfish_types.rs:576-592); a device-drawn resample gives e^-1 = 36.8 % of the reads multiplicity 0 and the
batched kernel runs them all.  For a device-drawn C3 resample: a second store from the 63.2 % of the rows with
row_w > 0, (i) its point-estimate pass, alone and as an un-batched replicate with its multiplicities, against the full
store's; (ii) what creating such a store costs.  usage: boot_subset_probe.py [c3|c2]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _ab  # noqa: E401,E402,F401
from oarfish_amd import synth, _lib
from oarfish_amd.types import DeviceStore
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
st = synth.make_config(wl)
R, T = st.n_reads, st.n_txps


def unbatched(d, W, n_it=300):
    d.set_option(_lib.OEM_OPT_BATCH_BOOTSTRAP, 0)
    d.bootstrap(1, row_w_all=W[None, :], max_iter=20, conv_thresh=0.0)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        out, infos = d.bootstrap(1, row_w_all=W[None, :], max_iter=n_it, conv_thresh=0.0)
        best = min(best, time.perf_counter() - t)
    d.set_option(_lib.OEM_OPT_BATCH_BOOTSTRAP, 1)
    return best / (n_it + 1) * 1e3, out[0]


with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, T) as d:
    W = d.bootstrap_weights(0x7e57, 0)
    d.time_m_step(200)
    full_pass = min(d.time_m_step(100) for _ in range(3))
    bp = d.time_bootstrap_passes(40)
    full_unb, full_counts = unbatched(d, W)
    print(f"{wl} full store: {R} reads, pass {full_pass:.4f} ms; batched pass {bp[0]:.4f} ms = {bp[0] / bp[1] * 1e3:.1f} us per "
          f"replicate-pass; un-batched replicate with multiplicities {full_unb * 1e3:.1f} us per pass (host time of the run / passes)")
keep = W > 0
print(f"resample: {int(keep.sum())} of {R} reads have multiplicity > 0 ({keep.mean():.4f}); max multiplicity {int(W.max())}")
t = time.perf_counter()
lens = np.diff(st.row_ptr).astype(np.int64)
kl = lens[keep]
rp = np.zeros(kl.size + 1, np.uint64); rp[1:] = np.cumsum(kl)
starts = st.row_ptr[:-1][keep].astype(np.int64)
idx = np.repeat(starts - rp[:-1].astype(np.int64), kl) + np.arange(int(rp[-1]), dtype=np.int64)
tid2, p2 = st.tid[idx], st.as_prob[idx]
W2 = W[keep]
print(f"host subset of the CSR: {time.perf_counter() - t:.2f} s (numpy; a device compaction is a prefix sum + gather)")
t = time.perf_counter()
d2 = DeviceStore(rp, tid2, p2, None, T)
t_create = time.perf_counter() - t
with d2:
    d2.time_m_step(200)
    sub_pass = min(d2.time_m_step(100) for _ in range(3))
    bp2 = d2.time_bootstrap_passes(40)
    sub_unb, sub_counts = unbatched(d2, W2)
    tiles = d2.info(_lib.OEM_INFO_TILES)
    print(f"{wl} subset store: {kl.size} reads, {int(rp[-1])} alignments, {tiles} tiles, created in {t_create * 1e3:.0f} ms (host arrays -> HBM + device layout)")
    print(f"   (i) pass {sub_pass:.4f} ms = {sub_pass / full_pass:.3f} of the full store's; un-batched replicate with multiplicities "
          f"{sub_unb * 1e3:.1f} us per pass (full store: {full_unb * 1e3:.1f}; batched: {bp[0] / bp[1] * 1e3:.1f} per replicate-pass)")
    print(f"   batched pass over the subset store (4 fresh resamples of ITS rows): {bp2[0]:.4f} ms = {bp2[0] / bp2[1] * 1e3:.1f} us per replicate-pass")
    err = np.abs(sub_counts - full_counts).max() / max(1.0, np.abs(full_counts).max())
    print(f"   same replicate through both stores: max |diff| / max count = {err:.2e}")
