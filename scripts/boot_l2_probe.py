#!/usr/bin/env python3
"""Does the batched pass follow the size of theta[T][4] against the 4 MiB L2 of an XCD?  The same 10 M reads over 200 k
(6.4 MB of theta per batch), 100 k (3.2 MB) and 50 k transcripts (1.6 MB): batched pass and point-estimate pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _ab  # noqa: E401,E402,F401
from oarfish_amd import synth, _lib
from oarfish_amd.types import DeviceStore
for T in (200_000, 100_000, 50_000):
    st = synth.make_store(10_000_000, T, 8.0, threads=32)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, T) as d:
        ms, slots, nbytes = d.time_bootstrap_passes(20)
        d.time_m_step(20)
        pm = min(d.time_m_step(50) for _ in range(3))
        print(f"T={T}: theta[T][4] {T * 32 / 1e6:.1f} MB, tiles {d.info(_lib.OEM_INFO_TILES)}, remote {d.info(_lib.OEM_INFO_REMOTE_ALIGNMENTS) / st.nnz:.3f}: "
              f"batched pass {ms:.4f} ms, point-estimate pass {pm:.4f} ms", flush=True)
