"""CPU oracle for the oarfish EM / bootstrap hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``oarfish_amd``) never does.

Parity status: *unpinned by the reference* (the reference has no EM tests and
cannot be built here); pinned instead by closed forms, invariants and the
independent NumPy restatement in :mod:`oracle.oracle_np`.

``oracle.c_oracle`` is a ctypes binding of ``oracle/oem_oracle.c`` (the
line-by-line C restatement of src/em.rs, src/bootstrap.rs).
"""
from . import c_oracle, oracle_np  # noqa: F401
