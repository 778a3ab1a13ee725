"""CPU tests of the C-ABI library: it builds, loads, exports every symbol the
header declares, validates arguments and fails loudly without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oarfish_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "oarfish_em.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(oem_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    declared = _header_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/oarfish_em.h but not exported"
    assert sorted(_lib.ABI_SYMBOLS) == declared
    assert L.oem_abi_version() == 2


def test_no_torch_types_in_abi():
    src = open(os.path.join(ROOT, "include", "oarfish_em.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    assert "torch" not in src and "at::" not in src and "#include <hip" not in src


def test_argument_validation_precedes_device_use():
    L = _lib.lib()
    h = C.c_void_p()
    rp = np.array([0, 2, 1], dtype=np.uint64)  # decreasing
    tid = np.array([0, 1], dtype=np.uint32)
    p = np.array([1.0, 1.0], dtype=np.float32)
    rc = L.oem_store_create(rp.ctypes.data, tid.ctypes.data, p.ctypes.data, None, 2, 2, 2, 0, None, C.byref(h))
    assert rc == _lib.OEM_ERR_ARG and b"non-decreasing" in L.oem_last_error()
    rp = np.array([0, 1, 2], dtype=np.uint64)
    tid = np.array([0, 7], dtype=np.uint32)  # tid >= n_txps
    rc = L.oem_store_create(rp.ctypes.data, tid.ctypes.data, p.ctypes.data, None, 2, 2, 2, 0, None, C.byref(h))
    assert rc == _lib.OEM_ERR_ARG and b"n_txps" in L.oem_last_error()
    rc = L.oem_store_create(None, tid.ctypes.data, p.ctypes.data, None, 2, 2, 2, 0, None, C.byref(h))
    assert rc == _lib.OEM_ERR_ARG
    assert L.oem_em_run(None, None, 10, 1e-3, 50, None, None) == _lib.OEM_ERR_ARG
    assert L.oem_comm_create(None, 3, 2, 0, C.byref(h)) == _lib.OEM_ERR_ARG
    # option words outside their documented values are refused, not read as some other value
    tid = np.array([0, 1], dtype=np.uint32)
    for field, bad in (("weight_coding", 3), ("layout_build", 2), ("reorder_rows", 3)):
        o = _lib.StoreOptsC()
        setattr(o, field, bad)
        rc = L.oem_store_create(rp.ctypes.data, tid.ctypes.data, p.ctypes.data, None, 2, 2, 2, 0, C.byref(o), C.byref(h))
        assert rc == _lib.OEM_ERR_ARG and field.encode() in L.oem_last_error(), field


def test_fails_loudly_without_a_device():
    """No CPU fallback: on a box without a HIP device every compute entry point errors."""
    if _lib.device_count() > 0:
        pytest.skip("a HIP device is present")
    import oarfish_amd
    rp = np.array([0, 1, 2], dtype=np.uint64)
    tid = np.array([0, 1], dtype=np.uint32)
    p = np.array([1.0, 1.0], dtype=np.float32)
    with pytest.raises(oarfish_amd.OemError) as ei:
        oarfish_amd.DeviceStore(rp, tid, p, None, 2)
    assert ei.value.code == _lib.OEM_ERR_NO_DEVICE
    st = oarfish_amd.InMemoryAlignmentStore.from_arrays(rp, tid, p)
    emi = oarfish_amd.EMInfo(eq_map=st, txp_info=[oarfish_amd.TranscriptInfo()] * 2)
    with pytest.raises(oarfish_amd.OemError):
        oarfish_amd.em(emi, 1)
    # the device coverage model likewise: no silent host path behind the *_device entry point
    tl = np.array([500, 700], dtype=np.uint64)
    se = np.array([10, 20], dtype=np.uint32), np.array([300, 400], dtype=np.uint32)
    out = np.zeros(2)
    rc = _lib.lib().oem_coverage_probs_device(rp.ctypes.data, tid.ctypes.data, se[0].ctypes.data, se[1].ctypes.data,
                                              tl.ctypes.data, 2, 2, 2, 100, 0, 2.0, 0, out.ctypes.data)
    assert rc == _lib.OEM_ERR_NO_DEVICE


def test_p2p_communicator_without_a_device_fails_cleanly():
    """A communicator that will exchange peer to peer is created without touching a device; allocating its
    exchange buffer needs one: an error code, never a crash, and an unconnected multi-rank communicator
    cannot be attached to a store."""
    if _lib.device_count() > 0:
        pytest.skip("a HIP device is present")
    L = _lib.lib()
    h = C.c_void_p()
    assert L.oem_comm_create(None, 0, 2, 0, C.byref(h)) == _lib.OEM_OK
    try:
        blob = (C.c_ubyte * _lib.OEM_P2P_HANDLE_BYTES)()
        assert L.oem_comm_p2p_export(h, 0, C.addressof(blob)) == _lib.OEM_ERR_ARG
        assert L.oem_comm_p2p_connect(h, C.addressof(blob)) == _lib.OEM_ERR_STATE        # nothing exported yet
        assert L.oem_comm_p2p_export(h, 1000, C.addressof(blob)) in (_lib.OEM_ERR_HIP, _lib.OEM_ERR_NO_DEVICE, _lib.OEM_ERR_OOM)
        assert L.oem_comm_set_option(h, 99, 0) == _lib.OEM_ERR_ARG
        assert L.oem_comm_set_option(h, _lib.OEM_COMM_OPT_P2P_MAX_BYTES, 0) == _lib.OEM_OK
        for shape in (0, 1, 2):
            assert L.oem_comm_set_option(h, _lib.OEM_COMM_OPT_P2P_SHAPE, shape) == _lib.OEM_OK
        assert L.oem_comm_set_option(h, _lib.OEM_COMM_OPT_P2P_SHAPE, 3) == _lib.OEM_ERR_ARG
    finally:
        L.oem_comm_destroy(h)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under oarfish_amd/ may reference it."""
    pkg = os.path.join(ROOT, "oarfish_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("SURVEY", ""), f"{f} mentions the oracle"


def _exported(path):
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return {ln.split()[-1] for ln in out.splitlines() if ln.strip()}


def test_product_library_exports_exactly_the_header(monkeypatch):
    """nm -D of liboarfish_em.so is the header's entry points and NOTHING else (no oem:: internals, no
    rocPRIM instantiations: the link uses a version script made from the header); the debug / test
    hooks and the environment-driven knobs exist only in liboarfish_em_testing.so (-DOEM_TESTING)."""
    import subprocess
    from oarfish_amd import build as _b

    prod, test = _exported(_b.LIB_PATH), _exported(_b.TESTING_LIB_PATH)
    assert prod == set(_header_symbols()), sorted(prod ^ set(_header_symbols()))
    hooks = set(_b.TESTING_HOOKS)
    assert test == set(_header_symbols()) | hooks, sorted(test ^ (set(_header_symbols()) | hooks))
    # the product's knob() never reads the environment: its object file does not even reference getenv
    def undefined(obj):
        out = subprocess.check_output(["nm", "-u", obj], text=True)
        return {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    obj = os.path.join(_b.OBJ, "oem_knobs.o")
    assert "getenv" not in undefined(obj)
    assert "getenv" in undefined(os.path.join(_b.OBJ, "oem_knobs.testing.o"))
    monkeypatch.setenv("OEM_SELFTEST_KNOB", "7")
    T = _lib.testing_lib()
    T.oem_debug_knob.restype, T.oem_debug_knob.argtypes = C.c_long, [C.c_char_p, C.c_long]
    assert T.oem_debug_knob(b"OEM_SELFTEST_KNOB", 3) == 7
    assert not hasattr(_lib.lib(), "oem_debug_knob")


def test_header_is_strict_c99(tmp_path):
    """The boundary is a C ABI: include/oarfish_em.h must compile as C99 with no C++ or HIP in sight."""
    import subprocess
    src = tmp_path / "use_header.c"
    src.write_text('#include "oarfish_em.h"\nint main(void) { oem_run_info i; oem_store_opts o; (void)i; (void)o; '
                   'return OEM_ABI_VERSION == oem_abi_version() ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(ROOT, "include"), str(src)])


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md's table says what each C entry point replaces in the reference: none may be missing."""
    from oarfish_amd import build
    doc = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    assert [s for s in build.header_symbols() if s not in doc] == []
