"""Builds the C-ABI libraries in-tree with hipcc for gfx950.

  liboarfish_em.so          the product: include/oarfish_em.h and nothing else
  liboarfish_em_testing.so  the same objects plus the test hooks of csrc/oem_testing.hip and the
                            env-driven knobs (-DOEM_TESTING); loaded only by tests/ and scripts/

Plain HIP/C++: no torch types, no pybind.  hipcc cross-compiles gfx950 code objects without a GPU,
so this runs in the CPU-only container too.  Every source is compiled to its own object (in
parallel, cached by mtime under csrc/_obj/) and the two libraries are linked from them.
"""
from __future__ import annotations

import os
import sys

if __name__ == "__main__" and sys.path and os.path.abspath(sys.path[0]) == os.path.dirname(os.path.abspath(__file__)):
    sys.path.pop(0)  # run as a script: keep this package's types.py from shadowing the stdlib module

import shutil      # noqa: E402
import subprocess  # noqa: E402
from concurrent.futures import ThreadPoolExecutor  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB_PATH = os.path.join(HERE, "liboarfish_em.so")
TESTING_LIB_PATH = os.path.join(HERE, "liboarfish_em_testing.so")

# sources of the product library
SOURCES = ["oem_api.hip", "oem_em_driver.hip", "oem_bootstrap.hip", "oem_cells.hip", "oem_timing.hip", "oem_kernels.hip", "oem_tile_kernels.hip", "oem_batch_kernels.hip",
           "oem_multi_kernels.hip", "oem_layout.cpp", "oem_layout_device.hip", "oem_layout_pack.hip", "oem_layout_dict.hip", "oem_coverage_device.hip",
           "oem_builder.cpp", "oem_comm.cpp", "oem_p2p.hip", "oem_knobs.cpp"]
# the testing library swaps these for their -DOEM_TESTING build and adds the hooks
TESTING_VARIANTS = ["oem_comm.cpp", "oem_knobs.cpp", "oem_tile_kernels.hip", "oem_batch_kernels.hip"]
TESTING_ONLY = ["oem_testing.hip"]
HEADERS = ["oem_internal.h", "oem_driver.h", "oem_layout.h", "oem_tile_common.h", "oem_lane_runs.h", os.path.join(INCLUDE, "oarfish_em.h")]

FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-munsafe-fp-atomics",  # hardware global_atomic_add_f64 / ds_add_f64, no CAS loops
    "-ffp-contract=off",    # keep (theta*w)*inv as written; parity is judged in f64
    "-Wall",
    "-Wno-unused-function",
    "-Wno-unused-value",
    "-Wno-unused-result",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; cannot build liboarfish_em.so")


def _header_paths():
    return [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]


def _obj_path(src: str, testing: bool) -> str:
    return os.path.join(OBJ, os.path.splitext(src)[0] + (".testing.o" if testing else ".o"))


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _jobs():
    """(source, testing?) of every object the two libraries are linked from."""
    jobs = [(s, False) for s in SOURCES]
    jobs += [(s, True) for s in TESTING_VARIANTS + TESTING_ONLY]
    return jobs


def needs_build() -> bool:
    hdr = _header_paths()
    objs = [_obj_path(s, t) for s, t in _jobs()]
    if any(_stale(_obj_path(s, t), [os.path.join(CSRC, s)] + hdr) for s, t in _jobs()):
        return True
    return _stale(LIB_PATH, objs) or _stale(TESTING_LIB_PATH, objs)


def _compile(src: str, testing: bool, verbose: bool) -> None:
    cmd = [_hipcc(), *FLAGS, "-I", INCLUDE, "-x", "hip", "-c", os.path.join(CSRC, src)]
    if testing:
        cmd.append("-DOEM_TESTING")
    if os.environ.get("OEM_KBATCH"):  # experiments only: slots of one chain of the batched bootstrap (default 4)
        cmd.append("-DOEM_KBATCH=" + os.environ["OEM_KBATCH"])
    if os.environ.get("OEM_EXTRA_DEFS"):  # experiments only (scripts/build_variant.sh): A/B switches of a kernel under test
        cmd += os.environ["OEM_EXTRA_DEFS"].split()
    out = _obj_path(src, testing)
    cmd += ["-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)


TESTING_HOOKS = ["oem_debug_layout_hash", "oem_debug_local_comm_create", "oem_test_reldiff_stress", "oem_debug_knob",
                 "oem_debug_tile_probe", "oem_debug_tile_e_probe_begin", "oem_debug_tile_e_probe_end",
                 "oem_debug_overlap_probe"]


def header_symbols() -> list:
    """Every function include/oarfish_em.h declares, in declaration order."""
    import re
    src = open(os.path.join(INCLUDE, "oarfish_em.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = []
    for name in re.findall(r"\b(oem_[a-z0-9_]+)\s*\(", src):
        if name not in out:
            out.append(name)
    return out


def _version_script(path: str, symbols) -> str:
    """The dynamic symbol table of a library is exactly `symbols`: oem:: internals, rocPRIM
    instantiations and kernel stubs stay local (nm -D shows the header's entry points and nothing else)."""
    txt = "{\n  global:\n" + "".join(f"    {s};\n" for s in symbols) + "  local:\n    *;\n};\n"
    if not os.path.exists(path) or open(path).read() != txt:
        with open(path, "w") as f:
            f.write(txt)
    return path


def _link(objs, out: str, verbose: bool, exports) -> None:
    vs = _version_script(os.path.join(OBJ, os.path.basename(out) + ".map"), exports)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-Wl,--version-script=" + vs,
           *objs, "-o", out + ".tmp", "-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile what is stale and link both libraries; returns the product library's path."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(OBJ, exist_ok=True)
    hdr = _header_paths()
    todo = [(s, t) for s, t in _jobs()
            if force or _stale(_obj_path(s, t), [os.path.join(CSRC, s)] + hdr)]
    if todo:
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as ex:
            list(ex.map(lambda j: _compile(j[0], j[1], verbose), todo))
    product = [_obj_path(s, False) for s in SOURCES]
    testing = [_obj_path(s, s in TESTING_VARIANTS) for s in SOURCES] + [_obj_path(s, True) for s in TESTING_ONLY]
    _link(product, LIB_PATH, verbose, header_symbols())
    _link(testing, TESTING_LIB_PATH, verbose, header_symbols() + TESTING_HOOKS)
    return LIB_PATH


MICROBENCH = os.path.join(os.path.dirname(HERE), "scripts", "microbench")


def build_microbench(force: bool = False) -> list:
    """The measurement helpers of scripts/ (stand-alone HIP programs: the FETCH_SIZE calibration
    stream of scripts/collect_profiles.sh, the atomics / LDS / gather microbenchmarks).  Built in-tree
    like the library, so they travel to the GPU box; the binaries are git-ignored."""
    out = []
    for name in ("stream", "atomics", "lds_atomics"):
        src, exe = os.path.join(MICROBENCH, name + ".hip"), os.path.join(MICROBENCH, name)
        if force or not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
            subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-O3", "-munsafe-fp-atomics", src, "-o", exe])
        out.append(exe)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--microbench" in sys.argv:
        print(build_microbench(force="--force" in sys.argv))
