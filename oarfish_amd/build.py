"""Builds liboarfish_em.so (the C-ABI library) in-tree with hipcc for gfx950.

The library is plain HIP/C++: no torch types, no pybind.  hipcc cross-compiles
gfx950 code objects without a GPU, so this runs in the CPU-only container too.
"""
from __future__ import annotations

import os
import sys

if __name__ == "__main__" and sys.path and os.path.abspath(sys.path[0]) == os.path.dirname(os.path.abspath(__file__)):
    sys.path.pop(0)  # run as a script: keep this package's types.py from shadowing the stdlib module

import shutil      # noqa: E402
import subprocess  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB_PATH = os.path.join(HERE, "liboarfish_em.so")

SOURCES = ["oem_api.hip", "oem_kernels.hip", "oem_tile_kernels.hip", "oem_batch_kernels.hip", "oem_multi_kernels.hip", "oem_layout.cpp", "oem_layout_device.hip", "oem_coverage_device.hip", "oem_builder.cpp", "oem_comm.cpp"]
HEADERS = ["oem_internal.h", "oem_layout.h", os.path.join(INCLUDE, "oarfish_em.h")]

FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
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


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [
        h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS
    ] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc(), *FLAGS, "-I", INCLUDE, "-x", "hip"]
    if os.environ.get("OEM_TILE_ABLATION"):  # profiling builds only: enables the OEM_TILE_ABLATE switches
        cmd.append("-DOEM_TILE_ABLATION")
        if os.environ.get("OEM_ABL_BYTEW"):
            cmd.append("-DOEM_ABL_BYTEW")
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIB_PATH + ".tmp", "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


MICROBENCH = os.path.join(os.path.dirname(HERE), "scripts", "microbench")


def build_microbench(force: bool = False) -> list:
    """The measurement helpers of scripts/ (stand-alone HIP programs: the FETCH_SIZE calibration
    stream of scripts/collect_profiles.sh, the atomics / LDS / gather microbenchmarks).  Built in-tree
    like the library, so they travel to the GPU box; the binaries are git-ignored."""
    out = []
    for name in ("stream", "atomics"):
        src, exe = os.path.join(MICROBENCH, name + ".hip"), os.path.join(MICROBENCH, name)
        if force or not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
            subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-O3", "-munsafe-fp-atomics", src, "-o", exe])
        out.append(exe)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--microbench" in sys.argv:
        print(build_microbench(force="--force" in sys.argv))
