"""The C++ host mirror (include/oarfish_em.hpp: InMemoryAlignmentStore / EMInfo / em::em / em::em_par /
em::bootstrap with the reference's names) compiled with g++ against liboarfish_em.so and the oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "host_mirror_test.cpp")
EXE = os.path.join(ROOT, "tests", "native", "host_mirror_test")


def _build():
    from oarfish_amd import build as b
    from oracle import c_oracle
    b.build()
    c_oracle.build()
    libdir = os.path.join(ROOT, "oarfish_amd")
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["g++", "-O1", "-std=c++17", SRC, "-o", EXE, "-L" + libdir, "-loarfish_em", "-L" + odir,
                           "-loem_oracle", "-Wl,-rpath," + libdir, "-Wl,-rpath," + odir, "-fopenmp"])


def _run():
    env = dict(os.environ)
    # one HIP runtime per process: resolve libamdhip64 the way the python process does
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    return subprocess.run([EXE], capture_output=True, text=True, env=env, timeout=300)


def test_cpp_mirror_builds_and_fails_loudly_without_device():
    from oarfish_amd import _lib
    _build()
    if _lib.device_count() > 0:
        pytest.skip("a HIP device is present; the gpu-marked test covers the mirror")
    r = _run()
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)       # OemError{OEM_ERR_NO_DEVICE}
    assert "no HIP device" in r.stdout


@pytest.mark.gpu
def test_cpp_mirror_parity_on_gpu():
    _build()
    r = _run()
    assert r.returncode == 0 and "PASS" in r.stdout, (r.returncode, r.stdout, r.stderr)
