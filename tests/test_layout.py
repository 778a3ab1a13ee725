"""CPU test of the tiled HBM layout (oarfish_amd/csrc/oem_layout.cpp): the product's
layout pass is compiled into a test helper that replays the tile kernels'
arithmetic on the host; the result must equal the oracle's m_step."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oarfish_amd import synth
from oracle import c_oracle
from tests.common import golden_names, load_golden

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "layout_emul.cpp")
LIB = os.path.join(HERE, "native", "liblayout_emul.so")


@pytest.fixture(scope="module")
def emul():
    dep = os.path.join(HERE, "..", "oarfish_amd", "csrc", "oem_layout.cpp")
    hdr = os.path.join(HERE, "..", "oarfish_amd", "csrc", "oem_layout.h")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(map(os.path.getmtime, (SRC, dep, hdr))):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-o", LIB, SRC])
    return C.CDLL(LIB)


def _run(emul, row_ptr, tid, p, cov, T, theta, row_w=None, problem_size=0):
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
    tid = np.ascontiguousarray(tid, dtype=np.uint32)
    p = np.ascontiguousarray(p, dtype=np.float32)
    cnt = np.zeros(T, dtype=np.float64)
    stats = np.zeros(5, dtype=np.uint64)
    vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    rc = emul.layout_emul_m_step(vp(row_ptr), vp(tid), vp(p), vp(cov), C.c_uint64(len(row_ptr) - 1),
                                 C.c_uint64(len(tid)), C.c_uint32(T), vp(theta), vp(row_w), vp(cnt), vp(stats),
                                 C.c_uint32(problem_size))
    assert rc == 0, f"layout self-check failed with code {rc}"
    return cnt, stats


@pytest.mark.parametrize("name", golden_names())
def test_layout_on_golden(emul, name):
    g = load_golden(name)
    T = g["n_txps"]
    rng = np.random.default_rng(3)
    theta = rng.lognormal(0, 1.5, size=T)
    theta[rng.random(T) < 0.1] = 0.0
    o = c_oracle.Store(g["row_ptr"], g["tid"], g["as_prob"], g["cov_prob"], T)
    got, _ = _run(emul, g["row_ptr"], g["tid"], g["as_prob"], g["cov_prob"], T, theta)
    np.testing.assert_allclose(got, c_oracle.m_step(o, theta), rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("coverage", [False, True])
def test_layout_large_window_overflow(emul, coverage):
    """T far beyond one window and one bucket: local + remote + several buckets."""
    st = synth.make_store(120_000, 30_000, seed=17, coverage=coverage, threads=2)
    rng = np.random.default_rng(4)
    theta = rng.lognormal(0, 2, size=st.n_txps)
    w = rng.poisson(1.0, st.n_reads).astype(np.uint32)
    o = c_oracle.Store(st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps)
    got, stats = _run(emul, st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps, theta)
    np.testing.assert_allclose(got, c_oracle.m_step(o, theta), rtol=1e-10, atol=1e-10)
    got_w, _ = _run(emul, st.row_ptr, st.tid, st.as_prob, st.cov_prob, st.n_txps, theta, w)
    np.testing.assert_allclose(got_w, c_oracle.m_step(o, theta, row_w=w), rtol=1e-10, atol=1e-10)
    n_tiles, n_local, n_remote, n_rows, w_slots = (int(x) for x in stats)
    assert n_local + n_remote == st.nnz and n_rows == st.n_reads
    assert n_remote < 0.3 * st.nnz            # ~17.5 % of the alignments are uniform-random targets
    assert w_slots < 1.15 * n_local           # SELL padding stays small


@pytest.mark.parametrize("what", ["sparse", "dense", "cells"])
def test_layout_wide_window(emul, monkeypatch, what):
    """Window cap kWinWide (2048 transcripts, chosen for sparse stores such as per-cell batches): same
    results, and on a sparse store many more reads per tile than with the narrow window."""
    ps = 0
    if what == "sparse":
        st = synth.make_store(40_000, 50_000, seed=23, threads=2)       # 0.8 reads per transcript
        rp, tid, p, T = st.row_ptr, st.tid, st.as_prob, st.n_txps
    elif what == "dense":
        st = synth.make_store(60_000, 2_500, seed=24, threads=2)
        rp, tid, p, T = st.row_ptr, st.tid, st.as_prob, st.n_txps
    else:
        n_cells, Tc = 4, 3_000
        cell_off, rp, tid, p = synth.make_cells(n_cells, 2_500, Tc, seed=25)
        cell_of_row = np.repeat(np.arange(n_cells), np.diff(cell_off.astype(np.int64)))
        tid = (tid.astype(np.int64) + np.repeat(cell_of_row, np.diff(rp.astype(np.int64))) * Tc).astype(np.uint32)
        T, ps = n_cells * Tc, Tc
    theta = np.random.default_rng(8).lognormal(0, 1.5, size=T)
    want = c_oracle.m_step(c_oracle.Store(rp, tid, p, None, T), theta)
    narrow, s_n = _run(emul, rp, tid, p, None, T, theta, problem_size=ps)
    monkeypatch.setenv("LAYOUT_EMUL_WIN_CAP", "2048")
    wide, s_w = _run(emul, rp, tid, p, None, T, theta, problem_size=ps)
    np.testing.assert_allclose(narrow, want, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(wide, want, rtol=1e-10, atol=1e-10)
    if what != "dense":
        assert int(s_w[0]) < 0.6 * int(s_n[0])      # fewer, fuller tiles


def test_layout_empty_rows_and_sparse_keys(emul):
    # empty reads are skipped; primaries far apart force many narrow tiles
    rp = np.array([0, 0, 2, 2, 3, 5, 5], dtype=np.uint64)
    tid = np.array([0, 49_999, 25_000, 7, 40_000], dtype=np.uint32)
    p = np.array([1.0, 0.5, 1.0, 0.3, 1.0], dtype=np.float32)
    theta = np.linspace(0.5, 2.0, 50_000)
    o = c_oracle.Store(rp, tid, p, None, 50_000)
    got, stats = _run(emul, rp, tid, p, None, 50_000, theta)
    np.testing.assert_allclose(got, c_oracle.m_step(o, theta), rtol=1e-12, atol=1e-14)
    assert int(stats[3]) == 3


def test_layout_per_cell_problems(emul):
    """Per-cell batch: cells concatenated over a virtual transcript space; tiles stay inside one cell."""
    n_cells, T = 5, 700
    cell_off, row_ptr, tid, p = synth.make_cells(n_cells, 2_500, T, seed=3)
    lens = np.diff(row_ptr.astype(np.int64))
    cell_of_row = np.repeat(np.arange(n_cells), np.diff(cell_off.astype(np.int64)))
    vtid = (tid.astype(np.int64) + np.repeat(cell_of_row, lens) * T).astype(np.uint32)
    rng = np.random.default_rng(6)
    theta = rng.lognormal(0, 1.5, size=n_cells * T)
    o = c_oracle.Store(row_ptr, vtid, p, None, n_cells * T)
    got, stats = _run(emul, row_ptr, vtid, p, None, n_cells * T, theta, problem_size=T)
    np.testing.assert_allclose(got, c_oracle.m_step(o, theta), rtol=1e-10, atol=1e-10)


def test_layout_hypothesis_random_stores(emul):
    """Property test: for arbitrary ragged stores -- empty reads, repeated
    transcripts inside a read, zero weights, 1..3000 transcripts, far-apart targets -- the tiled layout
    replayed on the host equals the oracle's m_step, with and without per-read multiplicities."""
    from hypothesis import given, settings, strategies as st_, HealthCheck

    @settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
    @given(st_.integers(1, 3000), st_.integers(0, 400), st_.integers(0, 2**31 - 1), st_.booleans())
    def run(T, R, seed, coverage):
        rng = np.random.default_rng(seed)
        lens = rng.integers(0, 9, size=R)
        if R and rng.random() < 0.3:
            lens[rng.integers(0, R)] = 100                      # one --best-n sized read
        rp = np.zeros(R + 1, dtype=np.uint64)
        np.cumsum(lens, out=rp[1:])
        nnz = int(rp[-1])
        centre = rng.integers(0, T, size=R)
        tid = np.empty(nnz, dtype=np.uint32)
        for i in range(R):
            s, e = int(rp[i]), int(rp[i + 1])
            near = (centre[i] + rng.integers(-4, 5, size=e - s)) % T   # repeats inside a read are allowed
            far = rng.integers(0, T, size=e - s)
            tid[s:e] = np.where(rng.random(e - s) < 0.8, near, far)
        p = np.exp(-rng.integers(0, 30, size=nnz).astype(np.float32) / np.float32(5)).astype(np.float32)
        p[rng.random(nnz) < 0.05] = 0.0
        cov = rng.uniform(0.0, 1.0, size=nnz) if coverage else None
        theta = rng.lognormal(0, 2, size=T)
        theta[rng.random(T) < 0.2] = 0.0
        w = rng.integers(0, 4, size=R).astype(np.uint32)
        o = c_oracle.Store(rp, tid, p, cov, T)
        got, stats = _run(emul, rp, tid, p, cov, T, theta)
        np.testing.assert_allclose(got, c_oracle.m_step(o, theta), rtol=1e-10, atol=1e-12)
        got_w, _ = _run(emul, rp, tid, p, cov, T, theta, w)
        np.testing.assert_allclose(got_w, c_oracle.m_step(o, theta, row_w=w), rtol=1e-10, atol=1e-12)
        assert int(stats[1]) + int(stats[2]) == nnz and int(stats[3]) == int((lens > 0).sum())

    run()
