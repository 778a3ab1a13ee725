"""Result writers: formats of write_function.rs / parquet_utils.rs."""
import json

import numpy as np
import pytest

from oarfish_amd import writers as W


def test_rust_display_matches_rust_float_formatting():
    # what `format!("{}", x)` prints for these f64 values
    cases = {3.0: "3", 0.1: "0.1", 1e-7: "0.0000001", 1e21: "1000000000000000000000", 123456.789: "123456.789",
             0.0: "0", 2.5e-5: "0.000025", 1.0 / 3.0: "0.3333333333333333", 1e16: "10000000000000000"}
    for x, s in cases.items():
        assert W.rust_display(x) == s
    assert W.rust_display(np.float32(0.1), f32=True) == "0.1"
    assert W.rust_display(np.float32(16777216.0), f32=True) == "16777216"
    assert W.rust_display(float("nan")) == "NaN" and W.rust_display(float("inf")) == "inf"
    for x in np.random.default_rng(0).lognormal(0, 8, size=200):
        assert float(W.rust_display(x)) == x and "e" not in W.rust_display(x)


def test_prob_display_decimals_reference_vectors():
    """The reference's own unit test (write_function.rs:346-359) as known-answer vectors."""
    import sys
    for thresh, want in [(1e-2, 3), (1e-3, 3), (0.5, 3), (1e-6, 6), (1e-4, 4), (1e-12, 9),
                         (sys.float_info.min, 9), (0.0, 9)]:
        assert W.prob_display_decimals(thresh) == want
    d = W.prob_display_decimals(1e-6)                      # write_function.rs:362-372
    assert any(c.isdigit() and c != "0" for c in f"{1e-6:.{d}f}")


def test_write_output_files(tmp_path):
    out = str(tmp_path / "sub" / "run1")
    names = ["t0", "t1|x", "t2"]
    W.write_output(out, {"num_reads": 7, "filter": {"a": 1.5}}, names, [100, 2500, 31], [3.0, 0.25, 1e-7],
                   [2, 0, 5], [3, 4, 4])
    assert json.load(open(out + ".meta_info.json"))["filter"]["a"] == 1.5
    assert open(out + ".quant").read() == "tname\tlen\tnum_reads\nt0\t100\t3\nt1|x\t2500\t0.25\nt2\t31\t0.0000001\n"
    # ambig = total - unique, saturating (write_function.rs:139-142)
    assert open(out + ".ambig_info.tsv").read() == "unique_reads\tambig_reads\ttotal_reads\n2\t1\t3\n0\t4\t4\n5\t0\t4\n"
    with pytest.raises(ValueError):
        W.write_output(out, {}, names, [1, 2], [0.0] * 3, [0] * 3, [0] * 3)


def test_infreps_parquet_roundtrip(tmp_path):
    import pyarrow.parquet as pq
    rng = np.random.default_rng(1)
    breps = rng.gamma(2.0, 3.0, size=(5, 37))
    path = W.write_infrep_file(str(tmp_path / "q"), breps)
    assert path.endswith("q.infreps.pq")
    f = pq.ParquetFile(path)
    assert [c for c in f.schema_arrow.names] == [f"bootstrap.{i}" for i in range(5)]
    assert all(not fld.nullable and str(fld.type) == "double" for fld in f.schema_arrow)
    assert f.metadata.row_group(0).column(0).compression == "ZSTD"
    assert "PLAIN" in f.metadata.row_group(0).column(0).encodings
    got = f.read().to_pandas().to_numpy().T
    assert np.array_equal(got, breps)                       # bit-exact f64


def test_prob_file_format(tmp_path):
    row_ptr = [0, 2, 3, 6]
    tid = [4, 1, 0, 2, 3, 4]
    probs = [0.75, 0.25, 1.0, -1.0, 0.5, 0.5]              # -1: below display_thresh, omitted
    path = W.write_out_prob(str(tmp_path / "p"), row_ptr, tid, probs, ["r0", "r1\0\0", "r2"],
                            [f"T{i}" for i in range(5)], 1e-4)
    lines = open(path).read().split("\n")
    assert lines[0] == "5\t3" and lines[1:6] == [f"T{i}" for i in range(5)]
    assert lines[6] == "r0\t2\t4\t1\t0.7500\t0.2500"
    assert lines[7] == "r1\t1\t0\t1.0000"
    assert lines[8] == "r2\t2\t3\t4\t0.5000\t0.5000"
    with pytest.raises(NotImplementedError):
        W.write_out_prob(str(tmp_path / "p"), row_ptr, tid, probs, ["a"] * 3, ["T"] * 5, 1e-4, compressed=True)


def test_single_cell_matrix_market(tmp_path):
    from scipy.io import mmread
    counts = np.array([[0.0, 1.5, 0.0, 2.0], [0.0, 0.0, 0.0, 0.0], [1e-3, 0.0, 7.0, 0.0]])
    r, c, v = W.cell_triplets(counts)
    assert v.dtype == np.float32 and list(r) == [0, 0, 2, 2] and list(c) == [1, 3, 0, 2]
    out = str(tmp_path / "sc")
    W.write_single_cell_output(out, {"k": 1}, ["a", "b", "c", "d"], ["AAAC", "AAAG", "AAAT"], 3, r, c, v)
    m = mmread(out + ".count.mtx").toarray()
    np.testing.assert_array_equal(m.astype(np.float32), counts.astype(np.float32))
    assert open(out + ".count.mtx").readline() == "%%MatrixMarket matrix coordinate real general\n"
    assert open(out + ".features.txt").read() == "a\nb\nc\nd\n"
    assert open(out + ".barcodes.txt").read() == "AAAC\nAAAG\nAAAT\n"


def test_read_short_quant_vec(tmp_path):
    """read_function.rs:9-85: projection onto the header order, 0 for missing, error for unknown names."""
    from oarfish_amd.bulk import read_short_quant_vec
    f = tmp_path / "quant.sf"
    f.write_text("Name\tLength\tEffectiveLength\tTPM\tNumReads\nt2\t900\t750.5\t12.5\t40.25\nt0\t300\t150\t1\t3\n")
    v = read_short_quant_vec(str(f), ["t0", "t1", "t2"])
    assert v.dtype == np.float64 and v.tolist() == [3.0, 0.0, 40.25]
    with pytest.raises(ValueError, match="didn't appear in the BAM header"):
        read_short_quant_vec(str(f), ["t0", "t1"])
    g = tmp_path / "bad.sf"
    g.write_text("Name\tNumReads\nt0\t3\n")
    with pytest.raises(ValueError, match="expected the columns"):
        read_short_quant_vec(str(g), ["t0"])
