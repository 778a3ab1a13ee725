"""Pure-Python restatement of AlignmentFilters::filter + add_filtered_group (TEST INFRASTRUCTURE ONLY).

Reference: src/util/oarfish_types.rs:955-1130 (filter), :718-738 (add_filtered_group),
:811-857 (DiscardTable).  Written independently of oarfish_amd/csrc/oem_builder.cpp, record by
record with Python loops (small cases only); f32 arithmetic through numpy.float32 and the C
library's expf (the function Rust's f32::exp lowers to on Linux).
"""
import ctypes
import ctypes.util
from dataclasses import dataclass, field

import numpy as np

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.expf.restype = ctypes.c_float
_libm.expf.argtypes = [ctypes.c_float]
f32 = np.float32
I32_MIN = -2 ** 31


@dataclass
class Filters:
    five_prime_clip: int = 2 ** 32 - 1
    three_prime_clip: int = 2 ** 62
    score_threshold: float = 0.95
    min_aligned_fraction: float = 0.5
    min_aligned_len: int = 50
    which_strand: int = 0          # 0 unknown, 1 forward, 2 reverse
    score_prob_denom: float = 5.0


@dataclass
class Rec:
    ref_id: int
    aln_start: int
    aln_end: int
    aln_span: int
    score: object = None           # None = no AS tag
    seq_len: object = None         # None = opt_sequence_len() is None
    unmapped: bool = False
    reverse: bool = False
    supp: bool = False


@dataclass
class Store:
    row_ptr: list = field(default_factory=lambda: [0])
    tid: list = field(default_factory=list)
    as_prob: list = field(default_factory=list)
    start: list = field(default_factory=list)
    end: list = field(default_factory=list)
    strand: list = field(default_factory=list)
    dt: dict = field(default_factory=lambda: dict(discard_5p=0, discard_3p=0, discard_score=0,
                                                  discard_aln_frac=0, discard_aln_len=0, discard_ori=0,
                                                  discard_supp=0, valid_best_aln=0, no_mapping=0,
                                                  no_valid_aln=0))


def add_group(st: Store, F: Filters, txp_len, ag):
    if not ag:                                                      # :677
        return 0
    dt = st.dt
    best, frac_best, len_best = I32_MIN, f32(0), 0                  # :963-969
    n_mapped_in = sum(1 for x in ag if not x.unmapped)              # :974
    seq_len = next((x.seq_len for x in ag if x.seq_len is not None), 0)  # :979-982
    kept = []
    for x in ag:                                                    # :985-1069
        if x.unmapped:
            continue
        score = x.score if x.score is not None else I32_MIN
        score = ((score + 2 ** 31) % 2 ** 32) - 2 ** 31              # `as i32`
        if F.which_strand == 2 and not x.reverse:
            dt["discard_ori"] += 1; continue
        if F.which_strand == 1 and x.reverse:
            dt["discard_ori"] += 1; continue
        if x.supp:
            dt["discard_supp"] += 1; continue
        if x.aln_span < F.min_aligned_len:
            dt["discard_aln_len"] += 1; continue
        if x.aln_end <= txp_len[x.ref_id] - F.three_prime_clip:
            dt["discard_3p"] += 1; continue
        if x.aln_start >= F.five_prime_clip:
            dt["discard_5p"] += 1; continue
        if score > best:
            best, len_best = score, x.aln_span
            frac_best = f32(x.aln_span) / f32(seq_len) if seq_len > 0 else f32(0)
        kept.append(x)
    if not kept or len_best == 0 or best <= 0:                      # :1071-1083
        dt["no_mapping" if n_mapped_in == 0 else "no_valid_aln"] += 1
        return 0
    if frac_best < f32(F.min_aligned_fraction):                     # :1084-1089
        dt["discard_aln_frac"] += 1
        return 0
    dt["valid_best_aln"] += 1
    mscore = f32(best)
    inv_max = f32(1.0) / mscore
    n = 0
    for x in kept:                                                  # :1107-1118
        sc = x.score if x.score is not None else 0
        sc = ((sc + 2 ** 31) % 2 ** 32) - 2 ** 31
        fscore = f32(sc)
        if not (fscore * inv_max >= f32(F.score_threshold)):
            dt["discard_score"] += 1
            continue
        fexp = (fscore - mscore) / f32(F.score_prob_denom)
        st.as_prob.append(f32(_libm.expf(ctypes.c_float(float(fexp)))))
        st.tid.append(x.ref_id); st.start.append(x.aln_start); st.end.append(x.aln_end)
        st.strand.append(1 if x.reverse else 0)
        n += 1
    if n:
        st.row_ptr.append(len(st.tid))                              # :733
    return n


# ---------------------------------------------------------------------------------------------
# Coverage model (bulk), restated independently of oem_builder.cpp:
#   TranscriptInfo::with_len_and_bin_width / add_interval   oarfish_types.rs:460-468, :496-538
#   logistic_prob / logstic_function / logistic             logistic_probability.rs:7-79
#   get_normalized_counts_and_lengths                       oarfish_types.rs:471-493
#   normalize_read_probs                                    normalize_probability.rs:5-74
# ---------------------------------------------------------------------------------------------
import math


def _rust_round(x):  # f64::round: half away from zero
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


def _f32(x):
    return float(np.float32(x))


def binomial_probability(cnt, length, distinct_rate):
    """binomial_probability.rs:7-168 with its f32 / f64 mix (cnt, length: f32 values held as floats)."""
    n = len(cnt)
    ZERO, MAXS = 1e-20, 709.0
    count_sum = np.float32(0.0)
    for c in cnt:
        count_sum = np.float32(count_sum + np.float32(c))                    # :14 (f32 sum)
    if float(count_sum) == 0.0 or distinct_rate == 0.0:                      # :19-25
        return [0.0] * n
    p = [0.0 if (c == 0.0 or l == 0.0) else c / (l * distinct_rate) for c, l in zip(cnt, length)]  # :27-43
    max_val = max(cnt)                                                        # :50
    m = [_f32(MAXS) if c == max_val else _f32((c * MAXS) / max_val) for c in cnt]  # :61-71
    sum_vec = np.float32(0.0)
    for v in m:
        sum_vec = np.float32(sum_vec + np.float32(v))                        # :72
    sum_vec = float(sum_vec)
    ln1 = math.lgamma(sum_vec + 1.0)                                         # :75
    res = []
    for pi, mi in zip(p, m):
        rest = _f32(np.float32(sum_vec) - np.float32(mi))                    # (sum_vec - count) in f32
        denom = math.lgamma(mi + 1.0) + math.lgamma(rest + 1.0)              # :76-79
        num2 = (math.log(pi) if pi > ZERO else math.log(ZERO)) * mi          # :82
        q = 1.0 - pi
        num3 = (math.log(q) if q > ZERO else math.log(ZERO)) * rest          # :89
        res.append(math.exp(ln1 - denom + num2 + num3))                      # :101
    tot = 0.0
    for r in res:
        tot += r                                                             # :120
    return [r / tot for r in res]                                            # :121-137


def coverage_probs(st: Store, txp_len, bin_width: int, growth_rate: float, model: str = "logistic"):
    T = len(txp_len)
    bins = [[0.0] * int(math.ceil(float(txp_len[t]) / float(bin_width))) for t in range(T)]
    total_weight = [0.0] * T
    U32 = 2 ** 32
    for j, t in enumerate(st.tid):                               # add_interval (:496-538)
        n = len(bins[t]); nf = float(n); tlen_f = float(txp_len[t])
        bw = _rust_round(tlen_f / nf)
        start, stop = st.start[j], st.end[j]
        start = min(start, stop)
        stop = max(start, stop)
        sb = int(math.floor((float(start) / tlen_f) * nf))
        eb = int(math.floor((float(stop) / tlen_f) * nf))
        for bi in range(sb, eb):
            cbs = int(float(bi) * bw) % U32
            cbe = int(min((float(bi) + 1.0) * bw, tlen_f)) % U32
            olap = ((min(stop, cbe) - max(start, cbs)) % U32) if start <= cbe else 0
            bins[t][bi] += float(olap) / float((cbe - cbs) % U32)
        total_weight[t] += 1.0
    prob = []
    for t in range(T):                                           # logistic_prob (:41-79)
        min_cov = total_weight[t] / 100.0
        b = [e + min_cov for e in bins[t]]
        counts = [float(np.float32(e)) for e in b]               # f32 counts (oarfish_types.rs:478)
        if model == "binomial":                                  # binomial_continuous_prob (:170-196)
            n = len(b)
            bwf = np.float32(_rust_round(float(txp_len[t]) / float(n)))
            lens = [_f32(min(np.float32((np.float32(i) + np.float32(1.0)) * bwf), np.float32(float(txp_len[t])))
                         - np.float32(np.float32(i) * bwf)) for i in range(n)]       # oarfish_types.rs:479-484
            rate = 0.0
            for c, l in zip(counts, lens):
                rate += c / l                                     # :184-188
            prob.append(binomial_probability(counts, lens, rate)); continue
        csum = 0.0
        for c in counts:
            csum += c
        if csum <= 1e-8:
            prob.append([0.0] * len(b)); continue
        expected = csum / len(b)
        pr = []
        for c in counts:
            diff = (expected - c) / expected
            r = 1.0 / (1.0 + math.exp(-growth_rate * diff))
            pr.append(min(max(r, 1e-8), 0.99999))
        prob.append(pr)
    out = [0.0] * len(st.tid)
    bl = float(bin_width)
    for r in range(len(st.row_ptr) - 1):                         # normalize_read_probs (:5-74)
        s = 0.0
        for j in range(st.row_ptr[r], st.row_ptr[r + 1]):
            t = st.tid[j]
            sa, ea, tlen = float(st.start[j]), float(st.end[j]), float(txp_len[t])
            sb = int(sa / bl)
            eb = min(int(ea / bl), len(prob[t]) - 1)
            if sb == eb:
                w = (ea - sa) / bl
                tw, cp = w, w * prob[t][sb]
            else:
                tw, cp = 0.0, 0.0
                for i in range(sb, eb):
                    w = (min(bl * float(i) + bl, tlen) - sa) / bl if i == sb else 1.0
                    tw += w
                    cp += w * prob[t][i]
            if math.isnan(cp) or math.isinf(cp):                  # :49-57: the only panic of this function
                raise ValueError("Error: Invalid result. normalize_read_probs function.")
            # f64 division as Rust does it: 0/0 is NaN, not an exception (:58); the NaN makes the row sum
            # fail `> 0` (:62) and the EM later drops the read (em.rs:115)
            out[j] = float(np.float64(cp) / np.float64(tw)) if tw != 0.0 else (float("nan") if cp == 0.0 else math.copysign(float("inf"), cp))
            s += out[j]
        d = s if s > 0 else 1.0
        for j in range(st.row_ptr[r], st.row_ptr[r + 1]):
            out[j] /= d
    return out
