// oem_builder.cpp -- host-side store builder: the step immediately before the EM path
// (SURVEY.md section 8f row 1).
//
// Reference: AlignmentFilters::filter (src/util/oarfish_types.rs:955-1130) produces, per read, the
// retained alignments and their conditional probabilities as_prob = expf((score - best) / D) in
// f32 (:1107-1113); InMemoryAlignmentStore::add_group / add_filtered_group (:672-685, :718-738)
// append them to the CSR the EM consumes.  This runs once per store on the host; nothing here
// touches the GPU.  BAM parsing stays out of scope: the caller supplies the record fields the
// AlnRecordLike trait exposes (:180-202).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "oem_internal.h"

struct oem_builder {
    oem_filters f;
    std::vector<uint64_t> txp_len;
    std::vector<uint64_t> row_ptr{0};           // boundaries, starts [0] (oarfish_types.rs:645)
    std::vector<uint32_t> tid, start, end;
    std::vector<uint8_t> strand;
    std::vector<float> as_prob;
    oem_discard_table dt{};
};

using namespace oem;

extern "C" int oem_builder_create(const oem_filters *filters, const uint64_t *txp_len, uint32_t n_txps,
                                  oem_builder **out)
{
    OEM_API_BEGIN
    if (!filters || !txp_len || !out || n_txps == 0) return fail(OEM_ERR_ARG, "oem_builder_create: bad argument");
    oem_builder *b = new (std::nothrow) oem_builder();
    if (!b) return fail(OEM_ERR_OOM, "oem_builder_create: host allocation failed");
    b->f = *filters;
    b->txp_len.assign(txp_len, txp_len + n_txps);
    *out = b;
    return OEM_OK;
    OEM_API_END("oem_builder_create")
}

extern "C" void oem_builder_destroy(oem_builder *b) { delete b; }

extern "C" int oem_builder_add_group(oem_builder *b, const oem_aln_record *ag, uint32_t n, uint32_t *out_kept)
{
    OEM_API_BEGIN
    if (!b || (n && !ag)) return fail(OEM_ERR_ARG, "oem_builder_add_group: NULL argument");
    if (out_kept) *out_kept = 0;
    if (n == 0) return OEM_OK;                                    // add_group: `if !ag.is_empty()` (:677)
    for (uint32_t i = 0; i < n; ++i)   // every argument error is reported before the discard table is touched
        if (!(ag[i].flags & OEM_REC_UNMAPPED) && ag[i].ref_id >= b->txp_len.size())
            return fail(OEM_ERR_ARG, "oem_builder_add_group: ref_id %u is not below n_txps", ag[i].ref_id);
    const oem_filters &F = b->f;
    oem_discard_table &dt = b->dt;

    int32_t best_retained_score = INT32_MIN;                      // :963
    float aln_frac_at_best_retained = 0.f;                        // :966
    uint32_t aln_len_at_best_retained = 0;                        // :969
    uint64_t n_mapped_in = 0;                                     // :974
    for (uint32_t i = 0; i < n; ++i) n_mapped_in += !(ag[i].flags & OEM_REC_UNMAPPED);
    uint32_t seq_len = 0;                                         // :979-982: first record that has a length
    for (uint32_t i = 0; i < n; ++i)
        if (ag[i].seq_len >= 0) { seq_len = (uint32_t)ag[i].seq_len; break; }

    std::vector<uint32_t> kept;                                   // ag.retain (:985-1069)
    kept.reserve(n);
    for (uint32_t i = 0; i < n; ++i) {
        const oem_aln_record &x = ag[i];
        if (x.flags & OEM_REC_UNMAPPED) continue;                 // :987
        const uint32_t aln_span = x.aln_span;                     // :991
        const int32_t score = (x.flags & OEM_REC_HAS_SCORE) ? (int32_t)x.score : INT32_MIN; // :994
        const bool is_rc = x.flags & OEM_REC_REVERSE;             // :997
        if (F.which_strand == 2 && !is_rc) { dt.discard_ori += 1; continue; }              // :1008-1011
        if (F.which_strand == 1 && is_rc) { dt.discard_ori += 1; continue; }               // :1013-1016
        if (x.flags & OEM_REC_SUPPLEMENTARY) { dt.discard_supp += 1; continue; }           // :1022-1026
        if (aln_span < F.min_aligned_len) { dt.discard_aln_len += 1; continue; }           // :1029-1033
        if ((int64_t)x.aln_end <= (int64_t)b->txp_len[x.ref_id] - F.three_prime_clip) {    // :1036-1041
            dt.discard_3p += 1;
            continue;
        }
        if (x.aln_start >= F.five_prime_clip) { dt.discard_5p += 1; continue; }            // :1044-1048
        if (score > best_retained_score) {                        // :1053-1063
            best_retained_score = score;
            aln_len_at_best_retained = aln_span;
            aln_frac_at_best_retained = seq_len > 0 ? (float)aln_span / (float)seq_len : 0.f;
        }
        kept.push_back(i);
    }
    if (kept.empty() || aln_len_at_best_retained == 0 || best_retained_score <= 0) {       // :1071-1083
        if (n_mapped_in == 0) dt.no_mapping += 1;
        else dt.no_valid_aln += 1;
        return OEM_OK;
    }
    if (aln_frac_at_best_retained < F.min_aligned_fraction) {     // :1084-1089
        dt.discard_aln_frac += 1;
        return OEM_OK;
    }
    dt.valid_best_aln += 1;                                       // :1092
    const float mscore = (float)best_retained_score;              // :1095
    const float inv_max_score = 1.0f / mscore;                    // :1096
    uint32_t n_kept = 0;
    for (uint32_t i : kept) {                                     // :1107-1118
        const oem_aln_record &x = ag[i];
        const int32_t sc = (x.flags & OEM_REC_HAS_SCORE) ? (int32_t)x.score : 0; // unwrap_or(0) (:1102)
        const float fscore = (float)sc;
        const bool score_ok = (fscore * inv_max_score) >= F.score_threshold;
        if (!score_ok) { dt.discard_score += 1; continue; }
        const float fexp = (fscore - mscore) / F.score_prob_denom;
        b->as_prob.push_back(expf(fexp));                         // f32 exp (:1113)
        b->tid.push_back(x.ref_id);                               // AlnInfo::from_aln_rec_like (:346-360)
        b->start.push_back(x.aln_start);
        b->end.push_back(x.aln_end);
        b->strand.push_back((x.flags & OEM_REC_REVERSE) ? 1 : 0);
        ++n_kept;
    }
    if (n_kept) b->row_ptr.push_back(b->tid.size());              // add_filtered_group (:724-735)
    if (out_kept) *out_kept = n_kept;
    return OEM_OK;
    OEM_API_END("oem_builder_add_group")
}

extern "C" int oem_builder_dims(const oem_builder *b, uint64_t *n_reads, uint64_t *nnz)
{
    OEM_API_BEGIN
    if (!b) return fail(OEM_ERR_ARG, "oem_builder_dims: builder is NULL");
    if (n_reads) *n_reads = b->row_ptr.size() - 1;
    if (nnz) *nnz = b->tid.size();
    return OEM_OK;
    OEM_API_END("oem_builder_dims")
}

extern "C" int oem_builder_discard_table(const oem_builder *b, oem_discard_table *out)
{
    OEM_API_BEGIN
    if (!b || !out) return fail(OEM_ERR_ARG, "oem_builder_discard_table: NULL argument");
    *out = b->dt;
    return OEM_OK;
    OEM_API_END("oem_builder_discard_table")
}

extern "C" int oem_builder_export(const oem_builder *b, uint64_t *row_ptr, uint32_t *tid, float *as_prob,
                                  uint32_t *start, uint32_t *end, uint8_t *strand)
{
    OEM_API_BEGIN
    if (!b) return fail(OEM_ERR_ARG, "oem_builder_export: builder is NULL");
    const size_t nnz = b->tid.size();
    if (row_ptr) std::memcpy(row_ptr, b->row_ptr.data(), sizeof(uint64_t) * b->row_ptr.size());
    if (tid && nnz) std::memcpy(tid, b->tid.data(), sizeof(uint32_t) * nnz);
    if (as_prob && nnz) std::memcpy(as_prob, b->as_prob.data(), sizeof(float) * nnz);
    if (start && nnz) std::memcpy(start, b->start.data(), sizeof(uint32_t) * nnz);
    if (end && nnz) std::memcpy(end, b->end.data(), sizeof(uint32_t) * nnz);
    if (strand && nnz) std::memcpy(strand, b->strand.data(), nnz);
    return OEM_OK;
    OEM_API_END("oem_builder_export")
}

// ---------------------------------------------------------------------------
// coverage model (bulk): oarfish_types.rs:460-538, logistic_probability.rs:7-79,
// normalize_probability.rs:5-74
// ---------------------------------------------------------------------------
// binomial_probability (binomial_probability.rs:7-168): per-bin Binomial(sum, p_i) pmf at the bin's
// count after rescaling the counts so that the largest is 709, normalised over the bins.  The f32 /
// f64 mix of the reference is kept (f32 sums and differences, f64 logs).  ln_gamma: statrs' Lanczos
// evaluation there, libm's lgamma here (same function, ~1e-15 relative apart).
static int binomial_probability(const std::vector<float> &cnt, const std::vector<float> &len, double distinct_rate,
                                std::vector<double> &out)
{
    const size_t n = cnt.size();
    const double kZero = 1e-20, kMaxScale = 709.0;
    out.assign(n, 0.0);
    float count_sum = 0.0f;
    for (float c : cnt) count_sum += c;                                // :14
    if (count_sum == 0.0f || distinct_rate == 0.0) return OEM_OK;      // :19-25
    std::vector<double> p(n);
    for (size_t i = 0; i < n; ++i)                                     // :27-43
        p[i] = (cnt[i] == 0.0f || len[i] == 0.0f) ? 0.0 : (double)cnt[i] / ((double)len[i] * distinct_rate);
    float max_val = std::nanf("");                                     // :50 (f32::max ignores a NaN operand)
    for (float c : cnt) max_val = std::isnan(max_val) ? c : (std::isnan(c) ? max_val : std::max(max_val, c));
    if (std::isnan(max_val)) return fail(OEM_ERR_STATE, "binomial_probability: max bin count is NaN (assert, :51)");
    std::vector<float> m(n);
    for (size_t i = 0; i < n; ++i)                                     // :61-71
        m[i] = cnt[i] == max_val ? (float)kMaxScale : (float)(((double)cnt[i] * kMaxScale) / (double)max_val);
    float sum_vec = 0.0f;
    for (float v : m) sum_vec += v;                                    // :72
    const double ln1 = std::lgamma((double)sum_vec + 1.0);             // :75
    double total = 0.0;
    for (size_t i = 0; i < n; ++i) {
        const double denom = std::lgamma((double)m[i] + 1.0) + std::lgamma((double)(sum_vec - m[i]) + 1.0); // :76-79
        const double num2 = (p[i] > kZero ? std::log(p[i]) : std::log(kZero)) * (double)m[i];              // :82
        const double q = 1.0 - p[i];
        const double num3 = (q > kZero ? std::log(q) : std::log(kZero)) * (double)(sum_vec - m[i]);        // :89
        const double res = std::exp(ln1 - denom + num2 + num3);        // :101
        if (std::isnan(num2) || std::isinf(num2) || std::isnan(num3) || std::isinf(num3) || std::isnan(res) ||
            std::isinf(res))                                           // the reference panics (:83-112)
            return fail(OEM_ERR_STATE, "binomial_probability: non-finite value at bin %zu", i);
        out[i] = res;
        total += res;                                                  // :120
    }
    for (size_t i = 0; i < n; ++i) {
        out[i] /= total;                                               // :124
        if (std::isnan(out[i])) return fail(OEM_ERR_STATE, "binomial_probability: normalised probability is NaN (:125-133)");
    }
    return OEM_OK;
}

enum class CovModel { Logistic, Binomial };

static int coverage_probs_impl(const oem_builder *b, uint32_t bin_width_u, CovModel model, double growth_rate,
                               double *out)
{
    if (!b || (!out && !b->tid.empty())) return fail(OEM_ERR_ARG, "oem_builder_coverage_probs: NULL argument");
    if (bin_width_u == 0) return fail(OEM_ERR_ARG, "coverage model with 0 bin width is not implemented (logistic_probability.rs:59, binomial_probability.rs:192)");
    const size_t T = b->txp_len.size(), nnz = b->tid.size();
    struct Txp { std::vector<double> bins, prob; double total_weight = 0.0, lenf = 0.0; };
    std::vector<Txp> txps(T);
    for (size_t t = 0; t < T; ++t) {                                   // with_len_and_bin_width (:460-468)
        txps[t].lenf = (double)b->txp_len[t];
        txps[t].bins.assign((size_t)std::ceil((double)b->txp_len[t] / (double)bin_width_u), 0.0);
    }
    // add_interval for every retained alignment, in store order (add_filtered_group, :725-728)
    for (size_t j = 0; j < nnz; ++j) {
        Txp &tx = txps[b->tid[j]];
        const size_t num_intervals = tx.bins.size();
        const double nf = (double)num_intervals, tlen_f = tx.lenf;
        const double bw = std::round(tlen_f / nf);                     // :501
        uint32_t start = b->start[j], stop = b->end[j];
        start = std::min(start, stop);                                 // :502
        stop = std::max(start, stop);                                  // :503
        const size_t start_bin = (size_t)std::floor(((double)start / tlen_f) * nf); // :504
        const size_t end_bin = (size_t)std::floor(((double)stop / tlen_f) * nf);    // :505
        if (start_bin > end_bin || end_bin > num_intervals)
            return fail(OEM_ERR_STATE, "add_interval: alignment [%u,%u) outside transcript %u of length %llu", start,
                        stop, b->tid[j], (unsigned long long)b->txp_len[b->tid[j]]);
        for (size_t bi = start_bin; bi < end_bin; ++bi) {              // :515-536 (end_bin itself is not visited)
            const double bidxf = (double)bi;
            const uint32_t cbs = (uint32_t)(bidxf * bw);
            const uint32_t cbe = (uint32_t)std::min((bidxf + 1.0) * bw, tlen_f);
            const uint32_t olap = start <= cbe ? std::min(stop, cbe) - std::max(start, cbs) : 0u; // :507-513 (u32)
            const double olfrac = (double)olap / (double)(uint32_t)(cbe - cbs);
            tx.bins[bi] += olfrac;
            if (olfrac > 1.0 + 2.220446049250313e-16)                  // :524-535: the reference panics here
                return fail(OEM_ERR_STATE, "coverage computation error at transcript %u bin %zu", b->tid[j], bi);
        }
        tx.total_weight += 1.0;                                        // :537 (weight 1.0, :727)
    }
    // logistic_prob (logistic_probability.rs:41-79) / binomial_continuous_prob (binomial_probability.rs:170-224)
    for (size_t t = 0; t < T; ++t) {
        Txp &tx = txps[t];
        const size_t n = tx.bins.size();
        if (n == 0) return fail(OEM_ERR_STATE, "transcript %zu has no coverage bins", t); // assert (:54)
        const double min_cov = tx.total_weight / 100.;                 // :55 / binomial :180
        for (double &e : tx.bins) e += min_cov;                        // :56
        // get_normalized_counts_and_lengths (oarfish_types.rs:471-493): f32 counts; the bin-width
        // assertion is reproduced because the reference would panic there
        const float bwf = (float)std::round(tx.lenf / (double)n);
        for (size_t bi = 0; bi < n; ++bi) {
            const float bs = (float)bi * bwf, be = std::min(((float)bi + 1.0f) * bwf, (float)tx.lenf);
            if (!(be > bs)) return fail(OEM_ERR_STATE, "transcript %zu: degenerate coverage bin %zu (assert, oarfish_types.rs:490)", t, bi);
        }
        tx.prob.assign(n, 0.0);
        if (model == CovModel::Binomial) {
            std::vector<float> cnt(n), len(n);
            for (size_t bi = 0; bi < n; ++bi) {
                cnt[bi] = (float)tx.bins[bi];
                len[bi] = std::min(((float)bi + 1.0f) * bwf, (float)tx.lenf) - (float)bi * bwf;
            }
            double distinct_rate = 0.0;                                // binomial_probability.rs:184-188
            for (size_t bi = 0; bi < n; ++bi) distinct_rate += (double)cnt[bi] / (double)len[bi];
            const int rc = binomial_probability(cnt, len, distinct_rate, tx.prob);
            if (rc != OEM_OK) return rc;
            continue;
        }
        double count_sum = 0.0;                                        // logstic_function (:13-39)
        for (double e : tx.bins) count_sum += (double)(float)e;
        if (count_sum <= 1e-8) continue;                               // :21-23
        const double expected = count_sum / (double)n;                 // :27
        for (size_t bi = 0; bi < n; ++bi) {
            const double diff = (expected - (double)(float)tx.bins[bi]) / expected;       // :32
            double r = 1.0 / (1.0 + std::exp(-growth_rate * diff));    // logistic (:7-10)
            r = r < 1e-8 ? 1e-8 : (r > 0.99999 ? 0.99999 : r);
            tx.prob[bi] = r;
        }
    }
    // normalize_read_probs (normalize_probability.rs:5-74)
    const double bin_length = (double)bin_width_u;
    const size_t R = b->row_ptr.size() - 1;
    for (size_t r = 0; r < R; ++r) {
        double nprob_sum = 0.0;
        for (uint64_t j = b->row_ptr[r]; j < b->row_ptr[r + 1]; ++j) {
            const Txp &tx = txps[b->tid[j]];
            const double start_aln = (double)b->start[j], end_aln = (double)b->end[j], tlen = (double)b->txp_len[b->tid[j]];
            const size_t start_bin = (size_t)(start_aln / bin_length);                     // :25
            const size_t end_bin = std::min((size_t)(end_aln / bin_length), tx.prob.size() - 1); // :26-27
            double total_weight = 0.0, cov_prob = 0.0;
            if (start_bin == end_bin) {                                // :33-35
                const double w = (end_aln - start_aln) / bin_length;
                total_weight = w;
                cov_prob = w * tx.prob[start_bin];
            } else {
                for (size_t i = start_bin; i < end_bin; ++i) {         // :37-46 (end_bin itself is not visited)
                    const double w = i == start_bin
                                         ? (std::min(bin_length * (double)i + bin_length, tlen) - start_aln) / bin_length
                                         : 1.0;
                    total_weight += w;
                    cov_prob += w * tx.prob[i];
                }
            }
            // :49-57: the reference panics on a non-finite cov_prob only.  A 0/0 expected value (zero-span
            // alignment, empty bin range) is not an error there: the NaN flows into the column, the row sum
            // fails `> 0` (:62), and the EM drops the read because its denominator is not > 1e-30 (em.rs:115).
            if (std::isnan(cov_prob) || std::isinf(cov_prob))
                return fail(OEM_ERR_STATE, "normalize_read_probs: invalid coverage probability for alignment %llu",
                            (unsigned long long)j);
            const double expected = cov_prob / total_weight;           // :58
            out[j] = expected;
            nprob_sum += expected;
        }
        const double denom = nprob_sum > 0.0 ? nprob_sum : 1.0;        // :62
        for (uint64_t j = b->row_ptr[r]; j < b->row_ptr[r + 1]; ++j) out[j] /= denom; // :65-69
    }
    return OEM_OK;
}

extern "C" int oem_builder_coverage_probs(const oem_builder *b, uint32_t bin_width, double growth_rate, double *out)
{
    OEM_API_BEGIN
    return coverage_probs_impl(b, bin_width, CovModel::Logistic, growth_rate, out);
    OEM_API_END("oem_builder_coverage_probs")
}

extern "C" int oem_builder_coverage_probs_binomial(const oem_builder *b, uint32_t bin_width, double *out)
{
    OEM_API_BEGIN
    return coverage_probs_impl(b, bin_width, CovModel::Binomial, 0.0, out);
    OEM_API_END("oem_builder_coverage_probs_binomial")
}

extern "C" int oem_builder_coverage_probs_device(const oem_builder *b, uint32_t bin_width, int model, double growth_rate,
                                                 int device, double *out)
{
    OEM_API_BEGIN
    if (!b) return fail(OEM_ERR_ARG, "oem_builder_coverage_probs_device: builder is NULL");
    return oem_coverage_probs_device(b->row_ptr.data(), b->tid.data(), b->start.data(), b->end.data(), b->txp_len.data(),
                                     b->row_ptr.size() - 1, b->tid.size(), (uint32_t)b->txp_len.size(), bin_width, model,
                                     growth_rate, device, out);
    OEM_API_END("oem_builder_coverage_probs_device")
}

extern "C" int oem_builder_store_create(const oem_builder *b, const double *cov_prob, int device,
                                        const oem_store_opts *opts, oem_store **out)
{
    OEM_API_BEGIN
    if (!b || !out) return fail(OEM_ERR_ARG, "oem_builder_store_create: NULL argument");
    return oem_store_create(b->row_ptr.data(), b->tid.data(), b->as_prob.data(), cov_prob,
                            b->row_ptr.size() - 1, b->tid.size(), (uint32_t)b->txp_len.size(), device, opts, out);
    OEM_API_END("oem_builder_store_create")
}
