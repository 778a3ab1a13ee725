// oem_builder.cpp -- host-side store builder: the step immediately before the EM path
// (SURVEY.md section 8f row 1).
//
// Reference: AlignmentFilters::filter (src/util/oarfish_types.rs:955-1130) produces, per read, the
// retained alignments and their conditional probabilities as_prob = expf((score - best) / D) in
// f32 (:1107-1113); InMemoryAlignmentStore::add_group / add_filtered_group (:672-685, :718-738)
// append them to the CSR the EM consumes.  This runs once per store on the host; nothing here
// touches the GPU.  BAM parsing stays out of scope: the caller supplies the record fields the
// AlnRecordLike trait exposes (:180-202).
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
    if (!filters || !txp_len || !out || n_txps == 0) return fail(OEM_ERR_ARG, "oem_builder_create: bad argument");
    oem_builder *b = new (std::nothrow) oem_builder();
    if (!b) return fail(OEM_ERR_OOM, "oem_builder_create: host allocation failed");
    b->f = *filters;
    b->txp_len.assign(txp_len, txp_len + n_txps);
    *out = b;
    return OEM_OK;
}

extern "C" void oem_builder_destroy(oem_builder *b) { delete b; }

extern "C" int oem_builder_add_group(oem_builder *b, const oem_aln_record *ag, uint32_t n, uint32_t *out_kept)
{
    if (!b || (n && !ag)) return fail(OEM_ERR_ARG, "oem_builder_add_group: NULL argument");
    if (out_kept) *out_kept = 0;
    if (n == 0) return OEM_OK;                                    // add_group: `if !ag.is_empty()` (:677)
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
        if (x.ref_id >= b->txp_len.size())
            return fail(OEM_ERR_ARG, "oem_builder_add_group: ref_id %u is not below n_txps", x.ref_id);
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
}

extern "C" int oem_builder_dims(const oem_builder *b, uint64_t *n_reads, uint64_t *nnz)
{
    if (!b) return fail(OEM_ERR_ARG, "oem_builder_dims: builder is NULL");
    if (n_reads) *n_reads = b->row_ptr.size() - 1;
    if (nnz) *nnz = b->tid.size();
    return OEM_OK;
}

extern "C" int oem_builder_discard_table(const oem_builder *b, oem_discard_table *out)
{
    if (!b || !out) return fail(OEM_ERR_ARG, "oem_builder_discard_table: NULL argument");
    *out = b->dt;
    return OEM_OK;
}

extern "C" int oem_builder_export(const oem_builder *b, uint64_t *row_ptr, uint32_t *tid, float *as_prob,
                                  uint32_t *start, uint32_t *end, uint8_t *strand)
{
    if (!b) return fail(OEM_ERR_ARG, "oem_builder_export: builder is NULL");
    const size_t nnz = b->tid.size();
    if (row_ptr) std::memcpy(row_ptr, b->row_ptr.data(), sizeof(uint64_t) * b->row_ptr.size());
    if (tid && nnz) std::memcpy(tid, b->tid.data(), sizeof(uint32_t) * nnz);
    if (as_prob && nnz) std::memcpy(as_prob, b->as_prob.data(), sizeof(float) * nnz);
    if (start && nnz) std::memcpy(start, b->start.data(), sizeof(uint32_t) * nnz);
    if (end && nnz) std::memcpy(end, b->end.data(), sizeof(uint32_t) * nnz);
    if (strand && nnz) std::memcpy(strand, b->strand.data(), nnz);
    return OEM_OK;
}

extern "C" int oem_builder_store_create(const oem_builder *b, const double *cov_prob, int device,
                                        const oem_store_opts *opts, oem_store **out)
{
    if (!b || !out) return fail(OEM_ERR_ARG, "oem_builder_store_create: NULL argument");
    return oem_store_create(b->row_ptr.data(), b->tid.data(), b->as_prob.data(), cov_prob,
                            b->row_ptr.size() - 1, b->tid.size(), (uint32_t)b->txp_len.size(), device, opts, out);
}
