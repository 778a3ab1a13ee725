// oarfish_em.hpp -- C++17 host-side mirror of the reference's EM interface over the C ABI.
//
// The reference is compiled code (Rust); no Rust toolchain exists in this image, so the host
// side above include/oarfish_em.h is mirrored here in C++ with the reference's own names and
// argument meaning (COMBINE-lab/oarfish v0.10.3):
//
//   oarfish::AlnInfo                 src/util/oarfish_types.rs:330-337
//   oarfish::AlignmentFilters        src/util/oarfish_types.rs:763-806 (only model_coverage, :792)
//   oarfish::TranscriptInfo          src/util/oarfish_types.rs:431-437 (len / lenf)
//   oarfish::InMemoryAlignmentStore  src/util/oarfish_types.rs:548-558, add_filtered_group :718-738,
//                                    len :562-564, total_len / num_aligned_reads :741-748
//   oarfish::EMInfo                  src/util/oarfish_types.rs:408-428
//   oarfish::em::em / em_par / bootstrap   src/em.rs:262 / :320 / :292
//
// Error behaviour: the reference's EM is infallible (it aborts on a panic, Cargo.toml:118); here a
// failed call throws oarfish::OemError carrying the oem_status and message.  Header-only; link
// against liboarfish_em.so.
#pragma once

#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "oarfish_em.h"

namespace oarfish {

struct OemError : std::runtime_error {
    int code;
    OemError(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

inline void check(int rc, const char *where)
{
    if (rc != OEM_OK) throw OemError(rc, std::string(where) + ": " + oem_last_error());
}

enum class Strand : uint8_t { Forward, Reverse, Unknown };

struct AlnInfo { // oarfish_types.rs:330-337
    uint32_t ref_id = 0;
    uint32_t start = 0;
    uint32_t end = 0;
    double prob = 0.0; // always 0.0 in the reference and never read (oarfish_types.rs:352)
    Strand strand = Strand::Forward;
    uint32_t alignment_span() const { return end - start; } // :341-343
};

struct AlignmentFilters {
    bool model_coverage = false; // oarfish_types.rs:792; selects the cov_prob column (em.rs:108)
};

struct TranscriptInfo { // oarfish_types.rs:431-437; the EM reads lenf only under --use-kde
    size_t len = 1;
    double total_weight = 0.0;
    double lenf = 1.0;
};

namespace detail {
struct StoreDeleter {
    void operator()(oem_store *s) const { oem_store_destroy(s); }
};
} // namespace detail

class InMemoryAlignmentStore { // oarfish_types.rs:548-558
public:
    AlignmentFilters filter_opts;
    std::vector<AlnInfo> alignments;
    std::vector<float> as_probabilities;
    std::vector<double> coverage_probabilities;

    explicit InMemoryAlignmentStore(AlignmentFilters fo = {}) : filter_opts(fo), boundaries_{0} {} // :637-649

    // :718-738 (the coverage intervals it also updates belong to the coverage model, out of scope)
    bool add_filtered_group(const std::vector<AlnInfo> &alns, const std::vector<float> &as_probs)
    {
        if (alns.empty()) return false;
        if (alns.size() != as_probs.size()) throw std::invalid_argument("add_filtered_group: sizes differ");
        alignments.insert(alignments.end(), alns.begin(), alns.end());
        as_probabilities.insert(as_probabilities.end(), as_probs.begin(), as_probs.end());
        coverage_probabilities.resize(alignments.size(), 0.0);
        boundaries_.push_back(alignments.size());
        device_.reset();
        return true;
    }
    size_t len() const { return boundaries_.size() - 1; }          // :562-564
    size_t num_aligned_reads() const { return len(); }             // :746-748
    size_t total_len() const { return alignments.size(); }         // :741-743
    // iter(): read i as (begin, end) into the three columns (:651-656)
    std::pair<size_t, size_t> read(size_t i) const { return {boundaries_[i], boundaries_[i + 1]}; }

    // The matrix resident in HBM (uploaded once, kept across em / bootstrap calls like the store
    // stays in RAM across bulk.rs:131-194).
    oem_store *device_store(size_t n_txps, int device = 0) const
    {
        if (!device_ || dev_txps_ != n_txps || dev_id_ != device) {
            std::vector<uint64_t> row_ptr(boundaries_.begin(), boundaries_.end());
            std::vector<uint32_t> tid(alignments.size());
            for (size_t j = 0; j < alignments.size(); ++j) tid[j] = alignments[j].ref_id;
            oem_store *h = nullptr;
            check(oem_store_create(row_ptr.data(), tid.data(), as_probabilities.data(),
                                   filter_opts.model_coverage ? coverage_probabilities.data() : nullptr, // em.rs:108
                                   len(), alignments.size(), (uint32_t)n_txps, device, nullptr, &h),
                  "oem_store_create");
            device_.reset(h);
            dev_txps_ = n_txps;
            dev_id_ = device;
        }
        return device_.get();
    }

private:
    std::vector<size_t> boundaries_; // private in the reference too (:555)
    mutable std::unique_ptr<oem_store, detail::StoreDeleter> device_;
    mutable size_t dev_txps_ = 0;
    mutable int dev_id_ = 0;
};

struct EMInfo { // oarfish_types.rs:408-428
    const InMemoryAlignmentStore *eq_map = nullptr;
    const std::vector<TranscriptInfo> *txp_info = nullptr;
    uint32_t max_iter = 1000;           // prog_opts.rs:532
    double convergence_thresh = 1e-3;   // prog_opts.rs:536
    std::optional<std::vector<double>> init_abundances;
    bool kde_model = false;             // hidden --use-kde: not supported (un-pinned kders crate)
    int device = 0;
};

namespace em {

namespace detail {
inline std::vector<double> run(const EMInfo &emi, uint32_t gate, oem_run_info *info)
{
    if (emi.kde_model) throw std::invalid_argument("kde_model is not supported");
    const size_t T = emi.txp_info->size();
    oem_store *s = emi.eq_map->device_store(T, emi.device);
    std::vector<double> out(T, 0.0);
    check(oem_em_run(s, emi.init_abundances ? emi.init_abundances->data() : nullptr, emi.max_iter,
                     emi.convergence_thresh, gate, out.data(), info),
          "oem_em_run");
    return out;
}
} // namespace detail

// em.rs:262-271: `_nthreads` is ignored there too
inline std::vector<double> em(const EMInfo &em_info, size_t /*nthreads*/, oem_run_info *info = nullptr)
{
    return detail::run(em_info, 50, info); // gate niter > 50 (em.rs:212)
}

// em.rs:320-447
inline std::vector<double> em_par(const EMInfo &em_info, size_t /*nthreads*/, oem_run_info *info = nullptr)
{
    return detail::run(em_info, 1, info); // gate niter > 1 (em.rs:399)
}

// em.rs:292-314; `seed` keys the device RNG (the reference seeds from the OS, em.rs:274),
// `row_weights` (num_boot x n_reads multiplicities) injects the resamples.
inline std::vector<std::vector<double>> bootstrap(const EMInfo &em_info, uint32_t num_boot, size_t /*nthreads*/,
                                                  uint64_t seed = 0, const uint32_t *row_weights = nullptr)
{
    if (em_info.kde_model) throw std::invalid_argument("kde_model is not supported");
    const size_t T = em_info.txp_info->size();
    oem_store *s = em_info.eq_map->device_store(T, em_info.device);
    std::vector<double> flat((size_t)num_boot * T, 0.0);
    check(oem_bootstrap(s, num_boot, seed, row_weights,
                        em_info.init_abundances ? em_info.init_abundances->data() : nullptr, em_info.max_iter,
                        em_info.convergence_thresh, flat.data(), nullptr),
          "oem_bootstrap");
    std::vector<std::vector<double>> out(num_boot);
    for (uint32_t b = 0; b < num_boot; ++b) out[b].assign(flat.begin() + (size_t)b * T, flat.begin() + (size_t)(b + 1) * T);
    return out;
}

} // namespace em
} // namespace oarfish
