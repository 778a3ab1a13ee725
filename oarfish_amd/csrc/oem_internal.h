// oem_internal.h -- shared declarations of the MI355X EM engine (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/oarfish_em.h"
#include "oem_layout.h"

namespace oem {

// ---------------------------------------------------------------------------
// error plumbing: every ABI entry point funnels through these
// ---------------------------------------------------------------------------
int fail(int code, const char *fmt, ...);

#define OEM_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return ::oem::fail(_e == hipErrorOutOfMemory ? OEM_ERR_OOM : OEM_ERR_HIP,      \
                               "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                               __FILE__, __LINE__);                                        \
    } while (0)

// The ABI promises never to throw across the boundary: every `extern "C" int` body sits between these.
#define OEM_API_BEGIN try {
#define OEM_API_END(name)                                                                          \
    }                                                                                              \
    catch (const std::bad_alloc &) { return ::oem::fail(OEM_ERR_OOM, "%s: host allocation failed", name); } \
    catch (const std::exception &e) { return ::oem::fail(OEM_ERR_STATE, "%s: %s", name, e.what()); } \
    catch (...) { return ::oem::fail(OEM_ERR_STATE, "%s: unknown C++ exception", name); }

#define OEM_TRY(expr)                 \
    do {                              \
        int _rc = (expr);             \
        if (_rc != OEM_OK) return _rc; \
    } while (0)

// Tuning / test switch (oem_knobs.cpp): the product library always returns `dflt`; only the
// test-only library built with -DOEM_TESTING reads the environment variable `name`.
long knob(const char *name, long dflt);

// ---------------------------------------------------------------------------
// Device-resident loop state of one EM problem (em.rs:169-170 rel_diff, niter).
// Lives in HBM so that the host never sits inside the iteration loop: the
// rel-diff kernel takes the reference's stopping decision on the device and
// every later launch of the same run turns into a no-op once `done` is set.
// ---------------------------------------------------------------------------
struct EmState {
    unsigned long long rel_bits; // running max of rel-diff this pass, as the bit pattern of a non-negative f64
    double last_rel;             // rel_diff of the last completed loop pass
    uint32_t niter;              // em.rs:170
    uint32_t n_passes;           // E/M passes executed so far
    uint32_t done;               // loop has exited (break or niter == max_iter)
    uint32_t converged;          // exit was through `break`
    uint32_t blocks_arrived;     // last-block election counter of the rel-diff kernel
    uint32_t pad[3];
};
static_assert(sizeof(EmState) == 48, "EmState layout");

// Parameters that do not change during a run.
struct EmParams {
    uint32_t n_txps;
    uint32_t max_iter;
    uint32_t min_iter_gate;
    double conv_thresh;
};

// ---------------------------------------------------------------------------
// The alignment store as laid out in HBM.
//
// v1 layout ("CSR"): row_ptr (u32 when nnz < 2^32, else u64), tid u32[nnz],
// w f32[nnz] (== as_prob; exact) or w64 f64[nnz] (== (f64)as_prob * cov_prob when
// the coverage model is on, em.rs:107-111).
// ---------------------------------------------------------------------------
struct DeviceCsr {
    uint64_t n_reads = 0;
    uint64_t nnz = 0;
    uint32_t n_txps = 0;
    bool wide_ptr = false; // row_ptr is u64
    bool w_is_f64 = false; // coverage model on
    void *row_ptr = nullptr;
    uint32_t *tid = nullptr;
    float *w32 = nullptr;
    double *w64 = nullptr;
};

// The tiled layout (oem_layout.h) resident in HBM.
struct DeviceTiled {
    bool present = false;
    bool built_on_device = false; // oem_layout_device.hip (else the host builder + upload)
    uint32_t n_tiles = 0;
    uint32_t n_buckets = 0;
    uint32_t win_cap = kWin; // kWin or kWinWide (sparse stores)
    uint32_t tile_rows = kTileRows; // reads per tile at most (<= 512: the batched tile kernel runs three per CU)
    uint64_t n_rows = 0;    // non-empty reads == length of perm
    uint64_t n_local = 0;
    uint64_t n_remote = 0;
    TileDesc *tiles = nullptr;
    uint32_t *perm = nullptr;
    uint32_t *codes = nullptr;
    float *w32 = nullptr;
    double *w64 = nullptr;
    uint32_t *r_tid = nullptr;
    float *r_w32 = nullptr;
    double *r_w64 = nullptr;
    uint16_t *r_row = nullptr;
    uint32_t *r_slot = nullptr;   // builders' form only: replaced by the per-tile slot table `sd`
    uint32_t *r_pk = nullptr;     // packed records (transcript - problem base | read << 22), when `packed`
    uint32_t *sd = nullptr;       // slot table (oem_layout_pack.hip)
    uint32_t n_sd = 0;
    bool packed = false;
    uint32_t problem_size = 0;    // transcripts per EM problem the records were packed against (0: one problem)
    uint16_t *q_dst = nullptr;
    uint32_t *bucket_base = nullptr;
    std::vector<uint32_t> h_bucket_base;
    double *queue = nullptr;      // n_remote f64: increments of the remote alignments, bucket-major
    uint32_t *row_w_perm = nullptr; // bootstrap multiplicities in permuted read order
    // dictionary-coded local weights (oem_layout_dict.hip), when the store has at most 256 distinct ones
    uint32_t dict_n = 0;           // entries of the table (0: not coded, the kernels read w32)
    bool dict_fused = false;       // <= 128 entries: the index sits in the spare bits of the window codes, no widx
    bool dict_words = false;       // 257 .. 1024 entries: 16-bit indices, two per word, in the geometry of the codes
    float *dict = nullptr;         // 1024 floats, ascending, [0] = 0.0
    uint32_t *widx = nullptr;      // four one-byte indices per word, SELL layout of the tiles (dict_words: see above)
    uint32_t *i_base = nullptr;    // n_tiles + 1: first index row of each tile
    uint8_t *r_wi = nullptr;       // n_remote: table index of each remote record's weight
};

// ---------------------------------------------------------------------------
// Batched bootstrap: kBatch replicates ("slots") share each pass over the matrix
// (oem_batch_kernels.hip).  Per-slot loop state, walked on the device:
// RUNNING -> FINAL (small abundances read as 0, one more pass) -> FINISHED.
// ---------------------------------------------------------------------------
// Whether the remote records of a store larger than the Infinity Cache are streamed non-temporally like its slices
// (1) or left to the caches (0): A/B switch of the tile kernels.
#ifndef OEM_REC_NT
#define OEM_REC_NT 0
#endif
#ifndef OEM_QUEUE_NT
#define OEM_QUEUE_NT 1 // k_em_tile writes its queue entries non-temporally (A/B)
#endif
#ifndef OEM_KBATCH
#define OEM_KBATCH 4
#endif
constexpr int kBatch = OEM_KBATCH; // slots of one chain: the replicates that share a pass over the matrix
constexpr int kChains = 2;         // chains running side by side, each with its own stream and buffers
enum : uint32_t { kPhaseRunning = 0, kPhaseFinal = 1, kPhaseFinished = 2 };

struct BatchState {
    unsigned long long rel_bits;
    double last_rel;
    uint32_t niter;
    uint32_t n_passes;
    uint32_t converged;
    uint32_t blocks_arrived; // only [0] is used
    uint32_t phase;
    uint32_t reserved0;
    uint32_t pad[2];
};
static_assert(sizeof(BatchState) == 48, "BatchState layout");

// k_reldiff_b keeps its workgroups' maxima in kBatchRelSlots x kBatch words, not in the slots' four state words: ~200
// workgroups x 4 atomics on ONE line ran at the rate of one hot address (0.08 G/s: 12 of the kernel's 16 us).
constexpr uint32_t kBatchRelSlots = 32;
struct BatchBuffers {
    hipStream_t stream = nullptr; // the chain's own stream (chain 0: the store's)
    uint32_t *d_row_w = nullptr;  // n_reads u32: the replicate being handed to a slot, caller order
    double *theta = nullptr;   // [T][kBatch]
    double *cnt = nullptr;     // [T][kBatch]  (flushes of the tile kernel and of the fold)
    double *out = nullptr;     // [kBatch][T]
    double *queue = nullptr;   // [n_remote][kBatch]
    BatchState *state = nullptr;
    unsigned long long *rel_slots = nullptr; // [kBatchRelSlots][kBatch] running maxima of k_reldiff_b (zero between passes)
    uint8_t *row_w = nullptr;  // [rows][kBatch], tile order: one byte per slot
    uint32_t *overflow = nullptr;
    BatchState *h_state = nullptr; // pinned
    double *h_out = nullptr;       // pinned [kBatch][T]
};

// Per-cell batch: the cells are laid out as one store over a concatenated transcript
// space (problem p owns transcripts [p*T, (p+1)*T)); every problem carries its own loop
// state and walks RUNNING -> FINAL -> FINISHED on the device (oem_multi_kernels.hip).
struct MultiBuffers {
    uint32_t n_problems = 0;
    uint32_t problem_size = 0;
    BatchState *state = nullptr;   // [n_problems]
    BatchState *h_state = nullptr; // host copy
    double *out = nullptr;         // [n_problems * problem_size]
    uint32_t *n_unfinished = nullptr; // device counter
    // live work of the batch, compacted by the host's look at the device state (k_multi_compact): the tiles and
    // the remote buckets of the cells that are still running, in their original order
    uint32_t *live_tiles = nullptr;   // [n_tiles]
    uint32_t *live_buckets = nullptr; // [n_buckets]
    uint32_t *d_live_counts = nullptr; // device: {n_live_tiles, n_live_buckets}
    uint32_t n_live_tiles = 0, n_live_buckets = 0; // host copies: the grids of the next launches
    bool live_valid = false;
    // Per-cell transcript compaction (oem_api.hip: compact_cells): a cell's transcripts are renumbered to the rank of
    // each among the transcripts that occur in the cell, and every cell owns txps_eff (= the largest such count)
    // consecutive ids of the store.  rank[c * txps_full + t] = rank of transcript t in cell c, or kNoRank.
    uint32_t *rank = nullptr;  // [n_problems * txps_full], device; NULL: no compaction (ids are c * T + t)
    uint32_t txps_full = 0;    // the caller's n_txps
    uint32_t txps_eff = 0;     // transcripts per cell in the store
};
constexpr uint32_t kNoRank = 0xffffffffu;

struct Comm; // oem_comm.cpp

} // namespace oem

struct oem_store {
    int device = 0;
    hipStream_t stream = nullptr;
    oem::DeviceCsr csr;
    oem::DeviceTiled tiled;
    // working set of one EM problem
    double *theta = nullptr;             // prev_counts, n_txps f64
    double *cnt = nullptr;               // curr_counts (rank-local partial sums until all-reduced)
    double *third = nullptr;             // the third count vector of the deferred stopping rule (run_em_deferred), lazily
    unsigned long long *rel_slots = nullptr; // its kRelSlots running maxima
    oem::EmState *d_state = nullptr;
    oem::EmState *h_state = nullptr;     // pinned
    uint32_t *d_row_w = nullptr;         // bootstrap multiplicities, n_reads u32
    double *h_pinned = nullptr;          // pinned staging, n_txps f64
    oem::BatchBuffers batch[oem::kChains]; // lazily allocated by the batched bootstrap
    oem::MultiBuffers multi;             // per-cell batches
    uint32_t bootstrap_first_replica = 0; // OEM_OPT_BOOTSTRAP_FIRST_REPLICA
    bool batch_bootstrap = true;         // OEM_OPT_BATCH_BOOTSTRAP (kChains chains of kBatch replicates per pass when applicable)
    // multi-GPU
    oem::Comm *comm = nullptr;
    uint64_t global_n_reads = 0;
    uint64_t global_row_offset = 0;
    // bookkeeping
    uint64_t hbm_bytes = 0;
    std::mutex mu;
};

namespace oem {

// kernels (oem_kernels.hip) -------------------------------------------------

// One E/M pass over rows [row_begin, row_end) of the store: cnt += E/M(theta).
// `cnt` must be zero on entry (the rel-diff kernel leaves it so).  With a
// non-null `state` the launch is a no-op once state->done is set.
int launch_em_pass(oem_store *s, const double *theta, double *cnt, const EmState *state,
                   const uint32_t *row_w, uint64_t row_begin, uint64_t row_end);

// rel-diff + swap + clear + stopping decision (em.rs:194-218 / :379-405):
// prev <- curr, curr <- 0, state updated by the last block to arrive.
int launch_reldiff_swap_clear(oem_store *s, double *prev, double *curr, EmState *state, EmParams p);

// em.rs:238-242: prev < 1e-5 -> 0; also zeroes curr for the final pass.
int launch_zero_small(oem_store *s, double *prev, double *curr, uint32_t n_txps);

// Tiled E/M pass over the whole store (oem_layout.h): tile kernel + remote-bucket kernel.
// row_w is in the caller's read order; it is permuted into tile order first.
// oem_layout_device.hip: the tiled layout built on the device from the resident CSR
int build_tiled_layout_device(oem_store *s, uint32_t problem_size, uint32_t win_cap, uint32_t tile_rows, bool *built);
int build_weight_dictionary(oem_store *s); // oem_layout_dict.hip
// oem_layout_pack.hip: slot table + packed remote records, after either builder
int pack_remote_records(oem_store *s, uint32_t problem_size, bool keep_unpacked);
// The stopping rule one pass behind (oem_em_driver.hip: run_em_deferred).  The pass that reads theta_i = cnt_{i-1} also
// takes the rel-diff of iteration i - 1 along: every tile workgroup compares its share of the transcripts between
// `prev` (theta_{i-1}) and theta_i, keeps its maximum in one of kRelSlots slots and zeroes its share of `prev` (which
// becomes the accumulator of pass i + 1); the first workgroup of the fold that follows applies the rule
// (em.rs:212-218) to the slots' maximum.  `decide` = 0 (pass 0: nothing to decide) or 1 + the index of the buffer that
// holds theta_i -- the final abundances if the rule says stop.
constexpr uint32_t kRelSlots = 64;
struct DeferredRelDiff {
    double *prev;              // theta_{i-1}; NULL on pass 0
    unsigned long long *slots; // kRelSlots running maxima (bit patterns of non-negative doubles)
    EmState *state;
    EmParams p;
    uint32_t decide;
};
int launch_em_pass_tiled(oem_store *s, const double *theta, double *cnt, const EmState *state,
                         const uint32_t *row_w_perm, const BatchState *problems = nullptr,
                         uint32_t problem_size = 0, bool skip_fold = false, const DeferredRelDiff *rd = nullptr);
// the last iteration of a deferred run that reaches max_iter: rel-diff of `prev` against `cur`, `prev` zeroed, the rule
// applied -- no pass (k_deferred_sweep, oem_tile_kernels.hip)
int launch_deferred_sweep(oem_store *s, double *prev, const double *cur, const DeferredRelDiff &rd);
// the start of a deferred run in one launch: theta_0 filled (or left: init_abundances), the other vectors, slots, state zeroed
int launch_deferred_init(oem_store *s, double *const bufs[3], double avg, bool fill);


// per-cell batches (oem_multi_kernels.hip)
int launch_multi_init(oem_store *s, double *theta, const uint64_t *d_problem_reads, const MultiBuffers &mb);
int launch_multi_expand(oem_store *s, const MultiBuffers &mb, double *full /* [n_problems * txps_full] */);
int launch_multi_fold_reldiff(oem_store *s, double *theta, double *cnt, const MultiBuffers &mb, EmParams p);
int launch_multi_reldiff(oem_store *s, double *theta, double *cnt, const MultiBuffers &mb, EmParams p);
int multi_compact_live(oem_store *s, MultiBuffers &mb); // rebuilds the live lists (synchronises the stream)
int launch_permute_row_w(oem_store *s, const uint32_t *row_w, uint32_t *row_w_perm);

// batched bootstrap (oem_batch_kernels.hip)
int launch_batch_pass(oem_store *s, const BatchBuffers &bb);
int launch_batch_reldiff(oem_store *s, const BatchBuffers &bb, EmParams p);
int launch_batch_reset_slot(oem_store *s, const BatchBuffers &bb, const double *d_init, double avg, uint32_t slot);
int launch_batch_pack_row_w(oem_store *s, const uint32_t *d_row_w, const BatchBuffers &bb, uint32_t slot,
                            uint32_t *d_overflow);

int launch_aux_counts(oem_store *s, uint32_t *d_unique, uint32_t *d_total);
int launch_assignment_probs(oem_store *s, const double *d_counts, double display_thresh, double *d_out);
int launch_fill(oem_store *s, double *p, double v, uint64_t n);
int launch_bootstrap_weights(oem_store *s, uint32_t *row_w, uint64_t n_local, uint64_t local_off,
                             uint64_t n_global, uint64_t seed, uint32_t replica, hipStream_t stream = nullptr);

// RCCL (oem_comm.cpp) ---------------------------------------------------------
int comm_allreduce_sum_f64(Comm *c, const double *send, double *recv, size_t count, hipStream_t st,
                           const EmState *state = nullptr);
bool comm_fuses_reldiff(const Comm *c, uint32_t n_txps);
int comm_reldiff_fused(Comm *c, double *prev, double *curr, EmState *state, EmParams prm, hipStream_t st);
bool comm_exchange_is_unconditional(const Comm *c, uint32_t n_txps);
int comm_check(Comm *c, hipStream_t st);

} // namespace oem
