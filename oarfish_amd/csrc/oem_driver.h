// oem_driver.h -- what the host-side drivers of the C ABI share (not part of the ABI): the store life cycle
// lives in oem_api.hip, the EM loop in oem_em_driver.hip, the bootstrap chains in oem_bootstrap.hip, the per-cell
// batches in oem_cells.hip and the HIP-event timing entry points in oem_timing.hip.
#pragma once

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "oem_internal.h"

namespace oem {

const char *last_error_text(); // this thread's message (oem_last_error)

int comm_rank(const Comm *c);
int comm_size(const Comm *c);
bool comm_exchanges(const Comm *c);

// OEM_VERBOSE=1: wall-clock breakdown of store creation on stderr (upload / layout diagnostics)
struct StageTimer {
    bool on = getenv("OEM_VERBOSE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[oem] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

int ensure_device(int device);

template <typename T>
int dev_alloc(T **p, size_t n, uint64_t *acct)
{
    *p = nullptr;
    const size_t bytes = (n ? n : 1) * sizeof(T);
    OEM_HIP(hipMalloc((void **)p, bytes));
    if (acct) *acct += bytes;
    return OEM_OK;
}

// oem_api.hip: checks of the caller's arrays, store creation and destruction
int validate_csr(const uint64_t *row_ptr, const uint32_t *tid, uint64_t n_reads, uint64_t nnz, uint32_t n_txps);
uint64_t zero_nan_rows(const uint64_t *row_ptr, const double *cov, uint64_t n_reads, uint64_t nnz, std::vector<double> *fixed);

// Per-cell batches: cell c's transcripts are relabelled to [c * cell_txps, (c + 1) * cell_txps) -- on
// the device, after the upload, instead of in a second host copy of the transcript ids.
struct CellRelabel {
    const uint64_t *cell_row_off;
    uint32_t n_cells;
    uint32_t cell_txps;
};
int create_store_impl(const uint64_t *row_ptr, const uint32_t *tid, const float *as_prob, const double *cov_prob,
                      uint64_t n_reads, uint64_t nnz, uint32_t n_txps, int device, const oem_store_opts *opts, oem_store *s,
                      const CellRelabel *relabel = nullptr);
void free_store(oem_store *s);

// oem_em_driver.hip: one EM run with the loop state on the device
struct RunArgs {
    const double *init = nullptr; // host, n_txps, or NULL
    const uint32_t *d_row_w = nullptr; // device multiplicities or NULL
    uint64_t row_begin = 0, row_end = 0;
    uint64_t total_reads = 0; // em.rs:154 total_weight
    uint32_t max_iter = 1000;
    double conv_thresh = 1e-3;
    uint32_t min_iter_gate = 50;
};
bool use_tiled(const oem_store *s, const RunArgs &a);
int enqueue_pass(oem_store *s, const RunArgs &a, const EmState *state);
int prepare_row_w(oem_store *s, const RunArgs &a);
int enqueue_iteration(oem_store *s, const RunArgs &a, const EmParams &p);
int run_em_device(oem_store *s, const RunArgs &a, oem_run_info *info);
bool deferred_reldiff_ok(const oem_store *s, const RunArgs &a);
int ensure_deferred(oem_store *s);
int enqueue_deferred_pass(oem_store *s, const RunArgs &a, const EmParams &p, double *const bufs[3], uint64_t i);
int copy_counts_out(oem_store *s, double *out);
int ensure_row_w(oem_store *s);

// A chunk of the loop as a hipGraph -- an experiment that stays reachable (OEM_GRAPH=1 in the test-only
// library), not the product path: replaying 16 iterations from an instantiated graph instead of launching
// their kernels one by one changes nothing measurable on MI355X (10 M reads: 0.2240 vs 0.2239 ms per
// iteration; 1 M reads: 38.1 vs 38.1-38.8 us, profiles/r03_notes.md) -- dependent launches on one stream
// already follow each other within ~1 us, and the host is far ahead of the device.  Nothing in an
// iteration carries a per-launch value (loop state, stopping rule and the peer-to-peer epoch live on the
// device), so one captured chunk serves a whole run.
constexpr uint32_t kGraphIters = 16;

struct ChunkGraph {
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    ChunkGraph() = default;
    ChunkGraph(const ChunkGraph &) = delete;
    ChunkGraph &operator=(const ChunkGraph &) = delete;
    ~ChunkGraph()
    {
        if (ge) hipGraphExecDestroy(ge);
        if (g) hipGraphDestroy(g);
    }
    bool ready() const { return ge != nullptr; }
};

// Captures `body` (kernel launches on `st` only) n times.  Returns OEM_OK with !out->ready() when the
// runtime declines (the caller then launches directly); an error only when `body` itself fails.
template <typename F>
int capture_chunk(hipStream_t st, uint32_t n, F &&body, ChunkGraph *out)
{
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        return OEM_OK;
    }
    int rc = OEM_OK;
    for (uint32_t k = 0; k < n && rc == OEM_OK; ++k) rc = body();
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &g);
    if (rc != OEM_OK || e != hipSuccess || !g) {
        if (g) hipGraphDestroy(g);
        (void)hipGetLastError();
        return rc;
    }
    hipGraphExec_t ge = nullptr;
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess || !ge) {
        hipGraphDestroy(g);
        (void)hipGetLastError();
        return OEM_OK;
    }
    out->g = g;
    out->ge = ge;
    return OEM_OK;
}

// RCCL calls are not captured (a row shard that exchanges through RCCL launches directly); the
// peer-to-peer exchange is plain kernels.
inline bool graph_ok(const oem_store *s, size_t exchange_count = 0)
{
    return knob("OEM_GRAPH", 0) != 0 &&
           !comm_exchange_is_unconditional(s->comm, exchange_count ? exchange_count : s->csr.n_txps);
}

// oem_bootstrap.hip
int ensure_batch(oem_store *s, int chain);
bool can_batch(const oem_store *s);
int agree_any(oem_store *s, bool *flag);

// oem_cells.hip: timing of the last oem_em_run_cells call of this thread
void cells_last_timing(double *loop_ms, uint64_t *batched_passes);

} // namespace oem
