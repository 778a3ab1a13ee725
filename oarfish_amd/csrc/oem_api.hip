// oem_api.hip -- the C ABI of include/oarfish_em.h: store management and the
// device-resident EM / bootstrap / per-cell drivers.
//
// Reference call sites this replaces (COMBINE-lab/oarfish v0.10.3):
//   bulk.rs:155-159   em::em / em::em_par      -> oem_em_run
//   bulk.rs:178-194   em::bootstrap            -> oem_bootstrap
//   single_cell.rs:139-160  per-cell em::em    -> oem_em_run_cells
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "oem_internal.h"

namespace oem {

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char t_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
    return code;
}

int comm_rank(const Comm *c);
int comm_size(const Comm *c);
bool comm_exchanges(const Comm *c);

namespace {

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

int ensure_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(OEM_ERR_NO_DEVICE,
                    "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n)
        return fail(OEM_ERR_ARG, "device ordinal %d out of range [0,%d)", device, n);
    OEM_HIP(hipSetDevice(device));
    return OEM_OK;
}

template <typename T>
int dev_alloc(T **p, size_t n, uint64_t *acct)
{
    *p = nullptr;
    const size_t bytes = (n ? n : 1) * sizeof(T);
    OEM_HIP(hipMalloc((void **)p, bytes));
    if (acct) *acct += bytes;
    return OEM_OK;
}

// Range checks of the caller's CSR, split over a few host threads (80 M alignments: 18 ms serial, the
// same order as the device layout build).  Reports the first offending index.
int validate_csr(const uint64_t *row_ptr, const uint32_t *tid, uint64_t n_reads, uint64_t nnz,
                 uint32_t n_txps)
{
    if (row_ptr[0] != 0) return fail(OEM_ERR_ARG, "row_ptr[0] must be 0 (oarfish_types.rs:645)");
    constexpr uint64_t kNone = ~0ull;
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (nt < 1 || n_reads + nnz < (1u << 20)) nt = 1;
    std::vector<uint64_t> bad_row(nt, kNone), bad_tid(nt, kNone);
    auto scan = [&](unsigned k) {
        const uint64_t r0 = n_reads * k / nt, r1 = n_reads * (k + 1) / nt;
        for (uint64_t i = r0; i < r1; ++i)
            if (row_ptr[i + 1] < row_ptr[i]) { bad_row[k] = i; break; }
        const uint64_t a0 = nnz * k / nt, a1 = nnz * (k + 1) / nt;
        for (uint64_t j = a0; j < a1; ++j)
            if (tid[j] >= n_txps) { bad_tid[k] = j; break; }
    };
    if (nt == 1) {
        scan(0);
    } else {
        std::vector<std::thread> th;
        for (unsigned k = 0; k < nt; ++k) th.emplace_back(scan, k);
        for (auto &t : th) t.join();
    }
    for (unsigned k = 0; k < nt; ++k)
        if (bad_row[k] != kNone)
            return fail(OEM_ERR_ARG, "row_ptr is not non-decreasing at read %llu", (unsigned long long)bad_row[k]);
    if (row_ptr[n_reads] != nnz)
        return fail(OEM_ERR_ARG, "row_ptr[n_reads]=%llu differs from nnz=%llu",
                    (unsigned long long)row_ptr[n_reads], (unsigned long long)nnz);
    for (unsigned k = 0; k < nt; ++k)
        if (bad_tid[k] != kNone)
            return fail(OEM_ERR_ARG, "tid[%llu]=%u is not below n_txps=%u", (unsigned long long)bad_tid[k],
                        tid[bad_tid[k]], n_txps);
    return OEM_OK;
}

// A NaN in the coverage column (normalize_read_probs lets the 0/0 of a zero-span alignment through,
// normalize_probability.rs:58) makes the read's denominator NaN in the reference, the test
// `denom > 1e-30` fails and the read contributes nothing (em.rs:115).  The kernels are branch-free
// (x * (c / denom), inv = 0 for a dropped read), where a NaN weight would survive as NaN * 0: such reads are
// therefore given all-zero coverage at upload, which drops them the same way.  Returns the reads found;
// `fixed` is filled (a copy of the column) only when there are any.
uint64_t zero_nan_rows(const uint64_t *row_ptr, const double *cov, uint64_t n_reads, uint64_t nnz,
                       std::vector<double> *fixed)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (nt < 1 || nnz < (1u << 20)) nt = 1;
    std::vector<std::vector<uint64_t>> bad(nt);
    auto scan = [&](unsigned k) {
        const uint64_t r0 = n_reads * k / nt, r1 = n_reads * (k + 1) / nt;
        for (uint64_t r = r0; r < r1; ++r)
            for (uint64_t j = row_ptr[r]; j < row_ptr[r + 1]; ++j)
                if (cov[j] != cov[j]) { bad[k].push_back(r); break; }
    };
    if (nt == 1) {
        scan(0);
    } else {
        std::vector<std::thread> th;
        for (unsigned k = 0; k < nt; ++k) th.emplace_back(scan, k);
        for (auto &t : th) t.join();
    }
    uint64_t n_bad = 0;
    for (auto &v : bad) n_bad += v.size();
    if (n_bad == 0) return 0;
    fixed->assign(cov, cov + nnz);
    for (auto &v : bad)
        for (uint64_t r : v)
            for (uint64_t j = row_ptr[r]; j < row_ptr[r + 1]; ++j) (*fixed)[j] = 0.0;
    return n_bad;
}

struct RunArgs {
    const double *init = nullptr; // host, n_txps, or NULL
    const uint32_t *d_row_w = nullptr; // device multiplicities or NULL
    uint64_t row_begin = 0, row_end = 0;
    uint64_t total_reads = 0; // em.rs:154 total_weight
    uint32_t max_iter = 1000;
    double conv_thresh = 1e-3;
    uint32_t min_iter_gate = 50;
};

bool use_tiled(const oem_store *s, const RunArgs &a)
{
    return s->tiled.present && a.row_begin == 0 && a.row_end == s->csr.n_reads;
}

// E/M pass theta -> cnt with whichever layout covers the request
int enqueue_pass(oem_store *s, const RunArgs &a, const EmState *state)
{
    if (use_tiled(s, a))
        return launch_em_pass_tiled(s, s->theta, s->cnt, state, a.d_row_w ? s->tiled.row_w_perm : nullptr);
    return launch_em_pass(s, s->theta, s->cnt, state, a.d_row_w, a.row_begin, a.row_end);
}

// bootstrap multiplicities arrive in the caller's read order; the tiles want them permuted
int prepare_row_w(oem_store *s, const RunArgs &a)
{
    if (a.d_row_w && use_tiled(s, a)) return launch_permute_row_w(s, a.d_row_w, s->tiled.row_w_perm);
    return OEM_OK;
}

// one loop iteration on the stream: E/M pass, (all-reduce), rel-diff/swap/clear
int enqueue_iteration(oem_store *s, const RunArgs &a, const EmParams &p)
{
    OEM_TRY(enqueue_pass(s, a, s->d_state));
    if (comm_exchanges(s->comm)) {
        // peer to peer: the sum over the shards happens inside the rel-diff kernel (oem_p2p.hip)
        if (comm_fuses_reldiff(s->comm, p.n_txps))
            return comm_reldiff_fused(s->comm, s->theta, s->cnt, s->d_state, p, s->stream);
        OEM_TRY(comm_allreduce_sum_f64(s->comm, s->cnt, s->cnt, p.n_txps, s->stream, s->d_state));
    }
    OEM_TRY(launch_reldiff_swap_clear(s, s->theta, s->cnt, s->d_state, p));
    return OEM_OK;
}

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
bool graph_ok(const oem_store *s, size_t exchange_count = 0)
{
    return knob("OEM_GRAPH", 0) != 0 &&
           !comm_exchange_is_unconditional(s->comm, exchange_count ? exchange_count : s->csr.n_txps);
}

// em.rs:144-255 / :320-447 with the loop state on the device.  On return the
// final counts are in s->cnt (device); *info filled from the device state.
int run_em_device(oem_store *s, const RunArgs &a, oem_run_info *info)
{
    const uint32_t T = s->csr.n_txps;
    EmParams p{T, a.max_iter, a.min_iter_gate, a.conv_thresh};

    if (a.init) {
        OEM_HIP(hipMemcpyAsync(s->theta, a.init, sizeof(double) * T, hipMemcpyHostToDevice, s->stream));
    } else {
        const double avg = (double)a.total_reads / (double)T; // em.rs:165
        OEM_TRY(launch_fill(s, s->theta, avg, T));
    }
    OEM_HIP(hipMemsetAsync(s->cnt, 0, sizeof(double) * T, s->stream));
    OEM_HIP(hipMemsetAsync(s->d_state, 0, sizeof(EmState), s->stream));
    std::memset(s->h_state, 0, sizeof(EmState));
    OEM_TRY(prepare_row_w(s, a));

    // The stopping rule cannot fire before niter > gate, so the first look at
    // the device state is due after gate+2 passes; afterwards every `kChunk`.
    // (with RCCL every launch of a finished run still costs a real all-reduce of zeros, so a row shard
    // that exchanges through it looks at the state every 4 iterations; the peer-to-peer exchange skips
    // itself on the device)
    uint64_t launched = 0;
    const uint64_t kChunk = comm_exchange_is_unconditional(s->comm, T) ? 4 : 16;
    ChunkGraph cg; // kGraphIters iterations, replayed (runs too short to repay the capture launch directly)
    if (graph_ok(s) && a.max_iter >= 4 * kGraphIters)
        OEM_TRY(capture_chunk(s->stream, kGraphIters, [&]() { return enqueue_iteration(s, a, p); }, &cg));
    while (launched < a.max_iter) {
        uint64_t chunk = launched == 0 ? (uint64_t)a.min_iter_gate + 2 : kChunk; // (a gate of u32::MAX must not wrap)
        if (chunk > a.max_iter - launched) chunk = a.max_iter - launched;
        if (chunk > 4096) chunk = 4096; // bound the work queued between two looks at the device state
        if (cg.ready()) {
            // whole graphs: the iterations launched beyond max_iter are no-ops (the loop ends itself on the device)
            chunk = (chunk + kGraphIters - 1) / kGraphIters * kGraphIters;
            for (uint64_t k = 0; k < chunk; k += kGraphIters) OEM_HIP(hipGraphLaunch(cg.ge, s->stream));
        } else {
            for (uint64_t k = 0; k < chunk; ++k) OEM_TRY(enqueue_iteration(s, a, p));
        }
        launched += chunk;
        OEM_HIP(hipMemcpyAsync(s->h_state, s->d_state, sizeof(EmState), hipMemcpyDeviceToHost, s->stream));
        OEM_HIP(hipStreamSynchronize(s->stream));
        OEM_TRY(comm_check(s->comm, s->stream));
        if (s->h_state->done) break;
    }

    OEM_TRY(launch_zero_small(s, s->theta, s->cnt, T));                                  // em.rs:238-242
    OEM_TRY(enqueue_pass(s, a, nullptr));                                                 // em.rs:245-252
    if (comm_exchanges(s->comm))
        OEM_TRY(comm_allreduce_sum_f64(s->comm, s->cnt, s->cnt, T, s->stream));
    if (info) {
        info->niter = s->h_state->niter;
        info->n_passes = s->h_state->n_passes + 1;
        info->converged = s->h_state->converged;
        info->reserved = 0;
        info->rel_diff = s->h_state->last_rel;
    }
    return OEM_OK;
}

int copy_counts_out(oem_store *s, double *out)
{
    const uint32_t T = s->csr.n_txps;
    OEM_HIP(hipMemcpyAsync(s->h_pinned, s->cnt, sizeof(double) * T, hipMemcpyDeviceToHost, s->stream));
    OEM_HIP(hipStreamSynchronize(s->stream));
    OEM_TRY(comm_check(s->comm, s->stream));
    std::memcpy(out, s->h_pinned, sizeof(double) * T);
    return OEM_OK;
}

int ensure_row_w(oem_store *s)
{
    if (!s->d_row_w) OEM_TRY(dev_alloc(&s->d_row_w, s->csr.n_reads, &s->hbm_bytes));
    return OEM_OK;
}

int ensure_batch(oem_store *s, int chain)
{
    BatchBuffers &b = s->batch[chain];
    if (b.theta) return OEM_OK;
    const size_t T = s->csr.n_txps;
    if (chain == 0) b.stream = s->stream;
    else OEM_HIP(hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking));
    OEM_TRY(dev_alloc(&b.d_row_w, s->csr.n_reads, &s->hbm_bytes));
    OEM_TRY(dev_alloc(&b.theta, T * kBatch, &s->hbm_bytes));
    OEM_TRY(dev_alloc(&b.cnt, T * kBatch, &s->hbm_bytes));
    OEM_TRY(dev_alloc(&b.out, T * kBatch, &s->hbm_bytes));
    OEM_TRY(dev_alloc(&b.queue, (size_t)s->tiled.n_remote * kBatch, &s->hbm_bytes));
    OEM_TRY(dev_alloc(&b.state, kBatch, &s->hbm_bytes));
    OEM_TRY(dev_alloc(&b.row_w, (size_t)s->tiled.n_rows * kBatch + 16, &s->hbm_bytes));
    OEM_HIP(hipMemsetAsync(b.row_w, 0, (size_t)s->tiled.n_rows * kBatch + 16, b.stream));
    // a slot that is never handed a replicate (n_boot < kBatch, the tail of a chain) is still swept by the
    // tile kernel's four-slot epoch: its columns must hold zeros, not whatever hipMalloc returned
    OEM_HIP(hipMemsetAsync(b.theta, 0, sizeof(double) * T * kBatch, b.stream));
    OEM_HIP(hipMemsetAsync(b.cnt, 0, sizeof(double) * T * kBatch, b.stream));
    OEM_HIP(hipMemsetAsync(b.out, 0, sizeof(double) * T * kBatch, b.stream));
    OEM_HIP(hipMemsetAsync(b.queue, 0, sizeof(double) * (size_t)s->tiled.n_remote * kBatch, b.stream));
    OEM_TRY(dev_alloc(&b.overflow, 1, &s->hbm_bytes));
    OEM_HIP(hipHostMalloc((void **)&b.h_state, sizeof(BatchState) * kBatch, hipHostMallocDefault));
    OEM_HIP(hipHostMalloc((void **)&b.h_out, sizeof(double) * T * kBatch, hipHostMallocDefault));
    return OEM_OK;
}

// The batch kernel takes narrow windows and byte multiplicities (f32 or f64 weights).
bool can_batch(const oem_store *s)
{
    return s->tiled.present && s->tiled.n_tiles > 0 && s->tiled.win_cap <= kWin;
}

// A decision that selects which collectives a row-sharded run issues must be the same on every
// rank: flag = 1 on any rank => 1 on all (one tiny all-reduce; a no-op without a communicator).
int agree_any(oem_store *s, bool *flag)
{
    if (!comm_exchanges(s->comm)) return OEM_OK;
    double *d = s->cnt; // scratch: the count vector is rebuilt by every run
    const double v = *flag ? 1.0 : 0.0;
    OEM_HIP(hipMemcpyAsync(d, &v, sizeof(double), hipMemcpyHostToDevice, s->stream));
    OEM_TRY(comm_allreduce_sum_f64(s->comm, d, d, 1, s->stream));
    double r = 0.0;
    OEM_HIP(hipMemcpyAsync(&r, d, sizeof(double), hipMemcpyDeviceToHost, s->stream));
    OEM_HIP(hipStreamSynchronize(s->stream));
    OEM_TRY(comm_check(s->comm, s->stream)); // a timed-out exchange would leave ranks disagreeing on the flag
    *flag = r != 0.0;
    return OEM_OK;
}

// What the chains of one oem_bootstrap call share: the replicates are handed out from one counter.
struct BootJob {
    uint32_t n_boot = 0;
    uint64_t seed = 0;
    const uint32_t *row_w_all = nullptr; // host, n_boot x R, or NULL
    const double *d_init = nullptr;      // device, or NULL => uniform
    uint32_t max_iter = 0;
    double conv_thresh = 0.0;
    double *out = nullptr;
    oem_run_info *infos = nullptr;
    std::atomic<uint32_t> next{0};
    std::mutex mu;                       // guards `fallback`
    std::vector<uint32_t> fallback;      // replicates with a multiplicity >= 256: one-per-pass path
};

// One chain of the rolling batch: kBatch slots share every pass over the matrix; a slot whose replicate
// has finished is handed the next replicate of the job at once, so the slots stay busy until the
// replicates run out (with fixed groups the pass count of a group is its largest, and every group pays
// its own set-up).  Every call of it runs on its own stream with its own buffers, so kChains of them run
// side by side (threads of oem_bootstrap): the streaming fold / rel-diff kernels of one chain overlap the
// tile kernel of the other (two chains: +10 % bootstraps/s at C3; three or four add nothing).
int run_bootstrap_chain(oem_store *s, int chain, BootJob *job)
{
    BatchBuffers &bb = s->batch[chain];
    hipStream_t st = bb.stream;
    const uint32_t T = s->csr.n_txps;
    const uint64_t R = s->csr.n_reads;
    const double avg = (double)s->global_n_reads / (double)T; // em.rs:154: the store's read count also for a replicate
    EmParams p{T, job->max_iter, 50u /* do_bootstrap -> do_em, em.rs:289,:212 */, job->conv_thresh};
    const bool sharded = comm_exchanges(s->comm);
    int slot_rep[kBatch];
    OEM_HIP(hipMemsetAsync(bb.cnt, 0, sizeof(double) * T * kBatch, st));
    for (int k = 0; k < kBatch; ++k) {
        slot_rep[k] = -1;
        std::memset(&bb.h_state[k], 0, sizeof(BatchState));
        bb.h_state[k].phase = kPhaseFinished;
    }
    OEM_HIP(hipMemcpyAsync(bb.state, bb.h_state, sizeof(BatchState) * kBatch, hipMemcpyHostToDevice, st));

    // hands slot k the next replicate that fits (or leaves it idle when none is left)
    auto load = [&](int k) -> int {
        for (;;) {
            const uint32_t rep = job->next.fetch_add(1);
            if (rep >= job->n_boot) return OEM_OK;
            if (job->row_w_all) {
                OEM_HIP(hipMemcpyAsync(bb.d_row_w, job->row_w_all + (size_t)rep * R, sizeof(uint32_t) * R, hipMemcpyHostToDevice, st));
            } else {
                OEM_TRY(launch_bootstrap_weights(s, bb.d_row_w, R, s->global_row_offset, s->global_n_reads, job->seed,
                                                 s->bootstrap_first_replica + rep, st)); // em.rs:274-276
            }
            OEM_HIP(hipMemsetAsync(bb.overflow, 0, sizeof(uint32_t), st));
            OEM_TRY(launch_batch_pack_row_w(s, bb.d_row_w, bb, (uint32_t)k, bb.overflow));
            uint32_t h_overflow = 0;
            OEM_HIP(hipMemcpyAsync(&h_overflow, bb.overflow, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            OEM_HIP(hipStreamSynchronize(st));
            if (sharded) OEM_TRY(comm_check(s->comm, st)); // (the passes queued before this hand-over exchanged)
            bool over = h_overflow != 0;
            OEM_TRY(agree_any(s, &over)); // row shards (one chain): every rank must route the replicate the same way
            if (over) { // a multiplicity >= 256: this replicate goes to the one-per-pass path
                std::lock_guard<std::mutex> lk(job->mu);
                job->fallback.push_back(rep);
                continue; // (the slot's byte column is rewritten by the next replicate it is handed)
            }
            OEM_TRY(launch_batch_reset_slot(s, bb, job->d_init, avg, (uint32_t)k));
            std::memset(&bb.h_state[k], 0, sizeof(BatchState));
            bb.h_state[k].phase = kPhaseRunning;
            OEM_HIP(hipMemcpyAsync(&bb.state[k], &bb.h_state[k], sizeof(BatchState), hipMemcpyHostToDevice, st));
            slot_rep[k] = (int)rep;
            return OEM_OK;
        }
    };
    for (int k = 0; k < kBatch; ++k) OEM_TRY(load(k));

    auto one_pass = [&]() -> int {
        OEM_TRY(launch_batch_pass(s, bb));
        if (sharded) OEM_TRY(comm_allreduce_sum_f64(s->comm, bb.cnt, bb.cnt, (size_t)T * kBatch, st));
        return launch_batch_reldiff(s, bb, p);
    };
    ChunkGraph cg; // kGraphIters batched passes, replayed (see capture_chunk)
    if (graph_ok(s, (size_t)T * kBatch) && job->max_iter >= 4 * kGraphIters)
        OEM_TRY(capture_chunk(st, kGraphIters, one_pass, &cg));
    bool first = true;
    for (;;) {
        bool busy = false;
        for (int k = 0; k < kBatch; ++k) busy = busy || slot_rep[k] >= 0;
        if (!busy) break;
        uint32_t chunk = first ? 52u : 16u; // (no slot can finish before its 53rd pass: gate 50)
        first = false;
        if (cg.ready()) {
            chunk = (chunk + kGraphIters - 1) / kGraphIters * kGraphIters;
            for (uint32_t i = 0; i < chunk; i += kGraphIters) OEM_HIP(hipGraphLaunch(cg.ge, st));
        } else {
            for (uint32_t i = 0; i < chunk; ++i) OEM_TRY(one_pass());
        }
        OEM_HIP(hipMemcpyAsync(bb.h_state, bb.state, sizeof(BatchState) * kBatch, hipMemcpyDeviceToHost, st));
        OEM_HIP(hipStreamSynchronize(st));
        // a peer that never arrived: the waits gave up and the passes summed stale slots -- an error, not replicates
        if (sharded) OEM_TRY(comm_check(s->comm, st));
        for (int k = 0; k < kBatch; ++k) {
            if (slot_rep[k] < 0 || bb.h_state[k].phase != kPhaseFinished) continue;
            const uint32_t rep = (uint32_t)slot_rep[k];
            OEM_HIP(hipMemcpyAsync(bb.h_out + (size_t)k * T, bb.out + (size_t)k * T, sizeof(double) * T,
                                   hipMemcpyDeviceToHost, st));
            OEM_HIP(hipStreamSynchronize(st));
            if (sharded) OEM_TRY(comm_check(s->comm, st));
            std::memcpy(job->out + (size_t)rep * T, bb.h_out + (size_t)k * T, sizeof(double) * T);
            if (job->infos) {
                job->infos[rep].niter = bb.h_state[k].niter;
                job->infos[rep].n_passes = bb.h_state[k].n_passes;
                job->infos[rep].converged = bb.h_state[k].converged;
                job->infos[rep].reserved = 0;
                job->infos[rep].rel_diff = bb.h_state[k].last_rel;
            }
            slot_rep[k] = -1;
            OEM_TRY(load(k));
        }
    }
    return OEM_OK;
}

// The batched bootstrap: kChains chains (one host thread each) over the one resident matrix.  A row-sharded
// store runs a single chain: its per-pass all-reduces must be issued in the same order on every rank.
int run_bootstrap_rolling(oem_store *s, uint32_t n_boot, uint64_t seed, const uint32_t *row_w_all, const double *init,
                          uint32_t max_iter, double conv_thresh, double *out, oem_run_info *infos,
                          std::vector<uint32_t> *fallback)
{
    const uint32_t T = s->csr.n_txps;
    BootJob job;
    job.n_boot = n_boot; job.seed = seed; job.row_w_all = row_w_all; job.max_iter = max_iter;
    job.conv_thresh = conv_thresh; job.out = out; job.infos = infos;
    int n_chains = comm_exchanges(s->comm) ? 1 : kChains;
    if (n_boot <= (uint32_t)kBatch) n_chains = 1; // one chain holds them all
    n_chains = (int)knob("OEM_BOOT_CHAINS", n_chains) < n_chains ? (int)knob("OEM_BOOT_CHAINS", n_chains) : n_chains;
    if (n_chains < 1) n_chains = 1;
    for (int c = 0; c < n_chains; ++c) OEM_TRY(ensure_batch(s, c));
    if (init) {
        OEM_HIP(hipMemcpyAsync(s->theta, init, sizeof(double) * T, hipMemcpyHostToDevice, s->stream));
        job.d_init = s->theta;
    }
    OEM_HIP(hipStreamSynchronize(s->stream)); // the init vector and the buffers' set-up are in place for every chain
    int rcs[kChains];
    std::string errs[kChains];
    for (int c = 0; c < kChains; ++c) rcs[c] = OEM_OK;
    auto body = [&](int c) {
        if (hipSetDevice(s->device) != hipSuccess) {
            rcs[c] = OEM_ERR_HIP;
            errs[c] = "hipSetDevice failed in a bootstrap chain";
            return;
        }
        try {
            rcs[c] = run_bootstrap_chain(s, c, &job);
        } catch (const std::exception &e) {
            rcs[c] = fail(OEM_ERR_OOM, "bootstrap chain: %s", e.what());
        } catch (...) {
            rcs[c] = fail(OEM_ERR_STATE, "bootstrap chain: unknown C++ exception");
        }
        if (rcs[c] != OEM_OK) errs[c] = t_err; // t_err is thread-local
    };
    // (a std::thread constructor that throws must not leave joinable threads behind: std::terminate)
    struct Joiner {
        std::vector<std::thread> th;
        ~Joiner() { for (auto &t : th) if (t.joinable()) t.join(); }
    } pool;
    int started = 1;
    try {
        for (int c = 1; c < n_chains; ++c) { pool.th.emplace_back(body, c); ++started; }
    } catch (...) { // the chains that did start (and chain 0 below) take all the replicates
    }
    body(0);
    for (auto &t : pool.th) t.join();
    n_chains = started;
    for (int c = 0; c < n_chains; ++c)
        if (rcs[c] != OEM_OK) return fail(rcs[c], "%s", errs[c].c_str());
    *fallback = job.fallback;
    std::sort(fallback->begin(), fallback->end());
    return OEM_OK;
}

void free_store(oem_store *s)
{
    if (!s) return;
    hipSetDevice(s->device);
    if (s->stream) hipStreamSynchronize(s->stream);
    hipFree(s->csr.row_ptr);
    hipFree(s->csr.tid);
    hipFree(s->csr.w32);
    hipFree(s->csr.w64);
    {
        oem::DeviceTiled &t = s->tiled;
        hipFree(t.dict); hipFree(t.widx); hipFree(t.i_base); hipFree(t.r_wi);
        hipFree(t.tiles); hipFree(t.perm); hipFree(t.codes); hipFree(t.w32);
        hipFree(t.w64); hipFree(t.r_tid); hipFree(t.r_w32); hipFree(t.r_w64); hipFree(t.r_row);
        hipFree(t.r_slot); hipFree(t.r_pk); hipFree(t.sd); hipFree(t.q_dst); hipFree(t.bucket_base);
        hipFree(t.queue);
        hipFree(t.row_w_perm);
    }
    for (int c = 0; c < oem::kChains; ++c) {
        oem::BatchBuffers &b = s->batch[c];
        if (c > 0 && b.stream) { hipStreamSynchronize(b.stream); hipStreamDestroy(b.stream); }
        hipFree(b.d_row_w);
        hipFree(b.theta); hipFree(b.cnt); hipFree(b.out); hipFree(b.queue); hipFree(b.state);
        hipFree(b.row_w); hipFree(b.overflow);
        if (b.h_state) hipHostFree(b.h_state);
        if (b.h_out) hipHostFree(b.h_out);
    }
    hipFree(s->multi.state); hipFree(s->multi.out); hipFree(s->multi.n_unfinished);
    hipFree(s->multi.live_tiles); hipFree(s->multi.live_buckets); hipFree(s->multi.d_live_counts);
    hipFree(s->theta);
    hipFree(s->cnt);
    hipFree(s->d_state);
    hipFree(s->d_row_w);
    if (s->h_state) hipHostFree(s->h_state);
    if (s->h_pinned) hipHostFree(s->h_pinned);
    if (s->stream) hipStreamDestroy(s->stream);
    delete s;
}

template <typename T, typename A>
int upload_vec(T **dst, const std::vector<T, A> &v, uint64_t *acct)
{
    OEM_TRY(dev_alloc(dst, v.size(), acct));
    if (!v.empty()) OEM_HIP(hipMemcpy(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return OEM_OK;
}

int upload_tiled(oem_store *s, const TiledHost &h)
{
    DeviceTiled &t = s->tiled;
    t.n_tiles = h.n_tiles;
    t.win_cap = h.win_cap;
    t.n_buckets = h.n_buckets;
    t.n_rows = h.n_rows;
    t.n_local = h.n_local;
    t.n_remote = h.n_remote;
    OEM_TRY(upload_vec(&t.tiles, h.tiles, &s->hbm_bytes));
    OEM_TRY(upload_vec(&t.perm, h.perm, &s->hbm_bytes));
    OEM_TRY(upload_vec(&t.codes, h.codes, &s->hbm_bytes));
    if (s->csr.w_is_f64) {
        OEM_TRY(upload_vec(&t.w64, h.w64, &s->hbm_bytes));
        OEM_TRY(upload_vec(&t.r_w64, h.r_w64, &s->hbm_bytes));
    } else {
        OEM_TRY(upload_vec(&t.w32, h.w32, &s->hbm_bytes));
        OEM_TRY(upload_vec(&t.r_w32, h.r_w32, &s->hbm_bytes));
    }
    OEM_TRY(upload_vec(&t.r_tid, h.r_tid, &s->hbm_bytes));
    OEM_TRY(upload_vec(&t.r_row, h.r_row, &s->hbm_bytes));
    OEM_TRY(upload_vec(&t.r_slot, h.r_slot, &s->hbm_bytes));
    OEM_TRY(upload_vec(&t.q_dst, h.q_dst, &s->hbm_bytes));
    OEM_TRY(upload_vec(&t.bucket_base, h.bucket_base, &s->hbm_bytes));
    t.h_bucket_base = h.bucket_base;
    OEM_TRY(dev_alloc(&t.queue, h.n_remote, &s->hbm_bytes));
    OEM_TRY(dev_alloc(&t.row_w_perm, h.n_rows, &s->hbm_bytes));
    t.present = true;
    return OEM_OK;
}

// Per-cell batches: cell c's transcripts are relabelled to [c * cell_txps, (c + 1) * cell_txps) -- on
// the device, after the upload, instead of in a second host copy of the transcript ids.
struct CellRelabel {
    const uint64_t *cell_row_off;
    uint32_t n_cells;
    uint32_t cell_txps;
};

__global__ __launch_bounds__(256) void k_narrow_u64(const unsigned long long *__restrict__ in, uint32_t *__restrict__ out,
                                                    uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = (uint32_t)in[i];
}

__global__ __launch_bounds__(256) void k_relabel_cells(const uint32_t *__restrict__ row_ptr, uint32_t *__restrict__ tid,
                                                       const unsigned long long *__restrict__ cell_row_off,
                                                       uint32_t n_cells, uint32_t cell_txps, uint64_t n_reads)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    uint32_t a = 0, b = n_cells; // last cell whose first read is <= r
    while (b - a > 1) {
        const uint32_t m = (a + b) >> 1;
        if (cell_row_off[m] <= r) a = m;
        else b = m;
    }
    const uint32_t add = a * cell_txps;
    for (uint32_t j = row_ptr[r]; j < row_ptr[r + 1]; ++j) tid[j] += add;
}

int create_store_layout(const uint64_t *row_ptr, const uint32_t *tid, const float *as_prob,
                        const double *cov_prob, uint64_t n_reads, uint64_t nnz, uint32_t n_txps,
                        int device, const oem_store_opts *opts, oem_store *s, const CellRelabel *relabel);

// upload + layout (either builder) + the slim remote records every kernel reads (oem_layout_pack.hip)
int create_store_impl(const uint64_t *row_ptr, const uint32_t *tid, const float *as_prob,
                      const double *cov_prob, uint64_t n_reads, uint64_t nnz, uint32_t n_txps,
                      int device, const oem_store_opts *opts, oem_store *s, const CellRelabel *relabel = nullptr)
{
    OEM_TRY(create_store_layout(row_ptr, tid, as_prob, cov_prob, n_reads, nnz, n_txps, device, opts, s, relabel));
    StageTimer tm;
    // (the test-only library keeps the builders' streams when asked to: the layout tests hash them)
    OEM_TRY(pack_remote_records(s, opts ? opts->problem_size : 0u, knob("OEM_KEEP_UNPACKED", 0) != 0));
    tm.lap("slot table + packed records");
    if (!opts || opts->weight_coding == 0) {
        OEM_TRY(build_weight_dictionary(s)); // <= 256 distinct f32 weights: one byte per local alignment
        tm.lap("weight dictionary");
    }
    return OEM_OK;
}

int create_store_layout(const uint64_t *row_ptr, const uint32_t *tid, const float *as_prob,
                        const double *cov_prob, uint64_t n_reads, uint64_t nnz, uint32_t n_txps,
                        int device, const oem_store_opts *opts, oem_store *s, const CellRelabel *relabel)
{
    s->device = device;
    StageTimer tm;
    OEM_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    DeviceCsr &m = s->csr;
    m.n_reads = n_reads;
    m.nnz = nnz;
    m.n_txps = n_txps;
    m.wide_ptr = nnz >= (1ull << 32);
    m.w_is_f64 = cov_prob != nullptr;

    // The caller-order CSR (row ranges of per-cell runs, aux counts, assignment probabilities) goes up
    // from a helper thread while this one builds the tiled layout: both are host-bound.
    auto upload_csr = [&]() -> int {
        if (m.wide_ptr) {
            uint64_t *d = nullptr;
            OEM_TRY(dev_alloc(&d, n_reads + 1, &s->hbm_bytes));
            m.row_ptr = d;
            OEM_HIP(hipMemcpy(d, row_ptr, sizeof(uint64_t) * (n_reads + 1), hipMemcpyHostToDevice));
        } else {
            // fewer than 2^32 alignments: the kernels walk u32 row pointers; narrowed on the device
            // (no second host copy of the array)
            unsigned long long *d64 = nullptr;
            OEM_HIP(hipMalloc((void **)&d64, sizeof(uint64_t) * (n_reads + 1)));
            uint32_t *d = nullptr;
            int rc = dev_alloc(&d, n_reads + 1, &s->hbm_bytes);
            m.row_ptr = d;
            hipError_t e = rc == OEM_OK ? hipMemcpy(d64, row_ptr, sizeof(uint64_t) * (n_reads + 1), hipMemcpyHostToDevice)
                                        : hipSuccess;
            if (rc == OEM_OK && e == hipSuccess) {
                const uint64_t n = n_reads + 1;
                uint64_t g = (n + 255) / 256;
                if (g > 4096) g = 4096;
                hipLaunchKernelGGL(k_narrow_u64, dim3((uint32_t)g), dim3(256), 0, s->stream, d64, d, n);
                e = hipStreamSynchronize(s->stream);
            }
            hipFree(d64);
            if (rc != OEM_OK) return rc;
            if (e != hipSuccess) return fail(OEM_ERR_HIP, "row_ptr upload failed: %s", hipGetErrorString(e));
        }
        OEM_TRY(dev_alloc(&m.tid, nnz, &s->hbm_bytes));
        OEM_HIP(hipMemcpy(m.tid, tid, sizeof(uint32_t) * nnz, hipMemcpyHostToDevice));
        if (m.w_is_f64) {
            // em.rs:107-111: prev * (p as f64) * cov; w = (p as f64) * cov is the
            // iteration-invariant factor (SURVEY.md 8a note 2: f64 keeps strict parity).
            std::vector<double> w(nnz);
            for (uint64_t j = 0; j < nnz; ++j) w[j] = (double)as_prob[j] * cov_prob[j];
            OEM_TRY(dev_alloc(&m.w64, nnz, &s->hbm_bytes));
            OEM_HIP(hipMemcpy(m.w64, w.data(), sizeof(double) * nnz, hipMemcpyHostToDevice));
        } else {
            OEM_TRY(dev_alloc(&m.w32, nnz, &s->hbm_bytes));
            OEM_HIP(hipMemcpy(m.w32, as_prob, sizeof(float) * nnz, hipMemcpyHostToDevice));
        }
        OEM_TRY(dev_alloc(&s->theta, n_txps, &s->hbm_bytes));
        OEM_TRY(dev_alloc(&s->cnt, n_txps, &s->hbm_bytes));
        OEM_TRY(dev_alloc(&s->d_state, 1, &s->hbm_bytes));
        OEM_HIP(hipHostMalloc((void **)&s->h_state, sizeof(EmState), hipHostMallocDefault));
        OEM_HIP(hipHostMalloc((void **)&s->h_pinned, sizeof(double) * (n_txps ? n_txps : 1), hipHostMallocDefault));
        return OEM_OK;
    };
    s->global_n_reads = n_reads;
    s->global_row_offset = 0;

    // default: lay the store out in primary-sorted tiles (oem_layout.h)
    const uint32_t reorder = opts ? opts->reorder_rows : 0;
    if (reorder == 1 || n_reads == 0) {
        OEM_TRY(upload_csr());
        tm.lap("caller-order CSR upload");
        return OEM_OK;
    }
    // Window cap of the tiles: sparse stores (few reads per transcript, e.g. per-cell batches) fill
    // their tiles only with a wide window; dense ones are faster with the narrow one and four copies.
    uint32_t win_cap = opts ? opts->window_cap : 0u;
    if (win_cap != kWin && win_cap != kWinWide) {
        // measured (scripts/wincap_ab.py): the wide cap wins on large sparse stores (2 M reads over 4 M
        // transcripts -10 %, a 625-cell batch -16 %), the narrow one on dense stores and on small ones,
        // which are latency-bound either way
        const bool sparse = n_reads < 2 * (uint64_t)(n_txps ? n_txps : 1);
        win_cap = (sparse && n_reads >= 1000000) ? kWinWide : kWin;
    }
    // host copy of the relabelled transcript ids, only for the host builder
    std::vector<uint32_t> vt;
    auto host_tids = [&]() -> const uint32_t * {
        if (!relabel) return tid;
        if (vt.empty() && nnz) {
            vt.resize(nnz);
            for (uint32_t c = 0; c < relabel->n_cells; ++c) {
                const uint64_t a0 = row_ptr[relabel->cell_row_off[c]], a1 = row_ptr[relabel->cell_row_off[c + 1]];
                for (uint64_t j = a0; j < a1; ++j) vt[j] = c * relabel->cell_txps + tid[j];
            }
        }
        return vt.data();
    };
    auto relabel_on_device = [&]() -> int {
        if (!relabel || nnz == 0) return OEM_OK;
        if (m.wide_ptr) return fail(OEM_ERR_ARG, "per-cell batch needs fewer than 2^32 alignments");
        unsigned long long *d_off = nullptr;
        OEM_HIP(hipMalloc((void **)&d_off, sizeof(unsigned long long) * ((size_t)relabel->n_cells + 1)));
        hipError_t e = hipMemcpy(d_off, relabel->cell_row_off, sizeof(unsigned long long) * ((size_t)relabel->n_cells + 1),
                                 hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_relabel_cells, dim3((uint32_t)((n_reads + 255) / 256)), dim3(256), 0, s->stream,
                               (const uint32_t *)m.row_ptr, m.tid, d_off, relabel->n_cells, relabel->cell_txps, n_reads);
            e = hipStreamSynchronize(s->stream);
        }
        hipFree(d_off);
        if (e != hipSuccess) return fail(OEM_ERR_HIP, "relabelling the cells failed: %s", hipGetErrorString(e));
        return OEM_OK;
    };
    // The layout is built on the device from the resident CSR (oem_layout_device.hip); the host
    // builder (oem_layout.cpp, the specification) takes the stores that one does not, or all of them
    // with oem_store_opts.layout_build = 1.
    if (!(opts && opts->layout_build == 1)) {
        OEM_TRY(upload_csr());
        OEM_TRY(relabel_on_device());
        tm.lap("caller-order CSR upload");
        bool built = false;
        OEM_TRY(build_tiled_layout_device(s, opts ? opts->problem_size : 0u, win_cap, &built));
        tm.lap("tiled layout build (device)");
        if (built) return OEM_OK;
        TiledHost h;
        const char *err = nullptr;
        if (build_tiled_layout(row_ptr, host_tids(), as_prob, cov_prob, n_reads, nnz, n_txps, &h, &err,
                               opts ? opts->problem_size : 0u, win_cap)) {
            OEM_TRY(upload_tiled(s, h));
        } else if (reorder == 2) {
            return fail(OEM_ERR_ARG, "oem_store_create: %s", err ? err : "cannot tile this store");
        }
        return OEM_OK;
    }
    int csr_rc = OEM_OK;
    char csr_err[sizeof(t_err)] = {0};
    std::thread up([&] {
        if (hipSetDevice(device) != hipSuccess) {
            csr_rc = OEM_ERR_HIP;
            snprintf(csr_err, sizeof(csr_err), "hipSetDevice(%d) failed in the upload thread", device);
            return;
        }
        csr_rc = upload_csr();
        if (csr_rc == OEM_OK) csr_rc = relabel_on_device();
        if (csr_rc != OEM_OK) snprintf(csr_err, sizeof(csr_err), "%s", t_err); // t_err is thread-local
    });
    TiledHost h;
    const char *err = nullptr;
    const bool tiled = build_tiled_layout(row_ptr, host_tids(), as_prob, cov_prob, n_reads, nnz, n_txps, &h, &err,
                                          opts ? opts->problem_size : 0u, win_cap);
    tm.lap("tiled layout build (host)");
    up.join();
    tm.lap("wait for the CSR upload");
    if (csr_rc != OEM_OK) return fail(csr_rc, "%s", csr_err);
    if (tiled) {
        OEM_TRY(upload_tiled(s, h));
        tm.lap("tiled layout upload");
    } else if (reorder == 2) {
        return fail(OEM_ERR_ARG, "oem_store_create: %s", err ? err : "cannot tile this store");
    }
    return OEM_OK;
}

} // namespace
} // namespace oem

using namespace oem;

// ---------------------------------------------------------------------------
// library
// ---------------------------------------------------------------------------
extern "C" int oem_abi_version(void) { return OEM_ABI_VERSION; }

extern "C" const char *oem_last_error(void) { return t_err; }

extern "C" int oem_device_count(int *out_count)
{
    OEM_API_BEGIN
    if (!out_count) return fail(OEM_ERR_ARG, "oem_device_count: out_count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    *out_count = n;
    return OEM_OK;
    OEM_API_END("oem_device_count")
}

// ---------------------------------------------------------------------------
// store
// ---------------------------------------------------------------------------
extern "C" int oem_store_create(const uint64_t *row_ptr, const uint32_t *tid, const float *as_prob,
                                const double *cov_prob, uint64_t n_reads, uint64_t nnz,
                                uint32_t n_txps, int device, const oem_store_opts *opts,
                                oem_store **out)
{
    OEM_API_BEGIN
    if (!out) return fail(OEM_ERR_ARG, "oem_store_create: out is NULL");
    *out = nullptr;
    if (!row_ptr) return fail(OEM_ERR_ARG, "oem_store_create: row_ptr is NULL");
    if (nnz > 0 && (!tid || !as_prob)) return fail(OEM_ERR_ARG, "oem_store_create: tid/as_prob is NULL");
    if (n_txps == 0) return fail(OEM_ERR_ARG, "oem_store_create: n_txps is 0");
    StageTimer tm;
    OEM_TRY(validate_csr(row_ptr, tid, n_reads, nnz, n_txps));
    std::vector<double> cov_fixed;
    if (cov_prob && zero_nan_rows(row_ptr, cov_prob, n_reads, nnz, &cov_fixed)) cov_prob = cov_fixed.data();
    tm.lap("validate_csr");
    OEM_TRY(ensure_device(device));
    oem_store *s = new (std::nothrow) oem_store();
    if (!s) return fail(OEM_ERR_OOM, "oem_store_create: host allocation failed");
    int rc = create_store_impl(row_ptr, tid, as_prob, cov_prob, n_reads, nnz, n_txps, device, opts, s);
    if (rc != OEM_OK) {
        free_store(s);
        return rc;
    }
    *out = s;
    return OEM_OK;
    OEM_API_END("oem_store_create")
}

extern "C" void oem_store_destroy(oem_store *store) { free_store(store); }

extern "C" int oem_store_dims(const oem_store *store, uint64_t *n_reads, uint64_t *nnz, uint32_t *n_txps)
{
    OEM_API_BEGIN
    if (!store) return fail(OEM_ERR_ARG, "oem_store_dims: store is NULL");
    if (n_reads) *n_reads = store->csr.n_reads;
    if (nnz) *nnz = store->csr.nnz;
    if (n_txps) *n_txps = store->csr.n_txps;
    return OEM_OK;
    OEM_API_END("oem_store_dims")
}

extern "C" int oem_store_set_option(oem_store *store, uint32_t option, uint64_t value)
{
    OEM_API_BEGIN
    if (!store) return fail(OEM_ERR_ARG, "oem_store_set_option: store is NULL");
    std::lock_guard<std::mutex> lk(store->mu);
    switch (option) {
    case OEM_OPT_BATCH_BOOTSTRAP: store->batch_bootstrap = value != 0; return OEM_OK;
    case OEM_OPT_BOOTSTRAP_FIRST_REPLICA:
        if (value > 0xffffffffull) return fail(OEM_ERR_ARG, "oem_store_set_option: replica index out of range");
        store->bootstrap_first_replica = (uint32_t)value;
        return OEM_OK;
    default: return fail(OEM_ERR_ARG, "oem_store_set_option: unknown option %u", option);
    }
    OEM_API_END("oem_store_set_option")
}

extern "C" int oem_store_info(const oem_store *store, uint32_t key, uint64_t *value)
{
    OEM_API_BEGIN
    if (!store || !value) return fail(OEM_ERR_ARG, "oem_store_info: NULL argument");
    switch (key) {
    case OEM_INFO_WEIGHT_DICT_ENTRIES: *value = store->tiled.present ? store->tiled.dict_n : 0u; return OEM_OK;
    case OEM_INFO_TILES: *value = store->tiled.present ? store->tiled.n_tiles : 0u; return OEM_OK;
    case OEM_INFO_REMOTE_ALIGNMENTS: *value = store->tiled.present ? store->tiled.n_remote : 0u; return OEM_OK;
    default: return fail(OEM_ERR_ARG, "oem_store_info: unknown key %u", key);
    }
    OEM_API_END("oem_store_info")
}

extern "C" int oem_store_bytes(const oem_store *store, uint64_t *hbm_bytes, uint64_t *algorithmic_bytes_per_pass)
{
    OEM_API_BEGIN
    if (!store) return fail(OEM_ERR_ARG, "oem_store_bytes: store is NULL");
    const DeviceCsr &m = store->csr;
    if (hbm_bytes) *hbm_bytes = store->hbm_bytes;
    if (algorithmic_bytes_per_pass)
        // nnz*(4 [tid] + 4|8 [w]) + (R+1)*4|8 [row_ptr] + 2*T*8 [theta read, cnt written]
        *algorithmic_bytes_per_pass = m.nnz * (4 + (m.w_is_f64 ? 8 : 4)) +
                                      (m.n_reads + 1) * (m.wide_ptr ? 8 : 4) + 2ull * m.n_txps * 8;
    return OEM_OK;
    OEM_API_END("oem_store_bytes")
}

// ---------------------------------------------------------------------------
// EM
// ---------------------------------------------------------------------------
extern "C" int oem_m_step(oem_store *s, const double *theta, const uint32_t *row_w, double *out_counts)
{
    OEM_API_BEGIN
    if (!s || !theta || !out_counts) return fail(OEM_ERR_ARG, "oem_m_step: NULL argument");
    std::lock_guard<std::mutex> lk(s->mu);
    OEM_TRY(ensure_device(s->device));
    const uint32_t T = s->csr.n_txps;
    OEM_HIP(hipMemcpyAsync(s->theta, theta, sizeof(double) * T, hipMemcpyHostToDevice, s->stream));
    OEM_HIP(hipMemsetAsync(s->cnt, 0, sizeof(double) * T, s->stream));
    const uint32_t *d_w = nullptr;
    if (row_w) {
        OEM_TRY(ensure_row_w(s));
        OEM_HIP(hipMemcpyAsync(s->d_row_w, row_w, sizeof(uint32_t) * s->csr.n_reads, hipMemcpyHostToDevice, s->stream));
        d_w = s->d_row_w;
    }
    RunArgs a;
    a.d_row_w = d_w;
    a.row_end = s->csr.n_reads;
    OEM_TRY(prepare_row_w(s, a));
    OEM_TRY(enqueue_pass(s, a, nullptr));
    if (comm_exchanges(s->comm))
        OEM_TRY(comm_allreduce_sum_f64(s->comm, s->cnt, s->cnt, T, s->stream));
    return copy_counts_out(s, out_counts);
    OEM_API_END("oem_m_step")
}

extern "C" int oem_em_run(oem_store *s, const double *init_abundances, uint32_t max_iter,
                          double conv_thresh, uint32_t min_iter_gate, double *out_counts,
                          oem_run_info *info)
{
    OEM_API_BEGIN
    if (!s || !out_counts) return fail(OEM_ERR_ARG, "oem_em_run: NULL argument");
    std::lock_guard<std::mutex> lk(s->mu);
    OEM_TRY(ensure_device(s->device));
    RunArgs a;
    a.init = init_abundances;
    a.row_begin = 0;
    a.row_end = s->csr.n_reads;
    a.total_reads = s->global_n_reads;
    a.max_iter = max_iter;
    a.conv_thresh = conv_thresh;
    a.min_iter_gate = min_iter_gate;
    OEM_TRY(run_em_device(s, a, info));
    return copy_counts_out(s, out_counts);
    OEM_API_END("oem_em_run")
}

// ---------------------------------------------------------------------------
// the steps right after the EM
// ---------------------------------------------------------------------------
extern "C" int oem_aux_counts(oem_store *s, uint32_t *out_unique, uint32_t *out_total)
{
    OEM_API_BEGIN
    if (!s || !out_unique || !out_total) return fail(OEM_ERR_ARG, "oem_aux_counts: NULL argument");
    std::lock_guard<std::mutex> lk(s->mu);
    OEM_TRY(ensure_device(s->device));
    const uint32_t T = s->csr.n_txps;
    uint32_t *d = nullptr;
    OEM_TRY(dev_alloc(&d, 2 * (size_t)T, nullptr));
    int rc = OEM_OK;
    if (hipMemsetAsync(d, 0, sizeof(uint32_t) * 2 * T, s->stream) != hipSuccess) rc = fail(OEM_ERR_HIP, "oem_aux_counts: memset failed");
    if (rc == OEM_OK) rc = launch_aux_counts(s, d, d + T);
    if (rc == OEM_OK && (hipMemcpyAsync(out_unique, d, sizeof(uint32_t) * T, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
                         hipMemcpyAsync(out_total, d + T, sizeof(uint32_t) * T, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
                         hipStreamSynchronize(s->stream) != hipSuccess))
        rc = fail(OEM_ERR_HIP, "oem_aux_counts: read-back failed");
    hipFree(d);
    return rc;
    OEM_API_END("oem_aux_counts")
}

extern "C" int oem_assignment_probs(oem_store *s, const double *counts, double display_thresh, double *out_prob)
{
    OEM_API_BEGIN
    if (!s || !counts || (s->csr.nnz && !out_prob)) return fail(OEM_ERR_ARG, "oem_assignment_probs: NULL argument");
    std::lock_guard<std::mutex> lk(s->mu);
    OEM_TRY(ensure_device(s->device));
    const uint32_t T = s->csr.n_txps;
    const uint64_t nnz = s->csr.nnz;
    double *d_out = nullptr;
    OEM_TRY(dev_alloc(&d_out, nnz, nullptr));
    int rc = OEM_OK;
    if (hipMemcpyAsync(s->theta, counts, sizeof(double) * T, hipMemcpyHostToDevice, s->stream) != hipSuccess)
        rc = fail(OEM_ERR_HIP, "oem_assignment_probs: upload failed");
    if (rc == OEM_OK) rc = launch_assignment_probs(s, s->theta, display_thresh, d_out);
    if (rc == OEM_OK && nnz &&
        (hipMemcpyAsync(out_prob, d_out, sizeof(double) * nnz, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
         hipStreamSynchronize(s->stream) != hipSuccess))
        rc = fail(OEM_ERR_HIP, "oem_assignment_probs: read-back failed");
    hipFree(d_out);
    return rc;
    OEM_API_END("oem_assignment_probs")
}

// ---------------------------------------------------------------------------
// bootstrap
// ---------------------------------------------------------------------------
extern "C" int oem_bootstrap_weights(oem_store *s, uint64_t seed, uint32_t replica, uint32_t *out_row_w)
{
    OEM_API_BEGIN
    if (!s || !out_row_w) return fail(OEM_ERR_ARG, "oem_bootstrap_weights: NULL argument");
    std::lock_guard<std::mutex> lk(s->mu);
    OEM_TRY(ensure_device(s->device));
    OEM_TRY(ensure_row_w(s));
    OEM_TRY(launch_bootstrap_weights(s, s->d_row_w, s->csr.n_reads, s->global_row_offset,
                                     s->global_n_reads, seed, replica));
    OEM_HIP(hipMemcpyAsync(out_row_w, s->d_row_w, sizeof(uint32_t) * s->csr.n_reads, hipMemcpyDeviceToHost, s->stream));
    OEM_HIP(hipStreamSynchronize(s->stream));
    return OEM_OK;
    OEM_API_END("oem_bootstrap_weights")
}

extern "C" int oem_bootstrap(oem_store *s, uint32_t n_boot, uint64_t seed, const uint32_t *row_w_all,
                             const double *init_abundances, uint32_t max_iter, double conv_thresh,
                             double *out, oem_run_info *infos)
{
    OEM_API_BEGIN
    if (!s || (n_boot && !out)) return fail(OEM_ERR_ARG, "oem_bootstrap: NULL argument");
    std::lock_guard<std::mutex> lk(s->mu);
    OEM_TRY(ensure_device(s->device));
    OEM_TRY(ensure_row_w(s));
    const uint32_t T = s->csr.n_txps;
    const uint64_t R = s->csr.n_reads;
    // the replicates that run one per pass: all of them, or those the rolling batch hands back
    std::vector<uint32_t> single;
    bool no_batch = !(s->batch_bootstrap && can_batch(s));
    OEM_TRY(agree_any(s, &no_batch)); // row shards tile their own blocks: all ranks batch, or none does
    if (!no_batch && max_iter >= 1 && n_boot >= 2) {
        OEM_TRY(run_bootstrap_rolling(s, n_boot, seed, row_w_all, init_abundances, max_iter, conv_thresh, out, infos,
                                      &single));
    } else {
        for (uint32_t b = 0; b < n_boot; ++b) single.push_back(b);
    }
    for (uint32_t b : single) {
        if (row_w_all) {
            OEM_HIP(hipMemcpyAsync(s->d_row_w, row_w_all + (uint64_t)b * R, sizeof(uint32_t) * R,
                                   hipMemcpyHostToDevice, s->stream));
        } else {
            OEM_TRY(launch_bootstrap_weights(s, s->d_row_w, R, s->global_row_offset, s->global_n_reads, seed,
                                             s->bootstrap_first_replica + b)); // em.rs:274-276
        }
        RunArgs a;
        a.init = init_abundances;
        a.d_row_w = s->d_row_w;
        a.row_begin = 0;
        a.row_end = R;
        a.total_reads = s->global_n_reads; // em.rs:154: still the store's read count
        a.max_iter = max_iter;
        a.conv_thresh = conv_thresh;
        a.min_iter_gate = 50;              // do_bootstrap -> do_em (em.rs:289, :212)
        OEM_TRY(run_em_device(s, a, infos ? &infos[b] : nullptr));
        OEM_TRY(copy_counts_out(s, out + (uint64_t)b * T));
    }
    return OEM_OK;
    OEM_API_END("oem_bootstrap")
}

// ---------------------------------------------------------------------------
// single-cell batch (v1: cells run back to back on the resident matrix)
// ---------------------------------------------------------------------------
namespace oem {
namespace {

// What the last oem_em_run_cells call of this thread spent in its batched EM loops (HIP events on the
// group's stream around the loop), for oem_cells_last_timing.
thread_local double t_cells_loop_ms = 0.0;
thread_local uint64_t t_cells_batched_passes = 0;

// All cells in one store over the concatenated transcript space; every pass serves every
// unfinished cell.  Returns *used = false (nothing done) when the batch form does not apply.
int run_cells_batched(const uint64_t *cell_row_off, uint32_t n_cells, const uint64_t *row_ptr,
                      const uint32_t *tid, const float *as_prob, const double *cov_prob, uint64_t n_reads,
                      uint64_t nnz, uint32_t n_txps, int device, uint32_t max_iter, double conv_thresh,
                      double *out, oem_run_info *infos, bool *used)
{
    *used = false;
    StageTimer tm;
    const uint64_t total_txps = (uint64_t)n_cells * n_txps;
    if (max_iter < 1 || n_cells < 2 || total_txps >= (1ull << 32) || n_reads >= (1ull << 32)) return OEM_OK;
    tm.lap("cells: group set-up");   // (the arrays were range-checked once by oem_em_run_cells)
    OEM_TRY(ensure_device(device));
    oem_store *s = new (std::nothrow) oem_store();
    if (!s) return fail(OEM_ERR_OOM, "oem_em_run_cells: host allocation failed");
    oem_store_opts opts;
    std::memset(&opts, 0, sizeof(opts));
    opts.reorder_rows = 0; // a batch that cannot be tiled falls through to the cell-by-cell path
    opts.problem_size = n_txps;
    // transcripts of cell p -> [p*T, (p+1)*T), relabelled on the device after the upload
    CellRelabel rl{cell_row_off, n_cells, n_txps};
    int rc = create_store_impl(row_ptr, tid, as_prob, cov_prob, n_reads, nnz, (uint32_t)total_txps, device, &opts, s, &rl);
    if (rc != OEM_OK) {
        free_store(s);
        return rc;
    }
    if (!s->tiled.present) { // e.g. a read with > 255 alignments inside one window: the serial path takes the group
        free_store(s);
        return OEM_OK;
    }
    *used = true;
    tm.lap("cells: store create");

    auto body = [&]() -> int {
        MultiBuffers &mb = s->multi;
        mb.n_problems = n_cells;
        mb.problem_size = n_txps;
        OEM_TRY(dev_alloc(&mb.state, n_cells, &s->hbm_bytes));
        OEM_TRY(dev_alloc(&mb.out, (size_t)total_txps, &s->hbm_bytes));
        OEM_TRY(dev_alloc(&mb.n_unfinished, 1, &s->hbm_bytes));
        std::vector<BatchState> hs(n_cells);
        std::vector<uint64_t> reads(n_cells);
        for (uint32_t c = 0; c < n_cells; ++c) {
            std::memset(&hs[c], 0, sizeof(BatchState));
            hs[c].phase = kPhaseRunning;
            reads[c] = cell_row_off[c + 1] - cell_row_off[c]; // the cell's own store.len() (single_cell.rs:122-130)
        }
        uint64_t *d_reads = nullptr;
        OEM_TRY(dev_alloc(&d_reads, n_cells, nullptr));
        int rc2 = OEM_OK;
        do {
            if (hipMemcpyAsync(d_reads, reads.data(), sizeof(uint64_t) * n_cells, hipMemcpyHostToDevice, s->stream) != hipSuccess ||
                hipMemcpyAsync(mb.state, hs.data(), sizeof(BatchState) * n_cells, hipMemcpyHostToDevice, s->stream) != hipSuccess ||
                hipMemcpyAsync(mb.n_unfinished, &n_cells, sizeof(uint32_t), hipMemcpyHostToDevice, s->stream) != hipSuccess ||
                hipMemsetAsync(s->cnt, 0, sizeof(double) * total_txps, s->stream) != hipSuccess) {
                rc2 = fail(OEM_ERR_HIP, "oem_em_run_cells: upload of the per-cell state failed");
                break;
            }
            if ((rc2 = launch_multi_init(s, s->theta, d_reads, mb)) != OEM_OK) break;
            EmParams p{n_txps, max_iter, 50u /* em::em, single_cell.rs:150 */, conv_thresh};
            if (hipMemsetAsync(mb.out, 0, sizeof(double) * total_txps, s->stream) != hipSuccess) {
                rc2 = fail(OEM_ERR_HIP, "oem_em_run_cells: clearing the result buffer failed");
                break;
            }
            const uint64_t total = (uint64_t)max_iter + 1; // loop passes + the final one (em.rs:245-252)
            // one workgroup per bucket folds the queue AND finishes the pass (k_multi_fold_reldiff); a
            // store without remote alignments has no buckets to own and takes the separate kernels
            const bool fused_fold = s->tiled.n_remote > 0 && s->tiled.n_buckets > 0 && knob("OEM_CELLS_FUSED_FOLD", 1) != 0;
            uint64_t launched = 0;
            uint32_t unfinished = n_cells, compacted_at = n_cells;
            hipEvent_t ev0 = nullptr, ev1 = nullptr;
            if (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess ||
                hipEventRecord(ev0, s->stream) != hipSuccess) {
                if (ev0) hipEventDestroy(ev0);
                if (ev1) hipEventDestroy(ev1);
                rc2 = fail(OEM_ERR_HIP, "oem_em_run_cells: event set-up failed");
                break;
            }
            auto one_pass = [&]() -> int {
                if (fused_fold) {
                    OEM_TRY(launch_em_pass_tiled(s, s->theta, s->cnt, nullptr, nullptr, mb.state, n_txps, true));
                    return launch_multi_fold_reldiff(s, s->theta, s->cnt, mb, p);
                }
                OEM_TRY(launch_em_pass_tiled(s, s->theta, s->cnt, nullptr, nullptr, mb.state, n_txps));
                return launch_multi_reldiff(s, s->theta, s->cnt, mb, p);
            };
            ChunkGraph cg; // kGraphIters batched passes (five to six kernels each), replayed
            if (graph_ok(s) && total >= 4 * kGraphIters) rc2 = capture_chunk(s->stream, kGraphIters, one_pass, &cg);
            while (rc2 == OEM_OK && launched < total && unfinished) {
                uint64_t chunk = launched == 0 ? 53 : 16;
                if (chunk > total - launched) chunk = total - launched;
                if (cg.ready()) { // (passes beyond `total` find every cell FINISHED: no-ops)
                    chunk = (chunk + kGraphIters - 1) / kGraphIters * kGraphIters;
                    for (uint64_t k = 0; k < chunk && rc2 == OEM_OK; k += kGraphIters)
                        if (hipGraphLaunch(cg.ge, s->stream) != hipSuccess) rc2 = fail(OEM_ERR_HIP, "oem_em_run_cells: graph launch failed");
                } else {
                    for (uint64_t k = 0; k < chunk && rc2 == OEM_OK; ++k) rc2 = one_pass();
                }
                if (rc2 != OEM_OK) break;
                launched += chunk;
                if (hipMemcpyAsync(&unfinished, mb.n_unfinished, sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
                    hipStreamSynchronize(s->stream) != hipSuccess) {
                    rc2 = fail(OEM_ERR_HIP, "oem_em_run_cells: state read-back failed");
                    break;
                }
                // cells have finished since the live lists were built: the next passes launch the live tiles and
                // buckets only (the lists stay supersets of the live work until the next look)
                if (unfinished && unfinished < compacted_at && !cg.ready() && knob("OEM_CELLS_COMPACT", 1) != 0) {
                    rc2 = multi_compact_live(s, mb);
                    compacted_at = unfinished;
                }
            }
            if (rc2 == OEM_OK && hipEventRecord(ev1, s->stream) == hipSuccess && hipEventSynchronize(ev1) == hipSuccess) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) {
                    t_cells_loop_ms += ms;
                    t_cells_batched_passes += launched;
                }
            }
            hipEventDestroy(ev0);
            hipEventDestroy(ev1);
            if (rc2 != OEM_OK) break;
            tm.lap("cells: EM loop");
            if (hipMemcpy(out, mb.out, sizeof(double) * total_txps, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(hs.data(), mb.state, sizeof(BatchState) * n_cells, hipMemcpyDeviceToHost) != hipSuccess) {
                rc2 = fail(OEM_ERR_HIP, "oem_em_run_cells: result read-back failed");
                break;
            }
            if (infos)
                for (uint32_t c = 0; c < n_cells; ++c) {
                    infos[c].niter = hs[c].niter;
                    infos[c].n_passes = hs[c].n_passes;
                    infos[c].converged = hs[c].converged;
                    infos[c].reserved = 0;
                    infos[c].rel_diff = hs[c].last_rel;
                }
        } while (false);
        hipFree(d_reads);
        return rc2;
    };
    rc = body();
    tm.lap("cells: read-back");
    free_store(s);
    tm.lap("cells: free");
    return rc;
}

// One group of consecutive cells [c0, c1): batched on the device when it can be (every pass over the
// resident store serves all unfinished cells), otherwise cell after cell over the caller-order CSR.
int run_cells_group(const uint64_t *cell_row_off, uint32_t c0, uint32_t c1, const uint64_t *row_ptr,
                    const uint32_t *tid, const float *as_prob, const double *cov_prob, uint32_t n_txps, int device,
                    uint32_t max_iter, double conv_thresh, double *out, oem_run_info *infos)
{
    const uint32_t n_cells = c1 - c0;
    const uint64_t r0 = cell_row_off[c0], r1 = cell_row_off[c1];
    const uint64_t a0 = row_ptr[r0], a1 = row_ptr[r1];
    const uint64_t n_reads = r1 - r0, nnz = a1 - a0;
    // The group's own offsets.  A group that starts at read 0 (the whole experiment, when it fits one group)
    // takes the caller's arrays as they are: rebasing 31 M row offsets of a 625-cell batch into a fresh
    // 250 MB vector cost ~60 ms of page faults, 7 % of the call.  Later groups rebase on a few threads.
    std::vector<uint64_t> off_v, rp_v;
    const uint64_t *off_p = cell_row_off + c0, *rp_p = row_ptr;
    if (r0 != 0 || a0 != 0) {
        off_v.resize((size_t)n_cells + 1);
        rp_v.resize(n_reads + 1);
        for (uint32_t c = 0; c <= n_cells; ++c) off_v[c] = cell_row_off[c0 + c] - r0;
        unsigned nt = std::thread::hardware_concurrency();
        if (nt > 16) nt = 16;
        if (nt < 1 || n_reads < (1u << 20)) nt = 1;
        auto rebase = [&](unsigned k) {
            const uint64_t b = (n_reads + 1) * k / nt, e = (n_reads + 1) * (k + 1) / nt;
            for (uint64_t r = b; r < e; ++r) rp_v[r] = row_ptr[r0 + r] - a0;
        };
        if (nt == 1) {
            rebase(0);
        } else {
            std::vector<std::thread> th;
            for (unsigned k = 0; k < nt; ++k) th.emplace_back(rebase, k);
            for (auto &t : th) t.join();
        }
        off_p = off_v.data();
        rp_p = rp_v.data();
    }
    const uint32_t *tid_g = tid ? tid + a0 : nullptr;
    const float *p_g = as_prob ? as_prob + a0 : nullptr;
    const double *cov_g = cov_prob ? cov_prob + a0 : nullptr;
    double *out_g = out + (uint64_t)c0 * n_txps;
    oem_run_info *infos_g = infos ? infos + c0 : nullptr;

    if (knob("OEM_SERIAL_CELLS", 0) == 0) { // testing build: force the cell-by-cell path
        bool used = false;
        int rcb = run_cells_batched(off_p, n_cells, rp_p, tid_g, p_g, cov_g, n_reads, nnz, n_txps, device,
                                    max_iter, conv_thresh, out_g, infos_g, &used);
        if (rcb != OEM_OK || used) return rcb;
    }
    // fallback (max_iter == 0, a single cell, or a group the tiler declines): cells one after another
    oem_store *s = nullptr;
    oem_store_opts opts;
    std::memset(&opts, 0, sizeof(opts));
    opts.reorder_rows = 1; // cells are row ranges of the caller-order CSR
    OEM_TRY(oem_store_create(rp_p, tid_g, p_g, cov_g, n_reads, nnz, n_txps, device, &opts, &s));
    int rc = OEM_OK;
    for (uint32_t c = 0; c < n_cells && rc == OEM_OK; ++c) {
        RunArgs a;
        a.row_begin = off_p[c];
        a.row_end = off_p[c + 1];
        a.total_reads = a.row_end - a.row_begin; // the cell's own store.len() (single_cell.rs:122-130)
        a.max_iter = max_iter;
        a.conv_thresh = conv_thresh;
        a.min_iter_gate = 50;                    // em::em (single_cell.rs:150)
        rc = run_em_device(s, a, infos_g ? &infos_g[c] : nullptr);
        if (rc == OEM_OK) rc = copy_counts_out(s, out_g + (uint64_t)c * n_txps);
    }
    free_store(s);
    return rc;
}

} // namespace
} // namespace oem

// ---------------------------------------------------------------------------
// single-cell batch
// ---------------------------------------------------------------------------
extern "C" int oem_em_run_cells(const uint64_t *cell_row_off, uint32_t n_cells, const uint64_t *row_ptr,
                                const uint32_t *tid, const float *as_prob, const double *cov_prob,
                                uint64_t n_reads, uint64_t nnz, uint32_t n_txps, int device,
                                uint32_t max_iter, double conv_thresh, double *out,
                                oem_run_info *infos)
{
    OEM_API_BEGIN
    if (!cell_row_off || !row_ptr || (n_cells && !out)) return fail(OEM_ERR_ARG, "oem_em_run_cells: NULL argument");
    if (n_txps == 0) return fail(OEM_ERR_ARG, "oem_em_run_cells: n_txps is 0");
    if (cell_row_off[0] != 0 || cell_row_off[n_cells] != n_reads)
        return fail(OEM_ERR_ARG, "oem_em_run_cells: cell_row_off must span [0, n_reads]");
    for (uint32_t c = 0; c < n_cells; ++c)
        if (cell_row_off[c + 1] < cell_row_off[c])
            return fail(OEM_ERR_ARG, "oem_em_run_cells: cell_row_off not non-decreasing at cell %u", c);
    if (nnz > 0 && (!tid || !as_prob)) return fail(OEM_ERR_ARG, "oem_em_run_cells: tid/as_prob is NULL");
    t_cells_loop_ms = 0.0;
    t_cells_batched_passes = 0;
    OEM_TRY(validate_csr(row_ptr, tid, n_reads, nnz, n_txps)); // all cells at once, on several host threads
    // a read with a NaN coverage probability is dropped (em.rs:115), on every path below: the batched
    // groups create their stores directly, not through oem_store_create
    std::vector<double> cov_fixed;
    if (cov_prob && zero_nan_rows(row_ptr, cov_prob, n_reads, nnz, &cov_fixed)) cov_prob = cov_fixed.data();

    // Cells are independent problems, so a large experiment is cut into groups of consecutive cells
    // that bound the batched store (transcript space < 2^32, <= 2^30 alignments, and the layout
    // builder's tile x bucket table); each group is one batched run on the device.
    const uint64_t max_group_nnz = (uint64_t)knob("OEM_CELLS_GROUP_NNZ", 1l << 30); // testing build: small groups
    uint32_t c0 = 0;
    while (c0 < n_cells) {
        uint32_t c1 = c0 + 1;
        while (c1 < n_cells) {
            const uint64_t cells = (uint64_t)(c1 + 1 - c0);
            const uint64_t reads = cell_row_off[c1 + 1] - cell_row_off[c0];
            const uint64_t gnnz = row_ptr[cell_row_off[c1 + 1]] - row_ptr[cell_row_off[c0]];
            const uint64_t buckets = (cells * n_txps + kBucket - 1) / kBucket;
            // tiles per group: ~300 reads per tile with the narrow window cap on sparse cells, ~700 with the
            // wide one that create_store_impl picks below 4 reads per transcript
            const bool wide = reads < 2 * cells * n_txps && reads >= 1000000; // as create_store_impl chooses
            const uint64_t tiles_est = reads / (wide ? 600 : 256) + 2 * cells;
            if (cells * n_txps >= (1ull << 32) || reads >= (1ull << 32) || gnnz > max_group_nnz ||
                tiles_est * buckets > (1ull << 28) || cells > 65535 /* gridDim.y of the per-cell kernels */)
                break;
            ++c1;
        }
        OEM_TRY(run_cells_group(cell_row_off, c0, c1, row_ptr, tid, as_prob, cov_prob, n_txps, device, max_iter,
                                conv_thresh, out, infos));
        c0 = c1;
    }
    return OEM_OK;
    OEM_API_END("oem_em_run_cells")
}

// ---------------------------------------------------------------------------
// multi-GPU
// ---------------------------------------------------------------------------
extern "C" int oem_store_attach_comm(oem_store *s, oem_comm *comm, uint64_t global_n_reads,
                                     uint64_t global_row_offset)
{
    OEM_API_BEGIN
    if (!s) return fail(OEM_ERR_ARG, "oem_store_attach_comm: store is NULL");
    if (global_row_offset + s->csr.n_reads > global_n_reads)
        return fail(OEM_ERR_ARG, "oem_store_attach_comm: shard [%llu,+%llu) exceeds %llu reads",
                    (unsigned long long)global_row_offset, (unsigned long long)s->csr.n_reads,
                    (unsigned long long)global_n_reads);
    if (comm && comm_size(reinterpret_cast<Comm *>(comm)) > 1 && !comm_exchanges(reinterpret_cast<Comm *>(comm)))
        return fail(OEM_ERR_STATE, "oem_store_attach_comm: a communicator of %d ranks with no backend connected "
                    "(RCCL unique id, or oem_comm_p2p_export + oem_comm_p2p_connect)", comm_size(reinterpret_cast<Comm *>(comm)));
    std::lock_guard<std::mutex> lk(s->mu);
    s->comm = reinterpret_cast<Comm *>(comm);
    s->global_n_reads = global_n_reads;
    s->global_row_offset = global_row_offset;
    return OEM_OK;
    OEM_API_END("oem_store_attach_comm")
}

// ---------------------------------------------------------------------------
// measurement
// ---------------------------------------------------------------------------
extern "C" int oem_time_m_step(oem_store *s, uint32_t n_launches, float *out_avg_ms)
{
    OEM_API_BEGIN
    if (!s || !out_avg_ms || n_launches == 0) return fail(OEM_ERR_ARG, "oem_time_m_step: bad argument");
    std::lock_guard<std::mutex> lk(s->mu);
    OEM_TRY(ensure_device(s->device));
    const uint32_t T = s->csr.n_txps;
    OEM_TRY(launch_fill(s, s->theta, (double)s->global_n_reads / (double)T, T));
    OEM_HIP(hipMemsetAsync(s->cnt, 0, sizeof(double) * T, s->stream));
    hipEvent_t e0, e1;
    OEM_HIP(hipEventCreate(&e0));
    OEM_HIP(hipEventCreate(&e1));
    RunArgs a;
    a.row_end = s->csr.n_reads;
    // one untimed launch to page the kernel in
    OEM_TRY(enqueue_pass(s, a, nullptr));
    // the passes are launched the way the loop launches them: from a graph, in chunks (counts are not
    // cleared in between: they only grow, the work does not change)
    ChunkGraph cg;
    constexpr uint32_t kPer = 10;
    if (graph_ok(s) && n_launches >= kPer && n_launches % kPer == 0)
        OEM_TRY(capture_chunk(s->stream, kPer, [&]() { return enqueue_pass(s, a, nullptr); }, &cg));
    if (cg.ready()) {
        OEM_HIP(hipGraphLaunch(cg.ge, s->stream)); // untimed: upload
        OEM_HIP(hipEventRecord(e0, s->stream));
        for (uint32_t k = 0; k < n_launches; k += kPer) OEM_HIP(hipGraphLaunch(cg.ge, s->stream));
    } else {
        OEM_HIP(hipEventRecord(e0, s->stream));
        for (uint32_t k = 0; k < n_launches; ++k) OEM_TRY(enqueue_pass(s, a, nullptr));
    }
    OEM_HIP(hipEventRecord(e1, s->stream));
    OEM_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    OEM_HIP(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *out_avg_ms = ms / (float)n_launches;
    return OEM_OK;
    OEM_API_END("oem_time_m_step")
}

extern "C" int oem_time_em_iters(oem_store *s, uint32_t n_iters, float *out_ms)
{
    OEM_API_BEGIN
    if (!s || !out_ms) return fail(OEM_ERR_ARG, "oem_time_em_iters: bad argument");
    std::lock_guard<std::mutex> lk(s->mu);
    OEM_TRY(ensure_device(s->device));
    const uint32_t T = s->csr.n_txps;
    RunArgs a;
    a.row_end = s->csr.n_reads;
    a.total_reads = s->global_n_reads;
    a.max_iter = n_iters;
    a.conv_thresh = -1.0; // rel_diff >= 0 is never < -1: no early exit (SURVEY.md 8a note 3)
    EmParams p{T, a.max_iter, 0xffffffffu, a.conv_thresh};
    OEM_TRY(launch_fill(s, s->theta, (double)a.total_reads / (double)T, T));
    OEM_HIP(hipMemsetAsync(s->cnt, 0, sizeof(double) * T, s->stream));
    OEM_HIP(hipMemsetAsync(s->d_state, 0, sizeof(EmState), s->stream));
    hipEvent_t e0, e1;
    OEM_HIP(hipEventCreate(&e0));
    OEM_HIP(hipEventCreate(&e1));
    OEM_HIP(hipEventRecord(e0, s->stream));
    ChunkGraph cg; // launched the way oem_em_run launches: chunks of kGraphIters iterations from a graph
    if (graph_ok(s) && n_iters >= kGraphIters && n_iters % kGraphIters == 0)
        OEM_TRY(capture_chunk(s->stream, kGraphIters, [&]() { return enqueue_iteration(s, a, p); }, &cg));
    if (cg.ready()) {
        OEM_HIP(hipGraphLaunch(cg.ge, s->stream)); // untimed: the first launch of an executable graph uploads it
        OEM_HIP(hipMemsetAsync(s->d_state, 0, sizeof(EmState), s->stream));
        OEM_HIP(hipEventRecord(e0, s->stream));
        for (uint32_t k = 0; k < n_iters; k += kGraphIters) OEM_HIP(hipGraphLaunch(cg.ge, s->stream));
        OEM_HIP(hipEventRecord(e1, s->stream));
        OEM_HIP(hipEventSynchronize(e1));
        float gms = 0.f;
        OEM_HIP(hipEventElapsedTime(&gms, e0, e1));
        hipEventDestroy(e0);
        hipEventDestroy(e1);
        OEM_TRY(comm_check(s->comm, s->stream));
        *out_ms = gms;
        return OEM_OK;
    }
    for (uint32_t k = 0; k < n_iters; ++k) OEM_TRY(enqueue_iteration(s, a, p));
    OEM_HIP(hipEventRecord(e1, s->stream));
    OEM_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    OEM_HIP(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *out_ms = ms;
    return OEM_OK;
    OEM_API_END("oem_time_em_iters")
}

extern "C" int oem_time_allreduce(oem_store *s, uint32_t n_calls, float *out_avg_us)
{
    OEM_API_BEGIN
    if (!s || !out_avg_us || n_calls == 0) return fail(OEM_ERR_ARG, "oem_time_allreduce: bad argument");
    std::lock_guard<std::mutex> lk(s->mu);
    OEM_TRY(ensure_device(s->device));
    if (!comm_exchanges(s->comm)) return fail(OEM_ERR_STATE, "oem_time_allreduce: no communicator attached");
    const uint32_t T = s->csr.n_txps;
    OEM_HIP(hipMemsetAsync(s->cnt, 0, sizeof(double) * T, s->stream));
    hipEvent_t e0, e1;
    OEM_HIP(hipEventCreate(&e0));
    OEM_HIP(hipEventCreate(&e1));
    OEM_TRY(comm_allreduce_sum_f64(s->comm, s->cnt, s->cnt, T, s->stream)); // untimed: first-use set-up
    OEM_HIP(hipEventRecord(e0, s->stream));
    for (uint32_t k = 0; k < n_calls; ++k) OEM_TRY(comm_allreduce_sum_f64(s->comm, s->cnt, s->cnt, T, s->stream));
    OEM_HIP(hipEventRecord(e1, s->stream));
    OEM_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    OEM_HIP(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    OEM_TRY(comm_check(s->comm, s->stream));
    *out_avg_us = ms * 1e3f / (float)n_calls;
    return OEM_OK;
    OEM_API_END("oem_time_allreduce")
}

extern "C" int oem_cells_last_timing(float *out_loop_ms, uint64_t *out_batched_passes)
{
    OEM_API_BEGIN
    if (out_loop_ms) *out_loop_ms = (float)t_cells_loop_ms;
    if (out_batched_passes) *out_batched_passes = t_cells_batched_passes;
    return OEM_OK;
    OEM_API_END("oem_cells_last_timing")
}

extern "C" int oem_time_bootstrap_passes(oem_store *s, uint32_t n_passes, float *out_avg_ms, uint32_t *out_slots,
                                         uint64_t *out_algorithmic_bytes)
{
    OEM_API_BEGIN
    if (!s || !out_avg_ms || n_passes == 0) return fail(OEM_ERR_ARG, "oem_time_bootstrap_passes: bad argument");
    std::lock_guard<std::mutex> lk(s->mu);
    OEM_TRY(ensure_device(s->device));
    if (!can_batch(s)) return fail(OEM_ERR_STATE, "oem_time_bootstrap_passes: this store runs its bootstraps one per pass");
    OEM_TRY(ensure_batch(s, 0));
    BatchBuffers &bb = s->batch[0];
    const uint32_t T = s->csr.n_txps;
    const uint64_t R = s->csr.n_reads;
    const double avg = (double)s->global_n_reads / (double)T;
    OEM_HIP(hipMemsetAsync(bb.cnt, 0, sizeof(double) * T * kBatch, s->stream));
    for (int k = 0; k < kBatch; ++k) { // every slot RUNNING on its own device-drawn resample
        OEM_TRY(launch_bootstrap_weights(s, bb.d_row_w, R, s->global_row_offset, s->global_n_reads, 0x7e57ull, (uint32_t)k));
        OEM_HIP(hipMemsetAsync(bb.overflow, 0, sizeof(uint32_t), s->stream));
        OEM_TRY(launch_batch_pack_row_w(s, bb.d_row_w, bb, (uint32_t)k, bb.overflow));
        OEM_TRY(launch_batch_reset_slot(s, bb, nullptr, avg, (uint32_t)k));
        std::memset(&bb.h_state[k], 0, sizeof(BatchState));
        bb.h_state[k].phase = kPhaseRunning;
    }
    OEM_HIP(hipMemcpyAsync(bb.state, bb.h_state, sizeof(BatchState) * kBatch, hipMemcpyHostToDevice, s->stream));
    EmParams p{T, 0xffffffffu, 0xffffffffu, -1.0}; // no slot ever stops (SURVEY.md 8a note 3)
    hipEvent_t e0, e1;
    OEM_HIP(hipEventCreate(&e0));
    OEM_HIP(hipEventCreate(&e1));
    OEM_TRY(launch_batch_pass(s, bb)); // one untimed pass
    OEM_TRY(launch_batch_reldiff(s, bb, p));
    auto one_pass = [&]() -> int {
        OEM_TRY(launch_batch_pass(s, bb));
        return launch_batch_reldiff(s, bb, p);
    };
    ChunkGraph cg; // launched the way oem_bootstrap launches its passes
    constexpr uint32_t kPer = 5;
    if (graph_ok(s) && n_passes % kPer == 0) OEM_TRY(capture_chunk(s->stream, kPer, one_pass, &cg));
    if (cg.ready()) {
        OEM_HIP(hipGraphLaunch(cg.ge, s->stream)); // untimed: upload
        OEM_HIP(hipEventRecord(e0, s->stream));
        for (uint32_t i = 0; i < n_passes; i += kPer) OEM_HIP(hipGraphLaunch(cg.ge, s->stream));
    } else {
        OEM_HIP(hipEventRecord(e0, s->stream));
        for (uint32_t i = 0; i < n_passes; ++i) OEM_TRY(one_pass());
    }
    OEM_HIP(hipEventRecord(e1, s->stream));
    OEM_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    OEM_HIP(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *out_avg_ms = ms / (float)n_passes;
    if (out_slots) *out_slots = kBatch;
    if (out_algorithmic_bytes) {
        // the matrix once per batched pass (nnz * (4 + 4|8) + row pointers), and per replicate the
        // row weights (R * 4) and theta read / counts written once per transcript (2 * T * 8)
        const DeviceCsr &m = s->csr;
        *out_algorithmic_bytes = m.nnz * (4 + (m.w_is_f64 ? 8 : 4)) + (m.n_reads + 1) * (m.wide_ptr ? 8 : 4) +
                                 (uint64_t)kBatch * (m.n_reads * 4 + 2ull * m.n_txps * 8);
    }
    return OEM_OK;
    OEM_API_END("oem_time_bootstrap_passes")
}
