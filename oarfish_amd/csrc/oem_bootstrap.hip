// oem_bootstrap.hip -- em::bootstrap (em.rs:292-314): the replicates of one call, as chains of batched passes
// over the resident matrix (oem_batch_kernels.hip) or one per pass through the point-estimate kernels.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "oem_driver.h"

namespace oem {

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
    OEM_TRY(dev_alloc(&b.rel_slots, (size_t)kBatchRelSlots * kBatch, &s->hbm_bytes));
    OEM_HIP(hipMemsetAsync(b.rel_slots, 0, sizeof(unsigned long long) * kBatchRelSlots * kBatch, b.stream));
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

namespace {

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
        if (rcs[c] != OEM_OK) errs[c] = last_error_text(); // (the message is thread-local)
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

} // namespace
} // namespace oem

using namespace oem;

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
