// oem_em_driver.hip -- em.rs:144-255 / :320-447 with the loop state on the device: one EM run over a resident
// store (point estimate, a resampled replicate, a row range of a per-cell store), and the C ABI entry points that
// are one such run or one pass: oem_m_step, oem_em_run, and the steps right after the EM (aux counts, posterior).
#include <cstring>

#include "oem_driver.h"

namespace oem {

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

// ---- the stopping rule one pass behind ------------------------------------------------------------------------
// An iteration of do_em is a pass and then a sweep over the two count vectors (rel-diff, swap, clear: em.rs:194-207)
// whose only product is one number and a decision.  On the device that sweep is a 6 us kernel between 0.15 ms passes
// -- 27 us passes at 1 M reads -- with a launch boundary on either side.  Here it rides on the NEXT pass: theta_{i+1}
// simply IS the vector pass i accumulated into (no swap), the tile workgroups of pass i + 1 each compare their share
// of theta_i and theta_{i+1} and zero their share of theta_i (it becomes the accumulator of pass i + 2: three
// vectors rotate), and the first workgroup of pass i + 1's fold applies the rule (DeferredRelDiff, oem_internal.h).
// If it says stop, pass i + 1 was speculative and its accumulator is dropped: the loop has run the reference's
// iterations, stopped where the reference stops, and theta is what the reference holds at that point; the cost is one
// pass per run that CONVERGES.  A run that reaches max_iter pays nothing: when pass max_iter would be due the loop ends
// whatever the comparison says (em.rs:181), so its last iteration is decided by a sweep over the two vectors alone
// (k_deferred_sweep) -- max_iter iterations are max_iter passes + one 3 us kernel.  Single-device stores on the tiled
// path; row shards keep the sweep of every iteration (it carries their exchange).
//
// What "speculative" obliges (the kernels rely on it, nothing else enforces it): (1) the workgroup that decides writes
// `done` while the other workgroups of the SAME fold launch may already have read it as 0 or still read it as 1, so the
// accumulator of a deciding pass can be folded in part -- it is `rest` below and is never read; (2) a tile workgroup
// must look at `done` BEFORE it zeroes its share of rd_prev (k_em_tile returns at the top), or the empty launches behind
// the decision would wipe the buffer the final pass accumulates into and, one launch later, the final theta.
#ifndef OEM_DEFERRED_DEFAULT
#define OEM_DEFERRED_DEFAULT 1 // (scripts/build_variant.sh: 0 builds the classic loop into the product library for an A/B)
#endif
bool deferred_reldiff_ok(const oem_store *s, const RunArgs &a)
{
    // (a store whose reads are all empty has a tiled layout without tiles: its passes launch nothing that could carry
    // the decision -- the classic loop's sweep returns the reference's zeros)
    return use_tiled(s, a) && s->tiled.n_tiles > 0 && !comm_exchanges(s->comm) && !graph_ok(s) &&
           knob("OEM_DEFERRED_RELDIFF", OEM_DEFERRED_DEFAULT) != 0;
}

int ensure_deferred(oem_store *s)
{
    if (!s->third) OEM_TRY(dev_alloc(&s->third, s->csr.n_txps, &s->hbm_bytes));
    if (!s->rel_slots) OEM_TRY(dev_alloc(&s->rel_slots, kRelSlots, &s->hbm_bytes));
    return OEM_OK;
}

// pass i of a deferred run: theta_i in bufs[i % 3], accumulator bufs[(i + 1) % 3], theta_{i-1} in bufs[(i + 2) % 3].
// i == max_iter is not a pass: iteration max_iter - 1 is decided by the sweep alone (the loop ends there either way).
int enqueue_deferred_pass(oem_store *s, const RunArgs &a, const EmParams &p, double *const bufs[3], uint64_t i)
{
    DeferredRelDiff rd{i > 0 ? bufs[(i + 2) % 3] : nullptr, s->rel_slots, s->d_state, p, 1u + (uint32_t)(i % 3)};
    if (i > 0 && i == p.max_iter) return launch_deferred_sweep(s, rd.prev, bufs[i % 3], rd);
    return launch_em_pass_tiled(s, bufs[i % 3], bufs[(i + 1) % 3], s->d_state, a.d_row_w ? s->tiled.row_w_perm : nullptr,
                                nullptr, 0, false, &rd);
}

static int run_em_deferred(oem_store *s, const RunArgs &a, oem_run_info *info)
{
    const uint32_t T = s->csr.n_txps;
    EmParams p{T, a.max_iter, a.min_iter_gate, a.conv_thresh};
    OEM_TRY(ensure_deferred(s));
    double *const bufs[3] = {s->theta, s->cnt, s->third};
    if (a.init) OEM_HIP(hipMemcpyAsync(bufs[0], a.init, sizeof(double) * T, hipMemcpyHostToDevice, s->stream));
    OEM_TRY(launch_deferred_init(s, bufs, (double)a.total_reads / (double)T, a.init == nullptr)); // em.rs:160-166
    std::memset(s->h_state, 0, sizeof(EmState));
    OEM_TRY(prepare_row_w(s, a));
    // iteration j is decided by launch j + 1: max_iter iterations take max_iter passes + the sweep that decides the last
    const uint64_t n_total = a.max_iter ? (uint64_t)a.max_iter + 1 : 0;
    // The host stays ONE chunk ahead of its look at the state: the next 16 passes are in the queue before it waits for
    // the copy taken behind the previous ones, so the device never idles for the host's round trip (passes launched
    // after the rule has fired return at once: at most a chunk of empty launches per run).
    uint64_t launched = 0;
    auto enqueue_chunk = [&]() -> int {
        uint64_t chunk = launched == 0 ? (uint64_t)a.min_iter_gate + 3 : 16;
        if (chunk > n_total - launched) chunk = n_total - launched;
        if (chunk > 4096) chunk = 4096;
        for (uint64_t k = 0; k < chunk; ++k) OEM_TRY(enqueue_deferred_pass(s, a, p, bufs, launched + k));
        launched += chunk;
        return OEM_OK;
    };
    hipEvent_t seen = nullptr;
    if (n_total) OEM_HIP(hipEventCreateWithFlags(&seen, hipEventDisableTiming));
    int rc = OEM_OK;
    if (n_total) {
        rc = enqueue_chunk();
        while (rc == OEM_OK) {
            // a copy of the state as the passes enqueued so far leave it ...
            if (hipMemcpyAsync(s->h_state, s->d_state, sizeof(EmState), hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
                hipEventRecord(seen, s->stream) != hipSuccess) {
                rc = fail(OEM_ERR_HIP, "oem_em_run: state copy failed");
                break;
            }
            // ... and the next chunk behind it, before the host waits for that copy
            const bool more = launched < n_total;
            if (more) rc = enqueue_chunk();
            if (rc != OEM_OK) break;
            if (hipEventSynchronize(seen) != hipSuccess) {
                rc = fail(OEM_ERR_HIP, "oem_em_run: waiting for the loop state failed");
                break;
            }
            if (s->h_state->done || !more) break;
        }
    }
    if (seen) hipEventDestroy(seen);
    if (rc != OEM_OK) return rc;
    if (n_total && !s->h_state->done) return fail(OEM_ERR_STATE, "the deferred stopping rule did not fire within max_iter passes and the last sweep");
    const uint32_t f = n_total ? s->h_state->pad[0] - 1u : 0u; // the buffer of the final abundances
    if (f > 2u) return fail(OEM_ERR_STATE, "the deferred stopping rule left no final buffer");
    // final: em.rs:238-252.  The buffer behind theta's was zeroed by the launch that decided; the one ahead holds that
    // pass's speculative counts (nothing, when the last sweep decided) and rests.
    double *theta = bufs[f], *cnt = bufs[(f + 2) % 3], *rest = bufs[(f + 1) % 3];
    OEM_TRY(launch_zero_small(s, theta, cnt, T));
    s->theta = theta;
    s->cnt = cnt;
    s->third = rest;
    OEM_TRY(enqueue_pass(s, a, nullptr));
    if (info) {
        info->niter = s->h_state->niter;
        info->n_passes = s->h_state->n_passes + 1;
        info->converged = s->h_state->converged;
        info->reserved = 0;
        info->rel_diff = s->h_state->last_rel;
    }
    return OEM_OK;
}

// em.rs:144-255 / :320-447 with the loop state on the device.  On return the
// final counts are in s->cnt (device); *info filled from the device state.
int run_em_device(oem_store *s, const RunArgs &a, oem_run_info *info)
{
    if (deferred_reldiff_ok(s, a)) return run_em_deferred(s, a, info);
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

} // namespace oem

using namespace oem;

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
