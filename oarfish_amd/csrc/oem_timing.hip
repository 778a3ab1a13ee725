// oem_timing.hip -- the measurement entry points of the ABI: HIP events on the store's stream around the very
// launches the drivers make (bench.py's roofline objects come from these).
#include <cstring>

#include "oem_driver.h"

using namespace oem;

namespace {
// the two events of a timed region; destroyed on every path out of the entry point
struct EventPair {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    EventPair() = default;
    EventPair(const EventPair &) = delete;
    EventPair &operator=(const EventPair &) = delete;
    ~EventPair()
    {
        if (e0) hipEventDestroy(e0);
        if (e1) hipEventDestroy(e1);
    }
    int create()
    {
        OEM_HIP(hipEventCreate(&e0));
        OEM_HIP(hipEventCreate(&e1));
        return OEM_OK;
    }
    // waits for e1; milliseconds between the two
    int elapsed(float *ms)
    {
        OEM_HIP(hipEventSynchronize(e1));
        OEM_HIP(hipEventElapsedTime(ms, e0, e1));
        return OEM_OK;
    }
};
} // namespace

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
    EventPair ev;
    OEM_TRY(ev.create());
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
        OEM_HIP(hipEventRecord(ev.e0, s->stream));
        for (uint32_t k = 0; k < n_launches; k += kPer) OEM_HIP(hipGraphLaunch(cg.ge, s->stream));
    } else {
        OEM_HIP(hipEventRecord(ev.e0, s->stream));
        for (uint32_t k = 0; k < n_launches; ++k) OEM_TRY(enqueue_pass(s, a, nullptr));
    }
    OEM_HIP(hipEventRecord(ev.e1, s->stream));
    float ms = 0.f;
    OEM_TRY(ev.elapsed(&ms));
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
    EventPair ev;
    OEM_TRY(ev.create());
    if (deferred_reldiff_ok(s, a)) { // as oem_em_run runs them: n_iters passes, the rule one pass behind, the last iteration decided by the sweep
        OEM_TRY(ensure_deferred(s));
        double *const bufs[3] = {s->theta, s->cnt, s->third};
        OEM_TRY(launch_deferred_init(s, bufs, (double)a.total_reads / (double)T, true));
        OEM_HIP(hipEventRecord(ev.e0, s->stream));
        for (uint64_t k = 0; k <= n_iters; ++k) OEM_TRY(enqueue_deferred_pass(s, a, p, bufs, k));
        OEM_HIP(hipEventRecord(ev.e1, s->stream));
        return ev.elapsed(out_ms);
    }
    OEM_TRY(launch_fill(s, s->theta, (double)a.total_reads / (double)T, T));
    OEM_HIP(hipMemsetAsync(s->cnt, 0, sizeof(double) * T, s->stream));
    OEM_HIP(hipMemsetAsync(s->d_state, 0, sizeof(EmState), s->stream));
    OEM_HIP(hipEventRecord(ev.e0, s->stream));
    ChunkGraph cg; // launched the way oem_em_run launches: chunks of kGraphIters iterations from a graph
    if (graph_ok(s) && n_iters >= kGraphIters && n_iters % kGraphIters == 0)
        OEM_TRY(capture_chunk(s->stream, kGraphIters, [&]() { return enqueue_iteration(s, a, p); }, &cg));
    if (cg.ready()) {
        OEM_HIP(hipGraphLaunch(cg.ge, s->stream)); // untimed: the first launch of an executable graph uploads it
        OEM_HIP(hipMemsetAsync(s->d_state, 0, sizeof(EmState), s->stream));
        OEM_HIP(hipEventRecord(ev.e0, s->stream));
        for (uint32_t k = 0; k < n_iters; k += kGraphIters) OEM_HIP(hipGraphLaunch(cg.ge, s->stream));
    } else {
        for (uint32_t k = 0; k < n_iters; ++k) OEM_TRY(enqueue_iteration(s, a, p));
    }
    OEM_HIP(hipEventRecord(ev.e1, s->stream));
    OEM_TRY(ev.elapsed(out_ms));
    return comm_check(s->comm, s->stream);
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
    EventPair ev;
    OEM_TRY(ev.create());
    OEM_TRY(comm_allreduce_sum_f64(s->comm, s->cnt, s->cnt, T, s->stream)); // untimed: first-use set-up
    OEM_HIP(hipEventRecord(ev.e0, s->stream));
    for (uint32_t k = 0; k < n_calls; ++k) OEM_TRY(comm_allreduce_sum_f64(s->comm, s->cnt, s->cnt, T, s->stream));
    OEM_HIP(hipEventRecord(ev.e1, s->stream));
    float ms = 0.f;
    OEM_TRY(ev.elapsed(&ms));
    OEM_TRY(comm_check(s->comm, s->stream));
    *out_avg_us = ms * 1e3f / (float)n_calls;
    return OEM_OK;
    OEM_API_END("oem_time_allreduce")
}
extern "C" int oem_cells_last_timing(float *out_loop_ms, uint64_t *out_batched_passes)
{
    OEM_API_BEGIN
    double ms = 0.0;
    uint64_t n = 0;
    cells_last_timing(&ms, &n);
    if (out_loop_ms) *out_loop_ms = (float)ms;
    if (out_batched_passes) *out_batched_passes = n;
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
    EventPair ev;
    OEM_TRY(ev.create());
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
        OEM_HIP(hipEventRecord(ev.e0, s->stream));
        for (uint32_t i = 0; i < n_passes; i += kPer) OEM_HIP(hipGraphLaunch(cg.ge, s->stream));
    } else {
        OEM_HIP(hipEventRecord(ev.e0, s->stream));
        for (uint32_t i = 0; i < n_passes; ++i) OEM_TRY(one_pass());
    }
    OEM_HIP(hipEventRecord(ev.e1, s->stream));
    float ms = 0.f;
    OEM_TRY(ev.elapsed(&ms));
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

