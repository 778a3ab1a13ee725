// oem_cells.hip -- single_cell.rs:139-160: an independent em::em per cell.  All cells of a group are laid out as
// ONE store over a concatenated transcript space and share every pass (oem_multi_kernels.hip); groups the tiler
// declines run cell after cell over the caller-order CSR.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "oem_driver.h"

namespace oem {
namespace {

#ifndef OEM_CELLS_HEAD_DIV
#define OEM_CELLS_HEAD_DIV 4 // a large single group is split head : rest = 1 : (div - 1); 0 = not split
#endif
constexpr uint32_t kCellsHeadDiv = OEM_CELLS_HEAD_DIV;


// What the last oem_em_run_cells call of this thread spent in its batched EM loops (HIP events on the
// group's stream around the loop), for oem_cells_last_timing.
thread_local double t_cells_loop_ms = 0.0;
thread_local uint64_t t_cells_batched_passes = 0;
// (the groups of one call may run on two host threads: they add into the call's accumulators under this lock,
// and the calling thread copies them into its thread-local pair at the end)
struct CellsTiming {
    std::mutex mu;
    std::vector<std::pair<double, double>> loops; // [begin, end) of every group's EM loop, ms on the host's steady clock
    uint64_t passes = 0;
    // Time during which at least one group's loop ran: groups of one call overlap on the device (two workers), and
    // the sum of their durations would count the shared time twice.
    double loop_ms()
    {
        std::sort(loops.begin(), loops.end());
        double total = 0.0, cur_b = 0.0, cur_e = -1.0;
        for (const auto &iv : loops) {
            if (cur_e < cur_b || iv.first > cur_e) {
                if (cur_e > cur_b) total += cur_e - cur_b;
                cur_b = iv.first;
                cur_e = iv.second;
            } else if (iv.second > cur_e) {
                cur_e = iv.second;
            }
        }
        if (cur_e > cur_b) total += cur_e - cur_b;
        return total;
    }
};
thread_local CellsTiming *t_timing = nullptr;

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
    // transcripts per cell IN THE STORE: the ones that occur in the cell, padded to the fullest cell's count
    // (oem_api.hip: k_cells_mark); the caller's n_txps where the batch was not compacted
    const uint64_t full_total = total_txps;
    const bool compacted = s->multi.rank != nullptr;
    if (compacted) n_txps = s->multi.txps_eff;
    const uint64_t store_total = (uint64_t)n_cells * n_txps;
    if (tm.on)
        fprintf(stderr, "[oem] cells: %u transcripts per cell in the store (of %u), %u tiles, %llu local + %llu remote alignments\n",
                n_txps, (unsigned)(full_total / n_cells), s->tiled.n_tiles, (unsigned long long)s->tiled.n_local,
                (unsigned long long)s->tiled.n_remote);

    auto body = [&]() -> int {
        MultiBuffers &mb = s->multi;
        mb.n_problems = n_cells;
        mb.problem_size = n_txps;
        OEM_TRY(dev_alloc(&mb.state, n_cells, &s->hbm_bytes));
        OEM_TRY(dev_alloc(&mb.out, (size_t)store_total, &s->hbm_bytes));
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
                hipMemsetAsync(s->cnt, 0, sizeof(double) * store_total, s->stream) != hipSuccess) {
                rc2 = fail(OEM_ERR_HIP, "oem_em_run_cells: upload of the per-cell state failed");
                break;
            }
            if ((rc2 = launch_multi_init(s, s->theta, d_reads, mb)) != OEM_OK) break;
            EmParams p{n_txps, max_iter, 50u /* em::em, single_cell.rs:150 */, conv_thresh};
            if (hipMemsetAsync(mb.out, 0, sizeof(double) * store_total, s->stream) != hipSuccess) {
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
                    if (t_timing) { // the loop ended just now and lasted `ms` on the device
                        const double end = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
                        std::lock_guard<std::mutex> lk(t_timing->mu);
                        t_timing->loops.emplace_back(end - (double)ms, end);
                        t_timing->passes += launched;
                    }
                }
            }
            hipEventDestroy(ev0);
            hipEventDestroy(ev1);
            if (rc2 != OEM_OK) break;
            tm.lap("cells: EM loop");
            const double *d_res = mb.out;
            double *d_full = nullptr;
            bool expanded_on_host = false;
            if (compacted && (knob("OEM_TEST_FAIL_FULL_ALLOC", 0) || hipMalloc((void **)&d_full, sizeof(double) * full_total) != hipSuccess)) {
                // no device memory for the expanded results: the compact ones and the rank table go to the host,
                // which expands them (a transcript that does not occur in a cell is 0)
                (void)hipGetLastError();
                d_full = nullptr;
                const size_t n_eff = (size_t)mb.n_problems * mb.txps_eff;
                std::vector<double> h_eff(n_eff);
                std::vector<uint32_t> h_rank((size_t)full_total);
                if (hipMemcpy(h_eff.data(), mb.out, sizeof(double) * n_eff, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(h_rank.data(), mb.rank, sizeof(uint32_t) * full_total, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(hs.data(), mb.state, sizeof(BatchState) * n_cells, hipMemcpyDeviceToHost) != hipSuccess) {
                    rc2 = fail(OEM_ERR_HIP, "oem_em_run_cells: read-back failed");
                    break;
                }
                for (size_t i = 0; i < (size_t)full_total; ++i)
                    out[i] = h_rank[i] == kNoRank ? 0.0 : h_eff[(i / mb.txps_full) * mb.txps_eff + h_rank[i]];
                expanded_on_host = true;
            }
            if (compacted && !expanded_on_host) { // expand to the caller's [cell][transcript] (the queue is done with: its memory is free by now)
                if ((rc2 = launch_multi_expand(s, mb, d_full)) != OEM_OK || hipStreamSynchronize(s->stream) != hipSuccess) {
                    hipFree(d_full);
                    if (rc2 == OEM_OK) rc2 = fail(OEM_ERR_HIP, "oem_em_run_cells: expanding the results failed");
                    break;
                }
                d_res = d_full;
            }
            const bool copied = expanded_on_host || hipMemcpy(out, d_res, sizeof(double) * full_total, hipMemcpyDeviceToHost) == hipSuccess;
            hipFree(d_full);
            if (!copied ||
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

namespace oem {
void cells_last_timing(double *loop_ms, uint64_t *batched_passes)
{
    if (loop_ms) *loop_ms = t_cells_loop_ms;
    if (batched_passes) *batched_passes = t_cells_batched_passes;
}
} // namespace oem

using namespace oem;

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
    StageTimer tm_all;
    OEM_TRY(validate_csr(row_ptr, tid, n_reads, nnz, n_txps)); // all cells at once, on several host threads
    tm_all.lap("cells: range checks");
    // a read with a NaN coverage probability is dropped (em.rs:115), on every path below: the batched
    // groups create their stores directly, not through oem_store_create
    std::vector<double> cov_fixed;
    if (cov_prob && zero_nan_rows(row_ptr, cov_prob, n_reads, nnz, &cov_fixed)) cov_prob = cov_fixed.data();

    // Cells are independent problems, so a large experiment is cut into groups of consecutive cells
    // that bound the batched store (transcript space < 2^32, <= 2^30 alignments, and the layout
    // builder's tile x bucket table); each group is one batched run on the device.
    const uint64_t max_group_nnz = (uint64_t)knob("OEM_CELLS_GROUP_NNZ", 1l << 30); // testing build: small groups
    std::vector<std::pair<uint32_t, uint32_t>> groups;
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
                tiles_est * buckets > (1ull << 29) || cells > 65535 /* gridDim.y of the per-cell kernels */)
                break;
            ++c1;
        }
        groups.emplace_back(c0, c1);
        c0 = c1;
    }
    // One large group only (BASELINE configs[4]'s slice of one GPU: 625 cells, 250 M alignments, 2 GB of caller arrays):
    // a quarter of the cells is cut off as a group of its own, so that the second worker uploads and lays out the rest
    // under the head's EM loop instead of the device idling through the whole upload and layout build (~70 ms of a
    // 0.67 s call).  Measured (scripts/cells_groups_exp.sh, three rounds): heads of 40 / 80 / 160 / 312 of 625 cells
    // +7 / -0.5 / -3.4 / -0.5 % against one group -- a small head's own loop runs its few tiles badly, two halves just
    // share the device.
    if (groups.size() == 1 && n_cells >= 64 && nnz >= (64ull << 20)) {
        const long head = knob("OEM_CELLS_HEAD", (long)(kCellsHeadDiv ? n_cells / kCellsHeadDiv : 0));
        if (head >= 2 && (uint32_t)head + 2 <= n_cells) {
            groups.clear();
            groups.emplace_back(0u, (uint32_t)head);
            groups.emplace_back((uint32_t)head, n_cells);
        }
    }
    // Groups are independent runs.  With several of them two host threads draw groups from one counter, each group
    // on its own stream: one group's upload, layout build and read-back run under the other's EM loop, and the tail
    // of a loop -- the few cells that run into max_iter, a handful of live tiles per pass -- shares the device with
    // the other group's full passes instead of leaving it idle (single_cell.rs:96-150 runs its cells on N worker
    // threads for the same reason).
    CellsTiming timing;
    std::atomic<size_t> next{0};
    constexpr int kMaxWorkers = 4;
    int n_workers = (int)knob("OEM_CELLS_WORKERS", 2);
    if (n_workers > kMaxWorkers) n_workers = kMaxWorkers;
    if ((size_t)n_workers > groups.size()) n_workers = (int)groups.size();
    if (n_workers < 1) n_workers = 1;
    int rcs[kMaxWorkers] = {OEM_OK, OEM_OK, OEM_OK, OEM_OK}; // each worker's own; read by the others only after the join
    std::atomic<bool> failed{false};                         // ... and this is what they stop on
    std::string errs[kMaxWorkers];
    auto work = [&](int wk) {
        t_timing = &timing;
        if (wk != 0 && hipSetDevice(device) != hipSuccess) {
            rcs[wk] = OEM_ERR_HIP;
            errs[wk] = "hipSetDevice failed in a per-cell worker";
            failed.store(true);
            return;
        }
        try {
            for (;;) {
                const size_t g = next.fetch_add(1);
                if (g >= groups.size() || failed.load()) break;
                rcs[wk] = run_cells_group(cell_row_off, groups[g].first, groups[g].second, row_ptr, tid, as_prob, cov_prob,
                                          n_txps, device, max_iter, conv_thresh, out, infos);
                if (rcs[wk] != OEM_OK) break;
            }
        } catch (const std::exception &e) {
            rcs[wk] = fail(OEM_ERR_OOM, "per-cell worker: %s", e.what());
        } catch (...) {
            rcs[wk] = fail(OEM_ERR_STATE, "per-cell worker: unknown C++ exception");
        }
        if (rcs[wk] != OEM_OK) failed.store(true);
        if (rcs[wk] != OEM_OK && errs[wk].empty()) errs[wk] = last_error_text(); // (the message is thread-local)
        t_timing = nullptr;
    };
    {
        struct Joiner { // (a std::thread constructor that throws must not leave joinable threads behind)
            std::vector<std::thread> th;
            ~Joiner() { for (auto &t : th) if (t.joinable()) t.join(); }
        } pool;
        try {
            for (int wk = 1; wk < n_workers; ++wk) pool.th.emplace_back(work, wk);
        } catch (...) { // fewer threads: the ones that started take all the groups
        }
        work(0);
    }
    tm_all.lap("cells: all groups");
    t_cells_loop_ms = timing.loop_ms();
    t_cells_batched_passes = timing.passes;
    for (int wk = 0; wk < kMaxWorkers; ++wk)
        if (rcs[wk] != OEM_OK) return fail(rcs[wk], "%s", errs[wk].c_str());
    return OEM_OK;
    OEM_API_END("oem_em_run_cells")
}
