// oem_api.hip -- the C ABI of include/oarfish_em.h: library entry points and the life cycle of a store
// (checks of the caller's arrays, upload, layout, destruction).  The drivers on top of a resident store are in
// oem_em_driver.hip (oem_em_run, oem_m_step and the steps right after the EM), oem_bootstrap.hip (oem_bootstrap),
// oem_cells.hip (oem_em_run_cells) and oem_timing.hip (the HIP-event timing entry points).
//
// Reference call sites the library replaces (COMBINE-lab/oarfish v0.10.3):
//   bulk.rs:155-159   em::em / em::em_par      -> oem_em_run
//   bulk.rs:178-194   em::bootstrap            -> oem_bootstrap
//   single_cell.rs:139-160  per-cell em::em    -> oem_em_run_cells
#include <algorithm>
#include <cstdarg>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "oem_driver.h"

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

const char *last_error_text() { return t_err; }

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
        hipFree(b.theta); hipFree(b.cnt); hipFree(b.out); hipFree(b.queue); hipFree(b.state); hipFree(b.rel_slots);
        hipFree(b.row_w); hipFree(b.overflow);
        if (b.h_state) hipHostFree(b.h_state);
        if (b.h_out) hipHostFree(b.h_out);
    }
    hipFree(s->multi.state); hipFree(s->multi.out); hipFree(s->multi.n_unfinished);
    hipFree(s->multi.live_tiles); hipFree(s->multi.live_buckets); hipFree(s->multi.d_live_counts);
    hipFree(s->multi.rank);
    hipFree(s->theta);
    hipFree(s->cnt);
    hipFree(s->third);
    hipFree(s->rel_slots);
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

namespace {

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

// ---- per-cell transcript compaction --------------------------------------------------------------------------
// A cell's reads touch a fraction of the annotation (single_cell.rs:139-160 runs every cell over ALL transcripts;
// the ones no alignment of the cell names keep count 0 from the first iteration on and take no part in a
// denominator or in the stopping rule -- em.rs:195-199 skips them once their abundance is 0, and at the first
// iteration their (0 - avg) / avg = -1 loses against the 0 the maximum starts from).  So a cell's transcript
// space in the batched store is the list of transcripts that OCCUR in it, in id order: windows span fewer ids
// (fuller tiles), and the per-pass sweep over counts and abundances shrinks with the list.  Every cell gets
// txps_eff = the longest list's length (uniform problem size: nothing else in the kernels changes); the results
// are expanded to the caller's [cell][transcript] on the way out.
__device__ __forceinline__ uint32_t cell_of_read(const unsigned long long *__restrict__ cell_row_off, uint32_t n_cells, uint64_t r)
{
    uint32_t a = 0, b = n_cells; // last cell whose first read is <= r
    while (b - a > 1) {
        const uint32_t m = (a + b) >> 1;
        if (cell_row_off[m] <= r) a = m;
        else b = m;
    }
    return a;
}

__global__ __launch_bounds__(256) void k_cells_mark(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ tid,
                                                    const unsigned long long *__restrict__ cell_row_off, uint32_t n_cells,
                                                    uint32_t cell_txps, uint64_t n_reads, uint32_t *__restrict__ rank)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const size_t base = (size_t)cell_of_read(cell_row_off, n_cells, r) * cell_txps;
    for (uint32_t j = row_ptr[r]; j < row_ptr[r + 1]; ++j) rank[base + tid[j]] = 1u; // (same value from every writer)
}

// one workgroup per cell: flags -> exclusive ranks (kNoRank for the transcripts that do not occur), count of the cell
constexpr int kRankT = 1024;
__global__ __launch_bounds__(kRankT) void k_cells_rank(uint32_t *__restrict__ rank, uint32_t cell_txps, uint32_t *__restrict__ counts)
{
    __shared__ uint32_t part[kRankT];
    uint32_t *rk = rank + (size_t)blockIdx.x * cell_txps;
    const uint32_t per = (cell_txps + kRankT - 1) / kRankT;
    const uint32_t i0 = threadIdx.x * per, i1 = i0 + per < cell_txps ? i0 + per : cell_txps;
    uint32_t mine = 0;
    for (uint32_t i = i0; i < i1; ++i) mine += rk[i];
    part[threadIdx.x] = mine;
    __syncthreads();
    for (uint32_t d = 1; d < kRankT; d <<= 1) { // inclusive scan (Hillis-Steele)
        const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t o = part[threadIdx.x] - mine;
    for (uint32_t i = i0; i < i1; ++i) rk[i] = rk[i] ? o++ : kNoRank;
    if (threadIdx.x == kRankT - 1) counts[blockIdx.x] = part[kRankT - 1];
}

__global__ __launch_bounds__(256) void k_cells_relabel_ranked(const uint32_t *__restrict__ row_ptr, uint32_t *__restrict__ tid,
                                                              const unsigned long long *__restrict__ cell_row_off, uint32_t n_cells,
                                                              uint32_t cell_txps, uint32_t txps_eff, uint64_t n_reads,
                                                              const uint32_t *__restrict__ rank)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t c = cell_of_read(cell_row_off, n_cells, r);
    const size_t base = (size_t)c * cell_txps;
    for (uint32_t j = row_ptr[r]; j < row_ptr[r + 1]; ++j) tid[j] = c * txps_eff + rank[base + tid[j]];
}



} // namespace

static int create_store_layout(const uint64_t *row_ptr, const uint32_t *tid, const float *as_prob,
                        const double *cov_prob, uint64_t n_reads, uint64_t nnz, uint32_t n_txps,
                        int device, const oem_store_opts *opts, oem_store *s, const CellRelabel *relabel);

// upload + layout (either builder) + the slim remote records every kernel reads (oem_layout_pack.hip)
int create_store_impl(const uint64_t *row_ptr, const uint32_t *tid, const float *as_prob,
                      const double *cov_prob, uint64_t n_reads, uint64_t nnz, uint32_t n_txps,
                      int device, const oem_store_opts *opts, oem_store *s, const CellRelabel *relabel)
{
    // weight_coding = 2 (opt-in): with the coverage model the iteration-invariant weight w = (p as f64) * cov
    // (em.rs:107-111) is rounded ONCE to f32 and the store is an f32 store -- 8 B per alignment instead of 12, the
    // kernels of the plain f32 stream instead of the f64 ones.  Every product and sum of the EM stays f64; only the
    // stored static factor carries a relative error of at most 2^-24 = 6e-8 (SURVEY.md 8a note 2; products below
    // 1.2e-38 go through f32's denormals, below 1.4e-45 to zero).  Without a coverage column it is coding 0.
    std::vector<float> w_rounded;
    oem_store_opts o2;
    if (opts && opts->weight_coding == 2) {
        o2 = *opts;
        o2.weight_coding = cov_prob ? 1u : 0u;
        opts = &o2;
        if (cov_prob) {
            w_rounded.resize(nnz);
            for (uint64_t j = 0; j < nnz; ++j) w_rounded[j] = (float)((double)as_prob[j] * cov_prob[j]);
            as_prob = w_rounded.data();
            cov_prob = nullptr;
        }
    }
    OEM_TRY(create_store_layout(row_ptr, tid, as_prob, cov_prob, n_reads, nnz, n_txps, device, opts, s, relabel));
    StageTimer tm;
    // (the test-only library keeps the builders' streams when asked to: the layout tests hash them)
    OEM_TRY(pack_remote_records(s, s->multi.txps_eff ? s->multi.txps_eff : (opts ? opts->problem_size : 0u), knob("OEM_KEEP_UNPACKED", 0) != 0));
    tm.lap("slot table + packed records");
    if (!opts || opts->weight_coding == 0) {
        OEM_TRY(build_weight_dictionary(s)); // <= 256 distinct f32 weights: one byte per local alignment
        tm.lap("weight dictionary");
    }
    return OEM_OK;
}

static int create_store_layout(const uint64_t *row_ptr, const uint32_t *tid, const float *as_prob,
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
        return OEM_OK;
    };
    // the vectors over the transcripts: after the upload, when a per-cell batch knows how many it keeps per cell
    // (a per-cell batch reads its results from its own buffer: no pinned staging vector of 300 MB for it)
    auto alloc_vectors = [&]() -> int {
        const uint32_t T = m.n_txps;
        OEM_TRY(dev_alloc(&s->theta, T, &s->hbm_bytes));
        OEM_TRY(dev_alloc(&s->cnt, T, &s->hbm_bytes));
        OEM_TRY(dev_alloc(&s->d_state, 1, &s->hbm_bytes));
        OEM_HIP(hipHostMalloc((void **)&s->h_state, sizeof(EmState), hipHostMallocDefault));
        OEM_HIP(hipHostMalloc((void **)&s->h_pinned, sizeof(double) * (T && !relabel ? T : 1), hipHostMallocDefault));
        return OEM_OK;
    };
    s->global_n_reads = n_reads;
    s->global_row_offset = 0;

    // default: lay the store out in primary-sorted tiles (oem_layout.h)
    const uint32_t reorder = opts ? opts->reorder_rows : 0;
    if (reorder == 1 || n_reads == 0) {
        OEM_TRY(upload_csr());
        OEM_TRY(alloc_vectors());
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
    if (win_cap == kWinWide) win_cap = kWinWideLds; // (the wide cap as the kernels hold it: oem_layout.h)
    // reads per tile: small stores are cut finer (oem_layout.h); per-cell batches are large by construction
    uint32_t tile_rows = (uint32_t)knob("OEM_TILE_ROWS", relabel || (opts && opts->problem_size) ? kTileRows : tile_rows_for(n_reads));
    tile_rows = tile_rows < 64u ? 64u : tile_rows > kTileRows ? kTileRows : (tile_rows & ~63u);
    s->tiled.tile_rows = tile_rows;
    // host copy of the relabelled transcript ids, only for the host builder
    bool relabelled_on_device = false;
    std::vector<uint32_t> vt;
    auto host_tids = [&]() -> const uint32_t * {
        if (!relabel) return tid;
        if (vt.empty() && nnz && relabelled_on_device) { // (compacted ids: as the device wrote them)
            vt.resize(nnz);
            if (hipMemcpy(vt.data(), m.tid, sizeof(uint32_t) * nnz, hipMemcpyDeviceToHost) != hipSuccess) vt.clear();
        }
        if (vt.empty() && nnz) {
            vt.resize(nnz);
            for (uint32_t c = 0; c < relabel->n_cells; ++c) {
                const uint64_t a0 = row_ptr[relabel->cell_row_off[c]], a1 = row_ptr[relabel->cell_row_off[c + 1]];
                for (uint64_t j = a0; j < a1; ++j) vt[j] = c * relabel->cell_txps + tid[j];
            }
        }
        return vt.data();
    };
    uint32_t problem_size = opts ? opts->problem_size : 0u; // (a compacted per-cell batch: its own, below)
    auto relabel_on_device = [&](bool compact) -> int {
        if (!relabel || nnz == 0) return OEM_OK;
        if (m.wide_ptr) return fail(OEM_ERR_ARG, "per-cell batch needs fewer than 2^32 alignments");
        unsigned long long *d_off = nullptr;
        uint32_t *d_counts = nullptr;
        OEM_HIP(hipMalloc((void **)&d_off, sizeof(unsigned long long) * ((size_t)relabel->n_cells + 1)));
        hipError_t e = hipMemcpy(d_off, relabel->cell_row_off, sizeof(unsigned long long) * ((size_t)relabel->n_cells + 1),
                                 hipMemcpyHostToDevice);
        const dim3 rgrid((uint32_t)((n_reads + 255) / 256));
        const size_t n_rank = (size_t)relabel->n_cells * relabel->cell_txps;
        if (e == hipSuccess && compact) {
            // (see k_cells_mark: a cell keeps the transcripts that occur in it, every cell as many ids as the
            // fullest one)
            e = knob("OEM_TEST_FAIL_RANK_ALLOC", 0) ? hipErrorOutOfMemory : hipMalloc((void **)&s->multi.rank, sizeof(uint32_t) * n_rank);
            if (e == hipSuccess) e = hipMalloc((void **)&d_counts, sizeof(uint32_t) * relabel->n_cells);
            if (e == hipErrorOutOfMemory) { // no room for the rank table: the batch keeps every cell's full id range
                hipFree(s->multi.rank);
                s->multi.rank = nullptr;
                hipFree(d_counts);
                d_counts = nullptr;
                (void)hipGetLastError();
                e = hipSuccess;
                compact = false;
            }
        }
        if (e == hipSuccess && compact) {
            e = hipMemsetAsync(s->multi.rank, 0, sizeof(uint32_t) * n_rank, s->stream);
            std::vector<uint32_t> counts(relabel->n_cells);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_cells_mark, rgrid, dim3(256), 0, s->stream, (const uint32_t *)m.row_ptr, m.tid, d_off,
                                   relabel->n_cells, relabel->cell_txps, n_reads, s->multi.rank);
                hipLaunchKernelGGL(k_cells_rank, dim3(relabel->n_cells), dim3(kRankT), 0, s->stream, s->multi.rank,
                                   relabel->cell_txps, d_counts);
                e = hipMemcpyAsync(counts.data(), d_counts, sizeof(uint32_t) * relabel->n_cells, hipMemcpyDeviceToHost, s->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
            }
            if (e == hipSuccess) {
                uint32_t eff = 1;
                for (uint32_t c : counts) eff = c > eff ? c : eff;
                s->multi.txps_full = relabel->cell_txps;
                s->multi.txps_eff = eff;
                problem_size = eff;
                m.n_txps = relabel->n_cells * eff;
                hipLaunchKernelGGL(k_cells_relabel_ranked, rgrid, dim3(256), 0, s->stream, (const uint32_t *)m.row_ptr, m.tid,
                                   d_off, relabel->n_cells, relabel->cell_txps, eff, n_reads, (const uint32_t *)s->multi.rank);
                e = hipStreamSynchronize(s->stream);
                relabelled_on_device = true;
            }
        } else if (e == hipSuccess) {
            hipLaunchKernelGGL(k_relabel_cells, rgrid, dim3(256), 0, s->stream,
                               (const uint32_t *)m.row_ptr, m.tid, d_off, relabel->n_cells, relabel->cell_txps, n_reads);
            e = hipStreamSynchronize(s->stream);
        }
        hipFree(d_off);
        hipFree(d_counts);
        if (e != hipSuccess) return fail(OEM_ERR_HIP, "relabelling the cells failed: %s", hipGetErrorString(e));
        return OEM_OK;
    };
    // The layout is built on the device from the resident CSR (oem_layout_device.hip); the host
    // builder (oem_layout.cpp, the specification) takes the stores that one does not, or all of them
    // with oem_store_opts.layout_build = 1.
    if (!(opts && opts->layout_build == 1)) {
        OEM_TRY(upload_csr());
        tm.lap("caller-order CSR upload");
        OEM_TRY(relabel_on_device(knob("OEM_CELLS_COMPACT_TXPS", 1) != 0));
        OEM_TRY(alloc_vectors());
        if (relabel) tm.lap("cells: transcripts relabelled");
        bool built = false;
        OEM_TRY(build_tiled_layout_device(s, problem_size, win_cap, tile_rows, &built));
        tm.lap("tiled layout build (device)");
        if (built) return OEM_OK;
        TiledHost h;
        const char *err = nullptr;
        if (build_tiled_layout(row_ptr, host_tids(), as_prob, cov_prob, n_reads, nnz, m.n_txps, &h, &err,
                               problem_size, win_cap, tile_rows)) {
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
        if (csr_rc == OEM_OK) csr_rc = relabel_on_device(false);
        if (csr_rc == OEM_OK) csr_rc = alloc_vectors();
        if (csr_rc != OEM_OK) snprintf(csr_err, sizeof(csr_err), "%s", t_err); // t_err is thread-local
    });
    TiledHost h;
    const char *err = nullptr;
    const bool tiled = build_tiled_layout(row_ptr, host_tids(), as_prob, cov_prob, n_reads, nnz, n_txps, &h, &err,
                                          opts ? opts->problem_size : 0u, win_cap, tile_rows);
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
    if (opts && opts->weight_coding > 2) return fail(OEM_ERR_ARG, "oem_store_create: weight_coding %u (0, 1 or 2)", opts->weight_coding);
    if (opts && opts->layout_build > 1) return fail(OEM_ERR_ARG, "oem_store_create: layout_build %u (0 or 1)", opts->layout_build);
    if (opts && opts->reorder_rows > 2) return fail(OEM_ERR_ARG, "oem_store_create: reorder_rows %u (0, 1 or 2)", opts->reorder_rows);
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
