// oem_multi_kernels.hip -- per-cell EM batched on the device.
//
// single_cell.rs:139-160: every cell is an independent em::em(&emi, 1) with
// init_abundances None over its own reads.  Here all cells share every pass over one
// resident store whose transcript space is the concatenation of the cells' (cell p owns
// transcripts [p*T, (p+1)*T)); the tile / fold kernels are the ordinary ones, and each
// cell walks the reference's loop on the device with its own state:
//   RUNNING -(em.rs:212 stopping rule / em.rs:181 max_iter)-> FINAL (em.rs:238-242 zero
//   small, em.rs:245-252 one more pass) -> FINISHED (counts parked in `out`).
#include "oem_internal.h"
#include "oem_lane_runs.h"

namespace oem {

namespace {

constexpr int kMT = 256;

// theta[p*T + i] = reads(p) / T   (em.rs:165 with the cell's own store.len())
// (T: transcripts per cell in the store; T_full: the caller's n_txps -- the reference's store.len() / n_txps, whatever
// the store keeps of the cell's transcripts)
__global__ __launch_bounds__(kMT) void k_multi_init(double *__restrict__ theta,
                                                    const uint64_t *__restrict__ problem_reads, uint32_t T, uint32_t T_full)
{
    const uint32_t p = blockIdx.y;
    const double avg = (double)problem_reads[p] / (double)T_full;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < T; i += gridDim.x * blockDim.x)
        theta[(size_t)p * T + i] = avg;
}

__global__ __launch_bounds__(kMT) void k_multi_expand(const double *__restrict__ out_eff, const uint32_t *__restrict__ rank,
                                                      uint32_t T_full, uint32_t T_eff, size_t n, double *__restrict__ full)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t r = rank[i];
        full[i] = r == kNoRank ? 0.0 : out_eff[(i / T_full) * T_eff + r];
    }
}

// rel-diff / swap / clear of one cell per blockIdx.y (em.rs:194-207); FINAL cells park their counts
__global__ __launch_bounds__(kMT) void k_multi_reldiff(double *__restrict__ theta, double *__restrict__ cnt,
                                                       double *__restrict__ out, BatchState *st, uint32_t T)
{
    const uint32_t p = blockIdx.y;
    const uint32_t phase = st[p].phase;
    if (phase == kPhaseFinished) return;
    double rel = 0.0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < T; i += gridDim.x * blockDim.x) {
        const size_t k = (size_t)p * T + i;
        const double cc = cnt[k];
        cnt[k] = 0.0;
        if (phase == kPhaseFinal) {
            out[k] = cc;                                      // em.rs:254
        } else {
            const double pc = theta[k];
            if (pc > OEM_MIN_READ_THRESH) rel = fmax(rel, (cc - pc) / pc); // em.rs:195-199
            theta[k] = cc;                                    // em.rs:204
        }
    }
    if (phase == kPhaseFinal) return;
    for (int off = 32; off > 0; off >>= 1) rel = fmax(rel, __shfl_xor(rel, off, 64));
    if ((threadIdx.x & 63) == 0 && rel > 0.0)
        atomicMax(&st[p].rel_bits, (unsigned long long)__double_as_longlong(rel));
}


// The fold kernel of a per-cell batch finishes the pass itself.  A cell's remote alignments spread over a
// handful of buckets with few entries each, so every bucket is owned by ONE workgroup: once the bucket's
// queue range is summed in LDS, counts = (what the tile kernels flushed into cnt) + (the LDS sums) are
// complete for these transcripts, and rel-diff / swap / clear (em.rs:194-207) or the parking of a FINAL
// cell's counts (em.rs:254) happen right here -- no flush atomics, no second sweep over theta and cnt.
constexpr int kFT = 1024;
constexpr uint32_t kMaxCellsPerBucket = 64; // beyond that (T < ~130) the per-cell maxima go straight to global atomics

__global__ __launch_bounds__(kFT) void k_multi_fold_reldiff(const uint32_t *__restrict__ bucket_base,
                                                            const double *__restrict__ queue,
                                                            const uint16_t *__restrict__ q_dst, double *__restrict__ theta,
                                                            double *__restrict__ cnt, double *__restrict__ out,
                                                            BatchState *st, uint32_t n_txps, uint32_t T,
                                                            const uint32_t *__restrict__ live_buckets)
{
    __shared__ double acc[kBucket];
    __shared__ unsigned long long cmax[kMaxCellsPerBucket];
    __shared__ uint32_t phase_l[kMaxCellsPerBucket];
    const uint32_t b = live_buckets ? live_buckets[blockIdx.x] : blockIdx.x;
    const uint32_t t0 = b * kBucket;
    uint32_t t1 = t0 + kBucket - 1;
    if (t1 >= n_txps) t1 = n_txps - 1;
    const uint32_t p0 = t0 / T, p1 = t1 / T, n_cells = p1 - p0 + 1;
    const bool small = n_cells <= kMaxCellsPerBucket;
    bool live = false;
    for (uint32_t p = p0; p <= p1; ++p) live = live || st[p].phase != kPhaseFinished;
    if (!live) return;
    if (small && threadIdx.x < n_cells) {
        cmax[threadIdx.x] = 0ull;
        phase_l[threadIdx.x] = st[p0 + threadIdx.x].phase;
    }
    for (uint32_t i = threadIdx.x; i < kBucket; i += kFT) acc[i] = 0.0;
    __syncthreads();
    const uint32_t q0 = bucket_base[b], q1 = bucket_base[b + 1];
    // four loads in flight per thread (a cell's bucket has a few entries per thread: one round trip instead of one
    // per entry); entries past the range are clamped to its last one and skipped at the point of use
    for (uint32_t o = q0 + threadIdx.x; o < q1; o += 4 * kFT) {
        double v[4];
        uint32_t d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t oo = o + k * kFT, oc = oo < q1 ? oo : q1 - 1;
            v[k] = __builtin_nontemporal_load(&queue[oc]); // (read once: the abundances and counts keep the caches)
            d[k] = __builtin_nontemporal_load(&q_dst[oc]);
        }
        const bool rep = keys_repeat(d[0]); // hot destinations: runs of equal ones summed on the vector ALU first (oem_lane_runs.h)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double vk[1] = {o + k * kFT < q1 ? v[k] : 0.0};
            if (rep) sum_runs_of_equal_keys<1, 1>(d[k], vk);
            if (vk[0] != 0.0) __hip_atomic_fetch_add(&acc[d[k]], vk[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    // the counts and abundances of this thread's window entries: requested together, then swept
    constexpr uint32_t kPerThread = kBucket / kFT;
    static_assert(kBucket % kFT == 0, "a thread sweeps kBucket / kFT window entries");
    double cv[kPerThread], pv[kPerThread];
#pragma unroll
    for (uint32_t k = 0; k < kPerThread; ++k) {
        const uint32_t i = threadIdx.x + k * kFT, t = t0 + i < n_txps ? t0 + i : n_txps - 1;
        cv[k] = cnt[t];
        pv[k] = theta[t];
    }
#pragma unroll
    for (uint32_t k = 0; k < kPerThread; ++k) {
        const uint32_t i = threadIdx.x + k * kFT;
        if (t0 + i >= n_txps) break;
        const uint32_t t = t0 + i, p = t / T;
        const uint32_t phase = small ? phase_l[p - p0] : st[p].phase;
        if (phase == kPhaseFinished) continue;
        const double cc = cv[k] + acc[i];
        cnt[t] = 0.0;                                             // em.rs:207
        if (phase == kPhaseFinal) {
            out[t] = cc;                                          // em.rs:254
        } else {
            const double pc = pv[k];
            theta[t] = cc;                                        // em.rs:204
            if (pc > OEM_MIN_READ_THRESH) {                       // em.rs:195-199 (signed, floored at 0 by the max)
                const double rel = (cc - pc) / pc;
                if (rel > 0.0) {
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(rel);
                    if (small) atomicMax(&cmax[p - p0], bits);
                    else atomicMax(&st[p].rel_bits, bits);
                }
            }
        }
    }
    if (!small) return;
    __syncthreads();
    if (threadIdx.x < n_cells && cmax[threadIdx.x] != 0ull) atomicMax(&st[p0 + threadIdx.x].rel_bits, cmax[threadIdx.x]);
}

// one thread per cell: the stopping rule (em.rs:212-218, :181)
__global__ __launch_bounds__(kMT) void k_multi_decide(BatchState *st, uint32_t n_problems, EmParams p,
                                                      uint32_t *n_unfinished)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_problems) return;
    BatchState s = st[i];
    if (s.phase == kPhaseFinished) return;
    if (s.phase == kPhaseFinal) {
        s.n_passes += 1;
        s.phase = kPhaseFinished;
        atomicSub(n_unfinished, 1u);
    } else {
        const double rel_diff = __longlong_as_double((long long)s.rel_bits);
        s.last_rel = rel_diff;
        s.n_passes += 1;
        if (rel_diff < p.conv_thresh && s.niter > p.min_iter_gate) {
            s.converged = 1;
            s.phase = kPhaseFinal;
        } else {
            s.niter += 1;
            if (s.niter >= p.max_iter) s.phase = kPhaseFinal;
        }
        s.rel_bits = 0ull;
    }
    st[i] = s;
}

// (em.rs:238-242, the zeroing of small abundances before the final pass, is not a sweep here: k_em_tile reads
// the abundances of a FINAL cell below the threshold as 0 on the way in.)

// Live-work compaction.  Cells are independent (single_cell.rs:139-160) and stop at their own iteration: half way
// through the loop most of a batch's tiles belong to finished cells, and launching them only to have their
// workgroups return costs a fixed ~0.1 ms per pass however few cells are live.  When the host's look at the device
// state (every 16 passes) finds fewer unfinished cells than at the last compaction, ONE workgroup rewrites the list
// of live items in their original order (the XCD-aware tile order of k_em_tile works on list positions): item i is
// live when one of the cells [first(i), last(i)] is not FINISHED.  kTiles: items are tiles (one cell each);
// otherwise remote buckets (the cells whose transcripts the bucket covers).
constexpr int kCT = 1024;
template <bool kTiles>
__global__ __launch_bounds__(kCT) void k_multi_compact(const TileDesc *__restrict__ tiles, uint32_t n_items,
                                                       const BatchState *__restrict__ st, uint32_t n_txps, uint32_t T,
                                                       uint32_t *__restrict__ live, uint32_t *__restrict__ n_live)
{
    __shared__ uint32_t part[kCT];
    const uint32_t per = (n_items + kCT - 1) / kCT;
    const uint32_t i0 = threadIdx.x * per, i1 = i0 + per < n_items ? i0 + per : n_items;
    auto is_live = [&](uint32_t i) -> bool {
        if (kTiles) return st[tiles[i].problem].phase != kPhaseFinished;
        const uint32_t t0 = i * kBucket;
        uint32_t t1 = t0 + kBucket - 1;
        if (t1 >= n_txps) t1 = n_txps - 1;
        for (uint32_t p = t0 / T; p <= t1 / T; ++p)
            if (st[p].phase != kPhaseFinished) return true;
        return false;
    };
    uint32_t mine = 0;
    for (uint32_t i = i0; i < i1; ++i) mine += is_live(i) ? 1u : 0u;
    part[threadIdx.x] = mine;
    __syncthreads();
    for (uint32_t d = 1; d < kCT; d <<= 1) { // inclusive scan (Hillis-Steele: 10 steps of one workgroup)
        const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t o = part[threadIdx.x] - mine;
    for (uint32_t i = i0; i < i1; ++i)
        if (is_live(i)) live[o++] = i;
    if (threadIdx.x == kCT - 1) *n_live = part[kCT - 1];
}

} // namespace

int launch_multi_init(oem_store *s, double *theta, const uint64_t *d_problem_reads, const MultiBuffers &mb)
{
    const uint32_t T = mb.problem_size;
    uint32_t gx = (T + kMT - 1) / kMT;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_multi_init, dim3(gx, mb.n_problems), dim3(kMT), 0, s->stream, theta, d_problem_reads, T,
                       mb.txps_full ? mb.txps_full : T);
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

// results of a compacted batch -> the caller's [cell][transcript]: a transcript that does not occur in the cell is 0
int launch_multi_expand(oem_store *s, const MultiBuffers &mb, double *full)
{
    const size_t n = (size_t)mb.n_problems * mb.txps_full;
    if (n == 0) return OEM_OK;
    size_t g = (n + kMT - 1) / kMT;
    if (g > 65535u * 16u) g = 65535u * 16u;
    hipLaunchKernelGGL(k_multi_expand, dim3((uint32_t)g), dim3(kMT), 0, s->stream, (const double *)mb.out, (const uint32_t *)mb.rank,
                       mb.txps_full, mb.txps_eff, n, full);
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

// fold + rel-diff in one kernel (see k_multi_fold_reldiff), then the per-cell state machine
int launch_multi_fold_reldiff(oem_store *s, double *theta, double *cnt, const MultiBuffers &mb, EmParams p)
{
    const DeviceTiled &t = s->tiled;
    const uint32_t T = mb.problem_size;
    const uint32_t gp = (mb.n_problems + kMT - 1) / kMT;
    const uint32_t *live = mb.live_valid ? mb.live_buckets : nullptr;
    const uint32_t n_b = live ? mb.n_live_buckets : t.n_buckets;
    if (n_b)
        hipLaunchKernelGGL(k_multi_fold_reldiff, dim3(n_b), dim3(kFT), 0, s->stream, t.bucket_base, t.queue, t.q_dst,
                           theta, cnt, mb.out, mb.state, s->csr.n_txps, T, live);
    hipLaunchKernelGGL(k_multi_decide, dim3(gp), dim3(kMT), 0, s->stream, mb.state, mb.n_problems, p, mb.n_unfinished);
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

int multi_compact_live(oem_store *s, MultiBuffers &mb)
{
    const DeviceTiled &t = s->tiled;
    if (!mb.live_tiles) {
        OEM_HIP(hipMalloc((void **)&mb.live_tiles, sizeof(uint32_t) * (t.n_tiles ? t.n_tiles : 1)));
        OEM_HIP(hipMalloc((void **)&mb.live_buckets, sizeof(uint32_t) * (t.n_buckets ? t.n_buckets : 1)));
        OEM_HIP(hipMalloc((void **)&mb.d_live_counts, sizeof(uint32_t) * 2));
    }
    hipLaunchKernelGGL((k_multi_compact<true>), dim3(1), dim3(kCT), 0, s->stream, t.tiles, t.n_tiles, mb.state,
                       s->csr.n_txps, mb.problem_size, mb.live_tiles, mb.d_live_counts);
    hipLaunchKernelGGL((k_multi_compact<false>), dim3(1), dim3(kCT), 0, s->stream, t.tiles, t.n_buckets, mb.state,
                       s->csr.n_txps, mb.problem_size, mb.live_buckets, mb.d_live_counts + 1);
    OEM_HIP(hipGetLastError());
    uint32_t h[2] = {0, 0};
    OEM_HIP(hipMemcpyAsync(h, mb.d_live_counts, sizeof(h), hipMemcpyDeviceToHost, s->stream));
    OEM_HIP(hipStreamSynchronize(s->stream));
    mb.n_live_tiles = h[0];
    mb.n_live_buckets = h[1];
    mb.live_valid = true;
    return OEM_OK;
}

int launch_multi_reldiff(oem_store *s, double *theta, double *cnt, const MultiBuffers &mb, EmParams p)
{
    const uint32_t T = mb.problem_size;
    uint32_t gx = (T + kMT - 1) / kMT;
    if (gx > 64) gx = 64;
    const uint32_t gp = (mb.n_problems + kMT - 1) / kMT;
    hipLaunchKernelGGL(k_multi_reldiff, dim3(gx, mb.n_problems), dim3(kMT), 0, s->stream, theta, cnt, mb.out,
                       mb.state, T);
    hipLaunchKernelGGL(k_multi_decide, dim3(gp), dim3(kMT), 0, s->stream, mb.state, mb.n_problems, p,
                       mb.n_unfinished);
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

} // namespace oem
