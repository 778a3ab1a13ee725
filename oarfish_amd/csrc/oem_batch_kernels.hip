// oem_batch_kernels.hip -- kBatch bootstrap replicates per pass over the resident matrix.
//
// em::bootstrap (em.rs:292-314) runs num_boot independent resampled EMs, each a serial
// do_em over random_sampling_iter (em.rs:273-290).  Every replicate streams the whole
// store again.  Here kBatch replicates share one pass over the tiled matrix: the
// weights w and window codes are read once and folded against kBatch abundance
// vectors (theta laid out [transcript][replicate], so one 32-byte access serves all
// replicates of a transcript), each read scaled by its per-replicate multiplicity
// c_ib ~ Multinomial(R; 1/R) (bootstrap.rs:7-16).
//
// Every replicate keeps its own loop state on the device and walks the reference's
// state machine by itself: RUNNING -(stopping rule em.rs:212 / max_iter em.rs:181)->
// FINAL (theta < 1e-5 read as 0, em.rs:238-242; one more pass, em.rs:245-252) ->
// FINISHED (counts parked in `out`, replicate ignored from then on).
#include <cstdlib>

#include "oem_internal.h"

namespace oem {

namespace {

constexpr int kB = kBatch;
constexpr int kBCh = 8;     // alignments per read kept in registers
constexpr int kFoldThreadsB = 1024;

__device__ __forceinline__ void lds_add(double *p, double v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct SliceRegsB {
    float w[kBCh];
    uint32_t c[kBCh / 2];
};

template <bool kNT, typename T>
__device__ __forceinline__ T ld_stream_b(const T *p)
{
    return kNT ? __builtin_nontemporal_load(p) : *p; // see ld_stream in oem_tile_kernels.hip
}

template <bool kNT>
__device__ __forceinline__ void load_slice_b(SliceRegsB &r, const float *__restrict__ wbase,
                                             const uint32_t *__restrict__ cbase, uint32_t lane,
                                             uint32_t width)
{
#pragma unroll
    for (int g = 0; g < kBCh / 2; ++g) {
        if ((uint32_t)(2 * g) < width) {
            r.w[2 * g] = ld_stream_b<kNT>(&wbase[(2 * g) * 64 + lane]);
            r.w[2 * g + 1] = ld_stream_b<kNT>(&wbase[(2 * g + 1) * 64 + lane]);
            r.c[g] = ld_stream_b<kNT>(&cbase[g * 64 + lane]);
        } else {
            r.w[2 * g] = 0.f;
            r.w[2 * g + 1] = 0.f;
            r.c[g] = 0u;
        }
    }
}

// Same structure as k_em_tile (oem_tile_kernels.hip): one workgroup per tile, one read per
// lane, slices prefetched one ahead, operands landed with one counted wait, count window in
// kCopies interleaved copies -- with every LDS / queue / theta entity carrying kB replicates.
// Window entry (c, b, copy p) lives at ((c * kB + b) * kCopies + p).
template <int kThreads, int kRem, int kCopies, int kMinWaves, bool kNT>
__global__ __launch_bounds__(kThreads, kMinWaves) void k_em_tile_b(
    const TileDesc *__restrict__ tiles, const uint32_t *__restrict__ codes,
    const float *__restrict__ w, const uint32_t *__restrict__ r_tid, const float *__restrict__ r_w,
    const uint16_t *__restrict__ r_row, const uint32_t *__restrict__ r_slot,
    double *__restrict__ queue /* [kB][n_remote] */, uint64_t n_remote,
    const double *__restrict__ theta /* [T][kB] */, double *__restrict__ cnt /* [T][kB] */,
    const BatchState *__restrict__ st, const uint8_t *__restrict__ row_w /* [rows][kB], tile order */)
{
    uint32_t act = 0; // replicates that still take part in this pass (RUNNING or FINAL)
    uint32_t fin = 0; // replicates on their final pass: theta < 1e-5 reads as 0 (em.rs:238-242)
#pragma unroll
    for (int b = 0; b < kB; ++b) {
        const uint32_t ph = st[b].phase;
        act |= (ph != kPhaseFinished) ? (1u << b) : 0u;
        fin |= (ph == kPhaseFinal) ? (1u << b) : 0u;
    }
    if (!act) return;
    auto th = [&](double v, int b) -> double {
        return (((fin >> b) & 1u) && v < OEM_MIN_READ_THRESH) ? 0.0 : v;
    };

    __shared__ double theta_l[kWin * kB];
    __shared__ double cnt_l[kWin * kB * kCopies];
    __shared__ double den_l[kTileRows * kB];

    constexpr uint32_t kWaves = kThreads / 64;
    constexpr uint32_t kPerWave = kTileSlices / kWaves;
    const TileDesc td = tiles[blockIdx.x];
    const uint32_t tx = threadIdx.x;
    const uint32_t lane = tx & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tx >> 6);
    const uint32_t copy = lane % kCopies;

    uint32_t woff[kPerWave], coff[kPerWave], wid[kPerWave];
    {
        uint32_t accw = td.w_base, accc = td.c_base;
#pragma unroll
        for (uint32_t i = 0; i < kTileSlices; ++i) {
            const uint32_t wi = td.width[i];
            if ((i % kWaves) == wave) {
                woff[i / kWaves] = accw;
                coff[i / kWaves] = accc;
                wid[i / kWaves] = wi;
            }
            accw += wi;
            accc += (wi + 1) >> 1;
        }
    }

    // remote alignments of this thread: x[b] = theta[t][b] * w (one 16-byte access serves both replicates)
    double rx[kRem][kB];
    uint32_t rrow[kRem], rslot[kRem];
    {
        uint32_t rt[kRem];
        float rw[kRem];
#pragma unroll
        for (int k = 0; k < kRem; ++k) {
            const uint32_t i = tx + k * kThreads;
            rt[k] = 0; rw[k] = 0.f; rrow[k] = 0; rslot[k] = 0;
            if (i < td.remote_cnt) {
                const uint32_t o = td.remote_begin + i;
                rt[k] = ld_stream_b<kNT>(&r_tid[o]);
                rw[k] = ld_stream_b<kNT>(&r_w[o]);
                rrow[k] = ld_stream_b<kNT>(&r_row[o]);
                rslot[k] = ld_stream_b<kNT>(&r_slot[o]);
            }
        }
#pragma unroll
        for (int k = 0; k < kRem; ++k) {
#pragma unroll
            for (int b = 0; b < kB; ++b) rx[k][b] = th(theta[(size_t)rt[k] * kB + b], b) * (double)rw[k];
        }
    }
    constexpr uint32_t kSets = kPerWave > 1 ? 2 : 1;
    SliceRegsB R[kSets];
    load_slice_b<kNT>(R[0], w + (size_t)woff[0] * 64, codes + (size_t)coff[0] * 64, lane, wid[0]);

    for (uint32_t i = tx; i < td.win_len * kB; i += kThreads) theta_l[i] = th(theta[(size_t)td.lo * kB + i], i % kB);
    for (uint32_t i = tx; i < td.win_len * kB * kCopies; i += kThreads) cnt_l[i] = 0.0;
    for (uint32_t i = tx; i < td.n_slices * 64 * kB; i += kThreads) den_l[i] = 0.0;
    __syncthreads();

    // remote phase A: denominators
#pragma unroll
    for (int k = 0; k < kRem; ++k)
        if (tx + k * kThreads < td.remote_cnt) {
#pragma unroll
            for (int b = 0; b < kB; ++b) lds_add(&den_l[rrow[k] * kB + b], rx[k][b]);
        }
    for (uint32_t i = tx + kRem * kThreads; i < td.remote_cnt; i += kThreads) { // overflow: park in the queue
        const uint32_t o = td.remote_begin + i;
        const uint32_t t = r_tid[o];
        const double wv = (double)r_w[o];
#pragma unroll
        for (int b = 0; b < kB; ++b) {
            const double x = th(theta[(size_t)t * kB + b], b) * wv;
            queue[(size_t)b * n_remote + r_slot[o]] = x;
            lds_add(&den_l[r_row[o] * kB + b], x);
        }
    }
    __syncthreads();

    // local alignments
#pragma unroll
    for (uint32_t q = 0; q < kPerWave; ++q) {
        const uint32_t s = wave + kWaves * q;
        if (q + 1 < kPerWave)
            load_slice_b<kNT>(R[(q + 1) % kSets], w + (size_t)woff[q + 1] * 64, codes + (size_t)coff[q + 1] * 64, lane,
                         wid[q + 1]);
        if (s >= td.n_slices) continue;
        const SliceRegsB &cur = R[q % kSets];
        // land this slice's operands with one counted wait (the next slice's loads stay in flight)
#pragma unroll
        for (int k = 0; k < kBCh; ++k) asm volatile("" ::"v"(cur.w[k]));
#pragma unroll
        for (int k = 0; k < kBCh / 2; ++k) asm volatile("" ::"v"(cur.c[k]));
        const uint32_t width = wid[q];
        const uint32_t rl = s * 64 + lane;
        const float *wbase = w + (size_t)woff[q] * 64;
        const uint32_t *cbase = codes + (size_t)coff[q] * 64;
        uint32_t mult = 0; // packed multiplicities of this read, one byte per replicate
        if (rl < td.n_rows) {
            const uint8_t *mp = row_w + (size_t)(td.row_base + rl) * kB;
#pragma unroll
            for (int b = 0; b < kB; ++b) mult |= (uint32_t)mp[b] << (8 * b);
        }
        double inv[kB];
#pragma unroll
        for (int b = 0; b < kB; ++b) {
            double denom = den_l[rl * kB + b];
#pragma unroll
            for (int k = 0; k < kBCh; ++k) {
                const uint32_t c = ((k & 1) ? (cur.c[k >> 1] >> 16) : (cur.c[k >> 1] & 0xffffu)) >> 3;
                const double wk = (uint32_t)k < width ? (double)cur.w[k] : 0.0;
                denom += theta_l[c * kB + b] * wk;                               // em.rs:111
            }
            for (uint32_t j = kBCh; j < width; ++j) {
                const uint32_t cc = cbase[(j >> 1) * 64 + lane];
                const uint32_t c = ((j & 1) ? (cc >> 16) : (cc & 0xffffu)) >> 3;
                denom += theta_l[c * kB + b] * (double)wbase[j * 64 + lane];
            }
            const double scale = (double)((mult >> (8 * b)) & 0xffu);
            inv[b] = ((act >> b) & 1u) && denom > OEM_EM_DENOM_THRESH ? scale / denom : 0.0; // em.rs:115
            den_l[rl * kB + b] = inv[b];
        }
#pragma unroll
        for (int k = 0; k < kBCh; ++k) {
            if ((uint32_t)k < width) {
                const uint32_t c = ((k & 1) ? (cur.c[k >> 1] >> 16) : (cur.c[k >> 1] & 0xffffu)) >> 3;
                const double wk = (double)cur.w[k];
#pragma unroll
                for (int b = 0; b < kB; ++b) {
                    const double v = theta_l[c * kB + b] * wk * inv[b];
                    if (v != 0.0) lds_add(&cnt_l[(c * kB + b) * kCopies + copy], v); // em.rs:128-129
                }
            }
        }
        for (uint32_t j = kBCh; j < width; ++j) {
            const uint32_t cc = cbase[(j >> 1) * 64 + lane];
            const uint32_t c = ((j & 1) ? (cc >> 16) : (cc & 0xffffu)) >> 3;
            const double wk = (double)wbase[j * 64 + lane];
#pragma unroll
            for (int b = 0; b < kB; ++b) {
                const double v = theta_l[c * kB + b] * wk * inv[b];
                if (v != 0.0) lds_add(&cnt_l[(c * kB + b) * kCopies + copy], v);
            }
        }
    }
    __syncthreads();

    // remote phase B: queue[b][slot] <- x_b * (c_ib / denom_ib)
#pragma unroll
    for (int k = 0; k < kRem; ++k) {
        const uint32_t i = tx + k * kThreads;
        if (i < td.remote_cnt) {
#pragma unroll
            for (int b = 0; b < kB; ++b)
                queue[(size_t)b * n_remote + rslot[k]] = rx[k][b] * den_l[rrow[k] * kB + b];
        }
    }
    for (uint32_t i = tx + kRem * kThreads; i < td.remote_cnt; i += kThreads) {
        const uint32_t o = td.remote_begin + i;
#pragma unroll
        for (int b = 0; b < kB; ++b) {
            const size_t qi = (size_t)b * n_remote + r_slot[o];
            queue[qi] = queue[qi] * den_l[r_row[o] * kB + b];
        }
    }
    // flush: the window is contiguous in [T][kB]
    for (uint32_t i = tx; i < td.win_len * kB; i += kThreads) {
        double v = 0.0;
#pragma unroll
        for (int p = 0; p < kCopies; ++p) v += cnt_l[i * kCopies + p];
        if (v != 0.0) unsafeAtomicAdd(&cnt[(size_t)td.lo * kB + i], v);
    }
}

// one workgroup per (replicate, bucket, group): streams queue[b][range] into an LDS window and
// flushes it into cnt2[b][.] (replicate-major, so the flush is line-coalesced)
__global__ __launch_bounds__(kFoldThreadsB) void k_remote_fold_b(
    const uint32_t *__restrict__ bucket_base, const double *__restrict__ queue, uint64_t n_remote,
    const uint16_t *__restrict__ q_dst, double *__restrict__ cnt2 /* [kB][T] */,
    const BatchState *__restrict__ st, uint32_t n_groups, uint32_t n_buckets, uint32_t n_txps)
{
    const uint32_t b = blockIdx.x / (n_groups * n_buckets);
    if (st[b].phase == kPhaseFinished) return;
    const uint32_t rest = blockIdx.x % (n_groups * n_buckets);
    const uint32_t bk = rest / n_groups, g = rest % n_groups;
    __shared__ double acc[kBucket];
    const uint32_t q0 = bucket_base[bk], q1 = bucket_base[bk + 1];
    const uint64_t span = q1 - q0;
    const uint32_t s0 = q0 + (uint32_t)(span * g / n_groups);
    const uint32_t s1 = q0 + (uint32_t)(span * (g + 1) / n_groups);
    if (s0 == s1) return;
    for (uint32_t i = threadIdx.x; i < kBucket; i += kFoldThreadsB) acc[i] = 0.0;
    __syncthreads();
    const double *qb = queue + (size_t)b * n_remote;
    uint32_t o = s0 + threadIdx.x;
    for (; o + 3 * kFoldThreadsB < s1; o += 4 * kFoldThreadsB) {
        double v[4];
        uint32_t d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = qb[o + k * kFoldThreadsB];
            d[k] = q_dst[o + k * kFoldThreadsB];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (v[k] != 0.0) lds_add(&acc[d[k]], v[k]);
    }
    for (; o < s1; o += kFoldThreadsB) {
        const double v = qb[o];
        if (v != 0.0) lds_add(&acc[q_dst[o]], v);
    }
    __syncthreads();
    const uint32_t base = bk * kBucket;
    for (uint32_t i = threadIdx.x; i < kBucket && base + i < n_txps; i += kFoldThreadsB) {
        const double v = acc[i];
        if (v != 0.0) unsafeAtomicAdd(&cnt2[(size_t)b * n_txps + base + i], v);
    }
}

// rel-diff / swap / clear / state machine for kB replicates (em.rs:194-218, :238-254)
//   curr_b[t] = cnt[t][b] + cnt2[b][t]
constexpr int kRelB = 1024; // as k_reldiff_swap_clear: few fat workgroups, the state-line atomics serialise
__global__ __launch_bounds__(kRelB) void k_reldiff_b(double *__restrict__ theta, double *__restrict__ cnt,
                                                   double *__restrict__ cnt2, double *__restrict__ out,
                                                   BatchState *st, EmParams p)
{
    uint32_t phase[kB];
    uint32_t any = 0;
#pragma unroll
    for (int b = 0; b < kB; ++b) {
        phase[b] = st[b].phase;
        any |= phase[b] != kPhaseFinished;
    }
    if (!any) return;
    double rel[kB];
#pragma unroll
    for (int b = 0; b < kB; ++b) rel[b] = 0.0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < p.n_txps; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int b = 0; b < kB; ++b) {
            if (phase[b] == kPhaseFinished) continue;
            const size_t ia = (size_t)i * kB + b, ib = (size_t)b * p.n_txps + i;
            const double cc = cnt[ia] + cnt2[ib];
            cnt[ia] = 0.0;
            cnt2[ib] = 0.0;
            if (phase[b] == kPhaseFinal) {
                out[ib] = cc;                               // em.rs:254
            } else {
                const double pc = theta[ia];
                if (pc > OEM_MIN_READ_THRESH) rel[b] = fmax(rel[b], (cc - pc) / pc); // em.rs:195-199
                theta[ia] = cc;                             // em.rs:204 (zeroing of small values: below)
            }
        }
    }
    __shared__ double smax[kRelB / 64][kB];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int b = 0; b < kB; ++b) {
        double r = rel[b];
        for (int off = 32; off > 0; off >>= 1) r = fmax(r, __shfl_xor(r, off, 64));
        if (lane == 0) smax[wv][b] = r;
    }
    __syncthreads();
    __shared__ bool is_last;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int b = 0; b < kB; ++b) {
            double m = smax[0][b];
            for (int i = 1; i < kRelB / 64; ++i) m = fmax(m, smax[i][b]);
            if (m > 0.0) atomicMax(&st[b].rel_bits, (unsigned long long)__double_as_longlong(m));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t ticket = atomicAdd(&st[0].blocks_arrived, 1u);
        is_last = (ticket == gridDim.x - 1);
    }
    __syncthreads();
    if (is_last && threadIdx.x == 0) {
#pragma unroll
        for (int b = 0; b < kB; ++b) {
            if (phase[b] == kPhaseFinished) continue;
            if (phase[b] == kPhaseFinal) {
                st[b].n_passes += 1;
                st[b].phase = kPhaseFinished;
                continue;
            }
            const unsigned long long bits =
                __hip_atomic_load(&st[b].rel_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double rel_diff = __longlong_as_double((long long)bits);
            st[b].last_rel = rel_diff;
            st[b].n_passes += 1;
            uint32_t niter = st[b].niter;
            if (rel_diff < p.conv_thresh && niter > p.min_iter_gate) { // em.rs:212
                st[b].converged = 1;
                st[b].phase = kPhaseFinal;
            } else {
                niter += 1;                                            // em.rs:218
                st[b].niter = niter;
                if (niter >= p.max_iter) st[b].phase = kPhaseFinal;    // em.rs:181
            }
            st[b].rel_bits = 0ull;
        }
        st[0].blocks_arrived = 0u;
    }
}

// (re)start of ONE slot of the rolling batch: theta[t][slot] = init / avg, its counts cleared
__global__ __launch_bounds__(256) void k_reset_slot_b(double *__restrict__ theta, double *__restrict__ cnt,
                                                      double *__restrict__ cnt2, const double *__restrict__ init,
                                                      double avg, uint32_t n_txps, uint32_t slot)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_txps; i += gridDim.x * blockDim.x) {
        theta[(size_t)i * kB + slot] = init ? init[i] : avg;
        cnt[(size_t)i * kB + slot] = 0.0;
        cnt2[(size_t)slot * n_txps + i] = 0.0;
    }
}

// multiplicities of kB replicates, caller order u32 [kB][R] -> tile order u8 [rows][kB];
// *overflow is set if a multiplicity does not fit a byte (the caller then falls back to the
// one-replicate-per-pass path)
__global__ __launch_bounds__(256) void k_pack_row_w_b(const uint32_t *__restrict__ row_w, uint64_t n_reads,
                                                      const uint32_t *__restrict__ perm, uint64_t n_rows,
                                                      uint8_t *__restrict__ out, uint32_t *overflow)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = perm[i];
#pragma unroll
        for (int b = 0; b < kB; ++b) {
            const uint32_t c = row_w[(uint64_t)b * n_reads + r];
            if (c > 255u) *overflow = 1u;
            out[i * kB + b] = (uint8_t)(c & 0xffu);
        }
    }
}

inline int grid_for(uint64_t n, int block, int max_blocks)
{
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (uint64_t)max_blocks) g = max_blocks;
    return (int)g;
}

} // namespace

int launch_batch_pass(oem_store *s, const BatchBuffers &bb)
{
    const DeviceTiled &t = s->tiled;
    if (t.n_tiles == 0) return OEM_OK;
    const int variant = (int)knob("OEM_BATCH_VARIANT", 0); // testing build only
    const uint64_t stream_bytes = (t.n_local + t.n_local / 8) * 6 + t.n_remote * 14;
    const bool nt = stream_bytes > (192ull << 20); // beyond the Infinity Cache: stream non-temporally
#define OEM_TILE_B_NT(TH, REM, NC, MW, NT)                                                                \
    hipLaunchKernelGGL((k_em_tile_b<TH, REM, NC, MW, NT>), dim3(t.n_tiles), dim3(TH), 0, s->stream,         \
                       t.tiles, t.codes, (const float *)t.w32, t.r_tid, (const float *)t.r_w32, t.r_row,    \
                       t.r_slot, bb.queue, t.n_remote, bb.theta, bb.cnt, bb.state, bb.row_w)
#define OEM_TILE_B(TH, REM, NC, MW)                                                                       \
    do {                                                                                                  \
        if (nt) OEM_TILE_B_NT(TH, REM, NC, MW, true);                                                     \
        else OEM_TILE_B_NT(TH, REM, NC, MW, false);                                                       \
    } while (0)
    switch (variant) {
    case 1: OEM_TILE_B(256, 6, 4, 2); break;
    case 2: OEM_TILE_B(512, 3, 2, 2); break;
    case 3: OEM_TILE_B(512, 3, 4, 2); break;
    case 4: OEM_TILE_B(256, 6, 1, 2); break;
    default: OEM_TILE_B(256, 6, 2, 2); break;
    }
#undef OEM_TILE_B
#undef OEM_TILE_B_NT
    OEM_HIP(hipGetLastError());
    if (t.n_remote > 0) {
        uint32_t n_groups = 256 / (t.n_buckets ? t.n_buckets : 1);
        const uint64_t per_bucket = t.n_remote / (t.n_buckets ? t.n_buckets : 1) + 1;
        const uint32_t max_useful = (uint32_t)((per_bucket + 32767) / 32768);
        if (n_groups > max_useful) n_groups = max_useful;
        if (n_groups < 1) n_groups = 1;
        hipLaunchKernelGGL(k_remote_fold_b, dim3(kB * t.n_buckets * n_groups), dim3(kFoldThreadsB), 0, s->stream,
                           t.bucket_base, bb.queue, t.n_remote, t.q_dst, bb.cnt2, bb.state, n_groups,
                           t.n_buckets, s->csr.n_txps);
        OEM_HIP(hipGetLastError());
    }
    return OEM_OK;
}

int launch_batch_reldiff(oem_store *s, const BatchBuffers &bb, EmParams p)
{
    const int grid = grid_for(p.n_txps, kRelB, 64);
    hipLaunchKernelGGL(k_reldiff_b, dim3(grid), dim3(kRelB), 0, s->stream, bb.theta, bb.cnt, bb.cnt2, bb.out,
                       bb.state, p);
    // (a replicate on its FINAL pass reads theta < 1e-5 as 0 inside k_em_tile_b: em.rs:238-242)
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

int launch_batch_reset_slot(oem_store *s, const BatchBuffers &bb, const double *d_init, double avg, uint32_t slot)
{
    const int grid = grid_for(s->csr.n_txps, 256, 256);
    hipLaunchKernelGGL(k_reset_slot_b, dim3(grid), dim3(256), 0, s->stream, bb.theta, bb.cnt, bb.cnt2, d_init, avg,
                       s->csr.n_txps, slot);
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

int launch_batch_pack_row_w(oem_store *s, const uint32_t *d_row_w_all, const BatchBuffers &bb, uint32_t *d_overflow)
{
    const uint64_t n = s->tiled.n_rows;
    if (n == 0) return OEM_OK;
    hipLaunchKernelGGL(k_pack_row_w_b, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s->stream, d_row_w_all,
                       s->csr.n_reads, s->tiled.perm, n, bb.row_w, d_overflow);
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

} // namespace oem
