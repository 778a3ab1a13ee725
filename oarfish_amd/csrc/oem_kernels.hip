// oem_kernels.hip -- gfx950 kernels of the EM engine.
//
// Reference semantics (COMBINE-lab/oarfish v0.10.3):
//   E/M pass                 src/em.rs:87-133 (m_step), :22-79 (m_step_par)
//   rel-diff / swap / clear  src/em.rs:194-218 (do_em), :379-405 (em_par)
//   zero small + final pass  src/em.rs:238-252
//   bootstrap resample       src/bootstrap.rs:7-16
// Nothing here is a translation of that code: the reference walks an AoS store
// on CPU threads; these kernels keep the matrix as (tid, w) streams in HBM and
// the loop state on the device.
#include <cstdlib>
#include "oem_internal.h"

namespace oem {

namespace {

constexpr int kBlock = 256;
constexpr int kRelBlock = 1024; // k_reldiff_swap_clear

__device__ __forceinline__ void atomic_add_f64(double *p, double v)
{
    // hardware global_atomic_add_f64 (no CAS loop); device scope.
    unsafeAtomicAdd(p, v);
}

// ---------------------------------------------------------------------------
// v1 E/M pass: one lane per read over the caller-order CSR.
// ---------------------------------------------------------------------------
template <typename PtrT, typename WT>
__global__ __launch_bounds__(kBlock) void k_em_pass_csr(
    const PtrT *__restrict__ row_ptr, const uint32_t *__restrict__ tid, const WT *__restrict__ w,
    const double *__restrict__ theta, double *__restrict__ cnt, const EmState *state,
    const uint32_t *__restrict__ row_w, uint64_t row_begin, uint64_t row_end)
{
    // Once the stopping rule has fired on the device the remaining launches of the
    // run are no-ops (the host only looks at the state every few passes).
    if (state && state->done) return;

    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t r = row_begin + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < row_end;
         r += stride) {
        double scale = 1.0;
        if (row_w) {
            const uint32_t c = row_w[r];
            if (c == 0) continue;
            scale = (double)c;
        }
        const uint64_t b = row_ptr[r], e = row_ptr[r + 1];
        double denom = 0.0;
        for (uint64_t j = b; j < e; ++j) denom += theta[tid[j]] * (double)w[j]; // em.rs:111
        if (denom > OEM_EM_DENOM_THRESH) {                                      // em.rs:115
            const double inv = scale / denom;
            for (uint64_t j = b; j < e; ++j) {
                const uint32_t t = tid[j];
                atomic_add_f64(&cnt[t], theta[t] * (double)w[j] * inv);         // em.rs:128-129
            }
        }
    }
}

// ---------------------------------------------------------------------------
// rel-diff + swap + clear + stopping rule, fused (em.rs:194-218 / :379-405).
// ---------------------------------------------------------------------------
template <int kRB>
__global__ __launch_bounds__(kRB) void k_reldiff_swap_clear(double *__restrict__ prev,
                                                               double *__restrict__ curr,
                                                               EmState *state, EmParams p)
{
    if (state->done) return;

    double rel = 0.0; // em.rs:169 / :234: starts at 0 => negative diffs never win
    // four elements per trip, all eight loads issued before the first use: the sweep is a handful of elements
    // per thread, and one element per trip made it a chain of round trips
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < p.n_txps; i0 += 4 * stride) {
        double pc[4], cc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = i0 + k * stride;
            const uint32_t ic = i < p.n_txps ? i : i0;
            pc[k] = prev[ic];
            cc[k] = curr[ic];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = i0 + k * stride;
            if (i < p.n_txps) {
                if (pc[k] > OEM_MIN_READ_THRESH) rel = fmax(rel, (cc[k] - pc[k]) / pc[k]); // em.rs:195-199 (signed)
                prev[i] = cc[k];              // em.rs:204 swap: prev_counts <- this pass's counts
                curr[i] = 0.0;                // em.rs:207 clear
            }
        }
    }
    // wave64 max, then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) rel = fmax(rel, __shfl_xor(rel, off, 64));
    __shared__ double smax[kRB / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) smax[wv] = rel;
    __syncthreads();
    __shared__ bool is_last;
    if (threadIdx.x == 0) {
        double m = smax[0];
        for (int i = 1; i < kRB / 64; ++i) m = fmax(m, smax[i]);
        // non-negative doubles order like their bit patterns.  Both this and the ticket below are
        // device-scope read-modify-write atomics, performed at the one point of coherence of their
        // line (memory side), and on gfx9 a no-return atomic is counted by vmcnt until it has been
        // performed there: draining vmcnt before taking the ticket means the maximum is in place
        // before the ticket can be observed, so the workgroup that draws the last ticket reads (with
        // an agent-scope atomic load) a maximum that contains every workgroup's.  A release fence
        // here would add an L2 write-back (buffer_wbl2) that publishes nothing this decision needs
        // (~3.5 us per workgroup tail, measured in round 1).  The election is hammered in isolation
        // by oem_test_reldiff_stress (test-only library): > 10^5 launches over 1..64 workgroups,
        // planted maxima, the decision workgroup's view compared bit for bit.
        if (m > 0.0) atomicMax(&state->rel_bits, (unsigned long long)__double_as_longlong(m));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t ticket = atomicAdd(&state->blocks_arrived, 1u);
        is_last = (ticket == gridDim.x - 1);
    }
    __syncthreads();
    if (is_last && threadIdx.x == 0) {
        const unsigned long long bits =
            __hip_atomic_load(&state->rel_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double rel_diff = __longlong_as_double((long long)bits);
        state->last_rel = rel_diff;
        state->n_passes += 1;
        uint32_t niter = state->niter;
        if (rel_diff < p.conv_thresh && niter > p.min_iter_gate) { // em.rs:212 / :399
            state->done = 1;
            state->converged = 1;
        } else {
            niter += 1;                                            // em.rs:218
            state->niter = niter;
            if (niter >= p.max_iter) state->done = 1;              // em.rs:181 loop condition
        }
        state->rel_bits = 0ull;                                    // em.rs:234
        state->blocks_arrived = 0u;
    }
}

// em.rs:238-242 on prev_counts; curr_counts is (re)zeroed for the final pass.
__global__ __launch_bounds__(kBlock) void k_zero_small(double *__restrict__ prev,
                                                       double *__restrict__ curr, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (prev[i] < OEM_MIN_READ_THRESH) prev[i] = 0.0;
        curr[i] = 0.0;
    }
}

__global__ __launch_bounds__(kBlock) void k_fill(double *p, double v, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = v;
}

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11), counter-based: draw k of replica b is a
// pure function of (seed, b, k), so every row shard draws the same resample.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
        const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += W0; k1 += W1;
    }
}

// bootstrap.rs:7-16 in multiplicity form: n_global draws from Uniform[0, n_global);
// this shard keeps the ones that fall into its rows [local_off, local_off+n_local).
__global__ __launch_bounds__(kBlock) void k_bootstrap_weights(uint32_t *row_w, uint64_t n_local,
                                                              uint64_t local_off, uint64_t n_global,
                                                              uint64_t seed, uint32_t replica)
{
    const uint64_t n_pairs = (n_global + 1) / 2;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_pairs;
         q += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), replica, 0x6f656d62u /* "oemb" */};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint64_t k = 2 * q + h;
            if (k >= n_global) break;
            const uint64_t r64 = ((uint64_t)c[2 * h] << 32) | c[2 * h + 1];
            // multiply-high maps 64 random bits onto [0, n): bias < n / 2^64
            const uint64_t idx = __umul64hi(r64, n_global);
            if (idx >= local_off && idx - local_off < n_local) atomicAdd(&row_w[idx - local_off], 1u);
        }
    }
}

// aux_counts.rs:23-50
template <typename PtrT>
__global__ __launch_bounds__(kBlock) void k_aux_counts(const PtrT *__restrict__ row_ptr,
                                                       const uint32_t *__restrict__ tid, uint64_t n_reads,
                                                       uint32_t *unique_count, uint32_t *total_count)
{
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads;
         r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = row_ptr[r], e = row_ptr[r + 1];
        const bool is_unique = (e - b) == 1;                       // aux_counts.rs:35
        for (uint64_t j = b; j < e; ++j) {
            atomicAdd(&total_count[tid[j]], 1u);
            if (is_unique) atomicAdd(&unique_count[tid[j]], 1u);
        }
    }
}

// write_function.rs:283-318, one lane per read over the caller-order CSR
template <typename PtrT, typename WT>
__global__ __launch_bounds__(kBlock) void k_assignment_probs(const PtrT *__restrict__ row_ptr,
                                                             const uint32_t *__restrict__ tid,
                                                             const WT *__restrict__ w,
                                                             const double *__restrict__ counts,
                                                             uint64_t n_reads, double display_thresh,
                                                             double *__restrict__ out)
{
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads;
         r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = row_ptr[r], e = row_ptr[r + 1];
        double denom = 0.0;
        for (uint64_t j = b; j < e; ++j) denom += counts[tid[j]] * (double)w[j];       // :286-291
        double denom2 = 0.0;
        for (uint64_t j = b; j < e; ++j) {                                              // :303-314
            double nprob = (counts[tid[j]] * (double)w[j]) / denom;
            if (nprob < 0.0) nprob = 0.0;
            if (nprob > 1.0) nprob = 1.0;                                               // clamp keeps NaN
            const bool keep = nprob >= display_thresh;
            out[j] = keep ? nprob : -1.0;
            if (keep) denom2 += nprob;
        }
        for (uint64_t j = b; j < e; ++j)                                                // :316-318
            if (out[j] >= 0.0) out[j] /= denom2;
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

// ---------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------
int launch_em_pass(oem_store *s, const double *theta, double *cnt, const EmState *state,
                   const uint32_t *row_w, uint64_t row_begin, uint64_t row_end)
{
    const DeviceCsr &m = s->csr;
    if (row_end <= row_begin) return OEM_OK;
    const int grid = grid_for(row_end - row_begin, kBlock, 256 * 32);
#define OEM_LAUNCH_CSR(PT, WT, wptr)                                                              \
    hipLaunchKernelGGL((k_em_pass_csr<PT, WT>), dim3(grid), dim3(kBlock), 0, s->stream,           \
                       (const PT *)m.row_ptr, m.tid, wptr, theta, cnt, state, row_w, row_begin,    \
                       row_end)
    if (m.wide_ptr) {
        if (m.w_is_f64) OEM_LAUNCH_CSR(uint64_t, double, (const double *)m.w64);
        else OEM_LAUNCH_CSR(uint64_t, float, (const float *)m.w32);
    } else {
        if (m.w_is_f64) OEM_LAUNCH_CSR(uint32_t, double, (const double *)m.w64);
        else OEM_LAUNCH_CSR(uint32_t, float, (const float *)m.w32);
    }
#undef OEM_LAUNCH_CSR
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

int launch_aux_counts(oem_store *s, uint32_t *d_unique, uint32_t *d_total)
{
    const DeviceCsr &m = s->csr;
    if (m.n_reads == 0) return OEM_OK;
    const int grid = grid_for(m.n_reads, kBlock, 256 * 16);
    if (m.wide_ptr)
        hipLaunchKernelGGL((k_aux_counts<uint64_t>), dim3(grid), dim3(kBlock), 0, s->stream,
                           (const uint64_t *)m.row_ptr, m.tid, m.n_reads, d_unique, d_total);
    else
        hipLaunchKernelGGL((k_aux_counts<uint32_t>), dim3(grid), dim3(kBlock), 0, s->stream,
                           (const uint32_t *)m.row_ptr, m.tid, m.n_reads, d_unique, d_total);
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

int launch_assignment_probs(oem_store *s, const double *d_counts, double display_thresh, double *d_out)
{
    const DeviceCsr &m = s->csr;
    if (m.n_reads == 0) return OEM_OK;
    const int grid = grid_for(m.n_reads, kBlock, 256 * 16);
#define OEM_LAUNCH_AP(PT, WT, wptr)                                                                  \
    hipLaunchKernelGGL((k_assignment_probs<PT, WT>), dim3(grid), dim3(kBlock), 0, s->stream,          \
                       (const PT *)m.row_ptr, m.tid, wptr, d_counts, m.n_reads, display_thresh, d_out)
    if (m.wide_ptr) {
        if (m.w_is_f64) OEM_LAUNCH_AP(uint64_t, double, (const double *)m.w64);
        else OEM_LAUNCH_AP(uint64_t, float, (const float *)m.w32);
    } else {
        if (m.w_is_f64) OEM_LAUNCH_AP(uint32_t, double, (const double *)m.w64);
        else OEM_LAUNCH_AP(uint32_t, float, (const float *)m.w32);
    }
#undef OEM_LAUNCH_AP
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

int launch_reldiff_swap_clear(oem_store *s, double *prev, double *curr, EmState *state, EmParams p)
{
    // 64 workgroups of 1024 threads: every workgroup ends with two device-scope atomics on the
    // one state line (max, then ticket) and those serialise, so fewer, fatter workgroups win
    // (MI355X, 200 k transcripts: 256 x 256 threads 11.3 us -> 64 x 1024 threads 8.1 us).
    const int grid = grid_for(p.n_txps, kRelBlock, 64);
    hipLaunchKernelGGL(k_reldiff_swap_clear<kRelBlock>, dim3(grid), dim3(kRelBlock), 0, s->stream, prev, curr, state, p);
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

int launch_zero_small(oem_store *s, double *prev, double *curr, uint32_t n_txps)
{
    const int grid = grid_for(n_txps, kBlock, 256);
    hipLaunchKernelGGL(k_zero_small, dim3(grid), dim3(kBlock), 0, s->stream, prev, curr, n_txps);
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

int launch_fill(oem_store *s, double *p, double v, uint64_t n)
{
    if (n == 0) return OEM_OK;
    const int grid = grid_for(n, kBlock, 1024);
    hipLaunchKernelGGL(k_fill, dim3(grid), dim3(kBlock), 0, s->stream, p, v, n);
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

int launch_bootstrap_weights(oem_store *s, uint32_t *row_w, uint64_t n_local, uint64_t local_off,
                             uint64_t n_global, uint64_t seed, uint32_t replica, hipStream_t stream)
{
    if (!stream) stream = s->stream;
    OEM_HIP(hipMemsetAsync(row_w, 0, sizeof(uint32_t) * n_local, stream));
    if (n_global == 0) return OEM_OK;
    const int grid = grid_for((n_global + 1) / 2, kBlock, 256 * 16);
    hipLaunchKernelGGL(k_bootstrap_weights, dim3(grid), dim3(kBlock), 0, stream, row_w, n_local,
                       local_off, n_global, seed, replica);
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

} // namespace oem
