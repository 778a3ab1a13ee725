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

namespace oem {

namespace {

constexpr int kMT = 256;

// theta[p*T + i] = reads(p) / T   (em.rs:165 with the cell's own store.len())
__global__ __launch_bounds__(kMT) void k_multi_init(double *__restrict__ theta,
                                                    const uint64_t *__restrict__ problem_reads, uint32_t T)
{
    const uint32_t p = blockIdx.y;
    const double avg = (double)problem_reads[p] / (double)T;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < T; i += gridDim.x * blockDim.x)
        theta[(size_t)p * T + i] = avg;
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
        s.zeroed = 0;
    }
    st[i] = s;
}

// em.rs:238-242 for the cells that just entered FINAL
__global__ __launch_bounds__(kMT) void k_multi_zero_small(double *__restrict__ theta, BatchState *st, uint32_t T)
{
    const uint32_t p = blockIdx.y;
    if (st[p].phase != kPhaseFinal || st[p].zeroed) return;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < T; i += gridDim.x * blockDim.x) {
        const size_t k = (size_t)p * T + i;
        if (theta[k] < OEM_MIN_READ_THRESH) theta[k] = 0.0;
    }
}
__global__ __launch_bounds__(kMT) void k_multi_mark_zeroed(BatchState *st, uint32_t n_problems)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_problems && st[i].phase == kPhaseFinal) st[i].zeroed = 1;
}

} // namespace

int launch_multi_init(oem_store *s, double *theta, const uint64_t *d_problem_reads, const MultiBuffers &mb)
{
    const uint32_t T = mb.problem_size;
    uint32_t gx = (T + kMT - 1) / kMT;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_multi_init, dim3(gx, mb.n_problems), dim3(kMT), 0, s->stream, theta, d_problem_reads, T);
    OEM_HIP(hipGetLastError());
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
    hipLaunchKernelGGL(k_multi_zero_small, dim3(gx, mb.n_problems), dim3(kMT), 0, s->stream, theta, mb.state, T);
    hipLaunchKernelGGL(k_multi_mark_zeroed, dim3(gp), dim3(kMT), 0, s->stream, mb.state, mb.n_problems);
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

} // namespace oem
