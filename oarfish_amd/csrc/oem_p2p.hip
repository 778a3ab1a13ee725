// oem_p2p.hip -- peer-to-peer all-reduce of the count vector over the row shards of one node, without RCCL.
//
// The only exchange of the EM path is the sum of the n_txps partial counts per E/M pass (SURVEY.md
// section 8e; in the reference the shared Vec<AtomicF64> of em.rs:338-341).  At 200 k transcripts that
// is 1.6 MB: latency-bound, and a ring (RCCL's default, 2 (N-1) steps, each bound by ONE xGMI link) is
// the wrong shape for it.  MI355X nodes are fully connected (7 links x ~153 GB/s per GPU), so here every
// rank publishes its partial vector in a buffer its peers have mapped (hipIpc memory handles; the same
// address space when the ranks are threads of one process) and the peers read it directly, all links in
// parallel.  Two shapes, by the number of ranks:
//
//   one-shot (N = 2, and short vectors): every rank reads the others' whole partials and adds in rank order.
//     k_p2p_publish   send -> own slot[parity]; the last workgroup raises this rank's flag A in every
//                     peer's flag block (a peer spins on its OWN memory)
//     k_p2p_reduce    waits for the flags, recv[i] = sum over ranks r = 0..N-1 of slot_r[parity][i]
//     k_p2p_reldiff   the same wait and sum fused into rel-diff / swap / clear / stopping rule
//                     (em.rs:194-218): the reduced vector is never written and read back
//   two-phase (N >= 3 and vectors of half a megabyte or more): one-shot pulls (N - 1) whole vectors through every rank's links -- 11 MB per rank and
//     pass at N = 8 -- where a reduce-scatter + all-gather pulls 2 (N - 1) / N of ONE vector (2.8 MB), spread
//     over the same N - 1 links: rank r sums slice r of all partials in rank order into its `red` buffer
//     (k_p2p_reduce_slice, after the flags A; raises flag B at every peer), and the last kernel (k_p2p_gather /
//     k_p2p_reldiff with kTwoPhase) waits for the flags B and reads every slice from its owner.  One more
//     flag round and one more small kernel for a quarter of the bytes per link at N = 8.
//   (oem_comm_set_option(OEM_COMM_OPT_P2P_SHAPE) forces either; bench.py times both next to RCCL on the node
//   it runs on and keeps the fastest.)
//
// Every element is added in rank order, by every rank (one-shot) or by its one owner (two-phase), so the
// reduced vector is bit-identical on all ranks and they take the identical stopping decision without a
// second exchange.  The buffers are double-buffered by the parity of a device-resident epoch counter: a rank
// overwrites slot[p] two exchanges later, after its own previous exchange has seen every peer's flag for the
// exchange in between, which a peer raises only after it has finished reading (stream order) -- no extra
// barrier; `red[p]` likewise (it is rewritten after the flags A of exchange e + 2, which a peer raises after
// its last kernel of exchange e + 1, hence e, has read it).  The epoch lives on the device and advances
// inside the kernels, so launches carry no per-call values and a chunk of iterations can be replayed from a
// hipGraph.  Launches of a finished run (EmState::done) skip the exchange on every rank alike, since the
// state they test is identical.
//
// Remote loads are round trips of microseconds: every kernel issues all the loads of a batch -- kP2PBatch
// elements x all ranks -- before the first add (the sum itself stays in rank order).
//
// A spinning wait is bounded (wall clock): a peer that never arrives sets an error flag that the host
// reports as OEM_ERR_STATE instead of hanging the GPU.
#include <time.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "oem_internal.h"

namespace oem {

constexpr int kP2PMaxRanks = 16;
constexpr uint32_t kP2PMagic = 0x6f703270u; // "op2p"
constexpr int kP2PBlock = 256;
constexpr long long kP2PTicksPerMs = 100000ll;       // wall_clock64() runs at 100 MHz
constexpr uint64_t kP2PDefaultTimeoutMs = 8000;       // OEM_COMM_OPT_P2P_TIMEOUT_MS
constexpr uint64_t kP2PSelfCheckTimeoutMs = 120000;   // the first exchange is also the ranks' rendezvous: wait long

// The region a rank shares with its peers (one allocation, one IPC handle).
struct P2PShared {
    unsigned long long flags[2][kP2PMaxRanks];  // A: [parity][source rank] = epoch of that rank's last publish
    unsigned long long flags_b[2][kP2PMaxRanks]; // B: ... of that rank's last reduced slice (two-phase)
    unsigned long long pad[512 - 4 * kP2PMaxRanks];
    // double slot[2][capacity] (partials), then double red[2][capacity] (reduced slices, indexed like the vector) follow
};
static_assert(sizeof(P2PShared) == 4096, "P2PShared header");

__host__ __device__ inline double *p2p_slot(P2PShared *s, uint64_t capacity, uint32_t parity)
{
    return reinterpret_cast<double *>(reinterpret_cast<char *>(s) + sizeof(P2PShared)) + (size_t)parity * capacity;
}
__host__ __device__ inline double *p2p_red(P2PShared *s, uint64_t capacity, uint32_t parity)
{
    return p2p_slot(s, capacity, 2u + parity);
}
// slice of rank r of a vector of n elements: [p2p_slice_begin(r), p2p_slice_begin(r + 1))
__host__ __device__ inline uint64_t p2p_slice_begin(uint64_t n, int r, int n_ranks) { return n * (uint64_t)r / (uint64_t)n_ranks; }

// rank-local control block (device memory, never shared)
struct P2PCtl {
    unsigned long long epoch; // completed exchanges
    uint32_t arrived_pub;     // last-workgroup tickets
    uint32_t arrived_red;
    uint32_t error;           // 1: a peer's flag did not arrive in time
    uint32_t arrived_slice;
    long long timeout_ticks;  // bound of one spinning wait (device wall clock)
    P2PShared *peer[kP2PMaxRanks]; // mapped regions, [rank] = own
};
static_assert(sizeof(P2PCtl) == 32 + 8 * kP2PMaxRanks, "P2PCtl layout");

struct P2P {
    int rank = 0, n_ranks = 1, device = 0;
    uint64_t capacity = 0; // doubles per slot
    P2PShared *self = nullptr;
    P2PCtl *ctl = nullptr;
    P2PCtl *h_ctl = nullptr; // pinned copy for error checks
    void *opened[kP2PMaxRanks] = {};  // hipIpcOpenMemHandle results to close
    bool connected = false;
    bool fine_grained = false; // the shared region is fine-grained device memory (else ordinary hipMalloc memory)
    bool self_check = false;   // OEM_COMM_OPT_P2P_SELF_CHECK: oem_comm_p2p_connect ends with a checked exchange
    uint64_t timeout_ms = kP2PDefaultTimeoutMs;
    int shape = 0; // OEM_COMM_OPT_P2P_SHAPE: 0 by the number of ranks, 1 one-shot, 2 two-phase
};

struct P2PBlob { // OEM_P2P_HANDLE_BYTES
    uint32_t magic, version;
    uint64_t pid, ptr, capacity;
    int32_t device, rank;
    hipIpcMemHandle_t handle;
    uint64_t nonce; // of the exporting PROCESS: ranks in different containers / pid namespaces can share a pid
    char pad[OEM_P2P_HANDLE_BYTES - 4 - 4 - 8 - 8 - 8 - 4 - 4 - sizeof(hipIpcMemHandle_t) - 8];
};
static_assert(sizeof(P2PBlob) == OEM_P2P_HANDLE_BYTES, "P2PBlob size");

namespace {

// Coherence by construction, without cache-wide maintenance.  Every access to a shared region -- partials
// and flags, own and peers' -- is a system-scope access (sc0 sc1: stores write through to memory, loads are
// served by memory), so nothing shared ever sits dirty or stale in a per-XCD L2; ordering is by waiting for
// the stores to be acknowledged (vmcnt) before a workgroup takes its ticket, and the flags go up after the
// last ticket.  The generic __threadfence_system() would do the same job with an L2 write-back AND an L2
// invalidate per wavefront (buffer_wbl2 + buffer_inv): measured, a 200 k-entry exchange of one rank with
// itself took 25 us that way (profiles/r03_notes.md).
__device__ __forceinline__ void sys_store_u64(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long sys_load_u64(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void sys_store_f64(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double sys_load_f64(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// every store this thread has issued is acknowledged by memory
__device__ __forceinline__ void stores_performed() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// The flags are raised with a system-scope RELEASE and consumed with a system-scope ACQUIRE on top of that: only the
// few threads that raise or wait for a flag execute them (one L2 write-back per exchange and one invalidate per
// waiting workgroup, not one per wavefront), and the pair is what the memory model asks of a flag that a peer
// DEVICE polls while this kernel runs -- the by-construction argument above has only ever been run on one device.
__device__ __forceinline__ void flag_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); }
__device__ __forceinline__ void flag_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); }

__global__ __launch_bounds__(kP2PBlock) void k_p2p_publish(const double *__restrict__ send, P2PCtl *ctl,
                                                           uint64_t capacity, uint64_t count, int rank, int n_ranks,
                                                           const EmState *state)
{
    if (state && state->done) return;
    const unsigned long long e = ctl->epoch + 1;
    P2PShared *self = ctl->peer[rank];
    double *slot = p2p_slot(self, capacity, (uint32_t)(e & 1));
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x)
        sys_store_f64(&slot[i], send[i]);
    stores_performed(); // this thread's part of the partial is in memory
    __syncthreads();
    __shared__ bool is_last;
    if (threadIdx.x == 0) {
        const uint32_t ticket = atomicAdd(&ctl->arrived_pub, 1u);
        is_last = ticket == gridDim.x - 1;
    }
    __syncthreads();
    if (is_last) { // every workgroup's part is in memory: raise this rank's flag at every peer
        if ((int)threadIdx.x < n_ranks && (int)threadIdx.x != rank) {
            flag_release();
            sys_store_u64(&ctl->peer[threadIdx.x]->flags[e & 1][rank], e);
        }
        if (threadIdx.x == 0) ctl->arrived_pub = 0u;
    }
}

// wait until every peer has raised its flag (A: published, B: slice reduced) for exchange `e`; the flags live
// in OUR region
__device__ __forceinline__ void p2p_wait(P2PCtl *ctl, unsigned long long e, int rank, int n_ranks, bool flag_b = false)
{
    // (once a wait has timed out the run is lost: later launches do not wait another 8 s each)
    if ((int)threadIdx.x < n_ranks && (int)threadIdx.x != rank &&
        !__hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        P2PShared *self = ctl->peer[rank];
        const unsigned long long *f = flag_b ? &self->flags_b[e & 1][threadIdx.x] : &self->flags[e & 1][threadIdx.x];
        const long long t0 = wall_clock64(), limit = ctl->timeout_ticks;
        while (sys_load_u64(f) < e) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > limit) {
                ctl->error = 1u;
                break;
            }
        }
        flag_acquire();
    }
    __syncthreads();
}

constexpr int kP2PBatch = 4; // elements per thread whose loads are all in flight together

// out[k] = sum over ranks, in rank order, of slot_r[parity][i[k]]: every load of the batch (kP2PBatch elements x
// n_ranks, the own slot too: it was written through, a cached copy may predate that) is issued before the
// first add -- a remote load is a round trip of microseconds, and rank after rank they would add up.
__device__ __forceinline__ void p2p_sum_batch(const P2PCtl *ctl, uint64_t capacity, uint32_t parity, const uint64_t (&i)[kP2PBatch],
                                              int n_ranks, double (&out)[kP2PBatch])
{
#pragma unroll
    for (int k = 0; k < kP2PBatch; ++k) out[k] = 0.0;
    constexpr int kGroup = 8; // ranks whose loads are in flight together (kGroup * kP2PBatch * 2 VGPRs)
    for (int r0 = 0; r0 < n_ranks; r0 += kGroup) {
        double v[kGroup][kP2PBatch];
#pragma unroll
        for (int g = 0; g < kGroup; ++g) {
            if (r0 + g < n_ranks) { // uniform
                const double *slot = p2p_slot(ctl->peer[r0 + g], capacity, parity);
#pragma unroll
                for (int k = 0; k < kP2PBatch; ++k) v[g][k] = sys_load_f64(&slot[i[k]]);
            }
        }
#pragma unroll
        for (int g = 0; g < kGroup; ++g)
            if (r0 + g < n_ranks) {
#pragma unroll
                for (int k = 0; k < kP2PBatch; ++k) out[k] += v[g][k]; // rank order: the same sum, bit for bit, wherever it is formed
            }
    }
}

// two-phase: out[k] = the reduced element i[k], read from the rank that owns its slice
__device__ __forceinline__ void p2p_gather_batch(const P2PCtl *ctl, uint64_t capacity, uint32_t parity, uint64_t count,
                                                 const uint64_t (&i)[kP2PBatch], int n_ranks, double (&out)[kP2PBatch])
{
#pragma unroll
    for (int k = 0; k < kP2PBatch; ++k) {
        // owner of element i = the largest r with floor(count r / N) <= i  <=>  r < (i + 1) N / count
        const int r = (int)(((i[k] + 1) * (uint64_t)n_ranks - 1) / count);
        out[k] = sys_load_f64(&p2p_red(ctl->peer[r], capacity, parity)[i[k]]);
    }
}

// two-phase, first half: after the flags A, this rank sums ITS slice of all partials into its `red` buffer
// and raises flag B at every peer
__global__ __launch_bounds__(kP2PBlock) void k_p2p_reduce_slice(P2PCtl *ctl, uint64_t capacity, uint64_t count, int rank,
                                                                int n_ranks, const EmState *state)
{
    if (state && state->done) return;
    const unsigned long long e = ctl->epoch + 1;
    p2p_wait(ctl, e, rank, n_ranks);
    const uint32_t parity = (uint32_t)(e & 1);
    const uint64_t b = p2p_slice_begin(count, rank, n_ranks), end = p2p_slice_begin(count, rank + 1, n_ranks);
    double *red = p2p_red(ctl->peer[rank], capacity, parity);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i0 = b + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < end; i0 += kP2PBatch * stride) {
        uint64_t idx[kP2PBatch];
        double sum[kP2PBatch];
#pragma unroll
        for (int k = 0; k < kP2PBatch; ++k) idx[k] = i0 + k * stride < end ? i0 + k * stride : i0;
        p2p_sum_batch(ctl, capacity, parity, idx, n_ranks, sum);
#pragma unroll
        for (int k = 0; k < kP2PBatch; ++k)
            if (i0 + k * stride < end) sys_store_f64(&red[idx[k]], sum[k]);
    }
    stores_performed();
    __syncthreads();
    __shared__ bool is_last;
    if (threadIdx.x == 0) {
        const uint32_t ticket = atomicAdd(&ctl->arrived_slice, 1u);
        is_last = ticket == gridDim.x - 1;
    }
    __syncthreads();
    if (is_last) {
        if ((int)threadIdx.x < n_ranks && (int)threadIdx.x != rank) {
            flag_release();
            sys_store_u64(&ctl->peer[threadIdx.x]->flags_b[e & 1][rank], e);
        }
        if (threadIdx.x == 0) ctl->arrived_slice = 0u;
    }
}

// recv = the reduced vector: summed here from all partials (one-shot) or gathered from the slice owners
template <bool kTwoPhase>
__global__ __launch_bounds__(kP2PBlock) void k_p2p_reduce(double *__restrict__ recv, P2PCtl *ctl, uint64_t capacity,
                                                          uint64_t count, int rank, int n_ranks, const EmState *state)
{
    if (state && state->done) return;
    const unsigned long long e = ctl->epoch + 1;
    p2p_wait(ctl, e, rank, n_ranks, kTwoPhase);
    const uint32_t parity = (uint32_t)(e & 1);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < count; i0 += kP2PBatch * stride) {
        uint64_t idx[kP2PBatch];
        double sum[kP2PBatch];
#pragma unroll
        for (int k = 0; k < kP2PBatch; ++k) idx[k] = i0 + k * stride < count ? i0 + k * stride : i0;
        if (kTwoPhase) p2p_gather_batch(ctl, capacity, parity, count, idx, n_ranks, sum);
        else p2p_sum_batch(ctl, capacity, parity, idx, n_ranks, sum);
#pragma unroll
        for (int k = 0; k < kP2PBatch; ++k)
            if (i0 + k * stride < count) recv[idx[k]] = sum[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t ticket = atomicAdd(&ctl->arrived_red, 1u);
        if (ticket == gridDim.x - 1) { // every workgroup has read `epoch` (at its start) and its share of the buffers
            ctl->arrived_red = 0u;
            __hip_atomic_store(&ctl->epoch, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The exchange fused into rel-diff / swap / clear / stopping rule (k_reldiff_swap_clear, oem_kernels.hip):
// curr is this rank's partial (already published), the reduced value is summed from the slots (one-shot) or
// read from the slice owners (two-phase).
template <int kRB, bool kTwoPhase>
__global__ __launch_bounds__(kRB) void k_p2p_reldiff(double *__restrict__ prev, double *__restrict__ curr, EmState *state,
                                                      EmParams p, P2PCtl *ctl, uint64_t capacity, int rank, int n_ranks)
{
    if (state->done) return;
    const unsigned long long e = ctl->epoch + 1;
    p2p_wait(ctl, e, rank, n_ranks, kTwoPhase);
    const uint32_t parity = (uint32_t)(e & 1);
    double rel = 0.0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, n = p.n_txps;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += kP2PBatch * stride) {
        uint64_t idx[kP2PBatch];
        double cc[kP2PBatch], pc[kP2PBatch];
#pragma unroll
        for (int k = 0; k < kP2PBatch; ++k) idx[k] = i0 + k * stride < n ? i0 + k * stride : i0;
#pragma unroll
        for (int k = 0; k < kP2PBatch; ++k) pc[k] = prev[idx[k]];
        if (kTwoPhase) p2p_gather_batch(ctl, capacity, parity, n, idx, n_ranks, cc);
        else p2p_sum_batch(ctl, capacity, parity, idx, n_ranks, cc);
#pragma unroll
        for (int k = 0; k < kP2PBatch; ++k)
            if (i0 + k * stride < n) {
                if (pc[k] > OEM_MIN_READ_THRESH) rel = fmax(rel, (cc[k] - pc[k]) / pc[k]); // em.rs:195-199
                prev[idx[k]] = cc[k];                                                    // em.rs:204
                curr[idx[k]] = 0.0;                                                      // em.rs:207
            }
    }
    for (int off = 32; off > 0; off >>= 1) rel = fmax(rel, __shfl_xor(rel, off, 64));
    __shared__ double smax[kRB / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) smax[wv] = rel;
    __syncthreads();
    __shared__ bool is_last;
    if (threadIdx.x == 0) {
        double m = smax[0];
        for (int i = 1; i < kRB / 64; ++i) m = fmax(m, smax[i]);
        if (m > 0.0) atomicMax(&state->rel_bits, (unsigned long long)__double_as_longlong(m));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (ordering argument: k_reldiff_swap_clear)
        const uint32_t ticket = atomicAdd(&state->blocks_arrived, 1u);
        is_last = (ticket == gridDim.x - 1);
    }
    __syncthreads();
    if (is_last && threadIdx.x == 0) {
        const unsigned long long bits = __hip_atomic_load(&state->rel_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
            if (niter >= p.max_iter) state->done = 1;              // em.rs:181
        }
        // a wait that gave up: the sums of this run are lost -- end it here instead of iterating on stale slots
        if (__hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) state->done = 1;
        state->rel_bits = 0ull;
        state->blocks_arrived = 0u;
        __hip_atomic_store(&ctl->epoch, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int grid_for_count(uint64_t n, int block, int max_blocks)
{
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (uint64_t)max_blocks) g = max_blocks;
    return (int)g;
}

} // namespace

// Identifies this PROCESS among the exporters of a node: the pid alone does not (ranks in separate containers or
// pid namespaces routinely share one), and a peer mistaken for a thread of this process would have its raw
// device pointer dereferenced instead of its IPC handle opened.
static uint64_t process_nonce()
{
    static const uint64_t nonce = [] {
        uint64_t v = 0;
        FILE *f = fopen("/dev/urandom", "rb");
        if (f) {
            if (fread(&v, sizeof(v), 1, f) != 1) v = 0;
            fclose(f);
        }
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        v ^= ((uint64_t)ts.tv_sec * 1000000007ull + (uint64_t)ts.tv_nsec) ^ ((uint64_t)(uintptr_t)&nonce << 17) ^ (uint64_t)getpid();
        return v ? v : 1ull;
    }();
    return nonce;
}

int p2p_self_check(P2P *p);
int p2p_checked_first_exchange(P2P *p);
int p2p_set_timeout_ms(P2P *p, uint64_t ms);
int p2p_allreduce(P2P *p, const double *send, double *recv, size_t count, hipStream_t st, const EmState *state);
int p2p_check(P2P *p, hipStream_t st);

int p2p_create(int rank, int n_ranks, int device, P2P **out)
{
    *out = nullptr;
    if (n_ranks > kP2PMaxRanks) return fail(OEM_ERR_ARG, "peer-to-peer exchange: at most %d ranks", kP2PMaxRanks);
    P2P *p = new (std::nothrow) P2P();
    if (!p) return fail(OEM_ERR_OOM, "peer-to-peer exchange: host allocation failed");
    p->rank = rank;
    p->n_ranks = n_ranks;
    p->device = device;
    *out = p;
    return OEM_OK;
}

void p2p_destroy(P2P *p)
{
    if (!p) return;
    hipSetDevice(p->device);
    hipDeviceSynchronize();
    for (int r = 0; r < kP2PMaxRanks; ++r)
        if (p->opened[r]) hipIpcCloseMemHandle(p->opened[r]);
    hipFree(p->self);
    hipFree(p->ctl);
    if (p->h_ctl) hipHostFree(p->h_ctl);
    delete p;
}

bool p2p_ready(const P2P *p) { return p && p->connected; }
uint64_t p2p_capacity(const P2P *p) { return p ? p->capacity : 0; }

int p2p_export(P2P *p, uint64_t capacity, void *out_blob)
{
    if (!p || !out_blob || capacity == 0) return fail(OEM_ERR_ARG, "oem_comm_p2p_export: bad argument");
    if (p->self) return fail(OEM_ERR_STATE, "oem_comm_p2p_export: already exported");
    OEM_HIP(hipSetDevice(p->device));
    const size_t bytes = sizeof(P2PShared) + 4 * capacity * sizeof(double); // partials and reduced slices, two parities each
    // FINE-GRAINED device memory: the region is written by peer devices and polled by this one while kernels
    // run, which is what fine-grained coherence is specified for (ordinary hipMalloc memory is coarse-grained:
    // coherent with other agents only at kernel boundaries -- it happened to work with several processes on one
    // device, where there is one L2 hierarchy, and is kept as the fall-back for a runtime that refuses the flag
    // or cannot export such an allocation).  Every access is a system-scope access either way (sys_store / sys_load).
    // (hipDeviceMallocUncached was tried in round 3: every access then goes to memory one lane at a time -- a
    // 200 k-entry exchange of ONE rank with itself took 25 us, profiles/r03_notes.md.)
    void *mem = nullptr;
    hipIpcMemHandle_t h;
    std::memset(&h, 0, sizeof(h));
    p->fine_grained = hipExtMallocWithFlags(&mem, bytes, hipDeviceMallocFinegrained) == hipSuccess && mem;
    if (p->fine_grained && p->n_ranks > 1 && hipIpcGetMemHandle(&h, mem) != hipSuccess) {
        hipFree(mem);
        mem = nullptr;
        p->fine_grained = false;
    }
    (void)hipGetLastError();
    if (!p->fine_grained) {
        OEM_HIP(hipMalloc(&mem, bytes));
        if (p->n_ranks > 1) {
            hipError_t e = hipIpcGetMemHandle(&h, mem);
            if (e != hipSuccess) {
                hipFree(mem);
                return fail(OEM_ERR_HIP, "hipIpcGetMemHandle: %s (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", hipGetErrorString(e));
            }
        }
    }
    OEM_HIP(hipMemset(mem, 0, bytes));
    p->self = static_cast<P2PShared *>(mem);
    p->capacity = capacity;
    OEM_HIP(hipMalloc((void **)&p->ctl, sizeof(P2PCtl)));
    OEM_HIP(hipMemset(p->ctl, 0, sizeof(P2PCtl)));
    OEM_HIP(hipHostMalloc((void **)&p->h_ctl, sizeof(P2PCtl), hipHostMallocDefault));
    OEM_HIP(hipDeviceSynchronize());
    P2PBlob b;
    std::memset(&b, 0, sizeof(b));
    b.magic = kP2PMagic;
    b.version = 1;
    b.pid = (uint64_t)getpid();
    b.ptr = (uint64_t)(uintptr_t)mem;
    b.capacity = capacity;
    b.device = p->device;
    b.rank = p->rank;
    b.handle = h;
    b.nonce = process_nonce();
    std::memcpy(out_blob, &b, sizeof(b));
    return OEM_OK;
}

int p2p_connect(P2P *p, const void *all_blobs)
{
    if (!p || !all_blobs) return fail(OEM_ERR_ARG, "oem_comm_p2p_connect: bad argument");
    if (!p->self) return fail(OEM_ERR_STATE, "oem_comm_p2p_connect: export first");
    if (p->connected) return fail(OEM_ERR_STATE, "oem_comm_p2p_connect: already connected");
    OEM_HIP(hipSetDevice(p->device));
    P2PCtl h;
    std::memset(&h, 0, sizeof(h));
    for (int r = 0; r < p->n_ranks; ++r) {
        P2PBlob b;
        std::memcpy(&b, static_cast<const char *>(all_blobs) + (size_t)r * sizeof(P2PBlob), sizeof(b));
        if (b.magic != kP2PMagic || b.version != 1 || b.rank != r)
            return fail(OEM_ERR_ARG, "oem_comm_p2p_connect: handle %d is not rank %d's export", r, r);
        if (b.capacity != p->capacity)
            return fail(OEM_ERR_ARG, "oem_comm_p2p_connect: rank %d exported %llu doubles, this rank %llu", r,
                        (unsigned long long)b.capacity, (unsigned long long)p->capacity);
        if (r == p->rank) {
            h.peer[r] = p->self;
        } else if (b.pid == (uint64_t)getpid() && b.nonce == process_nonce()) {
            // a rank of this very process (ranks as threads): one address space, no handle to open
            if (b.device != p->device) {
                int can = 0;
                OEM_HIP(hipDeviceCanAccessPeer(&can, p->device, b.device));
                if (!can) return fail(OEM_ERR_STATE, "device %d cannot access device %d", p->device, b.device);
                hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                    return fail(OEM_ERR_HIP, "hipDeviceEnablePeerAccess(%d): %s", b.device, hipGetErrorString(e));
                (void)hipGetLastError();
            }
            h.peer[r] = reinterpret_cast<P2PShared *>((uintptr_t)b.ptr);
        } else {
            void *m = nullptr;
            hipError_t e = hipIpcOpenMemHandle(&m, b.handle, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess)
                return fail(OEM_ERR_HIP, "hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
            p->opened[r] = m;
            h.peer[r] = static_cast<P2PShared *>(m);
        }
    }
    h.timeout_ticks = (long long)p->timeout_ms * kP2PTicksPerMs;
    OEM_HIP(hipMemcpy(p->ctl, &h, sizeof(h), hipMemcpyHostToDevice));
    p->connected = true;
    if (p->self_check) return p2p_checked_first_exchange(p);
    return OEM_OK;
}

// The checked first exchange with the rendezvous' long wait: the longer of the caller's bound (OEM_COMM_OPT_P2P_TIMEOUT_MS)
// and kP2PSelfCheckTimeoutMs.  Run at the end of connect (OEM_COMM_OPT_P2P_SELF_CHECK = 1 set before it), or on its
// own once the host knows that EVERY rank has mapped its peers (= 2 after connect: a rank whose hipIpcOpenMemHandle
// failed no longer leaves the others spinning in a kernel for the whole rendezvous bound).  A failure leaves the
// exchange disconnected and its error latch cleared.
int p2p_checked_first_exchange(P2P *p)
{
    if (!p2p_ready(p)) return fail(OEM_ERR_STATE, "peer-to-peer self check: the exchange is not connected");
    OEM_HIP(hipSetDevice(p->device));
    const uint64_t bound = p->timeout_ms > kP2PSelfCheckTimeoutMs ? p->timeout_ms : kP2PSelfCheckTimeoutMs;
    const long long ticks = (long long)bound * kP2PTicksPerMs;
    OEM_HIP(hipMemcpy(&p->ctl->timeout_ticks, &ticks, sizeof(ticks), hipMemcpyHostToDevice));
    const int rc = p2p_self_check(p);
    if (rc != OEM_OK) {
        p->connected = false; // the communicator falls back to RCCL (or reports that it has no backend)
        const uint32_t zero = 0;
        (void)hipMemcpy(&p->ctl->error, &zero, sizeof(zero), hipMemcpyHostToDevice);
        return rc;
    }
    return p2p_set_timeout_ms(p, p->timeout_ms);
}

int p2p_set_timeout_ms(P2P *p, uint64_t ms)
{
    if (!p) return OEM_OK;
    if (ms == 0 || ms > 3600000) return fail(OEM_ERR_ARG, "peer-to-peer exchange: timeout of %llu ms (1 .. 3 600 000)", (unsigned long long)ms);
    p->timeout_ms = ms;
    if (p->ctl) {
        OEM_HIP(hipSetDevice(p->device));
        const long long ticks = (long long)ms * kP2PTicksPerMs;
        OEM_HIP(hipMemcpy(&p->ctl->timeout_ticks, &ticks, sizeof(ticks), hipMemcpyHostToDevice));
    }
    return OEM_OK;
}
void p2p_set_self_check(P2P *p, bool on) { if (p) p->self_check = on; }
bool p2p_fine_grained(const P2P *p) { return p && p->fine_grained; }

// The first exchange, checked: every rank contributes a vector whose sum over the ranks is known in closed form,
// in both shapes, and compares what it reads back on the host.  It is also the ranks' rendezvous -- the wait of
// this exchange is long (a peer may still be creating its store), the waits of the EM loop are short.  Run by
// oem_comm_p2p_connect when OEM_COMM_OPT_P2P_SELF_CHECK is set (every rank is inside connect at the same time when
// the ranks are processes; ranks that are threads connected one after another by ONE thread must leave it off).
int p2p_self_check(P2P *p)
{
    const uint64_t n = p->capacity < 4096 ? p->capacity : 4096;
    std::vector<double> h(n);
    for (uint64_t i = 0; i < n; ++i) h[i] = (double)(p->rank + 1) * 0.5 + (double)(i % 977) * (double)(p->rank + 3);
    double *d = nullptr;
    OEM_HIP(hipMalloc((void **)&d, n * sizeof(double)));
    int rc = OEM_OK;
    const int saved = p->shape;
    for (int shape = 1; shape <= 2 && rc == OEM_OK; ++shape) {
        p->shape = shape;
        if (hipMemcpy(d, h.data(), n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { rc = fail(OEM_ERR_HIP, "peer-to-peer self check: upload failed"); break; }
        rc = p2p_allreduce(p, d, d, n, nullptr, nullptr);
        if (rc != OEM_OK) break;
        if (hipStreamSynchronize(nullptr) != hipSuccess) { rc = fail(OEM_ERR_HIP, "peer-to-peer self check: the exchange kernels failed"); break; }
        rc = p2p_check(p, nullptr);
        if (rc != OEM_OK) break;
        std::vector<double> got(n);
        if (hipMemcpy(got.data(), d, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(OEM_ERR_HIP, "peer-to-peer self check: read-back failed"); break; }
        const double N = (double)p->n_ranks;
        for (uint64_t i = 0; i < n; ++i) {
            // sum over r of (r + 1) / 2 + (i % 977) (r + 3): exact in f64 (small integers and halves)
            const double want = 0.5 * N * (N + 1.0) / 2.0 + (double)(i % 977) * (N * (N - 1.0) / 2.0 + 3.0 * N);
            if (got[i] != want) {
                rc = fail(OEM_ERR_STATE, "peer-to-peer self check (%s): element %llu is %.17g, the sum over %d ranks is %.17g",
                          shape == 1 ? "one-shot" : "two-phase", (unsigned long long)i, got[i], p->n_ranks, want);
                break;
            }
        }
    }
    p->shape = saved;
    hipFree(d);
    return rc;
}

// Two-phase costs one more flag round and one more small kernel (+6 us measured with 3-4 ranks on one device)
// and saves (1 - 2 / N) of a vector per link: at ~50 GB/s per xGMI link that pays from about half a megabyte.
bool p2p_two_phase(const P2P *p, uint64_t count)
{
    return p->shape == 2 || (p->shape == 0 && p->n_ranks >= 3 && count * sizeof(double) >= (512u << 10));
}
void p2p_set_shape(P2P *p, int shape) { if (p) p->shape = shape; }

// recv = sum over ranks of send (in place allowed); counts beyond the slot capacity go in pieces
int p2p_allreduce(P2P *p, const double *send, double *recv, size_t count, hipStream_t st, const EmState *state)
{
    if (!p2p_ready(p)) return fail(OEM_ERR_STATE, "peer-to-peer exchange is not connected");
    for (size_t off = 0; off < count; off += p->capacity) {
        const uint64_t n = count - off < p->capacity ? count - off : p->capacity;
        const bool two = p2p_two_phase(p, n);
        const int grid = grid_for_count(n, kP2PBlock, 128);
        hipLaunchKernelGGL(k_p2p_publish, dim3(grid), dim3(kP2PBlock), 0, st, send + off, p->ctl, p->capacity, n, p->rank,
                           p->n_ranks, state);
        const int rgrid = grid_for_count((n + kP2PBatch - 1) / kP2PBatch, kP2PBlock, 128);
        if (two) {
            const uint64_t slice = n / (uint64_t)p->n_ranks + 1;
            hipLaunchKernelGGL(k_p2p_reduce_slice, dim3(grid_for_count((slice + kP2PBatch - 1) / kP2PBatch, kP2PBlock, 64)),
                               dim3(kP2PBlock), 0, st, p->ctl, p->capacity, n, p->rank, p->n_ranks, state);
            hipLaunchKernelGGL(k_p2p_reduce<true>, dim3(rgrid), dim3(kP2PBlock), 0, st, recv + off, p->ctl, p->capacity, n,
                               p->rank, p->n_ranks, state);
        } else {
            hipLaunchKernelGGL(k_p2p_reduce<false>, dim3(rgrid), dim3(kP2PBlock), 0, st, recv + off, p->ctl, p->capacity, n,
                               p->rank, p->n_ranks, state);
        }
    }
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

// publish this rank's partial `curr`, then rel-diff / swap / clear / stopping rule on the reduced vector
int p2p_reldiff(P2P *p, double *prev, double *curr, EmState *state, EmParams prm, hipStream_t st)
{
    if (!p2p_ready(p) || prm.n_txps > p->capacity) return fail(OEM_ERR_STATE, "peer-to-peer exchange: not connected / too small");
    const uint64_t n = prm.n_txps;
    const int grid = grid_for_count(n, kP2PBlock, 128);
    hipLaunchKernelGGL(k_p2p_publish, dim3(grid), dim3(kP2PBlock), 0, st, curr, p->ctl, p->capacity, n, p->rank, p->n_ranks,
                       state);
    constexpr int kRB = 1024;
    const int rgrid = grid_for_count((n + kP2PBatch - 1) / kP2PBatch, kRB, 64);
    if (p2p_two_phase(p, n)) {
        const uint64_t slice = n / (uint64_t)p->n_ranks + 1;
        hipLaunchKernelGGL(k_p2p_reduce_slice, dim3(grid_for_count((slice + kP2PBatch - 1) / kP2PBatch, kP2PBlock, 64)),
                           dim3(kP2PBlock), 0, st, p->ctl, p->capacity, n, p->rank, p->n_ranks, state);
        hipLaunchKernelGGL((k_p2p_reldiff<kRB, true>), dim3(rgrid), dim3(kRB), 0, st, prev, curr, state, prm, p->ctl,
                           p->capacity, p->rank, p->n_ranks);
    } else {
        hipLaunchKernelGGL((k_p2p_reldiff<kRB, false>), dim3(rgrid), dim3(kRB), 0, st, prev, curr, state, prm, p->ctl,
                           p->capacity, p->rank, p->n_ranks);
    }
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

// after a stream synchronize: did a wait time out?
int p2p_check(P2P *p, hipStream_t st)
{
    if (!p2p_ready(p)) return OEM_OK;
    OEM_HIP(hipMemcpyAsync(p->h_ctl, p->ctl, sizeof(P2PCtl), hipMemcpyDeviceToHost, st));
    OEM_HIP(hipStreamSynchronize(st));
    if (p->h_ctl->error)
        return fail(OEM_ERR_STATE, "peer-to-peer exchange: a rank did not arrive within %.1f s (rank %d waited)",
                    (double)p->h_ctl->timeout_ticks / (double)(kP2PTicksPerMs * 1000), p->rank);
    return OEM_OK;
}

} // namespace oem
