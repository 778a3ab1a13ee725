// oem_comm.cpp -- communicator over the row shards of one node: RCCL, and the one-shot peer-to-peer
// exchange of oem_p2p.hip for the small (latency-bound) count vector.
//
// One process per GPU; each process owns one row shard of the alignment store
// and the only exchange of the path is the sum of the n_txps partial counts
// per E/M pass (in the reference this is the shared
// Vec<AtomicF64> of em.rs:338-341).  RCCL is loaded lazily with dlopen so a
// single-GPU user never pays for (or needs) librccl: if the host process has
// already loaded an RCCL (PyTorch ships one with the same SONAME) that copy is
// reused, so there is exactly one RCCL per process.
#include <dlfcn.h>

#include <condition_variable>
#include <cstring>
#include <memory>
#include <vector>

#include "oem_internal.h"

namespace oem {

namespace {

// Minimal slice of the RCCL API (rccl.h); types reduced to what crosses here.
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
constexpr int kNcclSum = 0;     // ncclSum
constexpr int kNcclFloat64 = 8; // ncclFloat64 / ncclDouble

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_api;
std::mutex g_api_mu;

int load_rccl()
{
    std::lock_guard<std::mutex> lk(g_api_mu);
    if (g_api.handle) return OEM_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); // reuse the host's copy if there is one
        if (h) break;
    }
    if (!h) {
        for (const char *n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
    }
    if (!h) return fail(OEM_ERR_RCCL, "cannot load librccl: %s", dlerror());
    RcclApi a;
    a.handle = h;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.CommCount = (decltype(a.CommCount))dlsym(h, "ncclCommCount"); // (optional: oem_comm_info only)
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce)
        return fail(OEM_ERR_RCCL, "librccl lacks a required symbol");
    g_api = a;
    return OEM_OK;
}

const char *nccl_err(ncclResult_t r)
{
    return g_api.GetErrorString ? g_api.GetErrorString(r) : "rccl error";
}

} // namespace

#ifdef OEM_TESTING
// Test backend (oem_debug_local_comm_create): the ranks are threads of ONE process that share a
// GPU, and the "collective" is a host-side rendezvous plus one summing kernel.  It exists so that the
// native sharded loop (global read count, per-pass exchange, identical stopping decision, sharded
// bootstrap) runs with several real shards on a 1-GPU box, where RCCL refuses two ranks per device.
struct LocalGroup {
    std::mutex mu;
    std::condition_variable cv;
    int n = 0, arrived = 0;
    uint64_t generation = 0;
    std::vector<const double *> send;
    std::vector<double *> recv;
    double *tmp = nullptr;
    size_t tmp_count = 0;
    ~LocalGroup() { hipFree(tmp); }
    void barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t g = generation;
        if (++arrived == n) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != g; });
    }
};

__global__ void k_local_sum(const double *const *send, int n, double *out, size_t count)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int r = 0; r < n; ++r) s += send[r][i];
        out[i] = s;
    }
}

#endif // OEM_TESTING

struct P2P; // oem_p2p.hip
int p2p_create(int rank, int n_ranks, int device, P2P **out);
void p2p_destroy(P2P *p);
bool p2p_ready(const P2P *p);
uint64_t p2p_capacity(const P2P *p);
int p2p_export(P2P *p, uint64_t capacity, void *out_blob);
int p2p_connect(P2P *p, const void *all_blobs);
int p2p_allreduce(P2P *p, const double *send, double *recv, size_t count, hipStream_t st, const EmState *state);
int p2p_reldiff(P2P *p, double *prev, double *curr, EmState *state, EmParams prm, hipStream_t st);
int p2p_check(P2P *p, hipStream_t st);
void p2p_set_shape(P2P *p, int shape);
int p2p_set_timeout_ms(P2P *p, uint64_t ms);
void p2p_set_self_check(P2P *p, bool on);
int p2p_checked_first_exchange(P2P *p);
bool p2p_fine_grained(const P2P *p);

// Vectors up to this size take the peer-to-peer exchange when it is connected (latency-bound: every
// rank reads N - 1 partials over its own links at once); larger ones (the batched bootstrap's
// 2 * T * 4 counts) are bandwidth-bound and go to RCCL when the communicator has one.
constexpr size_t kP2PMaxBytes = 4u << 20;

struct Comm {
    ncclComm_t comm = nullptr;
    P2P *p2p = nullptr;
    size_t p2p_max_bytes = kP2PMaxBytes; // OEM_COMM_OPT_P2P_MAX_BYTES (0: RCCL only)
    int p2p_shape = 0;                   // OEM_COMM_OPT_P2P_SHAPE (kept here too: it may be set before the export)
    uint64_t p2p_timeout_ms = 0;         // OEM_COMM_OPT_P2P_TIMEOUT_MS (0: the default)
    bool p2p_self_check = false;         // OEM_COMM_OPT_P2P_SELF_CHECK
    int rank = 0;
    int n_ranks = 1;
    int device = 0;
#ifdef OEM_TESTING
    std::shared_ptr<LocalGroup> local; // test-only library: ranks are threads of one process
#endif
};

#ifdef OEM_TESTING
static int local_allreduce(Comm *c, const double *send, double *recv, size_t count, hipStream_t st)
{
    LocalGroup &g = *c->local;
    OEM_HIP(hipStreamSynchronize(st)); // this rank's partial counts are complete
    g.send[c->rank] = send;
    g.recv[c->rank] = recv;
    g.barrier();
    int rc = OEM_OK;
    if (c->rank == 0) {
        do {
            if (g.tmp_count < count) {
                hipFree(g.tmp);
                g.tmp = nullptr;
                if (hipMalloc((void **)&g.tmp, count * sizeof(double)) != hipSuccess) { rc = fail(OEM_ERR_OOM, "local comm: scratch"); break; }
                g.tmp_count = count;
            }
            const double **d_ptrs = nullptr;
            if (hipMalloc((void **)&d_ptrs, sizeof(double *) * g.n) != hipSuccess) { rc = fail(OEM_ERR_OOM, "local comm: pointers"); break; }
            hipMemcpy(d_ptrs, g.send.data(), sizeof(double *) * g.n, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_local_sum, dim3(256), dim3(256), 0, st, (const double *const *)d_ptrs, g.n, g.tmp, count);
            for (int r = 0; r < g.n; ++r) hipMemcpyAsync(g.recv[r], g.tmp, count * sizeof(double), hipMemcpyDeviceToDevice, st);
            if (hipStreamSynchronize(st) != hipSuccess) rc = fail(OEM_ERR_HIP, "local comm: sum failed");
            hipFree(d_ptrs);
        } while (false);
    }
    g.barrier(); // every rank's recv is written
    return rc;
}
#endif // OEM_TESTING

static bool use_p2p(const Comm *c, size_t count)
{
    return c && p2p_ready(c->p2p) && (!c->comm || count * sizeof(double) <= c->p2p_max_bytes);
}

// `state` (optional): the launches of a finished run skip the exchange where that costs nothing -- the
// peer-to-peer kernels test it on the device, identically on every rank; RCCL calls are issued regardless.
int comm_allreduce_sum_f64(Comm *c, const double *send, double *recv, size_t count, hipStream_t st, const EmState *state)
{
#ifdef OEM_TESTING
    if (c && c->local) return local_allreduce(c, send, recv, count, st);
#endif
    if (use_p2p(c, count)) return p2p_allreduce(c->p2p, send, recv, count, st, state);
    if (!c || !c->comm) { // no exchange partner
        if (c && c->n_ranks > 1) return fail(OEM_ERR_STATE, "communicator of %d ranks has no connected backend", c->n_ranks);
        if (send != recv)
            OEM_HIP(hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, st));
        return OEM_OK;
    }
    ncclResult_t r = g_api.AllReduce(send, recv, count, kNcclFloat64, kNcclSum, c->comm, st);
    if (r != 0) return fail(OEM_ERR_RCCL, "ncclAllReduce: %s", nccl_err(r));
    return OEM_OK;
}

// The exchange of one loop iteration fused with rel-diff / swap / clear (peer-to-peer backend only).
bool comm_fuses_reldiff(const Comm *c, uint32_t n_txps)
{
#ifdef OEM_TESTING
    if (c && c->local) return false;
#endif
    return use_p2p(c, n_txps) && n_txps <= p2p_capacity(c->p2p);
}
int comm_reldiff_fused(Comm *c, double *prev, double *curr, EmState *state, EmParams prm, hipStream_t st)
{
    return p2p_reldiff(c->p2p, prev, curr, state, prm, st);
}
// true when a launch of a finished run costs a real collective (RCCL): the host then looks at the state more often
bool comm_exchange_is_unconditional(const Comm *c, uint32_t n_txps)
{
#ifdef OEM_TESTING
    if (c && c->local) return true;
#endif
    return c && c->comm && !use_p2p(c, n_txps);
}
// after a stream synchronize: a peer-to-peer wait that timed out is reported, not hung on
int comm_check(Comm *c, hipStream_t st)
{
    if (!c || !c->p2p) return OEM_OK;
    if (c->comm && c->p2p_max_bytes == 0) return OEM_OK; // switched to RCCL alone (e.g. after a failed self check)
    return p2p_check(c->p2p, st);
}

int comm_rank(const Comm *c) { return c ? c->rank : 0; }
int comm_size(const Comm *c) { return c ? c->n_ranks : 1; }
#ifdef OEM_TESTING
bool comm_exchanges(const Comm *c) { return c && (c->comm || c->local || p2p_ready(c->p2p)); }
#else
bool comm_exchanges(const Comm *c) { return c && (c->comm || p2p_ready(c->p2p)); }
#endif

} // namespace oem

using namespace oem;

static_assert(sizeof(ncclUniqueId) == OEM_UNIQUE_ID_BYTES, "unique id size");

extern "C" int oem_comm_unique_id(void *out_id)
{
    OEM_API_BEGIN
    if (!out_id) return fail(OEM_ERR_ARG, "oem_comm_unique_id: out_id is NULL");
    OEM_TRY(load_rccl());
    ncclUniqueId id;
    ncclResult_t r = g_api.GetUniqueId(&id);
    if (r != 0) return fail(OEM_ERR_RCCL, "ncclGetUniqueId: %s", nccl_err(r));
    std::memcpy(out_id, &id, sizeof(id));
    return OEM_OK;
    OEM_API_END("oem_comm_unique_id")
}

extern "C" int oem_comm_create(const void *unique_id, int rank, int n_ranks, int device,
                               oem_comm **out)
{
    OEM_API_BEGIN
    if (!out) return fail(OEM_ERR_ARG, "oem_comm_create: out is NULL");
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return fail(OEM_ERR_ARG, "oem_comm_create: rank %d of %d", rank, n_ranks);
    Comm *c = new (std::nothrow) Comm();
    if (!c) return fail(OEM_ERR_OOM, "oem_comm_create: host allocation failed");
    c->rank = rank;
    c->n_ranks = n_ranks;
    c->device = device;
    // n_ranks == 1 with a unique id still builds a real RCCL communicator (RCCL accepts one rank):
    // the single-GPU self test of the dlopen'ed entry points.  n_ranks == 1 without one is a no-op.
    // n_ranks > 1 without one is a communicator that will exchange peer to peer only
    // (oem_comm_p2p_export / _connect): no RCCL is loaded.
    if (unique_id) {
        int rc = load_rccl();
        if (rc != OEM_OK) { delete c; return rc; }
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) {
            delete c;
            return fail(OEM_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
        }
        ncclUniqueId id;
        std::memcpy(&id, unique_id, sizeof(id));
        ncclResult_t r = g_api.CommInitRank(&c->comm, n_ranks, id, rank);
        if (r != 0) { delete c; return fail(OEM_ERR_RCCL, "ncclCommInitRank: %s", nccl_err(r)); }
    }
    *out = reinterpret_cast<oem_comm *>(c);
    return OEM_OK;
    OEM_API_END("oem_comm_create")
}

#ifdef OEM_TESTING
// Test hook (not in the public header): n_ranks communicators of one process-local group.
extern "C" int oem_debug_local_comm_create(int n_ranks, int device, oem_comm **out /* [n_ranks] */)
{
    OEM_API_BEGIN
    if (n_ranks < 1 || !out) return fail(OEM_ERR_ARG, "oem_debug_local_comm_create: bad argument");
    auto g = std::make_shared<LocalGroup>();
    g->n = n_ranks;
    g->send.assign(n_ranks, nullptr);
    g->recv.assign(n_ranks, nullptr);
    for (int r = 0; r < n_ranks; ++r) {
        Comm *c = new (std::nothrow) Comm();
        if (!c) return fail(OEM_ERR_OOM, "oem_debug_local_comm_create: host allocation failed");
        c->rank = r;
        c->n_ranks = n_ranks;
        c->device = device;
        c->local = g;
        out[r] = reinterpret_cast<oem_comm *>(c);
    }
    return OEM_OK;
    OEM_API_END("oem_debug_local_comm_create")
}
#endif // OEM_TESTING

extern "C" int oem_comm_p2p_export(oem_comm *comm, uint64_t capacity, void *out_handle)
{
    OEM_API_BEGIN
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || !out_handle || capacity == 0) return fail(OEM_ERR_ARG, "oem_comm_p2p_export: bad argument");
    if (!c->p2p) OEM_TRY(p2p_create(c->rank, c->n_ranks, c->device, &c->p2p));
    p2p_set_shape(c->p2p, c->p2p_shape);
    p2p_set_self_check(c->p2p, c->p2p_self_check);
    if (c->p2p_timeout_ms) OEM_TRY(p2p_set_timeout_ms(c->p2p, c->p2p_timeout_ms));
    return p2p_export(c->p2p, capacity, out_handle);
    OEM_API_END("oem_comm_p2p_export")
}

extern "C" int oem_comm_p2p_connect(oem_comm *comm, const void *all_handles)
{
    OEM_API_BEGIN
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || !all_handles) return fail(OEM_ERR_ARG, "oem_comm_p2p_connect: bad argument");
    if (!c->p2p) return fail(OEM_ERR_STATE, "oem_comm_p2p_connect: call oem_comm_p2p_export first");
    return p2p_connect(c->p2p, all_handles);
    OEM_API_END("oem_comm_p2p_connect")
}

extern "C" int oem_comm_set_option(oem_comm *comm, uint32_t option, uint64_t value)
{
    OEM_API_BEGIN
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return fail(OEM_ERR_ARG, "oem_comm_set_option: comm is NULL");
    switch (option) {
    case OEM_COMM_OPT_P2P_MAX_BYTES: c->p2p_max_bytes = (size_t)value; return OEM_OK;
    case OEM_COMM_OPT_P2P_SHAPE:
        if (value > 2) return fail(OEM_ERR_ARG, "oem_comm_set_option: peer-to-peer shape is 0 (by rank count), 1 (one-shot) or 2 (two-phase)");
        c->p2p_shape = (int)value;
        p2p_set_shape(c->p2p, c->p2p_shape);
        return OEM_OK;
    case OEM_COMM_OPT_P2P_TIMEOUT_MS:
        if (value == 0 || value > 3600000) return fail(OEM_ERR_ARG, "oem_comm_set_option: timeout of 1 .. 3 600 000 ms");
        c->p2p_timeout_ms = value;
        return p2p_set_timeout_ms(c->p2p, value);
    case OEM_COMM_OPT_P2P_SELF_CHECK:
        if (value == 2) { // run the checked first exchange NOW (after connect, once every rank is known to be mapped)
            if (!c->p2p) return fail(OEM_ERR_STATE, "oem_comm_set_option: no peer-to-peer exchange to check");
            return p2p_checked_first_exchange(c->p2p);
        }
        if (value > 2) return fail(OEM_ERR_ARG, "oem_comm_set_option: self check is 0, 1 (at the end of connect) or 2 (now)");
        c->p2p_self_check = value != 0;
        p2p_set_self_check(c->p2p, c->p2p_self_check);
        return OEM_OK;
    default: return fail(OEM_ERR_ARG, "oem_comm_set_option: unknown option %u", option);
    }
    OEM_API_END("oem_comm_set_option")
}

extern "C" int oem_comm_info(const oem_comm *comm, uint32_t key, uint64_t *out)
{
    OEM_API_BEGIN
    const Comm *c = reinterpret_cast<const Comm *>(comm);
    if (!c || !out) return fail(OEM_ERR_ARG, "oem_comm_info: NULL argument");
    switch (key) {
    case OEM_COMM_INFO_RANKS: *out = (uint64_t)c->n_ranks; return OEM_OK;
    case OEM_COMM_INFO_RCCL_RANKS: {
        int n = 0;
        if (c->comm && g_api.CommCount) {
            const ncclResult_t r = g_api.CommCount(c->comm, &n);
            if (r != 0) return fail(OEM_ERR_RCCL, "ncclCommCount: %s", nccl_err(r));
        }
        *out = (uint64_t)n;
        return OEM_OK;
    }
    case OEM_COMM_INFO_P2P_CONNECTED: *out = c->p2p && p2p_ready(c->p2p) ? 1u : 0u; return OEM_OK;
    default: return fail(OEM_ERR_ARG, "oem_comm_info: unknown key %u", key);
    }
    OEM_API_END("oem_comm_info")
}

extern "C" void oem_comm_destroy(oem_comm *comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return;
    if (c->p2p) p2p_destroy(c->p2p);
    if (c->comm && g_api.CommDestroy) g_api.CommDestroy(c->comm);
    delete c;
}
