// oem_testing.hip -- hooks of the TEST-ONLY library (liboarfish_em_testing.so, -DOEM_TESTING).
//
// Nothing in this file is part of the product (liboarfish_em.so is linked without it) or of the
// public header.  tests/ and scripts/ use it to look inside a resident store (layout hashes), to
// hammer the stopping-rule kernel in isolation, and -- together with the OEM_TESTING build of
// oem_comm.cpp -- to run the row-sharded loop with several shards on one GPU.
#include <cstring>
#include <vector>

#include "oem_internal.h"

using namespace oem;

// Test hook (not in the public header): 64-bit hashes of the resident tiled layout, so that the
// device-built layout can be checked element for element against the host-built one.
// out[0..3] = n_tiles, n_rows, n_local, n_remote; out[4..13] = tiles, perm, codes, w, r_tid, r_w,
// r_row, r_slot, q_dst, bucket_base; out[14] (if asked for) = 1 when the device built it; returns OEM_ERR_STATE when the store has no tiled layout.
extern "C" int oem_debug_layout_hash(oem_store *s, uint64_t *out, uint32_t n_out)
{
    OEM_API_BEGIN
    if (!s || !out || n_out < 14) return fail(OEM_ERR_ARG, "oem_debug_layout_hash: bad argument");
    std::lock_guard<std::mutex> lk(s->mu);
    OEM_HIP(hipSetDevice(s->device));
    const DeviceTiled &t = s->tiled;
    if (!t.present) return fail(OEM_ERR_STATE, "oem_debug_layout_hash: no tiled layout");
    auto hash_dev = [&](const void *d, size_t bytes, uint64_t *h) -> int {
        std::vector<uint64_t> buf((bytes + 7) / 8, 0);
        if (bytes) OEM_HIP(hipMemcpy(buf.data(), d, bytes, hipMemcpyDeviceToHost));
        uint64_t x = 0x9e3779b97f4a7c15ull ^ bytes;
        for (uint64_t v : buf) { x ^= v; x *= 0xff51afd7ed558ccdull; x ^= x >> 29; }
        *h = x;
        return OEM_OK;
    };
    // array lengths follow from the descriptors: slices and remote records end with the last tile
    uint64_t w_slots = 0, c_slots = 0;
    if (t.n_tiles) {
        TileDesc last;
        OEM_HIP(hipMemcpy(&last, t.tiles + (t.n_tiles - 1), sizeof(last), hipMemcpyDeviceToHost));
        w_slots = last.w_base; c_slots = last.c_base;
        for (uint32_t i = 0; i < kTileSlices; ++i) { w_slots += last.width[i]; c_slots += (last.width[i] + 1u) / 2; }
    }
    out[0] = t.n_tiles; out[1] = t.n_rows; out[2] = t.n_local; out[3] = t.n_remote;
    const size_t wsz = s->csr.w_is_f64 ? 8 : 4;
    OEM_TRY(hash_dev(t.tiles, sizeof(TileDesc) * t.n_tiles, &out[4]));
    OEM_TRY(hash_dev(t.perm, 4 * t.n_rows, &out[5]));
    OEM_TRY(hash_dev(t.codes, 4 * (c_slots + 1) * 64, &out[6]));
    OEM_TRY(hash_dev(s->csr.w_is_f64 ? (const void *)t.w64 : (const void *)t.w32, wsz * (w_slots + 1) * 64, &out[7]));
    if (!t.r_tid || !t.r_row || !t.r_slot)
        return fail(OEM_ERR_STATE, "oem_debug_layout_hash: the builders' remote streams were dropped (set OEM_KEEP_UNPACKED=1)");
    OEM_TRY(hash_dev(t.r_tid, 4 * t.n_remote, &out[8]));
    OEM_TRY(hash_dev(s->csr.w_is_f64 ? (const void *)t.r_w64 : (const void *)t.r_w32, wsz * t.n_remote, &out[9]));
    OEM_TRY(hash_dev(t.r_row, 2 * t.n_remote, &out[10]));
    OEM_TRY(hash_dev(t.r_slot, 4 * t.n_remote, &out[11]));
    OEM_TRY(hash_dev(t.q_dst, 2 * t.n_remote, &out[12]));
    OEM_TRY(hash_dev(t.bucket_base, 4 * ((size_t)t.n_buckets + 1), &out[13]));
    if (n_out > 14) out[14] = t.built_on_device ? 1 : 0;
    if (n_out > 17) { // the slim form the kernels read (oem_layout_pack.hip)
        OEM_TRY(hash_dev(t.sd, 4 * (size_t)t.n_sd, &out[15]));
        out[16] = 0;
        if (t.packed) OEM_TRY(hash_dev(t.r_pk, 4 * t.n_remote, &out[16]));
        out[17] = t.packed ? 1 : 0;
    }
    return OEM_OK;
    OEM_API_END("oem_debug_layout_hash")
}


// Test hook: what oem::knob() returns in THIS library (the environment variable in the test-only build; the
// product's knob() is compiled without getenv and returns its default, checked on the object file).
extern "C" long oem_debug_knob(const char *name, long dflt) { return oem::knob(name, dflt); }

// ---------------------------------------------------------------------------
// Stress test of k_reldiff_swap_clear's last-block election (oem_kernels.hip): the stopping
// decision of every EM run (em.rs:194-218) is taken by the workgroup that draws the last ticket,
// from a running maximum the other workgroups published with device-scope atomics just before
// taking theirs.  Each launch gets a fresh (prev, curr) pair whose rel-diff maximum is known
// exactly -- one planted element at a pseudo-random position, (curr - prev) / prev =
// 0.75 + launch * 2^-20, every other element strictly below 0.5 -- and the decision workgroup's
// view of it (EmState::last_rel) is recorded after every launch.  out_last_rel[i] must equal the
// planted value bit for bit; a stale or partial maximum shows up as a smaller number.
// ---------------------------------------------------------------------------
namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void k_stress_case(double *__restrict__ prev, double *__restrict__ curr, uint32_t n,
                                                     uint32_t launch, uint32_t seed)
{
    const uint32_t pos = mix32(seed ^ (launch * 0x9e3779b9u)) % n;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        prev[i] = 1.0;
        const double below = (double)(mix32(i * 2654435761u + launch) & 0xffffu) * (0.5 / 65536.0);
        curr[i] = i == pos ? 1.75 + (double)launch * (1.0 / 1048576.0) : 1.0 + below;
    }
}

__global__ void k_stress_record(const EmState *state, double *out, uint32_t launch) { out[launch] = state->last_rel; }

} // namespace

extern "C" int oem_test_reldiff_stress(uint32_t n_txps, uint32_t n_launches, uint32_t seed, int device,
                                       double *out_last_rel /* n_launches */)
{
    OEM_API_BEGIN
    if (!n_txps || !n_launches || !out_last_rel) return fail(OEM_ERR_ARG, "oem_test_reldiff_stress: bad argument");
    OEM_HIP(hipSetDevice(device));
    oem_store s; // only the stream is used by the launcher
    s.device = device;
    OEM_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    double *prev = nullptr, *curr = nullptr, *rec = nullptr;
    EmState *st = nullptr;
    int rc = OEM_OK;
    do {
        if (hipMalloc((void **)&prev, sizeof(double) * n_txps) != hipSuccess ||
            hipMalloc((void **)&curr, sizeof(double) * n_txps) != hipSuccess ||
            hipMalloc((void **)&rec, sizeof(double) * n_launches) != hipSuccess ||
            hipMalloc((void **)&st, sizeof(EmState)) != hipSuccess ||
            hipMemsetAsync(st, 0, sizeof(EmState), s.stream) != hipSuccess) {
            rc = fail(OEM_ERR_OOM, "oem_test_reldiff_stress: device allocation failed");
            break;
        }
        EmParams p{n_txps, 0xffffffffu, 0xffffffffu, -1.0}; // never stops: rel_diff >= 0 is never < -1
        uint32_t g = (n_txps + 255) / 256;
        if (g > 1024) g = 1024;
        for (uint32_t i = 0; i < n_launches && rc == OEM_OK; ++i) {
            hipLaunchKernelGGL(k_stress_case, dim3(g), dim3(256), 0, s.stream, prev, curr, n_txps, i, seed);
            rc = launch_reldiff_swap_clear(&s, prev, curr, st, p);
            hipLaunchKernelGGL(k_stress_record, dim3(1), dim3(1), 0, s.stream, st, rec, i);
        }
        if (rc != OEM_OK) break;
        if (hipMemcpyAsync(out_last_rel, rec, sizeof(double) * n_launches, hipMemcpyDeviceToHost, s.stream) != hipSuccess ||
            hipStreamSynchronize(s.stream) != hipSuccess)
            rc = fail(OEM_ERR_HIP, "oem_test_reldiff_stress: read-back failed");
    } while (false);
    hipFree(prev); hipFree(curr); hipFree(rec); hipFree(st);
    hipStreamDestroy(s.stream);
    s.stream = nullptr;
    return rc;
    OEM_API_END("oem_test_reldiff_stress")
}
