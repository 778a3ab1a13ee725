// oem_tile_kernels.hip -- the E/M pass over the tiled layout (oem_layout.h).
//
// Semantics: em.rs:87-133 (m_step).  For every read: denom = sum_j theta[t_j]*w_j;
// if denom > 1e-30, cnt[t_j] += c_i * theta[t_j]*w_j / denom, c_i the read's
// bootstrap multiplicity (1 on the point-estimate path).
//
// k_em_tile      one workgroup per tile: theta window -> LDS, reads one per lane
//                (SELL-64), local increments by LDS f64 atomics, remote increments
//                written to the bucket-major queue, window flushed with
//                cache-line-coalesced global atomics.
// k_remote_fold  streams each bucket's queue range into an LDS window, flushes.
//
// HBM-bound by design (no MFMA: this is sparse gather/scatter).  Algorithmic
// bytes per pass are SURVEY.md section 8d's nnz*(4+4) + (R+1)*4 + 2*T*8.
#include "oem_internal.h"

namespace oem {

namespace {

constexpr int kTileThreads = 256;
constexpr int kFoldThreads = 1024;

__device__ __forceinline__ void lds_add_f64(double *p, double v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // ds_add_f64
}

template <typename WT>
__global__ __launch_bounds__(kTileThreads) void k_em_tile(
    const TileDesc *__restrict__ tiles, const SliceDesc *__restrict__ slices,
    const uint32_t *__restrict__ codes, const WT *__restrict__ w,
    const uint32_t *__restrict__ r_tid, const WT *__restrict__ r_w,
    const uint16_t *__restrict__ r_row, const uint32_t *__restrict__ r_slot,
    double *__restrict__ queue, const double *__restrict__ theta, double *__restrict__ cnt,
    const EmState *state, const uint32_t *__restrict__ row_w_perm)
{
    if (state && state->done) return;

    __shared__ double theta_l[kWin];
    __shared__ double cnt_l[kWin];
    __shared__ double den_l[kTileRows]; // remote part of the denominators, then row_w/denom

    const TileDesc td = tiles[blockIdx.x];
    const uint32_t tx = threadIdx.x;

    for (uint32_t i = tx; i < td.win_len; i += kTileThreads) {
        theta_l[i] = theta[td.lo + i];
        cnt_l[i] = 0.0;
    }
    for (uint32_t i = tx; i < td.n_slices * 64; i += kTileThreads) den_l[i] = 0.0;
    __syncthreads();

    // remote alignments, phase A: x = theta[t]*w parked in the queue, added to the read's denominator
    for (uint32_t i = tx; i < td.remote_cnt; i += kTileThreads) {
        const uint32_t o = td.remote_begin + i;
        const double x = theta[r_tid[o]] * (double)r_w[o];
        queue[r_slot[o]] = x;
        lds_add_f64(&den_l[r_row[o]], x);
    }
    __syncthreads();

    // local alignments: one read per lane, one slice per wavefront at a time
    const uint32_t lane = tx & 63u, wave = tx >> 6;
    for (uint32_t s = wave; s < td.n_slices; s += kTileThreads / 64) {
        const SliceDesc sd = slices[td.slice_begin + s];
        const uint32_t rl = s * 64 + lane;
        const WT *wp = w + (size_t)sd.w_off * 64 + lane;
        const uint32_t *cp = codes + (size_t)sd.c_off * 64 + lane;
        double denom = den_l[rl];
        for (uint32_t j = 0; j < sd.width; j += 2) {
            const uint32_t cc = cp[(size_t)(j >> 1) * 64];
            denom += theta_l[cc & 0xffffu] * (double)wp[(size_t)j * 64];           // em.rs:111
            if (j + 1 < sd.width) denom += theta_l[cc >> 16] * (double)wp[(size_t)(j + 1) * 64];
        }
        double scale = 1.0;
        if (row_w_perm) scale = rl < td.n_rows ? (double)row_w_perm[td.row_base + rl] : 0.0;
        const double inv = denom > OEM_EM_DENOM_THRESH ? scale / denom : 0.0;      // em.rs:115
        den_l[rl] = inv;
        if (inv != 0.0) {
            for (uint32_t j = 0; j < sd.width; j += 2) {
                const uint32_t cc = cp[(size_t)(j >> 1) * 64];
                const double w0 = (double)wp[(size_t)j * 64];
                if (w0 != 0.0) lds_add_f64(&cnt_l[cc & 0xffffu], theta_l[cc & 0xffffu] * w0 * inv); // em.rs:128-129
                if (j + 1 < sd.width) {
                    const double w1 = (double)wp[(size_t)(j + 1) * 64];
                    if (w1 != 0.0) lds_add_f64(&cnt_l[cc >> 16], theta_l[cc >> 16] * w1 * inv);
                }
            }
        }
    }
    __syncthreads();

    // remote alignments, phase B: queue <- x * (c_i / denom_i)
    for (uint32_t i = tx; i < td.remote_cnt; i += kTileThreads) {
        const uint32_t o = td.remote_begin + i;
        const uint32_t q = r_slot[o];
        queue[q] = queue[q] * den_l[r_row[o]];
    }

    // flush the window: consecutive lanes -> consecutive addresses (coalesced atomics)
    for (uint32_t i = tx; i < td.win_len; i += kTileThreads) {
        const double v = cnt_l[i];
        if (v != 0.0) unsafeAtomicAdd(&cnt[td.lo + i], v);
    }
}

__global__ __launch_bounds__(kFoldThreads) void k_remote_fold(
    const uint32_t *__restrict__ bucket_base, const double *__restrict__ queue,
    const uint16_t *__restrict__ q_dst, double *__restrict__ cnt, const EmState *state,
    uint32_t n_groups, uint32_t n_txps)
{
    if (state && state->done) return;
    __shared__ double acc[kBucket];
    const uint32_t b = blockIdx.x / n_groups, g = blockIdx.x % n_groups;
    const uint32_t q0 = bucket_base[b], q1 = bucket_base[b + 1];
    const uint64_t span = q1 - q0;
    const uint32_t s0 = q0 + (uint32_t)(span * g / n_groups);
    const uint32_t s1 = q0 + (uint32_t)(span * (g + 1) / n_groups);
    if (s0 == s1) return;
    for (uint32_t i = threadIdx.x; i < kBucket; i += kFoldThreads) acc[i] = 0.0;
    __syncthreads();
    for (uint32_t o = s0 + threadIdx.x; o < s1; o += kFoldThreads) {
        const double v = queue[o];
        if (v != 0.0) lds_add_f64(&acc[q_dst[o]], v);
    }
    __syncthreads();
    const uint32_t base = b * kBucket;
    for (uint32_t i = threadIdx.x; i < kBucket && base + i < n_txps; i += kFoldThreads) {
        const double v = acc[i];
        if (v != 0.0) unsafeAtomicAdd(&cnt[base + i], v);
    }
}

__global__ __launch_bounds__(256) void k_permute_row_w(const uint32_t *__restrict__ row_w,
                                                       const uint32_t *__restrict__ perm,
                                                       uint32_t *__restrict__ out, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = row_w[perm[i]];
}

} // namespace

int launch_em_pass_tiled(oem_store *s, const double *theta, double *cnt, const EmState *state,
                         const uint32_t *row_w_perm)
{
    const DeviceTiled &t = s->tiled;
    if (t.n_tiles == 0) return OEM_OK;
    if (s->csr.w_is_f64) {
        hipLaunchKernelGGL((k_em_tile<double>), dim3(t.n_tiles), dim3(kTileThreads), 0, s->stream,
                           t.tiles, t.slices, t.codes, (const double *)t.w64, t.r_tid,
                           (const double *)t.r_w64, t.r_row, t.r_slot, t.queue, theta, cnt, state,
                           row_w_perm);
    } else {
        hipLaunchKernelGGL((k_em_tile<float>), dim3(t.n_tiles), dim3(kTileThreads), 0, s->stream,
                           t.tiles, t.slices, t.codes, (const float *)t.w32, t.r_tid,
                           (const float *)t.r_w32, t.r_row, t.r_slot, t.queue, theta, cnt, state,
                           row_w_perm);
    }
    OEM_HIP(hipGetLastError());
    if (t.n_remote > 0) {
        // ~2 workgroups of 1024 threads per CU in total, split over the buckets by queue length
        uint32_t n_groups = 512 / (t.n_buckets ? t.n_buckets : 1);
        const uint64_t per_bucket = t.n_remote / (t.n_buckets ? t.n_buckets : 1) + 1;
        const uint32_t max_useful = (uint32_t)((per_bucket + 4095) / 4096);
        if (n_groups > max_useful) n_groups = max_useful;
        if (n_groups < 1) n_groups = 1;
        hipLaunchKernelGGL(k_remote_fold, dim3(t.n_buckets * n_groups), dim3(kFoldThreads), 0,
                           s->stream, t.bucket_base, t.queue, t.q_dst, cnt, state, n_groups,
                           s->csr.n_txps);
        OEM_HIP(hipGetLastError());
    }
    return OEM_OK;
}

int launch_permute_row_w(oem_store *s, const uint32_t *row_w, uint32_t *row_w_perm)
{
    const uint64_t n = s->tiled.n_rows;
    if (n == 0) return OEM_OK;
    uint64_t g = (n + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    hipLaunchKernelGGL(k_permute_row_w, dim3((uint32_t)g), dim3(256), 0, s->stream, row_w,
                       s->tiled.perm, row_w_perm, n);
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

} // namespace oem
