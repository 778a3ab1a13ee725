// oem_layout_device.hip -- the tiled layout of oem_layout.h built ON the device.
//
// build_tiled_layout (oem_layout.cpp) is the specification: this file produces the same arrays,
// element for element, from the caller-order CSR that is already resident in HBM, so the store is
// ready ~20x sooner than when a few host cores build it (10 M reads: ~0.65 s on the host).  The
// equality is a test (`oem_debug_layout_hash`, tests/test_gpu_parity.py); the host builder stays
// as the reference, and as the path for stores this one does not take (>= 2^31 reads or remote
// alignments, 64-bit row pointers).
//
//   A  anchor transcript of every read                       one thread per read
//   B  stable sort of the reads by anchor                    rocPRIM radix sort (hipcub)
//   C  tile cuts: next[i] for every position, then one thread walks the chain
//   D  per tile: local counts, read order inside the tile (two bitonic sorts in LDS), slice widths,
//      remote alignments per (tile, bucket)
//   E  prefix sums: slice / remote offsets per tile, bucket-major queue slot bases
//   F  per tile: fill weights + codes (column-major slices), emit remote (transcript, j) keys
//   G  segmented sort of the remote keys per tile, remote records + queue destinations
#include <hipcub/hipcub.hpp>

#include "oem_internal.h"

namespace oem {

namespace {

constexpr uint32_t kNoKey = 0xffffffffu;
constexpr int kLThreads = 256;

__device__ __forceinline__ uint32_t window_lo(uint32_t k0)
{
    uint32_t lo = k0 > kMargin ? k0 - kMargin : 0u;
    return lo & ~7u;
}

// ---- A ------------------------------------------------------------------------------------
template <typename WT>
__global__ __launch_bounds__(kLThreads) void k_anchor_keys(const uint32_t *__restrict__ row_ptr,
                                                           const uint32_t *__restrict__ tid,
                                                           const WT *__restrict__ w, uint32_t R,
                                                           uint32_t *__restrict__ key, uint32_t *__restrict__ iota,
                                                           uint32_t *n_empty)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    iota[r] = r;
    const uint32_t s = row_ptr[r], t = row_ptr[r + 1];
    if (s == t) {
        key[r] = kNoKey;
        atomicAdd(n_empty, 1u);
        return;
    }
    // (the read's transcripts are compared all against all: short reads keep them in a private LDS row)
    constexpr uint32_t kCacheK = 16;
    __shared__ uint32_t ct[kLThreads][kCacheK + 1];
    const bool cached = t - s <= kCacheK;
    if (cached)
        for (uint32_t j = s; j < t; ++j) ct[threadIdx.x][j - s] = tid[j];
    auto tid_of = [&](uint32_t j) -> uint32_t { return cached ? ct[threadIdx.x][j - s] : tid[j]; };
    uint32_t best = s, best_n = 0;
    double best_w = -1.0;
    for (uint32_t j = s; j < t; ++j) {
        const uint32_t tj = tid_of(j);
        uint32_t n = 0;
        for (uint32_t i = s; i < t; ++i) {
            const uint32_t ti = tid_of(i);
            const uint32_t d = ti > tj ? ti - tj : tj - ti;
            n += d <= kMargin;
        }
        const double wj = (double)w[j];
        if (n > best_n || (n == best_n && (wj > best_w || (wj == best_w && tj < tid_of(best))))) {
            best = j; best_n = n; best_w = wj;
        }
    }
    key[r] = tid_of(best);
}

// ---- C ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t tile_kmax(uint32_t k0, uint32_t problem_size, uint32_t win_cap)
{
    uint32_t kmax = window_lo(k0) + win_cap - kMargin - 1;
    if (problem_size) {
        const uint32_t pend = (k0 / problem_size + 1) * problem_size - 1;
        if (kmax > pend) kmax = pend;
    }
    return kmax;
}

__global__ __launch_bounds__(kLThreads) void k_next_cut(const uint32_t *__restrict__ skey, uint32_t n_rows,
                                                        uint32_t problem_size, uint32_t win_cap, uint32_t tile_rows,
                                                        uint32_t *__restrict__ next)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const uint32_t kmax = tile_kmax(skey[i], problem_size, win_cap);
    uint32_t lim = i + tile_rows;
    if (lim > n_rows) lim = n_rows;
    // first index in (i, lim) whose key exceeds kmax, else lim
    uint32_t a = i + 1, b = lim;
    while (a < b) {
        const uint32_t m = (a + b) >> 1;
        if (skey[m] <= kmax) a = m + 1;
        else b = m;
    }
    next[i] = a;
}

__global__ void k_walk_cuts(const uint32_t *__restrict__ next, uint32_t n_rows, uint32_t *__restrict__ tile_start,
                            uint32_t cap, uint32_t *n_tiles)
{
    if (threadIdx.x || blockIdx.x) return;
    uint32_t pos = 0, n = 0;
    while (pos < n_rows) {
        if (n < cap) tile_start[n] = pos;
        ++n;
        pos = next[pos];
    }
    if (n < cap) tile_start[n] = n_rows;
    *n_tiles = n;
}

// Per-cell batches: a tile never spans two problems (tile_kmax), so the first read of every problem starts a
// tile whatever came before it, and the chains of the problems can be walked side by side: one thread per
// problem, once to count its tiles and once (after a prefix sum) to write them.  The single chain of a
// 625-cell batch (52 k dependent loads) took 7.7 ms.
__global__ __launch_bounds__(kLThreads) void k_walk_cuts_problems(const uint32_t *__restrict__ next, const uint32_t *__restrict__ skey,
                                                                  uint32_t n_rows, uint32_t problem_size, uint32_t n_problems,
                                                                  uint32_t *__restrict__ counts, const uint32_t *__restrict__ offsets,
                                                                  uint32_t *__restrict__ tile_start)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_problems) return;
    auto first_at_least = [&](uint64_t k) -> uint32_t { // first row whose key is >= k
        uint32_t a = 0, b = n_rows;
        while (a < b) {
            const uint32_t m = (a + b) >> 1;
            if ((uint64_t)skey[m] < k) a = m + 1;
            else b = m;
        }
        return a;
    };
    const uint32_t begin = first_at_least((uint64_t)p * problem_size), end = first_at_least((uint64_t)(p + 1) * problem_size);
    uint32_t n = 0;
    for (uint32_t pos = begin; pos < end; pos = next[pos]) {
        if (tile_start) tile_start[offsets[p] + n] = pos;
        ++n;
    }
    if (!tile_start) counts[p] = n;
}
__global__ void k_finish_cuts(const uint32_t *__restrict__ offsets, uint32_t n_problems, uint32_t n_rows,
                              uint32_t *__restrict__ tile_start, uint32_t *n_tiles)
{
    if (threadIdx.x || blockIdx.x) return;
    const uint32_t n = offsets[n_problems]; // (the scan runs over n_problems + 1 entries: the last is the total)
    tile_start[n] = n_rows;
    *n_tiles = n;
}

// ---- D ------------------------------------------------------------------------------------
struct TileAux { // per tile, for the prefix sums of stage E
    uint32_t w_slots, c_slots, remote_cnt, pad;
};

__device__ __forceinline__ void bitonic_sort_u64(unsigned long long *a, uint32_t n_pow2)
{
    for (uint32_t k = 2; k <= n_pow2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < n_pow2; i += blockDim.x) {
                const uint32_t p = i ^ j;
                if (p > i) {
                    const unsigned long long x = a[i], y = a[p];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { a[i] = y; a[p] = x; }
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(kLThreads) void k_tile_pass1(
    const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ tid, const uint32_t *__restrict__ skey,
    const uint32_t *__restrict__ order, const uint32_t *__restrict__ tile_start, uint32_t n_txps,
    uint32_t n_buckets, uint32_t problem_size, uint32_t win_cap, TileDesc *__restrict__ tiles,
    TileAux *__restrict__ aux, uint32_t *__restrict__ perm, uint32_t *__restrict__ cnt_tb, uint32_t *too_wide)
{
    __shared__ unsigned long long sk[kTileRows];
    __shared__ uint32_t nloc_s[kTileRows];
    __shared__ uint32_t head[kTileRows];
    __shared__ uint32_t red[3];
    const uint32_t ti = blockIdx.x;
    const uint32_t p0 = tile_start[ti], p1 = tile_start[ti + 1], n = p1 - p0;
    const uint32_t k0 = skey[p0], lo = window_lo(k0);
    uint32_t win = skey[p1 - 1] - lo + kMargin + 1;
    if (win > win_cap) win = win_cap;
    if (lo + win > n_txps) win = n_txps - lo;
    if (threadIdx.x < 3) red[threadIdx.x] = 0;
    __syncthreads();
    uint32_t n2 = 1;
    while (n2 < n) n2 <<= 1;
    uint32_t my_remote = 0, my_max = 0;
    for (uint32_t i = threadIdx.x; i < n2; i += blockDim.x) {
        if (i < n) {
            const uint32_t r = order[p0 + i];
            uint32_t nloc = 0;
            for (uint32_t j = row_ptr[r]; j < row_ptr[r + 1]; ++j) {
                const uint32_t t = tid[j];
                if (t - lo < win) ++nloc; // wraps for t < lo
                else { ++my_remote; atomicAdd(&cnt_tb[(size_t)ti * n_buckets + t / kBucket], 1u); }
            }
            if (nloc > my_max) my_max = nloc;
            const uint32_t nl = nloc > 255u ? 255u : nloc;
            nloc_s[i] = nl;
            // sort 1: local count descending, anchor ascending, arrival order
            sk[i] = ((unsigned long long)(255u - nl) << 21) | ((unsigned long long)(skey[p0 + i] - lo) << 10) | i; // anchor - lo < kWinWide: 11 bits
        } else {
            sk[i] = ~0ull;
        }
    }
    atomicAdd(&red[0], my_remote);
    atomicMax(&red[1], my_max);
    __syncthreads();
    bitonic_sort_u64(sk, n2);
    // rank inside each (count, anchor) group = distance from the group's first position
    for (uint32_t q = threadIdx.x; q < n; q += blockDim.x)
        head[q] = (q == 0 || (sk[q] >> 10) != (sk[q - 1] >> 10)) ? q : 0u;
    __syncthreads();
    for (uint32_t off = 1; off < n; off <<= 1) { // inclusive max-scan
        uint32_t v[kTileRows / kLThreads];
        uint32_t c = 0;
        for (uint32_t q = threadIdx.x; q < n; q += blockDim.x, ++c) v[c] = q >= off ? max(head[q], head[q - off]) : head[q];
        __syncthreads();
        c = 0;
        for (uint32_t q = threadIdx.x; q < n; q += blockDim.x, ++c) head[q] = v[c];
        __syncthreads();
    }
    // sort 2: local count descending, rank ascending, anchor ascending
    {
        unsigned long long v[kTileRows / kLThreads];
        uint32_t c = 0;
        for (uint32_t q = threadIdx.x; q < n; q += blockDim.x, ++c) {
            const unsigned long long k1 = sk[q];
            const uint32_t idx = (uint32_t)(k1 & 1023u), anc = (uint32_t)(k1 >> 10) & 2047u;
            const uint32_t inv_nl = (uint32_t)(k1 >> 21);
            v[c] = ((unsigned long long)inv_nl << 31) | ((unsigned long long)(q - head[q]) << 21) |
                   ((unsigned long long)anc << 10) | idx;
        }
        __syncthreads();
        c = 0;
        for (uint32_t q = threadIdx.x; q < n; q += blockDim.x, ++c) sk[q] = v[c];
        __syncthreads();
    }
    bitonic_sort_u64(sk, n2);
    for (uint32_t q = threadIdx.x; q < n; q += blockDim.x) perm[p0 + q] = order[p0 + (uint32_t)(sk[q] & 1023u)];
    if (threadIdx.x == 0) {
        TileDesc td;
        memset(&td, 0, sizeof(td));
        td.n_rows = n;
        td.row_base = p0;
        td.lo = lo;
        td.win_len = win;
        td.remote_cnt = red[0];
        td.n_slices = (n + 63) / 64;
        td.problem = problem_size ? k0 / problem_size : 0u;
        uint32_t ws = 0, cs = 0;
        for (uint32_t s = 0; s < td.n_slices; ++s) {
            const uint32_t width = nloc_s[(uint32_t)(sk[s * 64] & 1023u)]; // first read of the slice is the longest
            td.width[s] = (uint8_t)width;
            ws += width;
            cs += (width + 1) / 2;
        }
        tiles[ti] = td;
        aux[ti] = TileAux{ws, cs, red[0], 0u};
        if (red[1] > 255u) atomicOr(too_wide, 1u);
    }
}

// ---- E ------------------------------------------------------------------------------------
// exclusive prefix sums over the tiles (one workgroup): w_base, c_base, remote_begin; totals out
__global__ __launch_bounds__(1024) void k_tile_offsets(TileDesc *__restrict__ tiles, const TileAux *__restrict__ aux,
                                                       uint32_t n_tiles, unsigned long long *totals /* [3] */)
{
    __shared__ unsigned long long part[3][1024];
    const uint32_t per = (n_tiles + blockDim.x - 1) / blockDim.x;
    const uint32_t b = threadIdx.x * per, e = min(n_tiles, b + per);
    unsigned long long s0 = 0, s1 = 0, s2 = 0;
    for (uint32_t i = b; i < e; ++i) { s0 += aux[i].w_slots; s1 += aux[i].c_slots; s2 += aux[i].remote_cnt; }
    part[0][threadIdx.x] = s0; part[1][threadIdx.x] = s1; part[2][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < 3) { // serial exclusive scan of 1024 partials per quantity
        unsigned long long acc = 0;
        for (uint32_t i = 0; i < blockDim.x; ++i) { const unsigned long long v = part[threadIdx.x][i]; part[threadIdx.x][i] = acc; acc += v; }
        totals[threadIdx.x] = acc;
    }
    __syncthreads();
    s0 = part[0][threadIdx.x]; s1 = part[1][threadIdx.x]; s2 = part[2][threadIdx.x];
    for (uint32_t i = b; i < e; ++i) {
        tiles[i].w_base = (uint32_t)s0; tiles[i].c_base = (uint32_t)s1; tiles[i].remote_begin = (uint32_t)s2;
        s0 += aux[i].w_slots; s1 += aux[i].c_slots; s2 += aux[i].remote_cnt;
    }
}

// bucket-major queue slots: slot base of (tile, bucket) = bucket_base[b] + remote alignments of earlier
// tiles in b.  Tiles are cut into kChunks chunks: column sums per chunk, scan, then the running pass.
constexpr uint32_t kChunks = 256;

__global__ __launch_bounds__(kLThreads) void k_bucket_chunk_sums(const uint32_t *__restrict__ cnt_tb, uint32_t n_tiles,
                                                                 uint32_t n_buckets, unsigned long long *__restrict__ csum)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x, ch = blockIdx.y;
    if (b >= n_buckets) return;
    const uint32_t per = (n_tiles + kChunks - 1) / kChunks, t0 = ch * per, t1 = min(n_tiles, t0 + per);
    unsigned long long s = 0;
    for (uint32_t t = t0; t < t1; ++t) s += cnt_tb[(size_t)t * n_buckets + b];
    csum[(size_t)ch * n_buckets + b] = s;
}

// one workgroup: totals per bucket -> bucket_base (exclusive scan over buckets), and csum becomes the
// slot base at the start of every (chunk, bucket)
__global__ __launch_bounds__(1024) void k_bucket_bases(unsigned long long *__restrict__ csum, uint32_t n_buckets,
                                                       uint32_t *__restrict__ bucket_base)
{
    __shared__ unsigned long long part[1024];
    const uint32_t per = (n_buckets + blockDim.x - 1) / blockDim.x;
    const uint32_t b0 = threadIdx.x * per, b1 = min(n_buckets, b0 + per);
    unsigned long long s = 0;
    for (uint32_t b = b0; b < b1; ++b)
        for (uint32_t ch = 0; ch < kChunks; ++ch) s += csum[(size_t)ch * n_buckets + b];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long acc = 0;
        for (uint32_t i = 0; i < blockDim.x; ++i) { const unsigned long long v = part[i]; part[i] = acc; acc += v; }
        bucket_base[n_buckets] = (uint32_t)acc;
    }
    __syncthreads();
    unsigned long long acc = part[threadIdx.x];
    for (uint32_t b = b0; b < b1; ++b) {
        bucket_base[b] = (uint32_t)acc;
        for (uint32_t ch = 0; ch < kChunks; ++ch) {
            const unsigned long long v = csum[(size_t)ch * n_buckets + b];
            csum[(size_t)ch * n_buckets + b] = acc;
            acc += v;
        }
    }
}

__global__ __launch_bounds__(kLThreads) void k_bucket_running(uint32_t *__restrict__ cnt_tb, uint32_t n_tiles,
                                                              uint32_t n_buckets, const unsigned long long *__restrict__ csum)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x, ch = blockIdx.y;
    if (b >= n_buckets) return;
    const uint32_t per = (n_tiles + kChunks - 1) / kChunks, t0 = ch * per, t1 = min(n_tiles, t0 + per);
    unsigned long long run = csum[(size_t)ch * n_buckets + b];
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t n = cnt_tb[(size_t)t * n_buckets + b];
        cnt_tb[(size_t)t * n_buckets + b] = (uint32_t)run; // becomes the slot base of (tile, bucket)
        run += n;
    }
}

// ---- F ------------------------------------------------------------------------------------
template <typename WT>
__global__ __launch_bounds__(kLThreads) void k_tile_fill(
    const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ tid, const WT *__restrict__ w_in,
    const uint32_t *__restrict__ key, const TileDesc *__restrict__ tiles, const uint32_t *__restrict__ perm,
    WT *__restrict__ w_out, uint32_t *__restrict__ codes, unsigned long long *__restrict__ rem_key,
    uint32_t *__restrict__ rem_row)
{
    __shared__ uint32_t woff[kTileSlices], coff[kTileSlices];
    __shared__ uint32_t rem_next;
    // a read's transcripts are looked at (local alignments + 1) times by the selection below: short reads keep
    // them in LDS (one private row per thread, odd stride), long ones go back to memory every time
    constexpr uint32_t kCacheK = 16;
    __shared__ uint32_t ct[kLThreads][kCacheK + 1];
    const TileDesc td = tiles[blockIdx.x];
    if (threadIdx.x == 0) {
        uint32_t a = td.w_base, c = td.c_base;
        for (uint32_t s = 0; s < kTileSlices; ++s) {
            woff[s] = a; coff[s] = c;
            a += td.width[s];
            c += (td.width[s] + 1u) >> 1;
        }
        rem_next = 0;
    }
    __syncthreads();
    const uint32_t lo = td.lo, win = td.win_len;
    for (uint32_t rl = threadIdx.x; rl < td.n_slices * 64; rl += blockDim.x) {
        const uint32_t s = rl >> 6, lane = rl & 63u, width = td.width[s];
        const size_t wb = (size_t)woff[s] * 64 + lane, cb = (size_t)coff[s] * 64 + lane;
        uint32_t nloc = 0;
        if (rl < td.n_rows) {
            const uint32_t r = perm[td.row_base + rl], anchor = key[r];
            const uint32_t j0 = row_ptr[r], j1 = row_ptr[r + 1];
            const bool cached = j1 - j0 <= kCacheK;
            if (cached)
                for (uint32_t j = j0; j < j1; ++j) ct[threadIdx.x][j - j0] = tid[j];
            auto tid_of = [&](uint32_t j) -> uint32_t { return cached ? ct[threadIdx.x][j - j0] : tid[j]; };
            // local alignments in the order (anchor first, then transcript, then position), by
            // repeated selection of the smallest key above the last one emitted
            unsigned long long last = 0; // keys are > 0
            bool first = true;
            uint32_t pending = 0, n_rem = 0;
            for (;;) {
                unsigned long long best = ~0ull;
                uint32_t best_j = 0;
                for (uint32_t j = j0; j < j1; ++j) {
                    const uint32_t t = tid_of(j);
                    if (t - lo >= win) { if (first) ++n_rem; continue; }
                    const unsigned long long k = ((unsigned long long)(t == anchor ? 0u : t - lo + 1u) << 32) | (j - j0 + 1u);
                    if (k > last && k < best) { best = k; best_j = j; }
                }
                if (first && n_rem) { // reserve this read's remote records, then write them
                    uint32_t o = td.remote_begin + atomicAdd(&rem_next, n_rem);
                    for (uint32_t j = j0; j < j1; ++j) {
                        const uint32_t t = tid_of(j);
                        if (t - lo >= win) {
                            rem_key[o] = ((unsigned long long)t << 32) | j;
                            rem_row[o] = rl;
                            ++o;
                        }
                    }
                }
                first = false;
                if (best == ~0ull) break;
                last = best;
                const uint32_t code = (tid_of(best_j) - lo) * 8u; // LDS byte offset
                w_out[wb + (size_t)nloc * 64] = w_in[best_j];
                if (nloc & 1u) codes[cb + (size_t)(nloc >> 1) * 64] = pending | (code << 16);
                else pending = code;
                ++nloc;
            }
            if (nloc & 1u) codes[cb + (size_t)(nloc >> 1) * 64] = pending;
        }
        // padding of the slice: weight 0, code 0
        for (uint32_t q = nloc; q < width; ++q) w_out[wb + (size_t)q * 64] = (WT)0;
        for (uint32_t g = (nloc + 1u) >> 1; g < (width + 1u) >> 1; ++g) codes[cb + (size_t)g * 64] = 0u;
    }
}

// ---- G ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kLThreads) void k_seg_bounds(const TileDesc *__restrict__ tiles, uint32_t n_tiles,
                                                          uint32_t *__restrict__ seg_b, uint32_t *__restrict__ seg_e)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tiles) return;
    seg_b[i] = tiles[i].remote_begin;
    seg_e[i] = tiles[i].remote_begin + tiles[i].remote_cnt;
}

template <typename WT>
__global__ __launch_bounds__(kLThreads) void k_remote_finish(
    const TileDesc *__restrict__ tiles, const unsigned long long *__restrict__ rem_key,
    const uint32_t *__restrict__ rem_row, const WT *__restrict__ w_in, const uint32_t *__restrict__ cnt_tb,
    uint32_t n_buckets, uint32_t *__restrict__ r_tid, WT *__restrict__ r_w, uint16_t *__restrict__ r_row,
    uint32_t *__restrict__ r_slot, uint16_t *__restrict__ q_dst)
{
    const uint32_t ti = blockIdx.x;
    const uint32_t b0 = tiles[ti].remote_begin, cnt = tiles[ti].remote_cnt;
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
        const unsigned long long k = rem_key[b0 + i];
        const uint32_t t = (uint32_t)(k >> 32), j = (uint32_t)k, b = t / kBucket;
        // first record of bucket b inside the tile (records are sorted by transcript)
        const unsigned long long lim = (unsigned long long)(b * kBucket) << 32;
        uint32_t a = 0, e = i;
        while (a < e) {
            const uint32_t m = (a + e) >> 1;
            if (rem_key[b0 + m] < lim) a = m + 1;
            else e = m;
        }
        const uint32_t slot = cnt_tb[(size_t)ti * n_buckets + b] + (i - a);
        r_tid[b0 + i] = t;
        r_w[b0 + i] = w_in[j];
        r_row[b0 + i] = (uint16_t)rem_row[b0 + i];
        r_slot[b0 + i] = slot;
        q_dst[slot] = (uint16_t)(t % kBucket);
    }
}

template <typename T>
__global__ __launch_bounds__(kLThreads) void k_zero_tail(T *p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (T)0;
}

struct Scratch { // freed on every exit path
    std::vector<void *> ptrs;
    template <typename T> int alloc(T **p, size_t n)
    {
        *p = nullptr;
        hipError_t e = hipMalloc((void **)p, (n ? n : 1) * sizeof(T));
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? OEM_ERR_OOM : OEM_ERR_HIP, "layout scratch: %s", hipGetErrorString(e));
        ptrs.push_back(*p);
        return OEM_OK;
    }
    ~Scratch() { for (void *p : ptrs) hipFree(p); }
};

template <typename T>
int out_alloc(T **p, size_t n, uint64_t *acct)
{
    *p = nullptr;
    const size_t bytes = (n ? n : 1) * sizeof(T);
    OEM_HIP(hipMalloc((void **)p, bytes));
    *acct += bytes;
    return OEM_OK;
}

template <typename WT>
int build_impl(oem_store *s, uint32_t problem_size, uint32_t win_cap, uint32_t tile_rows, const WT *w_in, WT **w_out, WT **r_w_out, bool *built)
{
    *built = false;
    const DeviceCsr &m = s->csr;
    DeviceTiled &t = s->tiled;
    hipStream_t st = s->stream;
    const uint32_t R = (uint32_t)m.n_reads, T = m.n_txps;
    const uint32_t *row_ptr = (const uint32_t *)m.row_ptr;
    const uint32_t n_buckets = (T + kBucket - 1) / kBucket;
    Scratch sc;

    // A + B
    uint32_t *key, *iota, *skey, *order, *d_small; // d_small: [0] n_empty [1] n_tiles [2] too_wide
    OEM_TRY(sc.alloc(&key, R));
    OEM_TRY(sc.alloc(&iota, R));
    OEM_TRY(sc.alloc(&skey, R));
    OEM_TRY(sc.alloc(&order, R));
    OEM_TRY(sc.alloc(&d_small, 4));
    OEM_HIP(hipMemsetAsync(d_small, 0, 16, st));
    hipLaunchKernelGGL(k_anchor_keys<WT>, dim3((R + kLThreads - 1) / kLThreads), dim3(kLThreads), 0, st, row_ptr, m.tid,
                       w_in, R, key, iota, d_small);
    OEM_HIP(hipGetLastError());
    {
        size_t tmp_bytes = 0;
        OEM_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key, skey, iota, order, (int)R, 0, 32, st));
        void *tmp;
        OEM_TRY(sc.alloc((char **)&tmp, tmp_bytes));
        OEM_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key, skey, iota, order, (int)R, 0, 32, st));
    }
    uint32_t h_small[4];
    OEM_HIP(hipMemcpyAsync(h_small, d_small, 16, hipMemcpyDeviceToHost, st));
    OEM_HIP(hipStreamSynchronize(st));
    const uint32_t n_rows = R - h_small[0];
    t.n_rows = n_rows;
    if (n_rows == 0) return OEM_OK; // nothing to tile: the caller keeps the CSR path

    // C
    uint32_t *next = iota; // reuse
    hipLaunchKernelGGL(k_next_cut, dim3((n_rows + kLThreads - 1) / kLThreads), dim3(kLThreads), 0, st, skey, n_rows,
                       problem_size, win_cap, tile_rows, next);
    OEM_HIP(hipGetLastError());
    uint32_t *tile_start;
    const uint32_t cap = n_rows + 1; // every tile holds at least one read
    OEM_TRY(sc.alloc(&tile_start, (size_t)cap + 1));
    const uint32_t n_problems = problem_size ? (uint32_t)(((uint64_t)T + problem_size - 1) / problem_size) : 0u;
    if (n_problems > 1) {
        uint32_t *counts, *offsets;
        OEM_TRY(sc.alloc(&counts, (size_t)n_problems + 1));
        OEM_TRY(sc.alloc(&offsets, (size_t)n_problems + 1));
        OEM_HIP(hipMemsetAsync(counts, 0, sizeof(uint32_t) * ((size_t)n_problems + 1), st));
        const dim3 pgrid((n_problems + kLThreads - 1) / kLThreads);
        hipLaunchKernelGGL(k_walk_cuts_problems, pgrid, dim3(kLThreads), 0, st, next, skey, n_rows, problem_size, n_problems,
                           counts, (const uint32_t *)nullptr, (uint32_t *)nullptr);
        size_t tmp_bytes = 0;
        OEM_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, counts, offsets, (int)n_problems + 1, st));
        void *tmp;
        OEM_TRY(sc.alloc((char **)&tmp, tmp_bytes ? tmp_bytes : 1));
        OEM_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, counts, offsets, (int)n_problems + 1, st));
        hipLaunchKernelGGL(k_walk_cuts_problems, pgrid, dim3(kLThreads), 0, st, next, skey, n_rows, problem_size, n_problems,
                           counts, offsets, tile_start);
        hipLaunchKernelGGL(k_finish_cuts, dim3(1), dim3(64), 0, st, offsets, n_problems, n_rows, tile_start, d_small + 1);
    } else {
        hipLaunchKernelGGL(k_walk_cuts, dim3(1), dim3(64), 0, st, next, n_rows, tile_start, cap, d_small + 1);
    }
    OEM_HIP(hipGetLastError());
    OEM_HIP(hipMemcpyAsync(h_small, d_small, 16, hipMemcpyDeviceToHost, st));
    OEM_HIP(hipStreamSynchronize(st));
    const uint32_t n_tiles = h_small[1];
    if ((uint64_t)n_tiles * n_buckets >= (1ull << 31)) return OEM_OK; // tile x bucket table too large: host path decides

    // D
    TileAux *aux;
    uint32_t *cnt_tb;
    OEM_TRY(sc.alloc(&aux, n_tiles));
    OEM_TRY(sc.alloc(&cnt_tb, (size_t)n_tiles * n_buckets));
    OEM_HIP(hipMemsetAsync(cnt_tb, 0, sizeof(uint32_t) * (size_t)n_tiles * n_buckets, st));
    OEM_TRY(out_alloc(&t.tiles, n_tiles, &s->hbm_bytes));
    OEM_TRY(out_alloc(&t.perm, n_rows, &s->hbm_bytes));
    hipLaunchKernelGGL(k_tile_pass1, dim3(n_tiles), dim3(kLThreads), 0, st, row_ptr, m.tid, skey, order, tile_start, T,
                       n_buckets, problem_size, win_cap, t.tiles, aux, t.perm, cnt_tb, d_small + 2);
    OEM_HIP(hipGetLastError());

    // E
    unsigned long long *totals, *csum;
    OEM_TRY(sc.alloc(&totals, 4));
    OEM_TRY(sc.alloc(&csum, (size_t)kChunks * n_buckets));
    hipLaunchKernelGGL(k_tile_offsets, dim3(1), dim3(1024), 0, st, t.tiles, aux, n_tiles, totals);
    OEM_TRY(out_alloc(&t.bucket_base, (size_t)n_buckets + 1, &s->hbm_bytes));
    const dim3 bgrid((n_buckets + kLThreads - 1) / kLThreads, kChunks);
    hipLaunchKernelGGL(k_bucket_chunk_sums, bgrid, dim3(kLThreads), 0, st, cnt_tb, n_tiles, n_buckets, csum);
    hipLaunchKernelGGL(k_bucket_bases, dim3(1), dim3(1024), 0, st, csum, n_buckets, t.bucket_base);
    hipLaunchKernelGGL(k_bucket_running, bgrid, dim3(kLThreads), 0, st, cnt_tb, n_tiles, n_buckets, csum);
    OEM_HIP(hipGetLastError());
    unsigned long long h_tot[3];
    OEM_HIP(hipMemcpyAsync(h_tot, totals, sizeof(h_tot), hipMemcpyDeviceToHost, st));
    OEM_HIP(hipMemcpyAsync(h_small, d_small, 16, hipMemcpyDeviceToHost, st));
    OEM_HIP(hipStreamSynchronize(st));
    if (h_small[2]) return OEM_OK; // a read with > 255 alignments inside one window: the host builder words the refusal
    const uint64_t w_slots = h_tot[0], c_slots = h_tot[1], n_remote = h_tot[2];
    if (w_slots >= (1ull << 32) || c_slots >= (1ull << 32) || n_remote >= (1ull << 31)) return OEM_OK; // host builder decides

    // F
    OEM_TRY(out_alloc(&t.codes, (size_t)(c_slots + 1) * 64, &s->hbm_bytes));
    OEM_TRY(out_alloc(w_out, (size_t)(w_slots + 1) * 64, &s->hbm_bytes));
    hipLaunchKernelGGL(k_zero_tail<uint32_t>, dim3(1), dim3(kLThreads), 0, st, t.codes + c_slots * 64, (size_t)64); // the slack row
    hipLaunchKernelGGL(k_zero_tail<WT>, dim3(1), dim3(kLThreads), 0, st, *w_out + w_slots * 64, (size_t)64);
    unsigned long long *rem_key, *rem_key2;
    uint32_t *rem_row, *rem_row2;
    OEM_TRY(sc.alloc(&rem_key, n_remote));
    OEM_TRY(sc.alloc(&rem_key2, n_remote));
    OEM_TRY(sc.alloc(&rem_row, n_remote));
    OEM_TRY(sc.alloc(&rem_row2, n_remote));
    hipLaunchKernelGGL(k_tile_fill<WT>, dim3(n_tiles), dim3(kLThreads), 0, st, row_ptr, m.tid, w_in, key, t.tiles,
                       t.perm, *w_out, t.codes, rem_key, rem_row);
    OEM_HIP(hipGetLastError());

    // G
    OEM_TRY(out_alloc(&t.r_tid, n_remote, &s->hbm_bytes));
    OEM_TRY(out_alloc(r_w_out, n_remote, &s->hbm_bytes));
    OEM_TRY(out_alloc(&t.r_row, n_remote, &s->hbm_bytes));
    OEM_TRY(out_alloc(&t.r_slot, n_remote, &s->hbm_bytes));
    OEM_TRY(out_alloc(&t.q_dst, n_remote, &s->hbm_bytes));
    if (n_remote) {
        uint32_t *seg_b, *seg_e;
        OEM_TRY(sc.alloc(&seg_b, n_tiles));
        OEM_TRY(sc.alloc(&seg_e, n_tiles));
        hipLaunchKernelGGL(k_seg_bounds, dim3((n_tiles + kLThreads - 1) / kLThreads), dim3(kLThreads), 0, st, t.tiles,
                           n_tiles, seg_b, seg_e);
        size_t tmp_bytes = 0;
        OEM_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, tmp_bytes, rem_key, rem_key2, rem_row, rem_row2,
                                                            (int)n_remote, (int)n_tiles, seg_b, seg_e, 0, 64, st));
        void *tmp;
        OEM_TRY(sc.alloc((char **)&tmp, tmp_bytes));
        OEM_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(tmp, tmp_bytes, rem_key, rem_key2, rem_row, rem_row2,
                                                            (int)n_remote, (int)n_tiles, seg_b, seg_e, 0, 64, st));
        hipLaunchKernelGGL(k_remote_finish<WT>, dim3(n_tiles), dim3(kLThreads), 0, st, t.tiles, rem_key2, rem_row2,
                           w_in, cnt_tb, n_buckets, t.r_tid, *r_w_out, t.r_row, t.r_slot, t.q_dst);
        OEM_HIP(hipGetLastError());
    }
    t.h_bucket_base.assign((size_t)n_buckets + 1, 0u);
    OEM_HIP(hipMemcpyAsync(t.h_bucket_base.data(), t.bucket_base, sizeof(uint32_t) * ((size_t)n_buckets + 1),
                           hipMemcpyDeviceToHost, st));
    OEM_HIP(hipStreamSynchronize(st));
    OEM_TRY(out_alloc(&t.queue, n_remote, &s->hbm_bytes));
    OEM_TRY(out_alloc(&t.row_w_perm, n_rows, &s->hbm_bytes));
    t.n_tiles = n_tiles;
    t.win_cap = win_cap;
    t.n_buckets = n_buckets;
    t.n_remote = n_remote;
    t.n_local = m.nnz - n_remote;
    t.present = true;
    t.built_on_device = true;
    *built = true;
    return OEM_OK;
}

} // namespace

// Builds s->tiled from s->csr on the device.  *built = false (and nothing allocated that matters)
// when this builder does not take the store; the caller then uses the host builder.
int build_tiled_layout_device(oem_store *s, uint32_t problem_size, uint32_t win_cap, uint32_t tile_rows, bool *built)
{
    *built = false;
    const DeviceCsr &m = s->csr;
    if (m.wide_ptr || m.n_reads == 0 || m.n_reads >= (1ull << 31) || m.nnz >= (1ull << 32)) return OEM_OK;
    int rc;
    if (m.w_is_f64) rc = build_impl<double>(s, problem_size, win_cap, tile_rows, m.w64, &s->tiled.w64, &s->tiled.r_w64, built);
    else rc = build_impl<float>(s, problem_size, win_cap, tile_rows, m.w32, &s->tiled.w32, &s->tiled.r_w32, built);
    if (rc != OEM_OK || !*built) { // leave no half-built layout behind
        DeviceTiled &t = s->tiled;
        hipFree(t.tiles); hipFree(t.perm); hipFree(t.codes); hipFree(t.w32); hipFree(t.w64); hipFree(t.r_tid);
        hipFree(t.r_w32); hipFree(t.r_w64); hipFree(t.r_row); hipFree(t.r_slot); hipFree(t.q_dst);
        hipFree(t.bucket_base); hipFree(t.queue); hipFree(t.row_w_perm);
        t = DeviceTiled();
        t.tile_rows = tile_rows;
    }
    return rc;
}

} // namespace oem
