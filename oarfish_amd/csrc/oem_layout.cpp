// oem_layout.cpp -- one-off host pass that lays the alignment store out for the
// tile kernels (see oem_layout.h).  Runs once per store at upload, the analogue
// of the store being built once by alignment_parser.rs before the EM starts.
#include "oem_layout.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

namespace oem {

namespace {

template <typename F>
void parallel_for(uint32_t n, F fn)
{
    static const unsigned nt_max = [] {
        const char *e = getenv("OEM_HOST_THREADS"); // layout-build threads (default: all cores, at most 128)
        unsigned n = e ? (unsigned)atoi(e) : std::thread::hardware_concurrency();
        if (n == 0) n = 4;
        if (!e && n > 128) n = 128;
        return n;
    }();
    unsigned nt = nt_max;
    if (nt > (n + 15) / 16) nt = (n + 15) / 16; // one chunk of 16 per thread at least
    if (n < 64 || nt == 1) {
        for (uint32_t i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<uint32_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const uint32_t b = next.fetch_add(16);
            if (b >= n) return;
            const uint32_t e = std::min(n, b + 16);
            for (uint32_t i = b; i < e; ++i) fn(i);
        }
    };
    std::vector<std::thread> th;
    th.reserve(nt);
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(worker);
    for (auto &t : th) t.join();
}

struct TileSizes {
    uint32_t n_slices = 0;
    uint64_t w_slots = 0;    // units of 64 weights
    uint64_t c_slots = 0;    // units of 64 packed code pairs
    uint32_t remote_cnt = 0;
    uint32_t win_len = 1;
};

struct RemoteRec {
    uint32_t tid;
    uint16_t row;
    uint64_t j; // index into the caller's arrays
};

} // namespace

bool build_tiled_layout(const uint64_t *row_ptr, const uint32_t *tid, const float *as_prob,
                        const double *cov_prob, uint64_t n_reads, uint64_t nnz, uint32_t n_txps,
                        TiledHost *out, const char **err, uint32_t problem_size, uint32_t win_cap,
                        uint32_t tile_rows)
{
    (void)nnz;
    if (n_reads >= (1ull << 32)) {
        *err = "tiled layout needs n_reads < 2^32";
        return false;
    }
    const bool verbose = getenv("OEM_VERBOSE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!verbose) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[oem]   layout: %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t_prev).count());
        t_prev = t1;
    };
    const uint32_t R = (uint32_t)n_reads;
    const uint32_t n_buckets = (n_txps + kBucket - 1) / kBucket;
    out->n_buckets = n_buckets;
    out->win_cap = win_cap;

    // 1. anchor transcript of every read: the alignment that has the most of the read's
    //    other alignments within +-kMargin transcripts (ties: larger weight, then smaller
    //    id).  For a read that maps inside one gene family this is the family; a stray
    //    high-scoring hit elsewhere does not drag the read away from it.
    std::vector<uint32_t> key(R);
    std::vector<uint32_t> hist((size_t)n_txps + 1, 0);
    {
        const uint32_t nblk = (R + 65535) / 65536;
        parallel_for(nblk, [&](uint32_t blk) {
            const uint32_t b = blk * 65536, e = std::min(R, b + 65536);
            for (uint32_t r = b; r < e; ++r) {
                const uint64_t s = row_ptr[r], t = row_ptr[r + 1];
                if (s == t) { key[r] = 0xffffffffu; continue; }
                uint64_t best = s;
                uint32_t best_n = 0;
                double best_w = -1.0;
                for (uint64_t j = s; j < t; ++j) {
                    const uint32_t tj = tid[j];
                    uint32_t n = 0;
                    for (uint64_t i = s; i < t; ++i) {
                        const uint32_t d = tid[i] > tj ? tid[i] - tj : tj - tid[i];
                        n += d <= kMargin;
                    }
                    const double w = cov_prob ? (double)as_prob[j] * cov_prob[j] : (double)as_prob[j];
                    if (n > best_n || (n == best_n && (w > best_w || (w == best_w && tj < tid[best])))) {
                        best = j; best_n = n; best_w = w;
                    }
                }
                key[r] = tid[best];
            }
        });
    }
    uint32_t n_rows = 0;
    for (uint32_t r = 0; r < R; ++r)
        if (key[r] != 0xffffffffu) { hist[key[r] + 1]++; ++n_rows; }
    for (uint32_t t = 0; t < n_txps; ++t) hist[t + 1] += hist[t];
    out->n_rows = n_rows;

    lap("anchors");
    // 2. stable counting sort of the non-empty reads by primary
    std::vector<uint32_t> order(n_rows);
    {
        std::vector<uint32_t> cur(hist.begin(), hist.end() - 1);
        for (uint32_t r = 0; r < R; ++r)
            if (key[r] != 0xffffffffu) order[cur[key[r]]++] = r;
    }

    lap("counting sort");
    // 3. tile boundaries: <= kTileRows reads, primaries within the window
    std::vector<uint32_t> tile_start; // positions in `order`
    std::vector<uint32_t> tile_lo;
    std::vector<uint32_t> tile_win; // window length: anchors + kMargin on both sides, never the full kWin by default
    {
        uint32_t pos = 0;
        while (pos < n_rows) {
            const uint32_t k0 = key[order[pos]];
            uint32_t lo = k0 > kMargin ? k0 - kMargin : 0;
            lo &= ~7u;
            uint32_t kmax = lo + win_cap - kMargin - 1;
            if (problem_size) { // stay inside the problem of the first read
                const uint32_t pend = (k0 / problem_size + 1) * problem_size - 1;
                if (kmax > pend) kmax = pend;
            }
            uint32_t end = pos + 1;
            while (end < n_rows && end - pos < tile_rows && key[order[end]] <= kmax) ++end;
            tile_start.push_back(pos);
            tile_lo.push_back(lo);
            // A stray alignment that merely happens to fall inside [lo, lo + kWin) is cheaper as a
            // remote record than as a reason to load, clear and flush a whole 2048-entry window.
            const uint32_t kend = key[order[end - 1]];
            uint32_t win = kend - lo + kMargin + 1;
            if (win > win_cap) win = win_cap;
            if (lo + win > n_txps) win = n_txps - lo;
            tile_win.push_back(win);
            pos = end;
        }
        tile_start.push_back(n_rows);
    }
    const uint32_t n_tiles = (uint32_t)tile_lo.size();
    out->n_tiles = n_tiles;
    out->tiles.assign(n_tiles, TileDesc{});
    out->perm.assign(n_rows, 0);

    lap("tile cuts");
    // 4. pass 1 (parallel over tiles): order reads by local count, measure sizes
    std::vector<TileSizes> sizes(n_tiles);
    std::atomic<bool> too_wide{false};
    std::vector<uint32_t> cnt_tb((size_t)n_tiles * n_buckets, 0u); // remote alignments per (tile, bucket)
    parallel_for(n_tiles, [&](uint32_t ti) {
        const uint32_t p0 = tile_start[ti], p1 = tile_start[ti + 1];
        const uint32_t n = p1 - p0, lo = tile_lo[ti], win = tile_win[ti];
        // (local count, anchor, rank among the reads with the same count and anchor, original read)
        struct RowKey { uint32_t nloc, anchor, rank, r; };
        std::vector<RowKey> rows(n);
        TileSizes sz;
        uint32_t max_code = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t r = order[p0 + i];
            uint32_t nloc = 0;
            for (uint64_t j = row_ptr[r]; j < row_ptr[r + 1]; ++j) {
                const uint32_t c = tid[j] - lo; // wraps for tid < lo
                if (c < win) { ++nloc; max_code = std::max(max_code, c); }
                else { ++sz.remote_cnt; ++cnt_tb[(size_t)ti * n_buckets + tid[j] / kBucket]; }
            }
            rows[i] = {nloc, key[r], 0, r};
        }
        // Slices want reads of equal length (SELL padding); within one length, reads with
        // the same anchor are dealt round-robin so that the 64 lanes of a wavefront add
        // into different window entries (same-address LDS atomics serialise).
        std::stable_sort(rows.begin(), rows.end(), [](const RowKey &a, const RowKey &b) {
            return a.nloc != b.nloc ? a.nloc > b.nloc : a.anchor < b.anchor;
        });
        for (uint32_t i = 1; i < n; ++i)
            if (rows[i].nloc == rows[i - 1].nloc && rows[i].anchor == rows[i - 1].anchor)
                rows[i].rank = rows[i - 1].rank + 1;
        std::stable_sort(rows.begin(), rows.end(), [](const RowKey &a, const RowKey &b) {
            return a.nloc != b.nloc ? a.nloc > b.nloc : a.rank < b.rank;
        });
        for (uint32_t i = 0; i < n; ++i) out->perm[p0 + i] = rows[i].r;
        sz.n_slices = (n + 63) / 64;
        for (uint32_t s = 0; s < sz.n_slices; ++s) {
            const uint32_t width = rows[s * 64].nloc; // sorted descending: first read is the longest
            sz.w_slots += width;
            sz.c_slots += (width + 1) / 2;
        }
        sz.win_len = max_code + 1;
        if (rows[0].nloc > 255) too_wide.store(true);
        sizes[ti] = sz;
    });

    if (too_wide.load()) {
        *err = "a read has more than 255 alignments inside one tile window";
        return false;
    }
    lap("pass 1 (sizes)");
    // 5. offsets
    uint64_t w_slots = 0, c_slots = 0, n_remote = 0;
    std::vector<uint64_t> w_base(n_tiles), c_base(n_tiles);
    for (uint32_t ti = 0; ti < n_tiles; ++ti) {
        TileDesc &td = out->tiles[ti];
        td.n_slices = sizes[ti].n_slices;
        td.w_base = (uint32_t)w_slots;
        td.c_base = (uint32_t)c_slots;
        td.n_rows = tile_start[ti + 1] - tile_start[ti];
        td.row_base = tile_start[ti];
        td.lo = tile_lo[ti];
        td.win_len = tile_win[ti];
        td.problem = problem_size ? key[order[tile_start[ti]]] / problem_size : 0;
        td.remote_begin = (uint32_t)n_remote;
        td.remote_cnt = sizes[ti].remote_cnt;
        w_base[ti] = w_slots;
        c_base[ti] = c_slots;
        w_slots += sizes[ti].w_slots;
        c_slots += sizes[ti].c_slots;
        n_remote += sizes[ti].remote_cnt;
    }
    if (w_slots >= (1ull << 32) || c_slots >= (1ull << 32) || n_remote >= (1ull << 32)) {
        *err = "store too large for 32-bit tile offsets";
        return false;
    }
    out->n_remote = n_remote;
    // one row of slack: the pair loads of an odd-width slice touch the next row
    // (not zero-filled here: every tile clears and fills its own range in pass 2)
    out->codes.resize((c_slots + 1) * 64);
    if (cov_prob) out->w64.resize((w_slots + 1) * 64);
    else out->w32.resize((w_slots + 1) * 64);
    std::memset(out->codes.data() + c_slots * 64, 0, 64 * sizeof(uint32_t)); // the slack row
    if (cov_prob) std::memset(out->w64.data() + w_slots * 64, 0, 64 * sizeof(double));
    else std::memset(out->w32.data() + w_slots * 64, 0, 64 * sizeof(float));
    out->r_tid.resize(n_remote);
    if (cov_prob) out->r_w64.resize(n_remote);
    else out->r_w32.resize(n_remote);
    out->r_row.resize(n_remote);
    out->r_slot.resize(n_remote);
    out->q_dst.resize(n_remote);
    // bucket-major queue: slot_base(tile, b) = bucket_base[b] + remote alignments of earlier tiles in b
    // (two row-major sweeps of the tile x bucket table)
    out->bucket_base.assign(n_buckets + 1, 0u);
    {
        std::vector<uint64_t> run(n_buckets, 0);
        for (uint32_t ti = 0; ti < n_tiles; ++ti) {
            const uint32_t *row = &cnt_tb[(size_t)ti * n_buckets];
            for (uint32_t b = 0; b < n_buckets; ++b) run[b] += row[b];
        }
        uint64_t acc = 0;
        for (uint32_t b = 0; b < n_buckets; ++b) {
            out->bucket_base[b] = (uint32_t)acc;
            const uint64_t n = run[b];
            run[b] = acc; // running slot base of the bucket
            acc += n;
        }
        out->bucket_base[n_buckets] = (uint32_t)acc;
        for (uint32_t ti = 0; ti < n_tiles; ++ti) {
            uint32_t *row = &cnt_tb[(size_t)ti * n_buckets];
            for (uint32_t b = 0; b < n_buckets; ++b) {
                const uint32_t n = row[b];
                row[b] = (uint32_t)run[b]; // becomes the slot base of (tile, bucket)
                run[b] += n;
            }
        }
    }

    lap("offsets");
    // 6. pass 2 (parallel over tiles): fill slices and remote records
    std::atomic<uint64_t> n_local{0};
    parallel_for(n_tiles, [&](uint32_t ti) {
        const TileDesc &td = out->tiles[ti];
        const uint32_t lo = td.lo, win = td.win_len;
        uint64_t woff = w_base[ti], coff = c_base[ti];
        // clear this tile's slices (padding lanes carry weight 0 and code 0; codes are OR-ed in)
        std::memset(out->codes.data() + coff * 64, 0, (size_t)sizes[ti].c_slots * 64 * sizeof(uint32_t));
        if (cov_prob) std::memset(out->w64.data() + woff * 64, 0, (size_t)sizes[ti].w_slots * 64 * sizeof(double));
        else std::memset(out->w32.data() + woff * 64, 0, (size_t)sizes[ti].w_slots * 64 * sizeof(float));
        std::vector<RemoteRec> rem;
        rem.reserve(td.remote_cnt);
        uint64_t nl = 0;
        TileDesc &tdw = out->tiles[ti];
        for (uint32_t s = 0; s < td.n_slices; ++s) {
            const uint32_t row0 = s * 64;
            const uint32_t lanes = std::min(64u, td.n_rows - row0);
            // width = local count of the first (longest) read
            uint32_t width = 0;
            {
                const uint32_t r = out->perm[td.row_base + row0];
                for (uint64_t j = row_ptr[r]; j < row_ptr[r + 1]; ++j)
                    if (tid[j] - lo < win) ++width;
            }
            tdw.width[s] = (uint8_t)width;
            for (uint32_t lane = 0; lane < lanes; ++lane) {
                const uint32_t rl = row0 + lane;
                const uint32_t r = out->perm[td.row_base + rl];
                // local alignments of the read: anchor first, the others by ascending transcript
                uint64_t loc[128];
                uint32_t jl = 0;
                for (uint64_t j = row_ptr[r]; j < row_ptr[r + 1]; ++j) {
                    if (tid[j] - lo < win) {
                        if (jl < 128) loc[jl] = j;
                        ++jl;
                    } else {
                        rem.push_back(RemoteRec{tid[j], (uint16_t)rl, j});
                    }
                }
                std::vector<uint64_t> big;
                uint64_t *lp = loc;
                if (jl > 128) { // longer than --best-n allows in the reference; keep it correct anyway
                    big.reserve(jl);
                    for (uint64_t j = row_ptr[r]; j < row_ptr[r + 1]; ++j)
                        if (tid[j] - lo < win) big.push_back(j);
                    lp = big.data();
                }
                std::sort(lp, lp + jl, [&](uint64_t a, uint64_t b) {
                    const bool aa = tid[a] == key[r], ab = tid[b] == key[r];
                    if (aa != ab) return aa;
                    return tid[a] != tid[b] ? tid[a] < tid[b] : a < b;
                });
                for (uint32_t q = 0; q < jl; ++q) {
                    const uint64_t j = lp[q];
                    const uint32_t c = tid[j] - lo;
                    const uint64_t wi = (woff + q) * 64 + lane;
                    if (cov_prob) out->w64[wi] = (double)as_prob[j] * cov_prob[j];
                    else out->w32[wi] = as_prob[j];
                    out->codes[(coff + q / 2) * 64 + lane] |= (c * 8u) << (16 * (q & 1)); // LDS byte offset
                }
                nl += jl;
            }
            woff += width;
            coff += (width + 1) / 2;
        }
        n_local.fetch_add(nl);
        // remote records ordered by destination bucket (then transcript)
        std::sort(rem.begin(), rem.end(), [](const RemoteRec &a, const RemoteRec &b) {
            return a.tid != b.tid ? a.tid < b.tid : a.j < b.j;
        });
        uint32_t cur_b = 0xffffffffu, slot = 0;
        for (uint32_t i = 0; i < rem.size(); ++i) {
            const uint64_t o = (uint64_t)td.remote_begin + i, j = rem[i].j;
            const uint32_t b = rem[i].tid / kBucket;
            if (b != cur_b) { cur_b = b; slot = cnt_tb[(size_t)ti * n_buckets + b]; }
            out->r_tid[o] = rem[i].tid;
            if (cov_prob) out->r_w64[o] = (double)as_prob[j] * cov_prob[j];
            else out->r_w32[o] = as_prob[j];
            out->r_row[o] = rem[i].row;
            out->r_slot[o] = slot;
            out->q_dst[slot] = (uint16_t)(rem[i].tid % kBucket);
            ++slot;
        }
    });
    out->n_local = n_local.load();
    lap("pass 2 (fill)");
    return true;
}

} // namespace oem
