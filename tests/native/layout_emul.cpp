// Test helper (NOT part of the product): builds the tiled layout with the product's
// own oem_layout.cpp and replays the two tile kernels' arithmetic on the host, so
// the layout pass and the tile algorithm can be checked against the oracle on a
// box without a GPU.  Compiled by tests/test_layout.py with g++.
#include "../../oarfish_amd/csrc/oem_layout.cpp"

#include <cstdio>

using namespace oem;

extern "C" int layout_emul_m_step(const uint64_t *row_ptr, const uint32_t *tid, const float *as_prob,
                                  const double *cov_prob, uint64_t n_reads, uint64_t nnz,
                                  uint32_t n_txps, const double *theta, const uint32_t *row_w,
                                  double *cnt, uint64_t *stats /* n_tiles, n_local, n_remote, n_rows, w_slots */,
                                  uint32_t problem_size)
{
    TiledHost h;
    const char *err = nullptr;
    const char *wc = getenv("LAYOUT_EMUL_WIN_CAP"); // tests pick the window cap (kWin or kWinWide)
    if (!build_tiled_layout(row_ptr, tid, as_prob, cov_prob, n_reads, nnz, n_txps, &h, &err, problem_size,
                            wc ? (uint32_t)atoi(wc) : kWin))
        return 1;
    if (problem_size) // a tile never mixes reads of two problems
        for (uint32_t ti = 0; ti < h.n_tiles; ++ti) {
            const TileDesc &td = h.tiles[ti];
            for (uint32_t i = 0; i < td.n_rows; ++i) {
                const uint32_t r = h.perm[td.row_base + i];
                for (uint64_t j = row_ptr[r]; j < row_ptr[r + 1]; ++j)
                    if (tid[j] / problem_size != td.problem) return 11;
            }
        }
    const bool f64 = cov_prob != nullptr;
    std::vector<double> queue(h.n_remote, -1.0);
    std::vector<uint8_t> slot_seen(h.n_remote, 0);
    // every read appears exactly once in perm
    {
        std::vector<uint8_t> seen(n_reads, 0);
        for (uint32_t r : h.perm) { if (r >= n_reads || seen[r]) return 2; seen[r] = 1; }
        for (uint64_t r = 0; r < n_reads; ++r)
            if (!seen[r] && row_ptr[r + 1] != row_ptr[r]) return 3;
    }
    for (uint32_t ti = 0; ti < h.n_tiles; ++ti) {
        const TileDesc &td = h.tiles[ti];
        if (td.win_len > h.win_cap || h.win_cap > kWinWide || td.n_rows > kTileRows || td.n_slices * 64 < td.n_rows) return 4;
        std::vector<double> theta_l(kWinWide, 0.0 / 0.0), cnt_l(kWinWide, 0.0), den_l(kTileRows, 0.0);
        for (uint32_t i = 0; i < td.win_len; ++i) { theta_l[i] = theta[td.lo + i]; }
        for (uint32_t i = 0; i < td.remote_cnt; ++i) {
            const uint32_t o = td.remote_begin + i;
            const double w = f64 ? h.r_w64[o] : (double)h.r_w32[o];
            const double x = theta[h.r_tid[o]] * w;
            if (h.r_slot[o] >= h.n_remote || slot_seen[h.r_slot[o]]) return 5;
            slot_seen[h.r_slot[o]] = 1;
            queue[h.r_slot[o]] = x;
            if (h.r_row[o] >= td.n_rows) return 6;
            den_l[h.r_row[o]] += x;
        }
        uint32_t w_off = td.w_base, c_off = td.c_base;
        for (uint32_t s = 0; s < td.n_slices; ++s) {
            struct { uint32_t w_off, c_off, width; } sd = {w_off, c_off, td.width[s]};
            w_off += sd.width;
            c_off += (sd.width + 1) / 2;
            for (uint32_t lane = 0; lane < 64; ++lane) {
                const uint32_t rl = s * 64 + lane;
                double denom = den_l[rl];
                for (uint32_t j = 0; j < sd.width; ++j) {
                    const uint32_t cc = h.codes[((size_t)sd.c_off + j / 2) * 64 + lane];
                    const uint32_t c = ((cc >> (16 * (j & 1))) & 0xffffu) / 8u;
                    const size_t wi = ((size_t)sd.w_off + j) * 64 + lane;
                    const double w = f64 ? h.w64[wi] : (double)h.w32[wi];
                    if (c >= td.win_len) return 7;
                    denom += theta_l[c] * w;
                }
                double scale = 1.0;
                if (row_w) scale = rl < td.n_rows ? (double)row_w[h.perm[td.row_base + rl]] : 0.0;
                const double inv = denom > 1e-30 ? scale / denom : 0.0;
                den_l[rl] = inv;
                if (inv != 0.0)
                    for (uint32_t j = 0; j < sd.width; ++j) {
                        const uint32_t cc = h.codes[((size_t)sd.c_off + j / 2) * 64 + lane];
                        const uint32_t c = ((cc >> (16 * (j & 1))) & 0xffffu) / 8u;
                        const size_t wi = ((size_t)sd.w_off + j) * 64 + lane;
                        const double w = f64 ? h.w64[wi] : (double)h.w32[wi];
                        if (w != 0.0) cnt_l[c] += theta_l[c] * w * inv;
                    }
            }
        }
        for (uint32_t i = 0; i < td.remote_cnt; ++i) {
            const uint32_t o = td.remote_begin + i;
            queue[h.r_slot[o]] *= den_l[h.r_row[o]];
        }
        for (uint32_t i = 0; i < td.win_len; ++i) cnt[td.lo + i] += cnt_l[i];
    }
    if (h.bucket_base.size() != h.n_buckets + 1 || h.bucket_base[h.n_buckets] != h.n_remote) return 8;
    for (uint32_t b = 0; b < h.n_buckets; ++b)
        for (uint32_t o = h.bucket_base[b]; o < h.bucket_base[b + 1]; ++o) {
            if (!slot_seen[o]) return 9;
            const uint32_t t = b * kBucket + h.q_dst[o];
            if (t >= n_txps) return 10;
            cnt[t] += queue[o];
        }
    if (stats) {
        stats[0] = h.n_tiles; stats[1] = h.n_local; stats[2] = h.n_remote; stats[3] = h.n_rows;
        stats[4] = (f64 ? h.w64.size() : h.w32.size()) - 64;
    }
    return 0;
}
