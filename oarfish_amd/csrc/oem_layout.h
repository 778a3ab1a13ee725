// oem_layout.h -- the tiled HBM layout of the alignment store ("v2").
//
// The reference keeps reads in arrival order and scatters into one shared
// Vec<AtomicF64> (em.rs:338-341).  On MI355X random device-scope f64 atomics
// run at ~24 G/s and a single hot address at ~0.08 G/s (profiles/
// r01_microbench_*), so the store is re-laid out once at upload such that
// almost every accumulation happens in LDS:
//
//   * reads are ordered by their *anchor transcript* (the alignment that has most
//     of the read's other alignments within +-kMargin ids) and cut into tiles of <= kTileRows reads whose primaries
//     span less than the tile's LDS window of kWin transcripts;
//   * inside a tile the reads are ordered by alignment count and stored as
//     SELL-64 slices: slice s holds 64 reads, one per lane, column-major, so a
//     wavefront walks 64 reads with fully coalesced loads and no cross-lane
//     reduction (entry j of lane l at (off + j) * 64 + l);
//   * an alignment whose transcript falls inside the tile window is "local":
//     16-bit window code + weight, accumulated with LDS atomics;
//   * the others are "remote": kept per tile, ordered by destination bucket
//     (kBucket transcripts each).  The tile kernel writes their increments into
//     a queue that is laid out bucket-major (slot of every remote alignment
//     fixed at upload; runs of one (tile, bucket) pair are contiguous), and a
//     second kernel streams each bucket's queue range into an LDS window.
//
// EM results are invariant to the order of reads (em.rs:97 sums over all reads)
// up to floating-point summation order.
#pragma once

#include <stdint.h>

#include <sys/mman.h>

#include <cstdlib>
#include <memory>
#include <new>
#include <utility>
#include <vector>

namespace oem {

constexpr uint32_t kTileRows = 1024;  // reads per tile (16 slices of 64)
constexpr uint32_t kWin = 512;        // transcripts per tile window (2 x 4 KiB of LDS); 8*kWin must fit 16 bits
constexpr uint32_t kWinWide = 2048;   // window cap of sparse stores as oem_store_opts.window_cap names it (few reads per transcript: per-cell batches)
#ifndef OEM_WIN_WIDE_LDS
#define OEM_WIN_WIDE_LDS 1984
#endif
constexpr uint32_t kWinWideLds = OEM_WIN_WIDE_LDS; // ... and what their tiles are cut for: theta + count window (2 x 15.5 KiB) + the per-read
                                       // denominators (8 KiB) + the 1 KiB weight table = 40 KiB, four workgroups per CU
constexpr uint32_t kMargin = 64;      // window slack on both sides of the primaries
constexpr uint32_t kBucket = 4096;    // transcripts per remote bucket (32 KiB of LDS; x 4 slots = 128 KiB in the batched bootstrap's fold)
constexpr uint32_t kBucketShift = 12;
static_assert((1u << kBucketShift) == kBucket, "bucket of a transcript = id >> kBucketShift");
constexpr uint32_t kPackRowShift = 22; // packed remote record: (transcript - problem base) | read-in-tile << 22
static_assert((kTileRows - 1) >> (32 - kPackRowShift) == 0, "the read index must fit above the transcript bits");

constexpr uint32_t kTileSlices = kTileRows / 64;

// Reads per tile by store size.  A pass over a small store is ONE round of workgroups and lasts as long as a tile
// lives; a wavefront walks its slices one after another, so a 125 k-read store (the row shard of one of eight ranks
// at 1 M reads) is 160 workgroups of four slices per wavefront on a chip with 1280 resident slots.  Cut into
// 256-read tiles the same store is 489 workgroups of one slice per wavefront: iteration 22.3 -> 20.0 us (250 k reads:
// 23.6 -> 22.5).  A tile costs ~7 us whatever it holds (descriptor, window load / clear / flush, three barriers,
// the remote round trips), so from 500 k reads on -- where the full tiles already cover the chip -- finer tiles
// only LOSE: 1 M reads 36.3 -> 39.5 -> 46.3 us at 512 / 256 reads per tile, 10 M reads 168 -> 213 -> 321 us
// (scripts/tile_rows_exp.py, profiles/r04_notes.md).
#ifndef OEM_TILE_ROWS_LARGE
#define OEM_TILE_ROWS_LARGE kTileRows // (A/B builds: -DOEM_TILE_ROWS_LARGE=512)
#endif
inline uint32_t tile_rows_for(uint64_t n_reads) { return n_reads <= (1ull << 18) ? 256u : OEM_TILE_ROWS_LARGE; }

// Everything a workgroup needs to know about its tile, fetched with one scalar
// load: with the slice widths in hand every wavefront derives the addresses of
// all its slices without a dependent descriptor load.
struct TileDesc { // 64 bytes
    uint32_t n_rows;       // reads in the tile
    uint32_t row_base;     // position of the tile's first read in the permuted order
    uint32_t lo;           // first transcript of the window
    uint32_t win_len;      // window entries actually used (<= kWin)
    uint32_t remote_begin; // first remote record of the tile
    uint32_t remote_cnt;
    uint32_t w_base;       // weights of slice s start at (w_base + sum_{i<s} width[i]) * 64
    uint32_t c_base;       // codes   of slice s start at (c_base + sum_{i<s} (width[i]+1)/2) * 64
    uint8_t width[kTileSlices]; // max local alignments of the 64 reads of each slice (0 = no slice)
    uint32_t n_slices;
    uint32_t problem;      // independent EM problem the tile belongs to (per-cell batches); 0 otherwise
    // filled by pack_remote_records (oem_layout_pack.hip), zero as the builders leave the descriptor:
    uint32_t sd_begin;     // the tile's slot table: sd[sd_begin + bucket - b_min] + (record index in the tile) = queue slot
    uint32_t b_min;        // bucket of the tile's first remote record
};
static_assert(sizeof(TileDesc) == 64, "TileDesc layout");
static_assert(kTileSlices == 16, "TileDesc::width is sized for 16 slices");

// Slice data: weights w[(off + j) * 64 + lane], j < width; codes packed in pairs,
// c[(coff + j/2) * 64 + lane] >> 16*(j&1) & 0xffff = 8 * (tid - lo) (an LDS byte offset).

// The big arrays are resized without being zero-filled (a serial memset of ~1 GB at 10 M reads);
// the layout pass writes every element it owns, padding included, from the thread that fills it.
// Large blocks are 2 MiB-aligned and advised towards transparent huge pages: the first touch of a
// fresh 4 KiB page from 100+ threads serialises in the kernel and dominated the fill pass.
template <typename T>
struct DefaultInitAlloc {
    using value_type = T;
    DefaultInitAlloc() = default;
    template <typename U> DefaultInitAlloc(const DefaultInitAlloc<U> &) noexcept {}
    template <typename U> struct rebind { using other = DefaultInitAlloc<U>; };
    T *allocate(size_t n)
    {
        const size_t bytes = n * sizeof(T);
        void *p = nullptr;
        if (bytes >= (4u << 20)) {
            const size_t huge = 2u << 20;
            p = std::aligned_alloc(huge, (bytes + huge - 1) / huge * huge);
            if (p) madvise(p, bytes, MADV_HUGEPAGE);
        } else {
            p = std::malloc(bytes ? bytes : 1);
        }
        if (!p) throw std::bad_alloc();
        return static_cast<T *>(p);
    }
    void deallocate(T *p, size_t) noexcept { std::free(p); }
    template <typename U> bool operator==(const DefaultInitAlloc<U> &) const noexcept { return true; }
    template <typename U> bool operator!=(const DefaultInitAlloc<U> &) const noexcept { return false; }
    template <typename U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <typename U, typename... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
template <typename T> using RawVec = std::vector<T, DefaultInitAlloc<T>>;

// Host-side result of the layout pass; all arrays are uploaded verbatim.
struct TiledHost {
    uint32_t n_tiles = 0;
    uint32_t n_buckets = 0;
    uint32_t win_cap = kWin; // window cap the tiles were cut for (kWin or kWinWide)
    uint64_t n_rows = 0;    // non-empty reads
    uint64_t n_local = 0;   // local alignments
    uint64_t n_remote = 0;  // remote alignments
    std::vector<TileDesc> tiles;
    std::vector<uint32_t> perm;     // permuted position -> original read index
    RawVec<uint32_t> codes;       // packed pairs of 16-bit window codes
    RawVec<float> w32;               // local weights (coverage off)
    RawVec<double> w64;             // local weights (coverage on)
    RawVec<uint32_t> r_tid;       // remote: transcript
    RawVec<float> r_w32;           // remote: weight
    RawVec<double> r_w64;
    RawVec<uint16_t> r_row;       // remote: read index inside its tile
    RawVec<uint32_t> r_slot;     // remote: slot in the bucket-major queue
    RawVec<uint16_t> q_dst;       // queue order: transcript - bucket * kBucket
    std::vector<uint32_t> bucket_base; // [n_buckets + 1]: queue range of every bucket
};

// Builds the layout from the caller's CSR.  w = (f64)as_prob * cov_prob when
// cov_prob != nullptr (em.rs:107-111), else as_prob (exact f32).
// Returns false (with `err`) if the store cannot be tiled (n_reads >= 2^32).
// `problem_size` > 0 declares the transcript space to be the concatenation of independent
// problems of that many transcripts each (per-cell EM, single_cell.rs:139-160): tiles then
// never mix reads of two problems.  `win_cap` (kWin or kWinWide) bounds the LDS window of a tile: sparse
// stores get more reads per tile from a wider window (the kernels then keep one count-window copy).
// `tile_rows` (a multiple of 64, <= kTileRows) caps the reads of a tile: small stores are cut finer so that
// their tiles still cover the chip (tile_rows_for above).
bool build_tiled_layout(const uint64_t *row_ptr, const uint32_t *tid, const float *as_prob,
                        const double *cov_prob, uint64_t n_reads, uint64_t nnz, uint32_t n_txps,
                        TiledHost *out, const char **err, uint32_t problem_size = 0, uint32_t win_cap = kWin,
                        uint32_t tile_rows = kTileRows);

} // namespace oem
