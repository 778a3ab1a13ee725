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

#include <vector>

namespace oem {

constexpr uint32_t kTileRows = 1024;  // reads per tile (16 slices of 64)
constexpr uint32_t kWin = 512;        // transcripts per tile window (2 x 4 KiB of LDS); 8*kWin must fit 16 bits
constexpr uint32_t kMargin = 64;      // window slack on both sides of the primaries
constexpr uint32_t kBucket = 8192;    // transcripts per remote bucket (64 KiB of LDS)

constexpr uint32_t kTileSlices = kTileRows / 64;

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
    uint32_t pad[2];
};
static_assert(sizeof(TileDesc) == 64, "TileDesc layout");
static_assert(kTileSlices == 16, "TileDesc::width is sized for 16 slices");

// Slice data: weights w[(off + j) * 64 + lane], j < width; codes packed in pairs,
// c[(coff + j/2) * 64 + lane] >> 16*(j&1) & 0xffff = 8 * (tid - lo) (an LDS byte offset).

// Host-side result of the layout pass; all arrays are uploaded verbatim.
struct TiledHost {
    uint32_t n_tiles = 0;
    uint32_t n_buckets = 0;
    uint64_t n_rows = 0;    // non-empty reads
    uint64_t n_local = 0;   // local alignments
    uint64_t n_remote = 0;  // remote alignments
    std::vector<TileDesc> tiles;
    std::vector<uint32_t> perm;     // permuted position -> original read index
    std::vector<uint32_t> codes;    // packed pairs of 16-bit window codes
    std::vector<float> w32;         // local weights (coverage off)
    std::vector<double> w64;        // local weights (coverage on)
    std::vector<uint32_t> r_tid;    // remote: transcript
    std::vector<float> r_w32;       // remote: weight
    std::vector<double> r_w64;
    std::vector<uint16_t> r_row;    // remote: read index inside its tile
    std::vector<uint32_t> r_slot;   // remote: slot in the bucket-major queue
    std::vector<uint16_t> q_dst;    // queue order: transcript - bucket * kBucket
    std::vector<uint32_t> bucket_base; // [n_buckets + 1]: queue range of every bucket
};

// Builds the layout from the caller's CSR.  w = (f64)as_prob * cov_prob when
// cov_prob != nullptr (em.rs:107-111), else as_prob (exact f32).
// Returns false (with `err`) if the store cannot be tiled (n_reads >= 2^32).
// `problem_size` > 0 declares the transcript space to be the concatenation of independent
// problems of that many transcripts each (per-cell EM, single_cell.rs:139-160): tiles then
// never mix reads of two problems.
bool build_tiled_layout(const uint64_t *row_ptr, const uint32_t *tid, const float *as_prob,
                        const double *cov_prob, uint64_t n_reads, uint64_t nnz, uint32_t n_txps,
                        TiledHost *out, const char **err, uint32_t problem_size = 0);

} // namespace oem
