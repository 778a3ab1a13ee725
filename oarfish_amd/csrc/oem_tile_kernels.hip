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
#include <cstdlib>
#include <vector>

#include "oem_internal.h"

// Phase timestamps of k_em_tile (test-only library): wave 0 of every workgroup stamps the device wall clock
// (100 MHz) at its phase boundaries into g_tile_probe[tile][16] -- scripts/tile_probe.py turns them into the
// per-phase account of profiles/r03_notes.md.  The product build has no probe code at all.
// Cost attribution (test-only library, OEM_TILE_EXP): parts of the kernel switched off -- wrong results, the time
// says what the part costs.  1 queue stores, 2 remote denominator atomics, 4 local scatter atomics, 8 local theta
// reads, 16 remote theta gathers, 32 window flush as plain stores, 64 no window flush
#ifdef OEM_TESTING
#define OEM_PROBE(i)                                                                                          \
    do {                                                                                                      \
        if (g_tile_probe && threadIdx.x == 0) g_tile_probe[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); \
    } while (0)
#define OEM_EXP(bit) ((exp_mask & (bit)) != 0u) // (exp_mask: g_tile_exp read once per kernel, an SGPR)
// the kernel cut short behind a phase (2048: at once, 1024: with the descriptor in hand, 128: behind the first barrier,
// 256: the second, 512: the third -- no queue stores, no flush): what the tiles cost up to there, at the real
// occupancy; 4096: no slice is requested, 8192: no record (scripts/tile_phase_exp.sh)
#define OEM_EXIT_AT(bit)                                                \
    do {                                                                \
        if (OEM_EXP(bit)) {                                             \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
            return;                                                     \
        }                                                               \
    } while (0)
#else
#define OEM_EXIT_AT(bit) do { } while (0)
#define OEM_PROBE(i) do { } while (0)
#define OEM_EXP(bit) false
#endif

#include "oem_tile_common.h"

namespace oem {

namespace {

constexpr int kFoldThreadsDefault = 1024;
constexpr int kFoldThreads = kFoldThreadsDefault; // (the launchers' block size; the kernel takes it as a template argument: the overlap probe folds with 256)
constexpr uint32_t kFoldEntriesPerGroup = 2 * kBucket; // queue entries that repay a fold workgroup's window clear + flush

#ifdef OEM_TESTING
__device__ unsigned long long *g_tile_probe = nullptr;
__device__ unsigned int g_tile_exp = 0;
#endif

template <typename WT, int kCh, int kRem, int kTileThreads, int kMinWaves, int kCopies, bool kNT, uint32_t kWinT, bool kPacked, int kDict, int kSetsT>
__global__ __launch_bounds__(kTileThreads, kMinWaves) void k_em_tile(
    const TileDesc *__restrict__ tiles, const uint32_t *__restrict__ codes,
    const WT *__restrict__ w, const uint32_t *__restrict__ r_a, const WT *__restrict__ r_w,
    const uint16_t *__restrict__ r_row, const uint32_t *__restrict__ sd,
    double *__restrict__ queue, const double *__restrict__ theta, double *__restrict__ cnt,
    const EmState *state, const uint32_t *__restrict__ row_w_perm,
    const BatchState *__restrict__ problems, uint32_t problem_size, uint32_t n_tiles,
    const uint32_t *__restrict__ widx, const uint32_t *__restrict__ i_base, const float *__restrict__ dict,
    const uint8_t *__restrict__ r_wi, const uint32_t *__restrict__ live_tiles,
    double *__restrict__ rd_prev, unsigned long long *__restrict__ rd_slots, uint32_t rd_chunk, uint32_t n_txps)
{
    __shared__ float dict_l[dict_entries<kDict>()]; // the distinct weights of a coded store (oem_layout_dict.hip)
    __shared__ double theta_l[kWinT]; // kWin, or kWinWideLds with one count-window copy (sparse stores)
    constexpr uint32_t kCntEntries = (kCopies > 1 && OEM_CNT_ENTRIES) ? OEM_CNT_ENTRIES : kWinT * (uint32_t)kCopies;
    __shared__ double cnt_l[kCntEntries];
    __shared__ double den_l[kTileRows]; // remote part of the denominators, then c_i/denom_i
#ifdef OEM_LDS_PAD // (A/B: fewer workgroups per CU by LDS)
    __shared__ double pad_l[OEM_LDS_PAD / 8];
    pad_l[threadIdx.x * 7 % (OEM_LDS_PAD / 8)] = 1.0;
    asm volatile("" ::"v"(pad_l[(threadIdx.x * 13 + 5) % (OEM_LDS_PAD / 8)]));
#endif

    OEM_PROBE(0);
#ifdef OEM_TESTING
    const uint32_t exp_mask = g_tile_exp;
#else
    constexpr uint32_t exp_mask = 0u;
#endif
    OEM_EXIT_AT(2048u); // (an empty workgroup: what dispatching the grid costs)
    // (the descriptor is requested before the run's state is looked at: two scalar loads in flight, not a chain)
    // Per-cell batch: the abundances of ALL cells together (300 MB for 625 cells) fit no cache, those of the cells
    // one XCD is working on do (a cell's 60 k transcripts are 480 KB of its 4 MiB L2) -- if the XCD works on few
    // cells at a time.  Block b runs on XCD b % 8 (observed placement: a matter of speed only), so block b takes
    // tile (b % 8) * n_tiles / 8 + b / 8: every XCD streams through its own eighth of the cells in order, and the
    // remote gathers hit its L2 instead of fetching a line from memory each (HBM reads of the kernel -45 %).
    // `live_tiles` (per-cell batch, later in the loop): the tiles of the cells that were unfinished when the host
    // last compacted the list, in tile order -- n_tiles is then the length of that list, and the grid follows it,
    // so a pass costs what its live cells cost (oem_multi_kernels.hip: k_multi_compact).
    uint32_t tile_index = blockIdx.x;
    // The rel-diff of the PREVIOUS iteration rides along (DeferredRelDiff, oem_internal.h) on workgroups of its own, the
    // LAST kRdBlocks of the grid: theta_{i-1} (`rd_prev`) against theta_i (what this pass reads as theta), their share of
    // the transcripts each, theta_{i-1} zeroed (it is the accumulator of pass i + 1: em.rs:207).  Dispatched behind
    // the last tile, they run in the slots the tile kernel's tail leaves idle.  (Rounds 5-6 gave every tile workgroup
    // a share, requested with its theta window and looked at behind its first barrier: 1.5-2 us of a 141 us iteration
    // at 10 M reads, 0.9 of 27.5 us at 1 M -- two more loads in front of every tile's records, and wave 0 late for
    // the second barrier.)  A workgroup must look at `done` BEFORE it zeroes anything: run_em_deferred.
    if (!problems && blockIdx.x >= n_tiles) {
        if (!rd_prev || (state && state->done)) return;
        const uint32_t rb = blockIdx.x - n_tiles;
        const uint32_t i0 = rb * rd_chunk, i1 = i0 + rd_chunk < n_txps ? i0 + rd_chunk : n_txps;
        double rel = 0.0;                                          // em.rs:194-201 (signed, floored at 0 by the maximum)
        for (uint32_t ib = i0 + threadIdx.x; ib < i1; ib += 4 * kTileThreads) {
            double pv[4], cv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { // (eight loads in flight before the first use)
                const uint32_t i = ib + k * kTileThreads, ic = i < i1 ? i : ib;
                pv[k] = rd_prev[ic];
                cv[k] = theta[ic];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t i = ib + k * kTileThreads;
                if (i < i1) {
                    if (pv[k] > OEM_MIN_READ_THRESH) rel = fmax(rel, (cv[k] - pv[k]) / pv[k]);
                    rd_prev[i] = 0.0;
                }
            }
        }
        for (int off = 32; off > 0; off >>= 1) rel = fmax(rel, __shfl_xor(rel, off, 64));
        if ((threadIdx.x & 63u) == 0 && rel > 0.0) // (non-negative doubles order like their bit patterns; no return value: nothing waits)
            atomicMax(&rd_slots[(rb * (kTileThreads / 64) + (threadIdx.x >> 6)) & (kRelSlots - 1u)], (unsigned long long)__double_as_longlong(rel));
        return;
    }
    if (problems) {
        tile_index = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
        if (tile_index >= n_tiles) return;
        if (live_tiles) tile_index = live_tiles[tile_index];
    }
    const TileDesc td = tiles[tile_index]; // one 64-byte scalar load
    const uint32_t ib = kDict == kWBytes ? i_base[tile_index] : 0u; // (requested with it)
    if (state && state->done) return;
    // per-cell batch: a FINISHED cell takes no part; a cell on its FINAL pass reads abundances below the
    // threshold as 0 (em.rs:238-242) -- done here, on the way in, instead of by a sweep over theta per pass
    const uint32_t phase = problems ? problems[td.problem].phase : (uint32_t)kPhaseRunning;
    if (phase == kPhaseFinished) return;
    const bool fin = phase == kPhaseFinal; // wave-uniform
    auto th = [&](double v) -> double { return (fin && v < OEM_MIN_READ_THRESH) ? 0.0 : v; };
    const uint32_t tx = threadIdx.x;
    const uint32_t lane = tx & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tx >> 6); // SGPR: slice control flow is scalar
    // Copies of the count window: as many as the tile's window leaves room for in the kWinT * kCopies entries of
    // cnt_l (a power of two, at most 1 << kMaxCopyShift).  A tile of a dense store spans few transcripts (C3: 1024
    // reads are ~20 transcripts + the margins, a window of ~150 of the 512 entries), and those are the tiles whose
    // lanes add into the same few entries: halving the copies (2 for 4) cost 21 % of the C3 pass, so the tiles that
    // can take 8 (C3 pass 0.1492 -> 0.146-0.147 ms; 16 or 32 copies: the same, profiles/r04_notes.md).
    uint32_t cs = 0;
    while (cs < kMaxCopyShift && (td.win_len << (cs + 1)) <= kCntEntries) ++cs;
    constexpr uint32_t kWaves = kTileThreads / 64;
    constexpr uint32_t kPerWave = kTileSlices / kWaves; // slices per wavefront
    // A tile's slices come in descending width, so dealing them round-robin gives wavefront 0 the widest of every
    // group of kWaves (37 alignment rows against wavefront 3's 27 at 8 alignments per read) and the tile waits for
    // it at the barrier.  Dealt boustrophedon -- wave, 2 kWaves - 1 - wave, 2 kWaves + wave, ... -- the widest
    // slice still comes first (fold_first) and the sums are within a few rows of each other.
    auto slice_of = [&](uint32_t q) -> uint32_t { return (q & 1u) ? (q + 1) * kWaves - 1 - wave : q * kWaves + wave; };

    // addresses of this wavefront's slices, from the widths alone (scalar prefix sums)
    uint32_t woff[kPerWave], coff[kPerWave], wid[kPerWave], ioff[kPerWave];
    {
        uint32_t accw = td.w_base, accc = td.c_base, acci = ib;
#pragma unroll
        for (uint32_t i = 0; i < kTileSlices; ++i) {
            const uint32_t wi = td.width[i];
            const uint32_t q = i / kWaves; // the group of kWaves slices i belongs to: one of them is this wavefront's
            if (i == slice_of(q)) {
                woff[q] = accw;
                coff[q] = accc;
                ioff[q] = acci;
                wid[q] = wi;
            }
            accw += wi;
            accc += (wi + 1) >> 1;
            acci += (wi + 3) >> 2;
        }
    }
    // (a slice's index words, when the weights are dictionary-coded: byte indices have their own row numbering, 16-bit
    // ones sit where the slice's code words sit)
    auto iptr = [&](uint32_t q) -> const uint32_t * {
        return kDict == kWBytes ? widx + (size_t)ioff[q] * 64 : kDict == kWWords ? widx + (size_t)coff[q] * 64 : nullptr;
    };

    OEM_PROBE(1); // descriptor in hand, slice addresses derived
    OEM_EXIT_AT(1024u);
    // ---- every long-latency load of the tile is issued here, before any use ---------
    // Register sets of the wavefront's slices.  Slice 0 -- the widest -- takes sets 0 and 1 (its alignments 0..7 and
    // 8..15, fold_first); slices 1 .. kTop are loaded HERE into sets 2 .., the others later, into the sets the fold
    // has finished with: the first of them into set 0 half way through slice 0's fold, the next into set 1 after it,
    // the third (two sets only) into set 0 after slice 1's fold.  With two sets that is one slice of look-ahead, and
    // every fold then waited a whole loaded round trip for operands requested one fold -- a microsecond of LDS work
    // -- earlier: the slices took ~2 us each whatever they did (profiles/r04_notes.md).  A coded slice is a handful
    // of registers (fused: its four code words), so those kernels keep all of them resident (kSets = 5).
    constexpr uint32_t kSets = kSetsT;
    static_assert(kSets >= 2 && kSets <= kPerWave + 1, "two sets for slice 0, at most one more per further slice");
    constexpr uint32_t kTop = kSets - 2 < kPerWave - 1 ? kSets - 2 : kPerWave - 1; // slices 1 .. kTop are loaded at the top
    // set a slice is folded from; late slice L (0-based) = slice kTop + 1 + L goes into set L % kSets
    auto set_of = [](uint32_t q) constexpr -> uint32_t { return q <= kTop ? q + 1 : (q - kTop - 1) % kSets; };
    constexpr uint32_t kLate0 = kTop + 1 < kPerWave ? kTop + 1 : 0; // the slice loaded at fold_first's hand-over (0: none)
    // the theta window depends on the descriptor alone: its loads go out first, so that (loads return in
    // order) waiting for them waits for nothing else
    constexpr uint32_t kPer = (kWinT + kTileThreads - 1) / kTileThreads;
    double tw[kPer];
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        const uint32_t i = tx + u * kTileThreads;
        tw[u] = theta[td.lo + (i < td.win_len ? i : 0u)];
    }
    constexpr uint32_t kDictPer = (dict_entries<kDict>() + kTileThreads - 1) / kTileThreads;
    float dict_v[kDictPer];
#pragma unroll
    for (uint32_t u = 0; u < kDictPer; ++u) // 1 KiB (4 KiB: 16-bit indices), L2-resident
        dict_v[u] = (kDict != kWPlain && tx + u * kTileThreads < (uint32_t)dict_entries<kDict>()) ? dict[tx + u * kTileThreads] : 0.0f;
    SliceRegs<WT, kCh> R[kSets];
    auto load_slices = [&]() {
        if (OEM_EXP(4096u)) return; // (cost attribution: no slice is requested)
        load_slice<WT, kCh, kNT, kDict>(R[0], w + (size_t)woff[0] * 64, codes + (size_t)coff[0] * 64, lane, wid[0], iptr(0));
        // alignments 8..15 of the first slice, into the second set (see fold_first)
        load_slice<WT, kCh, kNT, kDict>(R[1], w + ((size_t)woff[0] + kCh) * 64, codes + ((size_t)coff[0] + kCh / 2) * 64, lane,
                                        wid[0] > (uint32_t)kCh ? wid[0] - kCh : 0u,
                                        kDict == kWBytes ? iptr(0) + (kCh / 4) * 64 : kDict == kWWords ? iptr(0) + (kCh / 2) * 64 : nullptr);
#pragma unroll
        for (uint32_t q = 1; q <= kTop; ++q)
            load_slice<WT, kCh, kNT, kDict>(R[q + 1], w + (size_t)woff[q] * 64, codes + (size_t)coff[q] * 64, lane, wid[q], iptr(q));
    };

    double rx[kRem];      // theta[t] * w of this thread's remote alignments

    uint32_t rrow[kRem];  // their read (index inside the tile)
    uint32_t rslot[kRem]; // their slot in the bucket-major queue
    uint32_t rwi[kRem];   // coded stores: the table index of their weight (looked up in LDS behind the first barrier)
    constexpr bool kRemIdx = kDict == kWBytes || kDict == kWFused; // byte-coded stores: a remote record's weight is a table index byte too
    const uint32_t tid_base = td.problem * problem_size; // first transcript of the tile's EM problem (0: one problem)
    const uint32_t *sd_t = sd + td.sd_begin - td.b_min;  // slot of record i = sd_t[bucket of its transcript] + i
    {
        uint32_t rt[kRem];
        WT rw[kRem];
        uint32_t ri[kRem]; // coded stores: the remote weights are table indices too, one byte each (index 0 = 0.0)
        if (td.remote_cnt) { // wave-uniform
            // branch-free: out-of-range slots re-read the tile's last record and carry no weight,
            // so the loads issue back to back
            const uint32_t last = td.remote_cnt - 1;
#pragma unroll
            for (int k = 0; k < kRem; ++k) {
                const uint32_t i = tx + k * kTileThreads;
                const uint32_t o = td.remote_begin + (i < td.remote_cnt ? i : last);
                if (OEM_EXP(8192u)) { rt[k] = td.b_min << kBucketShift; rrow[k] = 0; ri[k] = 0u; rw[k] = (WT)0; continue; } // (no record is requested)
                ld_remote<kPacked, kNT && OEM_REC_NT>(r_a, r_row, o, tid_base, rt[k], rrow[k]);
                if (kRemIdx) ri[k] = ld_stream<kNT && OEM_REC_NT>(&r_wi[o]);
                else rw[k] = ld_stream<kNT && OEM_REC_NT>(&r_w[o]);
            }
            // (the records are requested ahead of the slices: loads return in order, so the gathers that hang on
            // the records go out one round trip after the kernel starts, not behind all the slices' data -- 1 % of
            // the pass now that every slice is requested up here)
            load_slices();
#pragma unroll
            for (int k = 0; k < kRem; ++k)
                if (tx + k * kTileThreads >= td.remote_cnt) { rw[k] = (WT)0; ri[k] = 0u; }
        } else {
            load_slices();
#pragma unroll
            for (int k = 0; k < kRem; ++k) { rt[k] = td.b_min << kBucketShift; rw[k] = (WT)0; ri[k] = 0u; rrow[k] = 0; }
        }
#pragma unroll
        for (int k = 0; k < kRem; ++k)
        {
            // (coded: the weight is looked up in the LDS copy of the table at the start of phase A -- as a global
            // gather it was one more vector-memory instruction per record in the prologue)
            rx[k] = th(OEM_EXP(16u) ? 1.0 : theta[rt[k]]) * (kRemIdx ? 1.0 : (double)rw[k]);
            rwi[k] = ri[k];
        }
        // the slots are wanted last (phase B): a few words per tile, cache-resident.  Branch-free and back to
        // back -- a lookup per branch made the compiler wait for each one in turn, six dependent round trips
        // (a tile without remote records reads the table's slack word)
        uint32_t sdv[kRem];
#pragma unroll
        for (int k = 0; k < kRem; ++k) sdv[k] = sd_t[rt[k] >> kBucketShift];
        const uint32_t last_i = td.remote_cnt ? td.remote_cnt - 1 : 0u;
#pragma unroll
        for (int k = 0; k < kRem; ++k) {
            const uint32_t i = tx + k * kTileThreads;
            rslot[k] = sdv[k] + (i < td.remote_cnt ? i : last_i);
        }
    }
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        const uint32_t i = tx + u * kTileThreads;
        if (i < td.win_len) theta_l[i] = th(tw[u]);
    }
    if (kDict != kWPlain) {
#pragma unroll
        for (uint32_t u = 0; u < kDictPer; ++u)
            if (tx + u * kTileThreads < (uint32_t)dict_entries<kDict>()) dict_l[tx + u * kTileThreads] = dict_v[u];
    }
    for (uint32_t i = tx; i < (td.win_len << cs); i += kTileThreads) cnt_l[i] = 0.0;
    for (uint32_t i = tx; i < td.n_slices * 64; i += kTileThreads) den_l[i] = 0.0;
    OEM_PROBE(2); // theta window landed and written to LDS, windows cleared (remote gathers may still be in flight)
    __syncthreads();
    OEM_PROBE(3);
    OEM_EXIT_AT(128u);

    // ---- remote alignments, phase A: denominators --------------------------------
    if (kRemIdx) {
#pragma unroll
        for (int k = 0; k < kRem; ++k) rx[k] *= (double)dict_l[rwi[k]]; // (index 0 = 0.0: a thread without a record)
    }
#pragma unroll
    for (int k = 0; k < kRem; ++k)
        if (tx + k * kTileThreads < td.remote_cnt && !OEM_EXP(2u)) lds_add_f64(&den_l[rrow[k]], rx[k]);
    for (uint32_t i = tx + kRem * kTileThreads; i < td.remote_cnt; i += kTileThreads) { // overflow: park in the queue
        const uint32_t o = td.remote_begin + i;
        uint32_t t, row;
        ld_remote<kPacked, false>(r_a, r_row, o, tid_base, t, row);
        const double x = th(theta[t]) * (kRemIdx ? (double)dict[r_wi[o]] : (double)r_w[o]);
        queue[sd_t[t >> kBucketShift] + i] = x;
        lds_add_f64(&den_l[row], x);
    }
    OEM_PROBE(4); // remote gathers landed, their denominator atomics issued
    __syncthreads();
    OEM_PROBE(5);
    OEM_EXIT_AT(256u);

    // ---- local alignments: one read per lane, all operands already in registers -----
    // slice 0: 16 register-resident alignments in sets 0 and 1; it releases set 0 to the first late slice half way
    if (wave < td.n_slices)
        fold_first<WT, kCh, kCopies, kNT, kDict>(R[0], R[1], wid[0], wave, lane, w + (size_t)woff[0] * 64, codes + (size_t)coff[0] * 64,
                                                 td, theta_l, cnt_l, den_l, row_w_perm, kLate0 != 0,
                                                 w + (size_t)woff[kLate0] * 64, codes + (size_t)coff[kLate0] * 64,
                                                 wid[kLate0], iptr(0), iptr(kLate0), dict_l, exp_mask, cs);
    else if (kLate0 != 0)
        load_slice<WT, kCh, kNT, kDict>(R[0], w + (size_t)woff[kLate0] * 64, codes + (size_t)coff[kLate0] * 64, lane, wid[kLate0], iptr(kLate0));
    OEM_PROBE(6);
    // slices 1..: before slice q is folded, the late slice whose set the previous fold has just released is requested
#pragma unroll
    for (uint32_t q = 1; q < kPerWave; ++q) {
        const uint32_t s = slice_of(q);
        constexpr uint32_t kNone = 0xffffffffu;
        const uint32_t late = kTop + 1 + q < kPerWave ? kTop + 1 + q : kNone; // late slice L = q: after fold q - 1, into set q % kSets
        if (late != kNone)
            load_slice<WT, kCh, kNT, kDict>(R[q % kSets], w + (size_t)woff[late != kNone ? late : 0] * 64,
                       codes + (size_t)coff[late != kNone ? late : 0] * 64, lane, wid[late != kNone ? late : 0], iptr(late != kNone ? late : 0));
        if (s < td.n_slices)
            fold_slice<WT, kCh, kCopies, kDict>(R[set_of(q)], wid[q], s, lane, w + (size_t)woff[q] * 64, codes + (size_t)coff[q] * 64, td,
                       theta_l, cnt_l, den_l, row_w_perm, iptr(q), dict_l, exp_mask, cs);
        OEM_PROBE(6 + q); // wave 0's slice q folded (its operands had to land first)
    }
    __syncthreads();
    OEM_PROBE(10);
    OEM_EXIT_AT(512u);

    // ---- remote alignments, phase B: queue <- x * (c_i / denom_i) ------------------
#pragma unroll
    for (int k = 0; k < kRem; ++k) {
        const uint32_t i = tx + k * kTileThreads;
        if (i < td.remote_cnt && !OEM_EXP(1u)) {
            if (OEM_QUEUE_NT) __builtin_nontemporal_store(rx[k] * den_l[rrow[k]], &queue[rslot[k]]);
            else queue[rslot[k]] = rx[k] * den_l[rrow[k]];
        }
    }
    for (uint32_t i = tx + kRem * kTileThreads; i < td.remote_cnt; i += kTileThreads) {
        const uint32_t o = td.remote_begin + i;
        uint32_t t, row;
        ld_remote<kPacked, false>(r_a, r_row, o, tid_base, t, row);
        const uint32_t q = sd_t[t >> kBucketShift] + i;
        queue[q] = queue[q] * den_l[row];
    }

    // ---- flush the window: consecutive lanes -> consecutive addresses ---------------
    // (a thread starts at copy `lane` of its entry: the lanes of a wavefront then read different banks)
    for (uint32_t i = tx; i < td.win_len; i += kTileThreads) {
        double v = 0.0;
        for (uint32_t p = 0; p < (1u << cs); ++p) v += cnt_l[(i << cs) + ((p + lane) & ((1u << cs) - 1u))];
        if (OEM_EXP(32u)) { if (v != 0.0) cnt[td.lo + i] = v; }      // (cost attribution: the flush as plain stores)
        else if (OEM_EXP(64u)) { if (v == -1.0) cnt[td.lo + i] = v; } // (... and not at all)
        else if (v != 0.0) unsafeAtomicAdd(&cnt[td.lo + i], v);
    }
    OEM_PROBE(11); // queue stores and window flush issued
}

// The stopping rule of the deferred rel-diff (em.rs:212-218, :181), by one wavefront: the maximum over the slots the
// tile workgroups of this pass filled, the slots reset for the next pass.
__device__ __forceinline__ void deferred_decide(unsigned long long *slots, EmState *state, EmParams p, uint32_t decide)
{
    const uint32_t lane = threadIdx.x & 63u;
    static_assert(kRelSlots == 64, "one slot per lane");
    unsigned long long bits = __hip_atomic_load(&slots[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    slots[lane] = 0ull;                                            // em.rs:234
    double rel = __longlong_as_double((long long)bits);
    for (int off = 32; off > 0; off >>= 1) rel = fmax(rel, __shfl_xor(rel, off, 64));
    if (lane != 0) return;
    state->last_rel = rel;
    state->n_passes += 1;
    uint32_t niter = state->niter;
    bool stop = false;
    if (rel < p.conv_thresh && niter > p.min_iter_gate) {          // em.rs:212 / :399
        state->converged = 1;
        stop = true;
    } else {
        niter += 1;                                                // em.rs:218
        state->niter = niter;
        stop = niter >= p.max_iter;                                // em.rs:181 loop condition
    }
    if (stop) {
        state->pad[0] = decide; // 1 + the buffer that holds the final abundances (theta of the pass that decides)
        state->done = 1;
    }
}

__global__ __launch_bounds__(64) void k_deferred_decide(unsigned long long *slots, EmState *state, EmParams p, uint32_t decide)
{
    if (state->done) return;
    deferred_decide(slots, state, p, decide);
}

// The LAST iteration of a run that reaches max_iter, decided without a pass.  When pass max_iter would be due the loop
// ends whatever the rel-diff says (em.rs:181) and theta_{max_iter} is final either way (the swap of em.rs:204 comes
// before the break of :212); only `converged`, `niter` and `rel_diff` of oem_run_info hang on that last comparison.
// So instead of a speculative pass whose counts are dropped (0.146 ms at C3, 27 us at C2) this sweep takes the rel-diff
// of theta_{N-1} (`prev`) against theta_N (`cur`) -- the tile workgroups' share of the work in k_em_tile -- zeroes
// `prev` (the accumulator of the final pass, em.rs:245) and its last workgroup applies the rule.  Election as in
// k_reldiff_swap_clear (oem_kernels.hip): the maxima are device-scope atomics drained before the ticket is taken.
constexpr int kSweepThreads = 256;
__global__ __launch_bounds__(kSweepThreads) void k_deferred_sweep(double *__restrict__ prev, const double *__restrict__ cur,
                                                                  unsigned long long *slots, EmState *state, EmParams p,
                                                                  uint32_t decide)
{
    if (state->done) return;
    double rel = 0.0;
    const uint32_t stride = gridDim.x * kSweepThreads;
    for (uint32_t i0 = blockIdx.x * kSweepThreads + threadIdx.x; i0 < p.n_txps; i0 += 4 * stride) {
        double pc[4], cc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { // (all eight loads in flight before the first use)
            const uint32_t i = i0 + k * stride, ic = i < p.n_txps ? i : i0;
            pc[k] = prev[ic];
            cc[k] = cur[ic];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = i0 + k * stride;
            if (i < p.n_txps) {
                if (pc[k] > OEM_MIN_READ_THRESH) rel = fmax(rel, (cc[k] - pc[k]) / pc[k]); // em.rs:195-199 (signed)
                prev[i] = 0.0;                                                             // em.rs:207
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) rel = fmax(rel, __shfl_xor(rel, off, 64));
    __shared__ double smax[kSweepThreads / 64];
    __shared__ bool is_last;
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = rel;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = smax[0];
        for (int i = 1; i < kSweepThreads / 64; ++i) m = fmax(m, smax[i]);
        if (m > 0.0) atomicMax(&slots[blockIdx.x & (kRelSlots - 1u)], (unsigned long long)__double_as_longlong(m));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        is_last = atomicAdd(&state->blocks_arrived, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (is_last && threadIdx.x < 64) {
        if (threadIdx.x == 0) state->blocks_arrived = 0u;
        deferred_decide(slots, state, p, decide);
    }
}

// kNTQ: the queue range is read non-temporally.  Measured both ways (profiles/r04_notes.md): a store that fits the
// Infinity Cache with room to spare (C2: 1 M reads) gains 2.7 % of its pass -- the entries are read once and theta and the
// counts keep the L2 -- while at C3 the fold finds the entries the tile kernel has just written in the caches, and
// reading past them costs 2.8 % (2.5 M reads: 1.6 %; 1.25 M: the same either way).
template <bool kNTQ, int kFoldThreads = oem::kFoldThreadsDefault>
__global__ __launch_bounds__(kFoldThreads) void k_remote_fold(
    const uint32_t *__restrict__ bucket_base, const double *__restrict__ queue,
    const uint16_t *__restrict__ q_dst, double *__restrict__ cnt, const EmState *state,
    uint32_t n_groups, uint32_t n_txps, const BatchState *__restrict__ problems, uint32_t problem_size,
    unsigned long long *rd_slots, EmState *rd_state, EmParams rd_p, uint32_t rd_decide)
{
    if (state && state->done) return;
    // (deferred stopping rule: the tile kernel of this pass has left the previous iteration's maxima in the slots)
    if (rd_decide && blockIdx.x == 0 && threadIdx.x < 64) deferred_decide(rd_slots, rd_state, rd_p, rd_decide);
    __shared__ double acc[kBucket];
    const uint32_t b = blockIdx.x / n_groups, g = blockIdx.x % n_groups;
    if (problems) { // per-cell batch: skip the bucket when every cell it touches is finished
        const uint32_t t0 = b * kBucket;
        uint32_t t1 = t0 + kBucket - 1;
        if (t1 >= n_txps) t1 = n_txps - 1;
        bool live = false;
        for (uint32_t p = t0 / problem_size; p <= t1 / problem_size; ++p)
            live = live || problems[p].phase != kPhaseFinished;
        if (!live) return;
    }
    const uint32_t q0 = bucket_base[b], q1 = bucket_base[b + 1];
    const uint64_t span = q1 - q0;
    const uint32_t s0 = q0 + (uint32_t)(span * g / n_groups);
    const uint32_t s1 = q0 + (uint32_t)(span * (g + 1) / n_groups);
    if (s0 == s1) return;
    for (uint32_t i = threadIdx.x; i < kBucket; i += kFoldThreads) acc[i] = 0.0;
    __syncthreads();
    // four independent loads in flight per thread before the LDS atomics; the last step's loads past the
    // range are clamped to its last entry and skipped at the point of use (a tail taken one entry per round trip
    // cost the small stores, whose threads have a handful of entries each, a microsecond per entry)
    constexpr int kDepth = 4; // (8 measured the same at both sizes)
    for (uint32_t o = s0 + threadIdx.x; o < s1; o += kDepth * kFoldThreads) {
        double v[kDepth];
        uint32_t d[kDepth];
#pragma unroll
        for (int k = 0; k < kDepth; ++k) {
            const uint32_t oo = o + k * kFoldThreads, oc = oo < s1 ? oo : s1 - 1;
            v[k] = ld_stream<kNTQ>(&queue[oc]);
            d[k] = ld_stream<kNTQ>(&q_dst[oc]);
        }
        // hot destinations: runs of equal ones are summed on the vector ALU first (see sum_runs_of_equal_keys; one
        // look at the trip's first entries decides for the trip: hot runs are thousands of entries long)
        const bool rep = keys_repeat(d[0]);
#pragma unroll
        for (int k = 0; k < kDepth; ++k) {
            double vk[1] = {o + k * kFoldThreads < s1 ? v[k] : 0.0};
            if (rep) sum_runs_of_equal_keys<1, 1>(d[k], vk);
            if (vk[0] != 0.0) lds_add_f64(&acc[d[k]], vk[0]);
        }
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

// Register blocking of k_em_tile (profiles/r01_notes.md .. r04_notes.md have the variants that were measured and
// dropped): 256 threads, 4 slices per wavefront, as many of them register-resident before the fold as the weight
// coding's register sets allow (tile_sets: all of them for the coded weights and the f32 stream, three sets for f64
// weights, two for the wide window), 8 local and 6 remote alignments per thread in registers; 4 to 8 interleaved
// count-window copies for the narrow window cap (by the tile's window), one for the wide cap of sparse stores
// (40 KiB LDS; same-address atomics are rare when few reads share a transcript).
constexpr uint32_t kRdBlocks = 64; // workgroups of the deferred rel-diff behind a pass's tiles (k_em_tile)
template <typename WT, bool kNT, bool kPacked, int kDict>
static void launch_tile(oem_store *s, const WT *w, const WT *r_w, const double *theta, double *cnt,
                        const EmState *state, const uint32_t *row_w_perm, const BatchState *problems, const DeferredRelDiff *rd)
{
    const DeviceTiled &t = s->tiled;
    const uint32_t *r_a = kPacked ? t.r_pk : t.r_tid;
    // per-cell batch: the compacted list of live tiles once the host has built one (oem_multi_kernels.hip)
    const uint32_t *live_tiles = problems && s->multi.live_valid ? s->multi.live_tiles : nullptr;
    const uint32_t n_tiles = live_tiles ? s->multi.n_live_tiles : t.n_tiles;
    if (n_tiles == 0) return;
    const uint32_t grid = problems ? (n_tiles + 7u) / 8u * 8u : n_tiles; // (per-cell batch: see the tile index in k_em_tile)
    double *rd_prev = rd && !problems ? rd->prev : nullptr;
    unsigned long long *rd_slots = rd ? rd->slots : nullptr;
    // (the deferred rel-diff: kRdBlocks more workgroups behind the tiles', a share of the transcripts each)
    const uint32_t n_rd = rd_prev ? kRdBlocks : 0u;
    const uint32_t n_txps = s->csr.n_txps, rd_chunk = n_rd ? (n_txps + n_rd - 1) / n_rd : 0u;
    if (t.win_cap > kWin)
        hipLaunchKernelGGL((k_em_tile<WT, 8, 6, 256, (sizeof(WT) == 4 ? OEM_WAVES_WIDE : 2), 1, kNT, kWinWideLds, kPacked, kDict, (sizeof(WT) == 4 ? OEM_SETS_WIDE : 2)>), dim3(grid + n_rd), dim3(256), 0, s->stream,
                           t.tiles, t.codes, w, r_a, r_w, t.r_row, t.sd, t.queue, theta, cnt, state,
                           row_w_perm, problems, t.problem_size, n_tiles, t.widx, t.i_base, t.dict, t.r_wi, live_tiles,
                           rd_prev, rd_slots, rd_chunk, n_txps);
    else
        hipLaunchKernelGGL((k_em_tile<WT, 8, 6, 256, tile_min_waves<WT, kDict>(), OEM_COPIES, kNT, kWin, kPacked, kDict, tile_sets<WT, kDict>()>), dim3(grid + n_rd), dim3(256), 0, s->stream,
                           t.tiles, t.codes, w, r_a, r_w, t.r_row, t.sd, t.queue, theta, cnt, state,
                           row_w_perm, problems, t.problem_size, n_tiles, t.widx, t.i_base, t.dict, t.r_wi, live_tiles,
                           rd_prev, rd_slots, rd_chunk, n_txps);
}

static uint32_t fold_groups(const DeviceTiled &t)
{
    // ~1 workgroup of 1024 threads per CU in total (each flushes a whole bucket window, so
    // fewer, longer-running workgroups mean fewer flush atomics) ...
    uint32_t n_groups = 256u / (t.n_buckets ? t.n_buckets : 1);
    // ... but every workgroup clears and flushes a whole window (32 KiB: 4096 transcripts), which only pays for
    // itself with a few queue entries per window entry to fold (1 M-read store, 8192-transcript windows: 32 Ki entries
    // 34.5 us, 16 Ki 33.1 us, 8 Ki 35.6 us per pass; the same two entries per window entry with the 4096 ones)
    const uint64_t per_bucket = t.n_remote / (t.n_buckets ? t.n_buckets : 1) + 1;
    const uint32_t max_useful = (uint32_t)((per_bucket + kFoldEntriesPerGroup - 1) / kFoldEntriesPerGroup);
    if (n_groups > max_useful) n_groups = max_useful;
    if (n_groups < 1) n_groups = 1;
    return n_groups;
}

// The fold of a pass's queue (k_remote_fold), on `stream`.
static int launch_remote_fold(oem_store *s, hipStream_t stream, double *cnt, const EmState *state, const BatchState *problems,
                              uint32_t problem_size, unsigned long long *rd_slots, EmState *rd_state, EmParams rd_p,
                              uint32_t rd_decide)
{
    const DeviceTiled &t = s->tiled;
    const uint64_t wsz = s->csr.w_is_f64 ? 8 : 4;
    const uint64_t stream_bytes = (t.n_local + t.n_local / 8) * (wsz + 2) + t.n_remote * (wsz + (t.packed ? 4 : 6));
    const uint32_t n_groups = fold_groups(t);
    if (stream_bytes > (96ull << 20)) // (2.5 M reads, 170 MB of streams: already better cached -- see kNTQ)
        hipLaunchKernelGGL(k_remote_fold<false>, dim3(t.n_buckets * n_groups), dim3(kFoldThreads), 0,
                           stream, t.bucket_base, t.queue, t.q_dst, cnt, state, n_groups,
                           s->csr.n_txps, problems, problem_size, rd_slots, rd_state, rd_p, rd_decide);
    else
        hipLaunchKernelGGL(k_remote_fold<true>, dim3(t.n_buckets * n_groups), dim3(kFoldThreads), 0,
                           stream, t.bucket_base, t.queue, t.q_dst, cnt, state, n_groups,
                           s->csr.n_txps, problems, problem_size, rd_slots, rd_state, rd_p, rd_decide);
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

int launch_em_pass_tiled(oem_store *s, const double *theta, double *cnt, const EmState *state,
                         const uint32_t *row_w_perm, const BatchState *problems, uint32_t problem_size, bool skip_fold,
                         const DeferredRelDiff *rd)
{
    const DeviceTiled &t = s->tiled;
    if (t.n_tiles == 0) return OEM_OK;
#ifdef OEM_TESTING
    { // (per-cell groups launch from two host threads, and a process may drive several devices)
        static std::mutex mu;
        static unsigned int current[64] = {0};
        const unsigned int want = (unsigned int)knob("OEM_TILE_EXP", 0);
        std::lock_guard<std::mutex> lk(mu);
        unsigned int &cur = current[s->device >= 0 && s->device < 64 ? s->device : 0];
        if (want != cur) {
            OEM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tile_exp), &want, sizeof(want)));
            cur = want;
        }
    }
#endif
    const bool f64w = s->csr.w_is_f64;
    // matrix bytes one pass streams; beyond the Infinity Cache they are loaded non-temporally
    const uint64_t wsz = f64w ? 8 : 4;
    const uint64_t stream_bytes = (t.n_local + t.n_local / 8) * (wsz + 2) + t.n_remote * (wsz + (t.packed ? 4 : 6));
    const long nt_knob = knob("OEM_TILE_NT", -1); // testing build: 0 never, 1 always
    const bool nt = nt_knob < 0 ? stream_bytes > (192ull << 20) : nt_knob != 0;
#define OEM_TILE(WT, NT, W, RW, DICT)                                                                       \
    do {                                                                                                   \
        if (t.packed) launch_tile<WT, NT, true, DICT>(s, W, RW, theta, cnt, state, row_w_perm, problems, rd);      \
        else launch_tile<WT, NT, false, DICT>(s, W, RW, theta, cnt, state, row_w_perm, problems, rd);              \
    } while (0)
    // (a store whose codes carry the fused index has no other way to be read; the knob -- testing build, A/B --
    // switches only the byte-stream coding off)
    const bool coded = !f64w && t.dict_n > 0 && !t.dict_fused && knob("OEM_NO_DICT", 0) == 0;
    const bool bytes = coded && !t.dict_words, words = coded && t.dict_words;
    if (f64w) {
        if (nt) OEM_TILE(double, true, t.w64, t.r_w64, kWPlain);
        else OEM_TILE(double, false, t.w64, t.r_w64, kWPlain);
    } else if (words) {
        if (nt) OEM_TILE(float, true, t.w32, t.r_w32, kWWords);
        else OEM_TILE(float, false, t.w32, t.r_w32, kWWords);
    } else if (t.dict_fused) {
        if (nt) OEM_TILE(float, true, t.w32, t.r_w32, kWFused);
        else OEM_TILE(float, false, t.w32, t.r_w32, kWFused);
    } else if (bytes) {
        if (nt) OEM_TILE(float, true, t.w32, t.r_w32, kWBytes);
        else OEM_TILE(float, false, t.w32, t.r_w32, kWBytes);
    } else {
        if (nt) OEM_TILE(float, true, t.w32, t.r_w32, kWPlain);
        else OEM_TILE(float, false, t.w32, t.r_w32, kWPlain);
    }
#undef OEM_TILE
    OEM_HIP(hipGetLastError());
    unsigned long long *rd_slots = rd ? rd->slots : nullptr;
    EmState *rd_state = rd ? rd->state : nullptr;
    const EmParams rd_p = rd ? rd->p : EmParams{0, 0, 0, 0.0};
    const uint32_t rd_decide = rd && rd->prev ? rd->decide : 0u;
    if (t.n_remote > 0 && !skip_fold) { // (the per-cell batch folds and finishes the pass in one kernel)
        OEM_TRY(launch_remote_fold(s, s->stream, cnt, state, problems, problem_size, rd_slots, rd_state, rd_p, rd_decide));
    } else if (rd_decide) { // no fold to carry the decision (a store without remote alignments)
        hipLaunchKernelGGL(k_deferred_decide, dim3(1), dim3(64), 0, s->stream, rd_slots, rd_state, rd_p, rd_decide);
        OEM_HIP(hipGetLastError());
    }
    return OEM_OK;
}

// The start of a deferred run in one launch (it was a fill, four memsets: five launch boundaries and ~50 us of host time
// in front of every run, 2.5 us per step of a 20-iteration one): theta_0 = `avg` everywhere (em.rs:165; `fill` false:
// the caller's init_abundances are already in b0), the two other vectors, the slots and the loop state zeroed.
__global__ __launch_bounds__(256) void k_deferred_init(double *__restrict__ b0, double *__restrict__ b1, double *__restrict__ b2,
                                                       double avg, bool fill, uint32_t n_txps, unsigned long long *slots,
                                                       EmState *state)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_txps; i += gridDim.x * 256u) {
        if (fill) b0[i] = avg;
        b1[i] = 0.0;
        b2[i] = 0.0;
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < kRelSlots) slots[threadIdx.x] = 0ull;
        static_assert(sizeof(EmState) % 4 == 0 && sizeof(EmState) / 4 <= 256, "EmState cleared by one workgroup");
        if (threadIdx.x < sizeof(EmState) / 4) reinterpret_cast<uint32_t *>(state)[threadIdx.x] = 0u;
    }
}

int launch_deferred_init(oem_store *s, double *const bufs[3], double avg, bool fill)
{
    const uint32_t T = s->csr.n_txps;
    uint32_t grid = (T + 1023) / 1024;
    if (grid > 512) grid = 512;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_deferred_init, dim3(grid), dim3(256), 0, s->stream, bufs[0], bufs[1], bufs[2], avg, fill, T, s->rel_slots,
                       s->d_state);
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

int launch_deferred_sweep(oem_store *s, double *prev, const double *cur, const DeferredRelDiff &rd)
{
    const uint32_t T = rd.p.n_txps;
    uint32_t grid = (T + 4 * kSweepThreads - 1) / (4 * kSweepThreads);
    if (grid > 256) grid = 256;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_deferred_sweep, dim3(grid), dim3(kSweepThreads), 0, s->stream, prev, cur, rd.slots, rd.state, rd.p, rd.decide);
    OEM_HIP(hipGetLastError());
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

#ifdef OEM_TESTING
// Test hook: one probed E/M pass (after an unprobed one); out = n_tiles x 16 wall-clock stamps (100 MHz).
extern "C" int oem_debug_tile_probe(oem_store *s, unsigned long long *out, uint64_t n_out)
{
    using namespace oem;
    OEM_API_BEGIN
    if (!s || !out || !s->tiled.present || n_out < (uint64_t)s->tiled.n_tiles * 16)
        return fail(OEM_ERR_ARG, "oem_debug_tile_probe: bad argument");
    std::lock_guard<std::mutex> lk(s->mu);
    OEM_HIP(hipSetDevice(s->device));
    const size_t n = (size_t)s->tiled.n_tiles * 16;
    unsigned long long *d = nullptr;
    OEM_HIP(hipMalloc((void **)&d, n * sizeof(unsigned long long)));
    OEM_HIP(hipMemset(d, 0, n * sizeof(unsigned long long)));
    const uint32_t T = s->csr.n_txps;
    OEM_TRY(launch_fill(s, s->theta, (double)s->global_n_reads / (double)T, T));
    OEM_HIP(hipMemsetAsync(s->cnt, 0, sizeof(double) * T, s->stream));
    OEM_TRY(launch_em_pass_tiled(s, s->theta, s->cnt, nullptr, nullptr));   // warm
    OEM_HIP(hipStreamSynchronize(s->stream));
    OEM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tile_probe), &d, sizeof(d)));
    int rc = launch_em_pass_tiled(s, s->theta, s->cnt, nullptr, nullptr);
    hipStreamSynchronize(s->stream);
    unsigned long long *null = nullptr;
    hipMemcpyToSymbol(HIP_SYMBOL(g_tile_probe), &null, sizeof(null));
    if (rc == OEM_OK && hipMemcpy(out, d, n * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(OEM_ERR_HIP, "oem_debug_tile_probe: read-back failed");
    hipFree(d);
    return rc;
    OEM_API_END("oem_debug_tile_probe")
}
// Test hook (scripts/overlap_probe.py): how much of a pass's fold hides under a tile kernel.  out_us[0] = tile kernel
// alone, [1] = fold alone, [2] = tile + fold on one stream (the pass), [3] = per iteration when fold i runs on a second
// stream behind tile i (an event) while tile i + 1 runs -- the results are meaningless, the time says what a fold
// that overlaps the next tiles would cost; [4] = the fold in 256-thread workgroups alone, [5] = that fold on the second
// stream under the next tile kernel, [6] = tile + that fold on one stream.  n launches each, HIP events on the
// store's stream; out_us has 7 entries.
extern "C" int oem_debug_overlap_probe(oem_store *s, uint32_t n, double *out_us)
{
    using namespace oem;
    OEM_API_BEGIN
    if (!s || !out_us || !n || !s->tiled.present || !s->tiled.n_remote)
        return fail(OEM_ERR_ARG, "oem_debug_overlap_probe: bad argument");
    std::lock_guard<std::mutex> lk(s->mu);
    OEM_HIP(hipSetDevice(s->device));
    const uint32_t T = s->csr.n_txps;
    OEM_TRY(launch_fill(s, s->theta, (double)s->global_n_reads / (double)T, T));
    OEM_HIP(hipMemsetAsync(s->cnt, 0, sizeof(double) * T, s->stream));
    hipStream_t s2 = nullptr;
    OEM_HIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1, ej;
    OEM_HIP(hipEventCreate(&e0));
    OEM_HIP(hipEventCreate(&e1));
    OEM_HIP(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    std::vector<hipEvent_t> ev(n);
    for (auto &e : ev) OEM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const EmParams p0{0, 0, 0, 0.0};
    // the same fold in workgroups of the tile kernel's shape (256 threads; four times the groups): a 1024-thread,
    // 32 KiB workgroup is not placed on a CU while 4-wavefront tile workgroups are pending -- the dispatcher refills
    // the slots a finished tile leaves with the next tile -- so the fat fold only ever ran in the tile kernel's tail
    auto thin_fold = [&](hipStream_t st) -> int {
        const DeviceTiled &t = s->tiled;
        const uint32_t n_groups = fold_groups(t) * 4;
        hipLaunchKernelGGL((k_remote_fold<false, 256>), dim3(t.n_buckets * n_groups), dim3(256), 0, st, t.bucket_base, t.queue,
                           t.q_dst, s->cnt, (const EmState *)nullptr, n_groups, s->csr.n_txps, (const BatchState *)nullptr, 0u,
                           (unsigned long long *)nullptr, (EmState *)nullptr, p0, 0u);
        OEM_HIP(hipGetLastError());
        return OEM_OK;
    };
    auto timed = [&](int mode, double *us) -> int {
        for (int rep = 0; rep < 2; ++rep) { // (the first round warms)
            OEM_HIP(hipEventRecord(e0, s->stream));
            for (uint32_t i = 0; i < n; ++i) {
                if (mode != 1 && mode != 4) OEM_TRY(launch_em_pass_tiled(s, s->theta, s->cnt, nullptr, nullptr, nullptr, 0, mode != 2));
                if (mode == 1) OEM_TRY(launch_remote_fold(s, s->stream, s->cnt, nullptr, nullptr, 0, nullptr, nullptr, p0, 0));
                if (mode == 4) OEM_TRY(thin_fold(s->stream));
                if (mode == 6) OEM_TRY(thin_fold(s->stream)); // (tile + thin fold on one stream)
                if (mode == 3 || mode == 5) {
                    OEM_HIP(hipEventRecord(ev[i], s->stream));
                    OEM_HIP(hipStreamWaitEvent(s2, ev[i], 0));
                    if (mode == 3) OEM_TRY(launch_remote_fold(s, s2, s->cnt, nullptr, nullptr, 0, nullptr, nullptr, p0, 0));
                    else OEM_TRY(thin_fold(s2));
                }
            }
            if (mode == 3 || mode == 5) {
                OEM_HIP(hipEventRecord(ej, s2));
                OEM_HIP(hipStreamWaitEvent(s->stream, ej, 0));
            }
            OEM_HIP(hipEventRecord(e1, s->stream));
            OEM_HIP(hipEventSynchronize(e1));
            float ms = 0.f;
            OEM_HIP(hipEventElapsedTime(&ms, e0, e1));
            *us = (double)ms * 1e3 / n;
        }
        return OEM_OK;
    };
    for (int r = 0; r < 50; ++r) OEM_TRY(launch_em_pass_tiled(s, s->theta, s->cnt, nullptr, nullptr)); // settle
    int rc = OEM_OK;
    for (int m = 0; m < 7 && rc == OEM_OK; ++m) rc = timed(m, &out_us[m]);
    hipStreamSynchronize(s->stream);
    hipStreamSynchronize(s2);
    for (auto &e : ev) hipEventDestroy(e);
    hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(ej);
    hipStreamDestroy(s2);
    return rc;
    OEM_API_END("oem_debug_overlap_probe")
}
#endif
