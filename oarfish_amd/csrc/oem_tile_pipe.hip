// oem_tile_pipe.hip -- the E/M pass of a large single-problem store as a SOFTWARE PIPELINE over tiles (round 5).
//
// Semantics: em.rs:87-133 (m_step), exactly as k_em_tile (oem_tile_kernels.hip) -- same layout, same fold
// (oem_tile_common.h), same queue and flush; only WHEN a tile's operands are requested differs.
//
// k_em_tile is one workgroup per tile: descriptor -> records / slices / theta window -> gathers -> LDS phases ->
// stores.  Every load of a tile is issued in its first microsecond and the memory pipe then idles through its
// LDS phases; the overlap comes from the other workgroups of the CU alone, and with everything switched off the
// skeleton streamed at 2.4 TB/s (profiles/r04_tile_cost_attribution.txt).  Here a workgroup is PERSISTENT (grid =
// resident slots, tiles b, b + G, b + 2 G, ...) and, while it folds tile k, requests tile k + 1:
//
//   barrier 1
//   F(k)   the wavefront's four slices of k; the register set a fold releases is handed to the same slice of k + 1
//          at once (no second copy of the slice registers: a slice of k + 1 has a whole tile's time to land);
//          after the second fold: request window / records of k + 1; at the end: descriptor of k + 2 (scalar)
//   barrier 2
//   G(k+1) gathers of k + 1 (its records landed under the folds)
//   B(k)   queue entries of k, window flush (the flush also leaves the count window zeroed)
//   W(k+1) theta window of k + 1 -> LDS
//   A(k+1) remote denominators of k + 1 (the gathers have landed under B and the flush); the denominators are
//          double-buffered: B(k) and A(k + 1) share a barrier interval
//
// Two barriers per tile (k_em_tile: three); 37 KiB of LDS and up to 128 VGPRs: four workgroups per CU, each with two tiles
// in flight.  The number of loads a wavefront issues per tile is STATIC (a slice narrower than a register set
// re-reads its first row, a tile with fewer records than register slots its last record): the waits the compiler
// places are counted (vmcnt(n): "all but the n youngest"), and a data-dependent number of younger loads makes it
// assume the fewest -- which would wait for the prefetch just issued.  The weight table of a coded store is read
// into LDS once per workgroup, not per tile.
#include "oem_internal.h"

#ifndef OEM_TESTING
// The product library carries no pipeline kernels: measured slower than k_em_tile (below), they exist for the tests and
// the measurement scripts.
namespace oem {
bool tile_pipeline_applies(const oem_store *, const BatchState *) { return false; }
int launch_tile_pipeline(oem_store *, const double *, double *, const EmState *, const uint32_t *, bool)
{
    return fail(OEM_ERR_STATE, "the pipelined tile walk is part of the test-only library");
}
} // namespace oem
#else

#define OEM_PROBE(i) do { } while (0)
#include "oem_tile_common.h"

namespace oem {

namespace {

// Phase timestamps (test-only library, scripts/pipe_probe.py): wave 0 of a workgroup stamps the 100 MHz device clock
// at the phase boundaries of every tile it walks into g_pipe_probe[tile][16].  The product build has no probe code.
#ifdef OEM_TESTING
__device__ unsigned long long *g_pipe_probe = nullptr;
#define OEM_PPROBE(i)                                                                                   \
    do {                                                                                                \
        if (g_pipe_probe && threadIdx.x == 0) g_pipe_probe[(size_t)tile * 16 + (i)] = wall_clock64(); \
    } while (0)
#else
#define OEM_PPROBE(i) do { } while (0)
#endif

constexpr uint32_t kPipeThreads = 256;
constexpr uint32_t kPipeWaves = kPipeThreads / 64;
constexpr uint32_t kPipePerWave = kTileSlices / kPipeWaves; // slices per wavefront
constexpr int kPipeCh = 8;                                    // local alignments per register set
constexpr int kPipeRem = 6;                                   // remote records per thread in registers
constexpr uint32_t kPipeSets = kPipePerWave + 1;              // two sets for the widest slice, one per further slice

#ifndef OEM_PIPE_WAVES
#define OEM_PIPE_WAVES 4 // workgroups per CU the launch bound leaves room for
#endif
// Cost attribution (scripts/build_variant.sh "-DOEM_PIPE_EXP=mask": wrong results, the time says what a part costs):
// 1 local scatter atomics, 2 pass-1 LDS reads, 4 remote theta gathers, 8 queue stores, 16 slice loads of k + 1,
// 32 remote denominator atomics, 64 window flush atomics
#ifndef OEM_PIPE_EXP
#define OEM_PIPE_EXP 0
#endif
#define OEM_PEXP(bit) ((OEM_PIPE_EXP & (bit)) != 0)

// where this wavefront's slices of a tile start (scalar prefix sums over the descriptor's widths)
struct SliceAddr {
    uint32_t woff[kPipePerWave], coff[kPipePerWave], ioff[kPipePerWave], wid[kPipePerWave];
};

// A tile's slices come in descending width and are dealt to the wavefronts boustrophedon (see k_em_tile).
__device__ __forceinline__ uint32_t pipe_slice_of(uint32_t q, uint32_t wave)
{
    return (q & 1u) ? (q + 1) * kPipeWaves - 1 - wave : q * kPipeWaves + wave;
}

__device__ __forceinline__ void slice_addrs(const TileDesc &td, uint32_t ib, uint32_t wave, SliceAddr &a)
{
    uint32_t accw = td.w_base, accc = td.c_base, acci = ib;
#pragma unroll
    for (uint32_t i = 0; i < kTileSlices; ++i) {
        const uint32_t wi = td.width[i];
        const uint32_t q = i / kPipeWaves;
        if (i == pipe_slice_of(q, wave)) {
            a.woff[q] = accw;
            a.coff[q] = accc;
            a.ioff[q] = acci;
            a.wid[q] = wi;
        }
        accw += wi;
        accc += (wi + 1) >> 1;
        acci += (wi + 3) >> 2;
    }
}

// The loads of one register set, ALWAYS kCh / 2 code words (and their weights / index words): pairs beyond the
// slice's width re-read the set's first row -- a line the neighbouring load fetches anyway -- and are zeroed by
// mask_slice when the set is folded.  (An empty slice reads the row behind it: the next slice's first row or the
// arrays' slack row.)
template <typename WT, int kCh, bool kNT, int kDict>
__device__ __forceinline__ void load_slice_static(SliceRegs<WT, kCh> &r, const WT *__restrict__ wbase,
                                                  const uint32_t *__restrict__ cbase, uint32_t lane, uint32_t width,
                                                  const uint32_t *__restrict__ ibase)
{
#pragma unroll
    for (int g = 0; g < kCh / 2; ++g) {
        const bool in = (uint32_t)(2 * g) < width; // wave-uniform
        const uint32_t gi = in ? (uint32_t)g : 0u;
        if (kDict == kWBytes) {
            if ((g & 1) == 0) r.wi[g >> 1] = ld_stream<kNT>(&ibase[(gi >> 1) * 64 + lane]);
        } else if (kDict == kWWords) {
            r.wi[g] = ld_stream<kNT>(&ibase[gi * 64 + lane]);
        } else if (kDict == kWPlain) {
            r.w[2 * g] = ld_stream<kNT>(&wbase[(2 * gi) * 64 + lane]);
            r.w[2 * g + 1] = ld_stream<kNT>(&wbase[(2 * gi + (in ? 1u : 0u)) * 64 + lane]);
        }
        r.c[g] = ld_stream<kNT>(&cbase[gi * 64 + lane]);
    }
}

// what load_slice (oem_tile_common.h) leaves in a set: nothing beyond the slice's width
template <typename WT, int kCh, int kDict>
__device__ __forceinline__ void mask_slice(SliceRegs<WT, kCh> &r, uint32_t width)
{
#pragma unroll
    for (int g = 0; g < kCh / 2; ++g) {
        const bool in = (uint32_t)(2 * g) < width; // wave-uniform
        if (kDict == kWBytes) {
            if ((g & 1) == 0) r.wi[g >> 1] = in ? r.wi[g >> 1] : 0u;
        } else if (kDict == kWWords) {
            r.wi[g] = in ? r.wi[g] : 0u;
        } else if (kDict == kWPlain) {
            r.w[2 * g] = in ? r.w[2 * g] : (WT)0;
            r.w[2 * g + 1] = in ? r.w[2 * g + 1] : (WT)0;
        }
        r.c[g] = in ? r.c[g] : 0u;
    }
}

// ---- the fold of the pipeline: every LDS read of a register set in flight before the first use ------------------
// fold_slice / fold_first (oem_tile_common.h) read theta and the weight of one alignment, wait, multiply, read the
// next: a chain of one LDS round trip per alignment, and with the memory latency gone from the tile (the pipeline)
// those round trips -- each queued behind the count-window atomics of the CU's other fifteen wavefronts -- are what
// a fold lasts (scripts/pipe_probe.py: ~2 us per slice of ~6 alignments).  Here a set's eight weights and eight
// abundances are requested back to back, then multiplied: one round trip per set.
template <typename WT, int kCh, int kDict>
__device__ __forceinline__ void set_products(const SliceRegs<WT, kCh> &r, uint32_t width, const double *theta_l,
                                             const float *dict_l, double (&x)[kCh])
{
    WT wf[kCh];
    double th[kCh];
#pragma unroll
    for (int k = 0; k < kCh; ++k) // (plain weights: the second element of the last pair of an odd width belongs to the next row)
        wf[k] = (kDict == kWPlain && (k & 1) && (uint32_t)k >= width) ? (WT)0 : OEM_PEXP(2) ? (WT)(r.c[k >> 1] & 0xffu) : slice_w<kDict>(r, k, dict_l);
#pragma unroll
    for (int k = 0; k < kCh; ++k) th[k] = OEM_PEXP(2) ? 1.0 + (double)r.c[k >> 1] : lds_ld(theta_l, code_off<kDict>(code_half(r.c[k >> 1], k & 1)));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < kCh; ++k) x[k] = th[k] * (double)wf[k]; // em.rs:111
}

// pass 2 of a set: cnt[t] += x * (c / denom), em.rs:128-129.  The count window is kept in 1 << cs interleaved copies
// (see fold_slice); `first`: entry 0 is the read's anchor, which whole wavefronts share inside a highly expressed
// transcript -- reduced across the wavefront, one lane adds.
template <int kCh, int kDict, typename WT>
__device__ __forceinline__ void set_scatter(const SliceRegs<WT, kCh> &r, const double (&x)[kCh], double inv, uint32_t width,
                                            uint32_t lane, double *cnt_l, uint32_t cs, bool first)
{
    const uint32_t copy_off = (lane & ((1u << cs) - 1u)) * 8u;
    if (OEM_PEXP(1)) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < kCh; ++k) acc += x[k] * inv;
        if (acc == -1.0) cnt_l[lane] = acc;
        return;
    }
#pragma unroll
    for (int k = 0; k < kCh; ++k) {
        if ((uint32_t)k < width) { // uniform
            const uint32_t off = code_off<kDict>(code_half(r.c[k >> 1], k & 1));
            const double v = x[k] * inv;
            if (k == 0 && first) {
                const uint32_t u = __builtin_amdgcn_readfirstlane(off);
                if (__all(off == u)) {
                    const double sum = wave_sum_f64(v);
                    if (lane == 0 && sum != 0.0) lds_add_f64(lds_at(cnt_l, u << cs), sum);
                } else if (v != 0.0) {
                    lds_add_f64(lds_at(cnt_l, (off << cs) + copy_off), v);
                }
            } else if (v != 0.0) {
                lds_add_f64(lds_at(cnt_l, (off << cs) + copy_off), v);
            }
        }
    }
}

// One slice (`lo`: its alignments 0..7; `hi`, kTwo: 8..15 -- the wavefront's first and widest slice); reads with
// more alignments than the sets hold reload the rest from memory (rare: > 8 / > 16 LOCAL alignments).
template <typename WT, int kCh, int kDict, bool kTwo>
__device__ __forceinline__ void fold_slice_b(const SliceRegs<WT, kCh> &lo, const SliceRegs<WT, kCh> &hi, uint32_t width, uint32_t s,
                                             uint32_t lane, const WT *__restrict__ wbase, const uint32_t *__restrict__ cbase,
                                             const TileDesc &td, const double *theta_l, double *cnt_l, double *den_l,
                                             const uint32_t *__restrict__ row_w_perm, const uint32_t *__restrict__ ibase,
                                             const float *dict_l, uint32_t cs)
{
    auto w_at = [&](uint32_t j) -> double {
        if (kDict == kWBytes) return (double)dict_l[(ibase[(j >> 2) * 64 + lane] >> (8 * (j & 3))) & 0xffu];
        if (kDict == kWWords) return (double)dict_l[code_half(ibase[(j >> 1) * 64 + lane], j & 1)];
        if (kDict == kWFused) return (double)dict_l[code_widx(cbase[(j >> 1) * 64 + lane], j & 1)];
        return (double)wbase[j * 64 + lane];
    };
    constexpr uint32_t kHeld = kTwo ? 2 * kCh : kCh;
    const uint32_t rl = s * 64 + lane;
    const bool two = kTwo && width > (uint32_t)kCh; // wave-uniform
    double x[kCh], xh[kCh];
    double denom = den_l[rl]; // the remote part (phase A)
    set_products<WT, kCh, kDict>(lo, width, theta_l, dict_l, x);
#pragma unroll
    for (int k = 0; k < kCh; ++k) denom += x[k];
    if (two) {
        set_products<WT, kCh, kDict>(hi, width - kCh, theta_l, dict_l, xh);
#pragma unroll
        for (int k = 0; k < kCh; ++k) denom += xh[k];
    }
    for (uint32_t j = kHeld; j < width; ++j) {
        const uint32_t cc = cbase[(j >> 1) * 64 + lane];
        denom += lds_ld(theta_l, code_off<kDict>(code_half(cc, j & 1))) * w_at(j);
    }
    double scale = 1.0;
    if (row_w_perm) scale = rl < td.n_rows ? (double)row_w_perm[td.row_base + rl] : 0.0;
    const double inv = denom > OEM_EM_DENOM_THRESH ? scale / denom : 0.0; // em.rs:115
    den_l[rl] = inv;
    set_scatter<kCh, kDict>(lo, x, inv, width, lane, cnt_l, cs, true);
    if (two) set_scatter<kCh, kDict>(hi, xh, inv, width - kCh, lane, cnt_l, cs, false);
    const uint32_t copy_off = (lane & ((1u << cs) - 1u)) * 8u;
    for (uint32_t j = kHeld; j < width; ++j) {
        const uint32_t cc = cbase[(j >> 1) * 64 + lane];
        const uint32_t off = code_off<kDict>(code_half(cc, j & 1));
        const double v = lds_ld(theta_l, off) * w_at(j) * inv;
        if (v != 0.0) lds_add_f64(lds_at(cnt_l, (off << cs) + copy_off), v);
    }
}

// a thread's share of a tile's remote records as loaded (packed: transcript | read << 22; weight or its table index) ...
template <typename WT, int kRem>
struct RecRegs {
    uint32_t pk[kRem];
    WT rw[kRem];
    uint32_t ri[kRem];
};
// ... and as the phases use them: theta[t] * w, the read inside the tile, the slot in the bucket-major queue
template <int kRem>
struct RemRegs {
    double rx[kRem];
    uint32_t rrow[kRem];
    uint32_t rslot[kRem];
};

template <typename WT, int kRem, bool kNT, bool kRemIdx>
__device__ __forceinline__ void load_records(RecRegs<WT, kRem> &rec, const TileDesc &td, uint32_t tx,
                                             const uint32_t *__restrict__ r_pk, const WT *__restrict__ r_w,
                                             const uint8_t *__restrict__ r_wi)
{
    // always kRem loads: slots beyond the tile's records re-read its last one (a tile without records: record 0 of
    // the store); neither takes part in anything (gather_records)
    const uint32_t last = td.remote_cnt ? td.remote_cnt - 1 : 0u;
    const uint32_t begin = td.remote_cnt ? td.remote_begin : 0u;
#pragma unroll
    for (int k = 0; k < kRem; ++k) {
        const uint32_t i = tx + k * kPipeThreads;
        const uint32_t o = begin + (i < td.remote_cnt ? i : last);
        rec.pk[k] = ld_stream<kNT>(&r_pk[o]);
        if (kRemIdx) rec.ri[k] = ld_stream<kNT>(&r_wi[o]);
        else rec.rw[k] = ld_stream<kNT>(&r_w[o]);
    }
}

template <typename WT, int kRem, bool kRemIdx>
__device__ __forceinline__ void gather_records(RemRegs<kRem> &rem, const RecRegs<WT, kRem> &rec, const TileDesc &td,
                                               uint32_t tx, const double *__restrict__ theta,
                                               const uint32_t *__restrict__ sd, const float *dict_l, uint32_t slack_slot)
{
    const uint32_t *sd_t = sd + td.sd_begin - td.b_min; // slot of record i = sd_t[bucket of its transcript] + i
    uint32_t rt[kRem];
#pragma unroll
    for (int k = 0; k < kRem; ++k) {
        // (a tile without records holds some other tile's record here: its bucket may lie outside the slot table)
        rt[k] = td.remote_cnt ? rec.pk[k] & ((1u << kPackRowShift) - 1u) : td.b_min << kBucketShift;
        rem.rrow[k] = td.remote_cnt ? rec.pk[k] >> kPackRowShift : 0u;
    }
#pragma unroll
    for (int k = 0; k < kRem; ++k) // em.rs:111, the weight first widened to f64 as in k_em_tile
        rem.rx[k] = (OEM_PEXP(4) ? 1.0 + (double)rt[k] : theta[rt[k]]) * (kRemIdx ? (double)dict_l[rec.ri[k]] : (double)rec.rw[k]);
    uint32_t sdv[kRem];
#pragma unroll
    for (int k = 0; k < kRem; ++k) sdv[k] = sd_t[rt[k] >> kBucketShift];
#pragma unroll
    for (int k = 0; k < kRem; ++k) {
        const uint32_t i = tx + k * kPipeThreads;
        rem.rslot[k] = i < td.remote_cnt ? sdv[k] + i : slack_slot; // (register slots without a record write the queue's slack entry)
    }
}

template <uint32_t kPer>
__device__ __forceinline__ void load_window(double (&tw)[kPer], const TileDesc &td, uint32_t tx, const double *__restrict__ theta)
{
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        const uint32_t i = tx + u * kPipeThreads;
        tw[u] = theta[td.lo + (i < td.win_len ? i : 0u)];
    }
}

__device__ __forceinline__ uint32_t copy_shift(const TileDesc &td, uint32_t cnt_entries)
{
    uint32_t cs = 0;
    while (cs < kMaxCopyShift && (td.win_len << (cs + 1)) <= cnt_entries) ++cs;
    return cs;
}

// A: the remote alignments' part of the denominators, den[read] += theta[t] * w (em.rs:98-112); records beyond a
// thread's register slots (a tile with more than kRem x 256 of them) park their product in the queue
template <typename WT, int kRem, bool kRemIdx>
__device__ __forceinline__ void remote_denominators(const RemRegs<kRem> &rem, const TileDesc &td, double *den, uint32_t tx,
                                                    const uint32_t *__restrict__ r_pk, const WT *__restrict__ r_w,
                                                    const uint8_t *__restrict__ r_wi, const double *__restrict__ theta,
                                                    const uint32_t *__restrict__ sd, double *__restrict__ queue, const float *dict_l)
{
#pragma unroll
    for (int k = 0; k < kRem; ++k)
        if (tx + k * kPipeThreads < td.remote_cnt && !OEM_PEXP(32)) lds_add_f64(&den[rem.rrow[k]], rem.rx[k]);
    const uint32_t *sd_t = sd + td.sd_begin - td.b_min;
    for (uint32_t i = tx + kRem * kPipeThreads; i < td.remote_cnt; i += kPipeThreads) {
        const uint32_t o = td.remote_begin + i;
        const uint32_t pk = r_pk[o];
        const uint32_t t = pk & ((1u << kPackRowShift) - 1u), row = pk >> kPackRowShift;
        const double x = theta[t] * (kRemIdx ? (double)dict_l[r_wi[o]] : (double)r_w[o]);
        queue[sd_t[t >> kBucketShift] + i] = x;
        lds_add_f64(&den[row], x);
    }
}

// what a workgroup carries from tile to tile
template <typename WT>
struct PipeState {
    TileDesc td;   // tile k (SGPRs)
    SliceAddr sa;
    TileDesc tdn;  // tile k + 1
    SliceAddr san;
    uint32_t ibn;
    SliceRegs<WT, kPipeCh> R[kPipeSets]; // slices of k, then -- set by set as the folds release them -- of k + 1
    RemRegs<kPipeRem> rem;               // remote records of k: gathered a tile ago
    uint32_t par;                        // which denominator buffer k uses
};

struct PipeArgs {
    const TileDesc *__restrict__ tiles;
    const uint32_t *__restrict__ codes;
    const uint32_t *__restrict__ r_pk;
    const uint32_t *__restrict__ sd;
    double *__restrict__ queue;
    const double *__restrict__ theta;
    double *__restrict__ cnt;
    const uint32_t *__restrict__ row_w_perm;
    const uint32_t *__restrict__ widx;
    const uint32_t *__restrict__ i_base;
    const uint8_t *__restrict__ r_wi;
    uint32_t n_tiles, slack_slot;
};

// One tile of the pipeline (see the head of the file).  kNext: there is a tile k + 1 to request.
template <bool kNext, typename WT, int kCopies, bool kNT, int kDict>
__device__ __forceinline__ void pipe_tile(PipeState<WT> &st, const PipeArgs &a, const WT *__restrict__ w, const WT *__restrict__ r_w,
                                          uint32_t tile, uint32_t stride, uint32_t tx, uint32_t lane, uint32_t wave,
                                          float *dict_l, double *theta_l, double *cnt_l, double (*den_l)[kTileRows])
{
    constexpr int kCh = kPipeCh, kRem = kPipeRem;
    constexpr bool kRemIdx = kDict == kWBytes || kDict == kWFused;
    constexpr uint32_t kCntEntries = kWin * (uint32_t)kCopies;
    constexpr uint32_t kPer = (kWin + kPipeThreads - 1) / kPipeThreads;
    constexpr uint32_t kRecAfter = 1; // the slice after whose fold the records of k + 1 are requested
    const TileDesc &td = st.td;
    const SliceAddr &sa = st.sa;
    double *den_k = den_l[st.par];
    const uint32_t cs = copy_shift(td, kCntEntries);
    auto iptr = [&](const SliceAddr &s, uint32_t q) -> const uint32_t * {
        return kDict == kWBytes ? a.widx + (size_t)s.ioff[q] * 64 : kDict == kWWords ? a.widx + (size_t)s.coff[q] * 64 : nullptr;
    };

    OEM_PPROBE(0);
    __syncthreads(); // ---- barrier 1: W(k) and A(k) are in LDS ------------------------------------------------
    OEM_PPROBE(1);
    const uint32_t *sd_t = a.sd + td.sd_begin - td.b_min;
    double *den_n = den_l[st.par ^ 1u];
    if (kNext) // (the other denominator buffer was last read by B(k - 1), ahead of the barrier)
        for (uint32_t i = tx; i < st.tdn.n_slices * 64; i += kPipeThreads) den_n[i] = 0.0;
    double twn[kPer];
    RecRegs<WT, kRem> recn;
    OEM_PPROBE(2);
    OEM_PPROBE(3);

    // F(k): the wavefront's slices, one read per lane; a set the fold releases goes to the same slice of k + 1
    {
        const uint32_t hi_w = sa.wid[0] > (uint32_t)kCh ? sa.wid[0] - kCh : 0u;
        __builtin_amdgcn_sched_barrier(0);
        mask_slice<WT, kCh, kDict>(st.R[0], sa.wid[0]);
        mask_slice<WT, kCh, kDict>(st.R[1], hi_w);
        if (wave < td.n_slices)
            fold_slice_b<WT, kCh, kDict, true>(st.R[0], st.R[1], sa.wid[0], wave, lane, w + (size_t)sa.woff[0] * 64,
                                               a.codes + (size_t)sa.coff[0] * 64, td, theta_l, cnt_l, den_k, a.row_w_perm, iptr(sa, 0),
                                               dict_l, cs);
        if (kNext && !OEM_PEXP(16)) {
            const SliceAddr &sn = st.san;
            const uint32_t hi_n = sn.wid[0] > (uint32_t)kCh ? sn.wid[0] - kCh : 0u;
            __builtin_amdgcn_sched_barrier(0);
            load_slice_static<WT, kCh, kNT, kDict>(st.R[0], w + (size_t)sn.woff[0] * 64, a.codes + (size_t)sn.coff[0] * 64, lane,
                                                   sn.wid[0], iptr(sn, 0));
            // (hi_n == 0: the set re-reads rows of the slice itself, in bounds)
            load_slice_static<WT, kCh, kNT, kDict>(st.R[1], w + ((size_t)sn.woff[0] + (hi_n ? kCh : 0)) * 64,
                                                   a.codes + ((size_t)sn.coff[0] + (hi_n ? kCh / 2 : 0)) * 64, lane, hi_n,
                                                   kDict == kWBytes   ? iptr(sn, 0) + (hi_n ? (kCh / 4) * 64 : 0)
                                                   : kDict == kWWords ? iptr(sn, 0) + (hi_n ? (kCh / 2) * 64 : 0)
                                                                      : nullptr);
            __builtin_amdgcn_sched_barrier(0);
        }
        OEM_PPROBE(4);
    }
#pragma unroll
    for (uint32_t q = 1; q < kPipePerWave; ++q) {
        const uint32_t s = pipe_slice_of(q, wave);
        __builtin_amdgcn_sched_barrier(0);
        mask_slice<WT, kCh, kDict>(st.R[q + 1], sa.wid[q]);
        if (s < td.n_slices)
            fold_slice_b<WT, kCh, kDict, false>(st.R[q + 1], st.R[q + 1], sa.wid[q], s, lane, w + (size_t)sa.woff[q] * 64,
                                                a.codes + (size_t)sa.coff[q] * 64, td, theta_l, cnt_l, den_k, a.row_w_perm, iptr(sa, q),
                                                dict_l, cs);
        if (kNext) {
            const SliceAddr &sn = st.san;
            __builtin_amdgcn_sched_barrier(0);
            if (!OEM_PEXP(16))
                load_slice_static<WT, kCh, kNT, kDict>(st.R[q + 1], w + (size_t)sn.woff[q] * 64, a.codes + (size_t)sn.coff[q] * 64, lane,
                                                       sn.wid[q], iptr(sn, q));
            // the window and the records of k + 1: not before the widest slice is folded (its products are the
            // register peak of the tile), two folds and a barrier ahead of the gathers that need them
            if (q == kRecAfter) {
                load_records<WT, kRem, kNT, kRemIdx>(recn, st.tdn, tx, a.r_pk, r_w, a.r_wi);
                load_window<kPer>(twn, st.tdn, tx, a.theta);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        OEM_PPROBE(4 + q);
    }
    // descriptor of k + 2: a scalar load that waits at the barrier with everyone else
    TileDesc tdnn;
    uint32_t ibnn = 0;
    if (kNext) {
        const uint32_t t2 = tile + 2 * stride < a.n_tiles ? tile + 2 * stride : a.n_tiles - 1;
        tdnn = a.tiles[t2];
        if (kDict == kWBytes) ibnn = a.i_base[t2];
    }
    __syncthreads(); // ---- barrier 2: every read's c / denom is in den_k, the count window is complete --------
    OEM_PPROBE(8);

    // G(k + 1): its records landed under the folds
    RemRegs<kRem> remn;
    if (kNext) {
        __builtin_amdgcn_sched_barrier(0);
        gather_records<WT, kRem, kRemIdx>(remn, recn, st.tdn, tx, a.theta, a.sd, dict_l, a.slack_slot);
        __builtin_amdgcn_sched_barrier(0);
    }
    OEM_PPROBE(9);
    // B(k): queue <- x * (c_i / denom_i)
#pragma unroll
    for (int k = 0; k < kRem; ++k)
        if (!OEM_PEXP(8)) __builtin_nontemporal_store(st.rem.rx[k] * den_k[st.rem.rrow[k]], &a.queue[st.rem.rslot[k]]);
    for (uint32_t i = tx + kRem * kPipeThreads; i < td.remote_cnt; i += kPipeThreads) {
        const uint32_t o = td.remote_begin + i;
        const uint32_t pk = a.r_pk[o];
        const uint32_t t = pk & ((1u << kPackRowShift) - 1u), row = pk >> kPackRowShift;
        const uint32_t q = sd_t[t >> kBucketShift] + i;
        a.queue[q] = a.queue[q] * den_k[row];
    }
    OEM_PPROBE(10);
    // flush the window (consecutive lanes -> consecutive addresses; a thread starts at copy `lane` of its entry) and
    // leave it zeroed for the next tile
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        const uint32_t i = tx + u * kPipeThreads;
        if (i < td.win_len) {
            double v = 0.0;
            for (uint32_t p = 0; p < (1u << cs); ++p) {
                double *e = &cnt_l[(i << cs) + ((p + lane) & ((1u << cs) - 1u))];
                v += *e;
                *e = 0.0;
            }
            if (v != 0.0 && !OEM_PEXP(64)) unsafeAtomicAdd(&a.cnt[td.lo + i], v);
        }
    }
    OEM_PPROBE(11);
    if (kNext) {
        // W(k + 1), A(k + 1)
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t i = tx + u * kPipeThreads;
            if (i < st.tdn.win_len) theta_l[i] = twn[u];
        }
        remote_denominators<WT, kRem, kRemIdx>(remn, st.tdn, den_n, tx, a.r_pk, r_w, a.r_wi, a.theta, a.sd, a.queue, dict_l);
        // rotate
        st.td = st.tdn;
        st.sa = st.san;
        st.rem = remn;
        st.par ^= 1u;
        st.tdn = tdnn;
        st.ibn = ibnn;
        slice_addrs(st.tdn, st.ibn, wave, st.san);
    }
    OEM_PPROBE(12);
}

template <typename WT, int kMinWaves, int kCopies, bool kNT, int kDict>
__global__ __launch_bounds__(kPipeThreads, kMinWaves) void k_em_tile_p(
    const TileDesc *__restrict__ tiles, const uint32_t *__restrict__ codes, const WT *__restrict__ w,
    const uint32_t *__restrict__ r_pk, const WT *__restrict__ r_w, const uint32_t *__restrict__ sd,
    double *__restrict__ queue, const double *__restrict__ theta, double *__restrict__ cnt, const EmState *state,
    const uint32_t *__restrict__ row_w_perm, uint32_t n_tiles, uint32_t n_remote, const uint32_t *__restrict__ widx,
    const uint32_t *__restrict__ i_base, const float *__restrict__ dict, const uint8_t *__restrict__ r_wi)
{
    __shared__ float dict_l[dict_entries<kDict>()];
    __shared__ double theta_l[kWin];
    __shared__ double cnt_l[kWin * kCopies];
    __shared__ double den_l[2][kTileRows];
    constexpr int kCh = kPipeCh, kRem = kPipeRem;
    constexpr bool kRemIdx = kDict == kWBytes || kDict == kWFused;
    constexpr uint32_t kPer = (kWin + kPipeThreads - 1) / kPipeThreads;

    uint32_t tile = blockIdx.x;        // (the launch keeps gridDim.x <= n_tiles)
    const uint32_t stride = gridDim.x;
    PipeState<WT> st;
    st.td = tiles[tile];
    const uint32_t ib = kDict == kWBytes ? i_base[tile] : 0u;
    {
        const uint32_t t1 = tile + stride < n_tiles ? tile + stride : n_tiles - 1;
        st.tdn = tiles[t1];
        st.ibn = kDict == kWBytes ? i_base[t1] : 0u;
    }
    if (state && state->done) return;
    const uint32_t tx = threadIdx.x;
    const uint32_t lane = tx & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tx >> 6);
    const PipeArgs a{tiles, codes, r_pk, sd, queue, theta, cnt, row_w_perm, widx, i_base, r_wi, n_tiles,
                     n_remote + blockIdx.x * kPipeWaves + wave}; // (the wavefront's own slack entry of the queue)
    auto iptr = [&](const SliceAddr &s, uint32_t q) -> const uint32_t * {
        return kDict == kWBytes ? widx + (size_t)s.ioff[q] * 64 : kDict == kWWords ? widx + (size_t)s.coff[q] * 64 : nullptr;
    };

    // ---- the first tile of the workgroup: nothing to hide its loads under --------------------------------------
    slice_addrs(st.td, ib, wave, st.sa);
    slice_addrs(st.tdn, st.ibn, wave, st.san);
    st.par = 0;
    double tw[kPer];
    RecRegs<WT, kRem> rec;
    load_records<WT, kRem, kNT, kRemIdx>(rec, st.td, tx, r_pk, r_w, r_wi);
    load_window<kPer>(tw, st.td, tx, theta);
    {
        const SliceAddr &s0 = st.sa;
        const uint32_t hi_w = s0.wid[0] > (uint32_t)kCh ? s0.wid[0] - kCh : 0u;
        load_slice_static<WT, kCh, kNT, kDict>(st.R[0], w + (size_t)s0.woff[0] * 64, codes + (size_t)s0.coff[0] * 64, lane, s0.wid[0], iptr(s0, 0));
        load_slice_static<WT, kCh, kNT, kDict>(st.R[1], w + ((size_t)s0.woff[0] + (hi_w ? kCh : 0)) * 64,
                                               codes + ((size_t)s0.coff[0] + (hi_w ? kCh / 2 : 0)) * 64, lane, hi_w,
                                               kDict == kWBytes   ? iptr(s0, 0) + (hi_w ? (kCh / 4) * 64 : 0)
                                               : kDict == kWWords ? iptr(s0, 0) + (hi_w ? (kCh / 2) * 64 : 0)
                                                                  : nullptr);
#pragma unroll
        for (uint32_t q = 1; q < kPipePerWave; ++q)
            load_slice_static<WT, kCh, kNT, kDict>(st.R[q + 1], w + (size_t)s0.woff[q] * 64, codes + (size_t)s0.coff[q] * 64, lane, s0.wid[q], iptr(s0, q));
    }
    // once per workgroup: the weight table, a zeroed count window
    if (kDict != kWPlain)
        for (uint32_t i = tx; i < (uint32_t)dict_entries<kDict>(); i += kPipeThreads) dict_l[i] = dict[i];
    for (uint32_t i = tx; i < kWin * (uint32_t)kCopies; i += kPipeThreads) cnt_l[i] = 0.0;
    for (uint32_t i = tx; i < st.td.n_slices * 64; i += kPipeThreads) den_l[0][i] = 0.0;
    __syncthreads(); // (the gathers read the table in LDS, the denominators are cleared)
    gather_records<WT, kRem, kRemIdx>(st.rem, rec, st.td, tx, theta, sd, dict_l, a.slack_slot);
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        const uint32_t i = tx + u * kPipeThreads;
        if (i < st.td.win_len) theta_l[i] = tw[u];
    }
    remote_denominators<WT, kRem, kRemIdx>(st.rem, st.td, den_l[0], tx, r_pk, r_w, r_wi, theta, sd, queue, dict_l);

    // ---- the pipeline ------------------------------------------------------------------------------------------
    while (tile + stride < n_tiles) {
        pipe_tile<true, WT, kCopies, kNT, kDict>(st, a, w, r_w, tile, stride, tx, lane, wave, dict_l, theta_l, cnt_l, den_l);
        tile += stride;
    }
    pipe_tile<false, WT, kCopies, kNT, kDict>(st, a, w, r_w, tile, stride, tx, lane, wave, dict_l, theta_l, cnt_l, den_l);
}

template <typename WT, bool kNT, int kDict>
void launch_pipe(oem_store *s, const WT *w, const WT *r_w, const double *theta, double *cnt, const EmState *state,
                 const uint32_t *row_w_perm, uint32_t grid)
{
    const DeviceTiled &t = s->tiled;
    hipLaunchKernelGGL((k_em_tile_p<WT, OEM_PIPE_WAVES, OEM_COPIES, kNT, kDict>), dim3(grid), dim3(kPipeThreads), 0, s->stream, t.tiles,
                       t.codes, w, t.r_pk, r_w, t.sd, t.queue, theta, cnt, state, row_w_perm, t.n_tiles, (uint32_t)t.n_remote, t.widx,
                       t.i_base, t.dict, t.r_wi);
}

} // namespace

#ifdef OEM_TESTING
int pipe_probe_set(unsigned long long *d) // oem_debug_tile_probe (oem_tile_kernels.hip)
{
    OEM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pipe_probe), &d, sizeof(d)));
    return OEM_OK;
}
#endif

// resident workgroup slots of the store's device (testing build: OEM_PIPE_SLOTS forces a few, so that small stores
// walk many tiles per workgroup)
static uint32_t pipe_slots(const oem_store *s)
{
    static std::mutex mu;
    static int cus_of[64] = {0};
    int cus = 256;
    {
        std::lock_guard<std::mutex> lk(mu);
        const int d = s->device >= 0 && s->device < 64 ? s->device : 0;
        if (cus_of[d] == 0) {
            int v = 0;
            cus_of[d] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, s->device) == hipSuccess && v > 0) ? v : 256;
        }
        cus = cus_of[d];
    }
    const long forced = knob("OEM_PIPE_SLOTS", 0);
    const uint32_t slots = forced > 0 ? (uint32_t)forced : (uint32_t)cus * OEM_PIPE_WAVES;
    return slots < kPipeMaxGrid ? slots : kPipeMaxGrid;
}

// Whether a store's E/M pass runs as the pipeline.  MEASURED AND NOT SHIPPED (profiles/r05_notes.md): at C3 the
// pipeline's tile iteration is 13.5 us against k_em_tile's 15.5 us tile lifetime, but its 128 VGPRs / 37 KiB allow
// four workgroups per CU against five, and persistent workgroups with a static stride leave 12-18 % of their slots
// idle (tile costs differ by 2x; the hardware dispatcher balances k_em_tile's workgroups for free): pass 0.176 ms
// against 0.146.  The product library therefore never takes it (knob() returns the default there); the test-only
// library does with OEM_TILE_PIPE=1 -- one EM problem, the narrow window, f32 or coded weights, packed records.
bool tile_pipeline_applies(const oem_store *s, const BatchState *problems)
{
    const DeviceTiled &t = s->tiled;
    if (problems || t.win_cap > kWin || s->csr.w_is_f64 || !t.packed || t.n_remote == 0 || t.n_tiles == 0) return false;
    return knob("OEM_TILE_PIPE", 0) == 1;
}

int launch_tile_pipeline(oem_store *s, const double *theta, double *cnt, const EmState *state, const uint32_t *row_w_perm, bool nt)
{
    const DeviceTiled &t = s->tiled;
    const uint32_t slots = pipe_slots(s);
    // every workgroup walks the same number of tiles (+- 1): rounds = ceil(tiles / slots), grid = ceil(tiles / rounds)
    const uint32_t rounds = (t.n_tiles + slots - 1) / slots;
    const uint32_t grid = (t.n_tiles + rounds - 1) / rounds;
    static_assert(kPipeWaves == 4, "kQueueSlack: four wavefronts per workgroup");
    const bool coded = t.dict_n > 0 && !t.dict_fused && knob("OEM_NO_DICT", 0) == 0;
    const bool bytes = coded && !t.dict_words, words = coded && t.dict_words;
#define OEM_PIPE(DICT)                                                                                  \
    do {                                                                                                \
        if (nt) launch_pipe<float, true, DICT>(s, t.w32, t.r_w32, theta, cnt, state, row_w_perm, grid);   \
        else launch_pipe<float, false, DICT>(s, t.w32, t.r_w32, theta, cnt, state, row_w_perm, grid);     \
    } while (0)
    if (words) OEM_PIPE(kWWords);
    else if (t.dict_fused) OEM_PIPE(kWFused);
    else if (bytes) OEM_PIPE(kWBytes);
    else OEM_PIPE(kWPlain);
#undef OEM_PIPE
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

} // namespace oem
#endif // OEM_TESTING
