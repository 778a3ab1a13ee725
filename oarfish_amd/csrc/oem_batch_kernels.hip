// oem_batch_kernels.hip -- kBatch bootstrap replicates per pass over the resident matrix.
//
// em::bootstrap (em.rs:292-314) runs num_boot independent resampled EMs, each a serial
// do_em over random_sampling_iter (em.rs:273-290); every replicate streams the whole store
// again.  Here kBatch replicates ("slots") share one pass over the tiled matrix, and kChains such
// batches run side by side on their own HIP streams (oem_api.hip): one chain's streaming fold /
// rel-diff kernels run under another's tile kernel.
//
// What amortises and what does not.  The matrix streams (weights, window codes, remote
// records: ~0.65 GB at 10 M reads) and the L2 requests of the remote theta gathers are paid
// once per pass whatever the number of slots: theta is laid out [transcript][slot], so the
// kEB = 4 slots of an epoch sit in one 32-byte piece of a transcript's kBatch * 8 bytes.  The
// per-slot work (LDS traffic of the local alignments, 8 bytes of queue per remote alignment) is
// not shared, and measured it is what the pass costs: ~150 us per slot at 4, 8 or 16 slots
// (profiles/r02_notes.md) -- hence kBatch = 4 (one epoch) and two chains, which overlap.
//
// Factored increments (round 4).  em.rs:119-130 adds theta_t * w / denom per alignment; the sum over the
// reads of a tile factors as theta_t * sum_i (w_it * c_i / denom_i), so the scatter pass adds u = w * (c / denom)
// -- no second LDS read of theta per (alignment, slot) -- and theta is multiplied in ONCE per window entry when
// the window is flushed (and once per bucket entry by the fold, for the remote alignments: a queue entry is
// w * c / denom).  Same value up to rounding (one multiplication moved across the sum); it halves the LDS reads
// of the local phase and frees the 24 registers that held theta * w of the remote records across it.
// A remote alignment's queue entry is ONE 32-byte piece [record][slot] (one store instead of four into four
// planes), and k_remote_fold_b folds the four slots of a bucket in one workgroup (4096 transcripts x 4 slots x
// 8 B = 128 KiB of LDS): the destinations are read once, not once per slot, and the counts of both kernels land
// in the one [T][slot] array.
//
// k_em_tile_e: one workgroup per tile, the tile's matrix data loaded ONCE into registers,
// then kE "epochs" of kEB = 4 slots each run over it: per epoch the theta window, the count
// window and the denominators of 4 slots live in LDS (64 KiB per workgroup => two workgroups
// per CU overlap one tile's memory phases with the other's LDS phases), one read per lane as
// in k_em_tile (oem_tile_kernels.hip).  Inside an epoch lane l works on slot (j + l) mod 4 at
// step j ("rotated" slot order): lanes that add into the same transcript in the same
// instruction hit four different addresses, the job the four interleaved count-window copies
// do in k_em_tile, with no extra LDS.
//
// Every slot keeps its own loop state on the device and walks the reference's state machine
// by itself: RUNNING -(stopping rule em.rs:212 / max_iter em.rs:181)-> FINAL (theta < 1e-5
// read as 0, em.rs:238-242; one more pass, em.rs:245-252) -> FINISHED (counts parked in
// `out`, slot ignored from then on, handed the next replicate by the host).
#include <atomic>

#include "oem_internal.h"
#include "oem_lane_runs.h"

namespace oem {

namespace {

// phase timestamps of k_em_tile_e (test-only library; see OEM_PROBE in oem_tile_kernels.hip)
#ifdef OEM_TESTING
__device__ unsigned long long *g_tile_e_probe = nullptr;
#define OEM_PROBE_E(i)                                                                                            \
    do {                                                                                                          \
        if (g_tile_e_probe && threadIdx.x == 0) g_tile_e_probe[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); \
    } while (0)
// cost attribution (test-only library, OEM_TILE_EXP): parts of the kernel switched off -- the results are wrong, the
// time says what the part costs.  1 queue stores, 2 remote denominator atomics, 4 local scatter atomics,
// 8 local theta reads of the denominators, 16 remote theta gathers
__device__ unsigned int g_tile_e_exp = 0;
#define OEM_EXP_E(bit) ((exp_mask & (bit)) != 0u)
#else
#define OEM_PROBE_E(i) do { } while (0)
#define OEM_EXP_E(bit) false
#endif

#ifndef OEM_E_MULT_NT
#define OEM_E_MULT_NT 0 // the reads' multiplicity words streamed non-temporally (A/B)
#endif
#ifndef OEM_E_RD_GROUP
#define OEM_E_RD_GROUP 2 // alignments whose LDS reads (x 4 slots) are in flight together in pass 1 of fold_slice_e
#endif
#ifndef OEM_E_WR_GROUP
#define OEM_E_WR_GROUP 1 // alignments whose LDS atomics are issued together in pass 2
#endif
#ifndef OEM_E_QUEUE_SC
#define OEM_E_QUEUE_SC 0 // a queue piece's stores: 0 non-temporal, 1 `sc1`, 2 `sc0 sc1` (write-through: the line leaves the L2), 3 plain
#endif
constexpr int kB = kBatch;
constexpr int kEB = 4;             // slots per epoch
constexpr int kE = kB / kEB;       // epochs per pass
constexpr int kBCh = 8;            // local alignments per read kept in registers
constexpr int kTileThreadsE = 512; // 8 wavefronts, 2 slices each
constexpr int kRemE = 3;           // remote alignments per thread kept in registers
constexpr int kFoldThreadsB = 1024;
// Count window of k_em_tile_e: [c][copy][b].  The rotated slot order makes the four slots of an entry four
// addresses; lanes l and l + 4 k still meet on one.  The 14 KiB that two 64 KiB workgroups leave of a CU's LDS hold
// further copies for the tiles whose window is short enough (a dense store's tiles: few transcripts, the ones whose
// lanes add into the same entries).  The theta window and the count copies share one pool of 46 KiB -- theta takes
// what the tile's window needs, the copies the rest: 8 copies up to 163 window entries, 4 up to 294, 2 up to 490;
// lane l adds into copy (l >> 2) mod copies.  (k_em_tile: 2 count-window copies instead of 4 cost 21 % of its
// pass; here 1 -> 2 | 4 copies took 6 % off the batched pass, profiles/r04_notes.md.)
constexpr uint32_t kPoolE = 5888; // 46 KiB: theta + counts 46 + denominators 32 = 78 KiB per workgroup
constexpr uint32_t kMaxCopyShiftE = 3;
// (Three workgroups per CU on tiles of <= 512 reads -- half the denominators, 52 KiB, 70 VGPRs -- were built and measured
// in round 5: the batched pass 5 % SLOWER, a tile's window costs the same for half the reads; removed in round 6.)
template <uint32_t kRows> struct TileShapeE {
    static_assert(kRows == kTileRows, "one tile shape");
    static constexpr uint32_t pool = kPoolE;
    static constexpr int rem = kRemE;
    static constexpr int min_waves = 4;
};
static_assert(kPoolE >= 2 * kWin * kEB, "theta and one copy of the widest window must fit");
static_assert(kB % kEB == 0 && kE >= 1, "kBatch must be a multiple of the epoch width");
static_assert((kB & (kB - 1)) == 0 && kB <= 16, "row multiplicities are packed kB bytes per read");

__device__ __forceinline__ void lds_add(double *p, double v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // ds_add_f64
}
__device__ __forceinline__ double lds_ld_b(const double *base, uint32_t byte_off)
{
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ double *lds_at_b(double *base, uint32_t byte_off)
{
    return reinterpret_cast<double *>(reinterpret_cast<char *>(base) + byte_off);
}

// LDS byte offset of a 16-bit window code.  A store with at most 128 distinct weights carries a table index in the
// spare bits of the pair of codes (bits 0..2 and 12..15 of either half: oem_layout_dict.hip, read by k_em_tile); this kernel reads the f32 weight
// stream and only has to look past them (the batched kernel serves narrow-window stores: offsets are bits 3..11).
__device__ __forceinline__ uint32_t code_off_b(uint32_t half) { return half & 0x0ff8u; }
// kFused (a store of <= 128 distinct weights, as_prob alone): no weight stream, the weight of a local alignment is the
// table entry its code's spare bits name (entry 0 = 0.0: padding needs no masking), the table sits in LDS; a remote
// record's weight is the entry its index byte names (DeviceTiled::r_wi).  The same f32 values as the stream.
constexpr uint32_t kDictE = 128;
__device__ __forceinline__ uint32_t code_widx_b(uint32_t c, int h) // (see code_widx in oem_tile_common.h)
{
    const uint32_t r = h ? ((c >> 26) | (c << 6)) : ((c >> 10) | (c << 22));
    return (r >> 2) & 0x7fu;
}

// The 32-byte queue piece of a remote record (its four slots).  Non-temporal stores keep the line in the XCD's L2
// until it is evicted; `sc1` stores write through and drop it (MI355X_MICROARCH.md, "stores of each flavour").
__device__ __forceinline__ void store_queue_piece(double *qp, const double (&qv)[4])
{
#if OEM_E_QUEUE_SC == 1 || OEM_E_QUEUE_SC == 2
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const d2_t lo = {qv[0], qv[1]}, hi = {qv[2], qv[3]};
#if OEM_E_QUEUE_SC == 1
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" ::"v"(qp), "v"(lo), "v"(hi) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc0 sc1" ::"v"(qp), "v"(lo), "v"(hi) : "memory");
#endif
#elif OEM_E_QUEUE_SC == 3
#pragma unroll
    for (int b = 0; b < 4; ++b) qp[b] = qv[b];
#else
#pragma unroll
    for (int b = 0; b < 4; ++b) __builtin_nontemporal_store(qv[b], &qp[b]);
#endif
}

template <typename WT>
struct SliceRegsB {
    WT w[kBCh];
    uint32_t c[kBCh / 2];
};

template <bool kNT, typename T>
__device__ __forceinline__ T ld_stream_b(const T *p)
{
    return kNT ? __builtin_nontemporal_load(p) : *p; // see ld_stream in oem_tile_kernels.hip
}

template <bool kNT, typename WT, bool kFused = false>
__device__ __forceinline__ void load_slice_b(SliceRegsB<WT> &r, const WT *__restrict__ wbase,
                                             const uint32_t *__restrict__ cbase, uint32_t lane,
                                             uint32_t width)
{
#pragma unroll
    for (int g = 0; g < kBCh / 2; ++g) {
        if ((uint32_t)(2 * g) < width) {
            if (!kFused) {
                r.w[2 * g] = ld_stream_b<kNT>(&wbase[(2 * g) * 64 + lane]);
                r.w[2 * g + 1] = ld_stream_b<kNT>(&wbase[(2 * g + 1) * 64 + lane]);
            }
            r.c[g] = ld_stream_b<kNT>(&cbase[g * 64 + lane]);
        } else {
            r.w[2 * g] = (WT)0;
            r.w[2 * g + 1] = (WT)0;
            r.c[g] = 0u;
        }
    }
    // (The second element of the last pair of an odd-width slice belongs to the next read and must carry no
    // weight: fold_slice_e masks it at the point of use.  Masking it HERE made the compiler wait for every pair
    // right behind its loads -- four serialised round trips per register set in the kernel's prologue.)
}

// Alignments beyond the kBCh a register set holds are reloaded by both passes of the fold.  One alignment at a
// time that is two synchronous loads per alignment (the widest slice of a tile has ~8 of them, and the
// wavefront that owns it holds up the tile's barriers); four at a time it is one round trip per four.  Rows
// past the slice's width are clamped to its last row and carry no weight.
template <typename WT, bool kFused>
__device__ __forceinline__ void load_over4(const WT *__restrict__ wbase, const uint32_t *__restrict__ cbase, uint32_t lane,
                                           uint32_t i0 /* even */, uint32_t width, WT wv[4], uint32_t off[4], const float *dict_l)
{
    const uint32_t lastp = (width - 1) >> 1;
    const uint32_t p0 = i0 >> 1, p1 = p0 + 1 <= lastp ? p0 + 1 : lastp;
    const uint32_t c0 = cbase[p0 * 64 + lane], c1 = cbase[p1 * 64 + lane];
    if (kFused) {
        wv[0] = (WT)dict_l[code_widx_b(c0, 0)];
        wv[1] = (WT)dict_l[code_widx_b(c0, 1)];
        wv[2] = (WT)dict_l[code_widx_b(c1, 0)];
        wv[3] = (WT)dict_l[code_widx_b(c1, 1)];
    } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const uint32_t i = i0 + m;
            wv[m] = wbase[(i < width ? i : width - 1) * 64 + lane];
        }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
        if (i0 + m >= width) wv[m] = (WT)0;
    off[0] = code_off_b(c0) * kEB;
    off[1] = code_off_b(c0 >> 16) * kEB;
    off[2] = code_off_b(c1) * kEB;
    off[3] = code_off_b(c1 >> 16) * kEB;
}

// The fold of one slice for the four slots of an epoch: denominators (pass 1), c_ib / denom_ib, scatter (pass 2,
// which derives its LDS addresses again from the packed operands: that keeps the kernel inside 128 VGPRs).
// kHasHi: the slice's alignments 8..15 sit in a second register set (`hi`) -- the widest slice of the
// wavefront, see fold_first in oem_tile_kernels.hip; the first set is handed to the NEXT slice's loads as soon
// as the scatter is done with it (`next_*`).  Alignments beyond the register-resident ones are reloaded four at
// a time by both passes.
template <typename WT, bool kNT, bool kHasHi, uint32_t kRows, bool kFused>
__device__ __forceinline__ void fold_slice_e(SliceRegsB<WT> &lo, SliceRegsB<WT> &hi, uint32_t width, uint32_t mq,
                                             uint32_t rl, uint32_t lane, const WT *__restrict__ wbase,
                                             const uint32_t *__restrict__ cbase, const double *theta_l, double *cnt_l,
                                             double *den_l, const uint32_t (&rot8)[kEB], uint32_t act_e, bool load_next,
                                             const WT *__restrict__ next_w, const uint32_t *__restrict__ next_c,
                                             uint32_t next_width, uint32_t exp_mask, uint32_t cs, uint32_t cpy,
                                             const float *dict_l)
{
    constexpr uint32_t kReg = kHasHi ? 2 * kBCh : kBCh; // register-resident alignments
    auto wt = [&](const SliceRegsB<WT> &r, int k) -> double {
        if (kFused) return (double)dict_l[code_widx_b(r.c[k >> 1], k & 1)];
        return (double)r.w[k];
    };
    // Every use of the slice's registers stays below this point: without the pins the compiler hoists the
    // f32 -> f64 conversions of the weights up to their loads and waits for each pair right behind them.
#pragma unroll
    for (int k = 0; k < kBCh; ++k) if (!kFused) asm volatile("" : "+v"(lo.w[k]));
#pragma unroll
    for (int k = 0; k < kBCh / 2; ++k) asm volatile("" : "+v"(lo.c[k]));
    if (kHasHi) {
#pragma unroll
        for (int k = 0; k < kBCh; ++k) if (!kFused) asm volatile("" : "+v"(hi.w[k]));
#pragma unroll
        for (int k = 0; k < kBCh / 2; ++k) asm volatile("" : "+v"(hi.c[k]));
    }
    // Pass 1 in the slots' natural order: the four slots of a window entry are one 32-byte piece, read as two 16-byte
    // halves -- two LDS instructions per alignment and no address arithmetic, against four 8-byte reads in the lanes'
    // rotated slot order with an address each.  The rotation only matters to the atomics of pass 2: c / denom of
    // the four slots is put into the lane's order once per slice.
    double dn[kEB];
#pragma unroll
    for (int b = 0; b < kEB; ++b) dn[b] = den_l[b * kRows + rl];
    static_assert(kEB == 4, "two 16-byte halves of a [c][4] entry");
    auto add4 = [&](uint32_t off, double wk) {
        const char *e = reinterpret_cast<const char *>(theta_l) + (OEM_EXP_E(8u) ? lane * 32u : off);
        const double2 lo2 = *reinterpret_cast<const double2 *>(e), hi2 = *reinterpret_cast<const double2 *>(e + 16);
        dn[0] += lo2.x * wk;   // em.rs:111
        dn[1] += lo2.y * wk;
        dn[2] += hi2.x * wk;
        dn[3] += hi2.y * wk;
    };
#pragma unroll
    for (int k = 0; k < kBCh; ++k) {
        const uint32_t off = code_off_b((k & 1) ? lo.c[k >> 1] >> 16 : lo.c[k >> 1]) * kEB;
        add4(off, ((k & 1) && (uint32_t)k >= width) ? 0.0 : wt(lo, k));
        if ((k & (OEM_E_RD_GROUP - 1)) == OEM_E_RD_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
    }
    if (kHasHi && width > (uint32_t)kBCh) { // wave-uniform
#pragma unroll
        for (int k = 0; k < kBCh; ++k) {
            const uint32_t off = code_off_b((k & 1) ? hi.c[k >> 1] >> 16 : hi.c[k >> 1]) * kEB;
            add4(off, ((k & 1) && (uint32_t)(k + kBCh) >= width) ? 0.0 : wt(hi, k));
            if ((k & (OEM_E_RD_GROUP - 1)) == OEM_E_RD_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (uint32_t i0 = kReg; i0 < width; i0 += 4) { // reads with more alignments than the registers hold
        WT wv[4];
        uint32_t off4[4];
        load_over4<WT, kFused>(wbase, cbase, lane, i0, width, wv, off4, dict_l);
#pragma unroll
        for (int m = 0; m < 4; ++m) add4(off4[m], (double)wv[m]);
    }
    double inv_n[kEB];
#pragma unroll
    for (int b = 0; b < kEB; ++b) {
        const double scale = (double)((mq >> (8 * b)) & 0xffu);
        inv_n[b] = (((act_e >> b) & 1u) && dn[b] > OEM_EM_DENOM_THRESH) ? scale / dn[b] : 0.0; // em.rs:115
        den_l[b * kRows + rl] = inv_n[b];
    }
    double inv[kEB]; // inv[j]: of the slot this lane works on at step j, (j + lane) mod 4
#pragma unroll
    for (int j = 0; j < kEB; ++j) {
        const uint32_t b = rot8[j] >> 3;
        inv[j] = b == 0 ? inv_n[0] : b == 1 ? inv_n[1] : b == 2 ? inv_n[2] : inv_n[3];
    }
#pragma unroll
    for (int k = 0; k < kBCh; ++k) if (!kFused) asm volatile("" : "+v"(lo.w[k]));
#pragma unroll
    for (int k = 0; k < kBCh / 2; ++k) asm volatile("" : "+v"(lo.c[k]));
#pragma unroll
    for (int k = 0; k < kBCh; ++k) {
        if ((uint32_t)k < width) { // wave-uniform
            const uint32_t off = code_off_b((k & 1) ? lo.c[k >> 1] >> 16 : lo.c[k >> 1]) * kEB;
            const double wk = wt(lo, k);
#pragma unroll
            for (int j = 0; j < kEB; ++j) {
                const double v = wk * inv[j]; // (theta is multiplied in when the window is flushed)
                if (v != 0.0 && !OEM_EXP_E(4u)) lds_add(lds_at_b(cnt_l, (off << cs) + cpy + rot8[j]), v);    // em.rs:128-129
            }
        }
        if ((k & (OEM_E_WR_GROUP - 1)) == OEM_E_WR_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
    }
    // the first set is free: the next slice's loads go out under the rest of this fold
    if (kHasHi && load_next) load_slice_b<kNT, WT, kFused>(lo, next_w, next_c, lane, next_width);
    if (kHasHi && width > (uint32_t)kBCh) {
#pragma unroll
        for (int k = 0; k < kBCh; ++k) {
            if ((uint32_t)(k + kBCh) < width) { // wave-uniform
                const uint32_t off = code_off_b((k & 1) ? hi.c[k >> 1] >> 16 : hi.c[k >> 1]) * kEB;
                const double wk = wt(hi, k);
#pragma unroll
                for (int j = 0; j < kEB; ++j) {
                    const double v = wk * inv[j];
                    if (v != 0.0 && !OEM_EXP_E(4u)) lds_add(lds_at_b(cnt_l, (off << cs) + cpy + rot8[j]), v);
                }
            }
            if ((k & (OEM_E_WR_GROUP - 1)) == OEM_E_WR_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (uint32_t i0 = kReg; i0 < width; i0 += 4) {
        WT wv[4];
        uint32_t off4[4];
        load_over4<WT, kFused>(wbase, cbase, lane, i0, width, wv, off4, dict_l);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const double wk = (double)wv[m];
#pragma unroll
            for (int j = 0; j < kEB; ++j) {
                const double v = wk * inv[j];
                if (v != 0.0 && !OEM_EXP_E(4u)) lds_add(lds_at_b(cnt_l, (off4[m] << cs) + cpy + rot8[j]), v);
            }
        }
    }
}

// LDS layouts of one epoch (b = slot inside the epoch, c = window entry, r = read of the tile):
//   theta_l, cnt_l : [c][b]  byte (c * kEB + b) * 8 = code * kEB + b * 8   (code = 8 c, as stored)
//   den_l          : [b][r]  remote part of the denominators, then c_ib / denom_ib
// WT = float (as_prob alone: exact) or double (as_prob * cov_prob, the coverage model: em.rs:107-111)
// a remote record and its queue slot: see ld_remote in oem_tile_kernels.hip / oem_layout_pack.hip
template <bool kPacked, bool kNT>
__device__ __forceinline__ void ld_remote_b(const uint32_t *__restrict__ r_a, const uint16_t *__restrict__ r_row, uint32_t o,
                                            uint32_t tid_base, uint32_t &t, uint32_t &row)
{
    if (kPacked) {
        const uint32_t pk = ld_stream_b<kNT>(&r_a[o]);
        t = tid_base + (pk & ((1u << kPackRowShift) - 1u));
        row = pk >> kPackRowShift;
    } else {
        t = ld_stream_b<kNT>(&r_a[o]);
        row = ld_stream_b<kNT>(&r_row[o]);
    }
}

template <bool kNT, typename WT, bool kPacked, uint32_t kRows, bool kFused>
__global__ __launch_bounds__(kTileThreadsE, TileShapeE<kRows>::min_waves) void k_em_tile_e(
    const TileDesc *__restrict__ tiles, const uint32_t *__restrict__ codes,
    const WT *__restrict__ w, const uint32_t *__restrict__ r_a, const WT *__restrict__ r_w,
    const uint16_t *__restrict__ r_row, const uint32_t *__restrict__ sd, uint32_t problem_size,
    double *__restrict__ queue /* [n_remote][kB]: w * c / denom of every remote alignment, the slots side by side */,
    const double *__restrict__ theta /* [T][kB] */, double *__restrict__ cnt /* [T][kB] */,
    const BatchState *__restrict__ st, const uint8_t *__restrict__ row_w /* [rows][kB], tile order */,
    const float *__restrict__ dict, const uint8_t *__restrict__ r_wi /* kFused: the weight table, the records' indices */)
{
    uint32_t act = 0; // slots that take part in this pass (RUNNING or FINAL)
    uint32_t fin = 0; // slots on their final pass: theta < 1e-5 reads as 0 (em.rs:238-242)
#pragma unroll
    for (int b = 0; b < kB; ++b) {
        const uint32_t ph = st[b].phase;
        act |= (ph != kPhaseFinished) ? (1u << b) : 0u;
        fin |= (ph == kPhaseFinal) ? (1u << b) : 0u;
    }
    if (!act) return;
    OEM_PROBE_E(0);
#ifdef OEM_TESTING
    const uint32_t exp_mask = g_tile_e_exp; // (read once: an SGPR)
#else
    constexpr uint32_t exp_mask = 0u;
#endif

    constexpr uint32_t kPool = TileShapeE<kRows>::pool;
    constexpr int kRem = TileShapeE<kRows>::rem; // remote alignments per thread kept in registers
    __shared__ __attribute__((aligned(16))) double pool_l[kPool]; // theta window [win_len][kEB], then the count copies [win_len][copies][kEB]
    __shared__ double den_l[kEB * kRows];
    __shared__ float dict_l[kFused ? kDictE : 1];

    constexpr uint32_t kWaves = kTileThreadsE / 64;
    constexpr uint32_t kSlices = kRows / 64;
    constexpr uint32_t kPerWave = kSlices / kWaves;
    const uint32_t tx = threadIdx.x;
    const uint32_t lane = tx & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tx >> 6);

    // slices come in descending width: wavefront w takes slice w and slice 15 - w (the widest with the
    // narrowest), not w and w + 8 -- the tile's barriers wait for the wavefront with the most rows
    auto slice_of = [&](uint32_t q) -> uint32_t { return (q & 1u) ? (q + 1) * kWaves - 1 - wave : q * kWaves + wave; };
    // (one workgroup per tile.  Persistent workgroups -- 512 of them striding over the tiles, so that a tile's queue
    // stores and flush atomics drain under the next tile's loads -- were measured at +13 % on the batched pass,
    // profiles/r04_notes.md: like k_em_tile in round 3, the hardware dispatcher does this better.)
    const TileDesc td = tiles[blockIdx.x];
    uint32_t woff[kPerWave], coff[kPerWave], wid[kPerWave];
    {
        uint32_t accw = td.w_base, accc = td.c_base;
#pragma unroll
        for (uint32_t i = 0; i < kSlices; ++i) {
            const uint32_t wi = td.width[i];
            if (i == slice_of(i / kWaves)) {
                woff[i / kWaves] = accw;
                coff[i / kWaves] = accc;
                wid[i / kWaves] = wi;
            }
            accw += wi;
            accc += (wi + 1) >> 1;
        }
    }

    OEM_PROBE_E(1);
    // ---- the tile's matrix data, loaded once and kept in registers over all epochs ----------
    // the theta window of the (only) epoch depends on the descriptor alone: its loads go out before everything
    // else, so that waiting for them waits for nothing else (loads return in order)
    constexpr uint32_t kPerT = (kWin * kEB + kTileThreadsE - 1) / kTileThreadsE;
    double tw0[kPerT];
    if (kE == 1) {
#pragma unroll
        for (uint32_t u = 0; u < kPerT; ++u) {
            const uint32_t i = tx + u * kTileThreadsE;
            const uint32_t ic = i < td.win_len * kEB ? i : 0u;
            tw0[u] = theta[((size_t)td.lo + ic / kEB) * kB + (ic % kEB)];
        }
    }
    float dict_v = 0.f; // (the weight table, an entry per thread: on its way with the window)
    if (kFused && tx < kDictE) dict_v = dict[tx];
    // One epoch (kBatch = 4, what ships): the second register set first holds alignments 8..15 of the wavefront's
    // first -- widest -- slice, and its second slice is loaded into the first set half way through that fold
    // (fold_slice_e).  Builds with several epochs keep both slices resident over all of them.
    // (tiles of <= 512 reads: one slice per wavefront, alignments 8..15 in the second set, nothing to hand over)
    constexpr bool kHandOver = kE == 1 && kPerWave == 2;
    constexpr bool kHiOnly = kE == 1 && kPerWave == 1;
    SliceRegsB<WT> R[kPerWave < 2 ? 2 : kPerWave];
    // (requested behind the records: loads return in order, and the records -- left to the caches, OEM_REC_NT -- are what
    // the theta gathers hang on; the slices stream from memory and are wanted two barriers later)
    auto load_slices = [&]() {
        load_slice_b<kNT, WT, kFused>(R[0], w + (size_t)woff[0] * 64, codes + (size_t)coff[0] * 64, lane, wid[0]);
        if (kHandOver || kHiOnly) {
            load_slice_b<kNT, WT, kFused>(R[1], w + ((size_t)woff[0] + kBCh) * 64, codes + ((size_t)coff[0] + kBCh / 2) * 64, lane,
                                          wid[0] > (uint32_t)kBCh ? wid[0] - kBCh : 0u);
        } else {
#pragma unroll
            for (uint32_t q = 1; q < kPerWave; ++q)
                load_slice_b<kNT, WT, kFused>(R[q], w + (size_t)woff[q] * 64, codes + (size_t)coff[q] * 64, lane, wid[q]);
        }
    };
    uint32_t rt[kRem], rrow[kRem], rslot[kRem];
    WT rw[kRem];
    uint32_t ri[kRem];
    const uint32_t tid_base = td.problem * problem_size;
    const uint32_t *sd_t = sd + td.sd_begin - td.b_min; // slot of record i = sd_t[bucket of its transcript] + i
    if (td.remote_cnt) { // wave-uniform
        // branch-free: a thread without a record re-reads the tile's last one and gives it no weight, so the loads
        // issue back to back (a load per branch is a round trip each)
        const uint32_t last = td.remote_cnt - 1;
#pragma unroll
        for (int k = 0; k < kRem; ++k) {
            const uint32_t i = tx + k * kTileThreadsE;
            const uint32_t o = td.remote_begin + (i < td.remote_cnt ? i : last);
            ld_remote_b<kPacked, kNT && OEM_REC_NT>(r_a, r_row, o, tid_base, rt[k], rrow[k]);
            if (kFused) ri[k] = ld_stream_b<kNT && OEM_REC_NT>(&r_wi[o]);
            else rw[k] = ld_stream_b<kNT && OEM_REC_NT>(&r_w[o]);
        }
        load_slices();
        // (kFused: the weight is read from the LDS copy of the table behind the first barrier -- index 0 = 0.0 for a
        // thread without a record; as a global gather it was one more vector-memory instruction per record up here)
#pragma unroll
        for (int k = 0; k < kRem; ++k)
            if (tx + k * kTileThreadsE >= td.remote_cnt) { rw[k] = (WT)0; ri[k] = 0u; }
    } else {
        load_slices();
#pragma unroll
        for (int k = 0; k < kRem; ++k) { rt[k] = td.b_min << kBucketShift; rw[k] = (WT)0; ri[k] = 0u; rrow[k] = 0; }
    }
#pragma unroll
    for (int k = 0; k < kRem; ++k) rslot[k] = 0;
    {   // branch-free and back to back (a lookup per branch is a dependent round trip each); a thread without a
        // record reads the tile's first table word (or the table's slack word when the tile has no records)
        uint32_t sdv[kRem];
#pragma unroll
        for (int k = 0; k < kRem; ++k)
            sdv[k] = sd_t[tx + k * kTileThreadsE < td.remote_cnt ? rt[k] >> kBucketShift : td.b_min];
#pragma unroll
        for (int k = 0; k < kRem; ++k) rslot[k] = sdv[k] + tx + k * kTileThreadsE;
    }
    // slot of this lane at step j of an epoch: (j + lane) mod kEB
    uint32_t rot8[kEB]; // byte offset of that slot inside a [c][b] window entry
#pragma unroll
    for (int j = 0; j < kEB; ++j) rot8[j] = ((j + lane) & (kEB - 1)) * 8u;
    uint32_t cs = 0; // copies of the count window = 1 << cs (wave-uniform), this lane's at byte cpy of an entry
    while (cs < kMaxCopyShiftE && td.win_len * kEB * (1u + (2u << cs)) <= kPool) ++cs;
    double *const theta_l = pool_l;
    double *const cnt_l = pool_l + td.win_len * kEB;
    const uint32_t cpy = ((lane >> 2) & ((1u << cs) - 1u)) * (kEB * 8u);

#pragma unroll 1
    for (int e = 0; e < kE; ++e) {
        const uint32_t act_e = (act >> (e * kEB)) & ((1u << kEB) - 1u);
        const uint32_t fin_e = (fin >> (e * kEB)) & ((1u << kEB) - 1u);
        if (!act_e) continue; // no slot of this epoch is running (wave-uniform)
        auto th = [&](double v, uint32_t b) -> double {
            return (((fin_e >> b) & 1u) && v < OEM_MIN_READ_THRESH) ? 0.0 : v;
        };
        const size_t eoff = (size_t)e * kEB; // first slot of the epoch

        // multiplicities of this wavefront's reads, one byte per slot
        uint32_t mult[kPerWave];
#pragma unroll
        for (uint32_t q = 0; q < kPerWave; ++q) {
            const uint32_t rl = slice_of(q) * 64 + lane;
            mult[q] = rl < td.n_rows ? ld_stream_b<OEM_E_MULT_NT != 0>(reinterpret_cast<const uint32_t *>(row_w + (size_t)(td.row_base + rl) * kB + eoff)) : 0u;
        }
        // multiplicities of the reads of this thread's remote records (4 KB per tile: cache-resident).  A read
        // that a slot's resample did not draw (1/e of them) takes no part in that slot's pass: its remote
        // denominators are never read (no LDS atomics for them) and its c / denom is 0, so its queue entries
        // are written as zeros with the 32-byte piece they share with the other slots.
        uint32_t rmult[kRem]; // (loaded branch-free and back to back; a thread without a record discards its word)
#pragma unroll
        for (int k = 0; k < kRem; ++k)
            rmult[k] = *reinterpret_cast<const uint32_t *>(row_w + (size_t)(td.row_base + rrow[k]) * kB + eoff);
#pragma unroll
        for (int k = 0; k < kRem; ++k)
            if (tx + k * kTileThreadsE >= td.remote_cnt) rmult[k] = 0u;
        // remote alignments: theta[t][b] of the epoch's four slots is one 32-byte piece.  The gathers go out here and
        // are not looked at before phase A: the theta window is written and the windows are cleared while they are
        // in flight.  (Measured and dropped, profiles/r04_notes.md: the first slice's local denominators summed in
        // that shadow too, and dead (read, slot) pairs reading a conflict-free word of their own: +1 % and +2 %.)
        double rx[kRem][kEB];
#pragma unroll
        for (int k = 0; k < kRem; ++k) {
            const double *tp = theta + (size_t)rt[k] * kB + eoff;
#pragma unroll
            for (int b = 0; b < kEB; ++b) rx[k][b] = OEM_EXP_E(16u) ? 1.0 : tp[b];
        }
        if (kFused && tx < kDictE) dict_l[tx] = dict_v;
        if (kE == 1) {
#pragma unroll
            for (uint32_t u = 0; u < kPerT; ++u) {
                const uint32_t i = tx + u * kTileThreadsE;
                if (i < td.win_len * kEB) theta_l[i] = th(tw0[u], i % kEB);
            }
        } else {
            for (uint32_t i = tx; i < td.win_len * kEB; i += kTileThreadsE)
                theta_l[i] = th(theta[((size_t)td.lo + i / kEB) * kB + eoff + (i % kEB)], i % kEB);
        }
        for (uint32_t i = tx; i < ((td.win_len * kEB) << cs); i += kTileThreadsE) cnt_l[i] = 0.0;
        for (uint32_t i = tx; i < td.n_slices * 64; i += kTileThreadsE) {
#pragma unroll
            for (int b = 0; b < kEB; ++b) den_l[b * kRows + i] = 0.0;
        }
        OEM_PROBE_E(2);
        __syncthreads();
        OEM_PROBE_E(3);

        // ---- remote phase A: denominators ------------------------------------------------
        if (kFused) {
#pragma unroll
            for (int k = 0; k < kRem; ++k) rw[k] = (WT)dict_l[ri[k]];
        }
#pragma unroll
        for (int k = 0; k < kRem; ++k)
            if (rmult[k]) {
                const double wv = (double)rw[k];
#pragma unroll
                for (int b = 0; b < kEB; ++b)
                    if (((rmult[k] >> (8 * b)) & 0xffu) && !OEM_EXP_E(2u)) lds_add(&den_l[b * kRows + rrow[k]], th(rx[k][b], b) * wv);
            }
        for (uint32_t i = tx + kRem * kTileThreadsE; i < td.remote_cnt; i += kTileThreadsE) { // beyond the register-resident records
            const uint32_t o = td.remote_begin + i;
            uint32_t t, row;
            ld_remote_b<kPacked, false>(r_a, r_row, o, tid_base, t, row);
            const double *tp = theta + (size_t)t * kB + eoff;
            const double wv = kFused ? (double)dict[r_wi[o]] : (double)r_w[o];
#pragma unroll
            for (int b = 0; b < kEB; ++b) lds_add(&den_l[b * kRows + row], th(tp[b], b) * wv);
        }
        OEM_PROBE_E(4);
        __syncthreads();
        OEM_PROBE_E(5);

        // ---- local alignments: one read per lane, slots in rotated order ---------------------
        if (kHiOnly) {
            if (wave < td.n_slices)
                fold_slice_e<WT, kNT, true, kRows, kFused>(R[0], R[1], wid[0], mult[0], wave * 64 + lane, lane, w + (size_t)woff[0] * 64,
                                                   codes + (size_t)coff[0] * 64, theta_l, cnt_l, den_l, rot8, act_e, false,
                                                   nullptr, nullptr, 0u, exp_mask, cs, cpy, dict_l);
            OEM_PROBE_E(6);
            OEM_PROBE_E(7);
        } else if (kHandOver) {
            const uint32_t s0 = slice_of(0), s1 = slice_of(kPerWave - 1);
            if (s0 < td.n_slices)
                fold_slice_e<WT, kNT, true, kRows, kFused>(R[0], R[1], wid[0], mult[0], s0 * 64 + lane, lane, w + (size_t)woff[0] * 64,
                                            codes + (size_t)coff[0] * 64, theta_l, cnt_l, den_l, rot8, act_e, true,
                                            w + (size_t)woff[kPerWave - 1] * 64, codes + (size_t)coff[kPerWave - 1] * 64, wid[kPerWave - 1], exp_mask, cs, cpy, dict_l);
            else
                load_slice_b<kNT, WT, kFused>(R[0], w + (size_t)woff[kPerWave - 1] * 64, codes + (size_t)coff[kPerWave - 1] * 64, lane, wid[kPerWave - 1]);
            OEM_PROBE_E(6);
            if (s1 < td.n_slices)
                fold_slice_e<WT, kNT, false, kRows, kFused>(R[0], R[0], wid[kPerWave - 1], mult[kPerWave - 1], s1 * 64 + lane, lane, w + (size_t)woff[kPerWave - 1] * 64,
                                             codes + (size_t)coff[kPerWave - 1] * 64, theta_l, cnt_l, den_l, rot8, act_e, false,
                                             nullptr, nullptr, 0u, exp_mask, cs, cpy, dict_l);
            OEM_PROBE_E(7);
        } else {
#pragma unroll 1
            for (uint32_t q = 0; q < kPerWave; ++q) {
                const uint32_t s = slice_of(q);
                if (s >= td.n_slices) continue;
                SliceRegsB<WT> cur;
                uint32_t width = wid[0], mq = mult[0], wo = woff[0], co = coff[0];
#pragma unroll
                for (int k = 0; k < kBCh; ++k) cur.w[k] = R[0].w[k];
#pragma unroll
                for (int k = 0; k < kBCh / 2; ++k) cur.c[k] = R[0].c[k];
#pragma unroll
                for (uint32_t qq = 1; qq < kPerWave; ++qq)
                    if (q == qq) { // wave-uniform
                        width = wid[qq]; mq = mult[qq]; wo = woff[qq]; co = coff[qq];
#pragma unroll
                        for (int k = 0; k < kBCh; ++k) cur.w[k] = R[qq].w[k];
#pragma unroll
                        for (int k = 0; k < kBCh / 2; ++k) cur.c[k] = R[qq].c[k];
                    }
                fold_slice_e<WT, kNT, false, kRows, kFused>(cur, cur, width, mq, s * 64 + lane, lane, w + (size_t)wo * 64,
                                             codes + (size_t)co * 64, theta_l, cnt_l, den_l, rot8, act_e, false, nullptr,
                                             nullptr, 0u, exp_mask, cs, cpy, dict_l);
            }
        }
        __syncthreads();
        OEM_PROBE_E(8);

        // ---- remote phase B: queue[record][.] <- w * (c_ib / denom_ib), the epoch's slots as one 32-byte piece ----
#pragma unroll
        for (int k = 0; k < kRem; ++k) {
            if (tx + k * kTileThreadsE < td.remote_cnt && !OEM_EXP_E(1u)) {
                double *qp = queue + (size_t)rslot[k] * kB + eoff;
                const double wv = (double)rw[k];
                double qv[kEB];
#pragma unroll
                for (int b = 0; b < kEB; ++b) qv[b] = wv * den_l[b * kRows + rrow[k]];
                store_queue_piece(qp, qv);
            }
        }
        for (uint32_t i = tx + kRem * kTileThreadsE; i < td.remote_cnt; i += kTileThreadsE) {
            const uint32_t o = td.remote_begin + i;
            uint32_t t, row;
            ld_remote_b<kPacked, false>(r_a, r_row, o, tid_base, t, row);
            double *qp = queue + (size_t)(sd_t[t >> kBucketShift] + i) * kB + eoff;
            const double wv = kFused ? (double)dict[r_wi[o]] : (double)r_w[o];
#pragma unroll
            for (int b = 0; b < kEB; ++b) __builtin_nontemporal_store(wv * den_l[b * kRows + row], &qp[b]);
        }
        // ---- flush the epoch's window: [c][b] -> cnt[lo + c][eoff + b], theta multiplied in here --------
        for (uint32_t i = tx; i < td.win_len * kEB; i += kTileThreadsE) {
            double sum = 0.0;
            for (uint32_t p = 0; p < (1u << cs); ++p) sum += cnt_l[((((i / kEB) << cs) + p) * kEB) + (i % kEB)];
            const double v = sum * theta_l[i];
            if (v != 0.0) unsafeAtomicAdd(&cnt[((size_t)td.lo + i / kEB) * kB + eoff + (i % kEB)], v);
        }
        OEM_PROBE_E(9);
        if (e + 1 < kE) __syncthreads(); // the next epoch re-initialises the windows
    }
}

// One workgroup per (bucket, group[, four slots of kB]): streams its range of queue entries -- the 32-byte piece of
// its four slots -- into an LDS window of kBucket transcripts x 4 slots (128 KiB: one workgroup per CU), then
// flushes theta * sum into cnt[t][slot] (theta is multiplied in here, once per bucket entry: see "factored
// increments" at the top).  The destinations q_dst are read once for the four slots.  A thread takes 16 bytes
// (two slots) of an entry, consecutive lanes consecutive pieces, eight loads in flight: with one 16-wavefront
// workgroup per CU it takes 128 KiB in flight per CU to keep the memory system busy (4 x 8 bytes per thread, the
// first version, ran the kernel at its latency, not at its bytes).
constexpr int kFS = 4;       // slots per fold workgroup
constexpr int kFoldDepth = 8; // loads in flight per thread
__global__ __launch_bounds__(kFoldThreadsB) void k_remote_fold_b(
    const uint32_t *__restrict__ bucket_base, const double *__restrict__ queue /* [n_remote][kB] */,
    const uint16_t *__restrict__ q_dst, const double *__restrict__ theta /* [T][kB] */, double *__restrict__ cnt /* [T][kB] */,
    const BatchState *__restrict__ st, uint32_t n_groups, uint32_t n_txps)
{
    const uint32_t s_first = blockIdx.y * kFS; // first slot of this workgroup
    uint32_t act = 0, fin = 0;
#pragma unroll
    for (int b = 0; b < kFS; ++b) {
        const uint32_t ph = st[s_first + b].phase;
        act |= (ph != kPhaseFinished) ? (1u << b) : 0u;
        fin |= (ph == kPhaseFinal) ? (1u << b) : 0u;
    }
    if (!act) return;
    const uint32_t bk = blockIdx.x / n_groups, g = blockIdx.x % n_groups;
    extern __shared__ double acc[]; // [kBucket][kFS]
    const uint32_t q0 = bucket_base[bk], q1 = bucket_base[bk + 1];
    const uint64_t span = q1 - q0;
    const uint32_t s0 = q0 + (uint32_t)(span * g / n_groups);
    const uint32_t s1 = q0 + (uint32_t)(span * (g + 1) / n_groups);
    if (s0 == s1) return;
    for (uint32_t i = threadIdx.x; i < kBucket * kFS; i += kFoldThreadsB) acc[i] = 0.0;
    __syncthreads();
    const uint32_t half = threadIdx.x & 1u;               // which two of the four slots
    constexpr uint32_t kEntriesPerStep = kFoldThreadsB / 2; // entries one load of the workgroup covers
    struct alignas(16) D2 { double x, y; };
    const D2 *q2 = reinterpret_cast<const D2 *>(queue);
    auto piece = [&](uint32_t o) -> const D2 * { return q2 + ((size_t)o * kB + s_first) / 2 + half; };
    // Two register sets: the loads of the next step are in flight while this step's values go into the LDS window
    // (one 16-wavefront workgroup per CU: nothing else on the CU hides a wave's round trip).  Entries past the
    // range are clamped to its last one and skipped at the point of use.
    constexpr uint32_t kStep = kFoldDepth * kEntriesPerStep;
    auto load = [&](D2 (&v)[kFoldDepth], uint32_t (&d)[kFoldDepth], uint32_t o) {
#pragma unroll
        for (int k = 0; k < kFoldDepth; ++k) {
            const uint32_t oo = o + k * kEntriesPerStep, oc = oo < s1 ? oo : s1 - 1;
            const D2 *pp = piece(oc); // (read once: non-temporal, the window's theta and counts stay in the L2)
            v[k].x = __builtin_nontemporal_load(&pp->x);
            v[k].y = __builtin_nontemporal_load(&pp->y);
            d[k] = __builtin_nontemporal_load(&q_dst[oc]);
        }
    };
    auto consume = [&](const D2 (&v)[kFoldDepth], const uint32_t (&d)[kFoldDepth], uint32_t o) {
        // hot destinations: runs of equal ones are summed on the vector ALU first (oem_lane_runs.h; lanes 2 i and 2 i + 1
        // hold the two halves of entry i, so a run's lanes are two apart)
        const bool rep = keys_repeat<2>(d[0]);
#pragma unroll
        for (int k = 0; k < kFoldDepth; ++k) {
            const bool in = o + k * kEntriesPerStep < s1;
            double xy[2] = {in ? v[k].x : 0.0, in ? v[k].y : 0.0};
            if (rep) sum_runs_of_equal_keys<2, 2>(d[k], xy);
            double *a = &acc[d[k] * kFS + 2 * half];
            if (xy[0] != 0.0) lds_add(a, xy[0]);
            if (xy[1] != 0.0) lds_add(a + 1, xy[1]);
        }
    };
    D2 va[kFoldDepth], vb[kFoldDepth];
    uint32_t da[kFoldDepth], db[kFoldDepth];
    const uint32_t lane_off = threadIdx.x >> 1;
    uint32_t qb = s0; // (uniform: every thread of the workgroup makes the same trips)
    load(va, da, qb + lane_off);
    for (;;) {
        const uint32_t b1 = qb + kStep;
        const bool more1 = b1 < s1;
        if (more1) load(vb, db, b1 + lane_off);
        consume(va, da, qb + lane_off);
        if (!more1) break;
        const uint32_t b2 = b1 + kStep;
        const bool more2 = b2 < s1;
        if (more2) load(va, da, b2 + lane_off);
        consume(vb, db, b1 + lane_off);
        if (!more2) break;
        qb = b2;
    }
    __syncthreads();
    // flush: thread i takes window elements i, i + 1024, ...: (transcript, slot) pairs in [t][slot] order -- for
    // kB == kFS consecutive lanes on consecutive doubles of theta and cnt.  Four elements per round trip.
    const uint32_t base = bk * kBucket;
    const uint32_t slot = threadIdx.x % kFS; // (kFoldThreadsB is a multiple of kFS)
    static_assert(kFoldThreadsB % kFS == 0, "slot of a thread = thread index mod kFS");
    if (!((act >> slot) & 1u)) return;       // (a finished slot's entries are zeros: nothing of it to flush)
    const bool final_slot = (fin >> slot) & 1u;
    for (uint32_t i0 = threadIdx.x; i0 < kBucket * kFS; i0 += 4 * kFoldThreadsB) {
        double v[4], th[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = i0 + k * kFoldThreadsB, t = base + i / kFS;
            v[k] = (i < kBucket * kFS && t < n_txps) ? acc[i] : 0.0;
            th[k] = v[k] != 0.0 ? theta[(size_t)t * kB + s_first + slot] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = i0 + k * kFoldThreadsB, t = base + i / kFS;
            if (final_slot && th[k] < OEM_MIN_READ_THRESH) th[k] = 0.0; // em.rs:238-242, as k_em_tile_e reads it
            const double x = v[k] * th[k];
            if (x != 0.0) unsafeAtomicAdd(&cnt[(size_t)t * kB + s_first + slot], x);
        }
    }
}

// rel-diff / swap / clear / state machine for kB slots (em.rs:194-218, :238-254); curr_b[t] = cnt[t][b]
constexpr int kRelB = 1024; // as k_reldiff_swap_clear: few fat workgroups, the state-line atomics serialise
__global__ __launch_bounds__(kRelB) void k_reldiff_b(double *__restrict__ theta, double *__restrict__ cnt,
                                                   double *__restrict__ out, BatchState *st, EmParams p,
                                                   unsigned long long *rel_slots)
{
    uint32_t running = 0, final_ = 0; // SGPR masks
#pragma unroll
    for (int b = 0; b < kB; ++b) {
        const uint32_t ph = st[b].phase;
        running |= (ph == kPhaseRunning) ? (1u << b) : 0u;
        final_ |= (ph == kPhaseFinal) ? (1u << b) : 0u;
    }
    if (!(running | final_)) return;
    // One (transcript, slot) element per thread per step, consecutive threads on consecutive elements of
    // the [T][kB] arrays (fully coalesced).  The stride
    // is a multiple of kB, so a thread stays on ONE slot, b = thread index mod kB, and the maxima of the kB
    // slots are the kB residue classes of the lanes.
    static_assert((kRelB % kB) == 0 && (64 % kB) == 0, "slot of a lane = lane mod kB");
    const uint32_t b = threadIdx.x % kB;
    const bool is_final = (final_ >> b) & 1u, is_live = ((running | final_) >> b) & 1u;
    double rel = 0.0;
    const size_t n_elem = (size_t)p.n_txps * kB;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if (is_live) { // (the loads of four elements go out before the first is looked at: the sweep is latency, not bytes)
        for (size_t j0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j0 < n_elem; j0 += 4 * stride) {
            double cc[4], pc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t j = j0 + k * stride, jc = j < n_elem ? j : j0;
                cc[k] = cnt[jc];
                pc[k] = is_final ? 0.0 : theta[jc];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t j = j0 + k * stride;
                if (j >= n_elem) break;
                cnt[j] = 0.0;
                if (is_final) {
                    out[(size_t)b * p.n_txps + j / kB] = cc[k];                           // em.rs:254
                } else {
                    if (pc[k] > OEM_MIN_READ_THRESH) rel = fmax(rel, (cc[k] - pc[k]) / pc[k]); // em.rs:195-199
                    theta[j] = cc[k];                           // em.rs:204 (zeroing of small values: k_em_tile_e reads them as 0)
                }
            }
        }
    }
    __shared__ double smax[kRelB / 64][kB];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int off = 32; off >= kB; off >>= 1) rel = fmax(rel, __shfl_xor(rel, off, 64)); // within the lanes of one slot
    if (lane < kB) smax[wv][lane] = rel;
    __syncthreads();
    __shared__ bool is_last;
    if (threadIdx.x == 0) {
        // the workgroup's maxima go to ITS row of the slot table (kBatchRelSlots rows of kB words): ~200 workgroups x kB
        // atomics on the four state words were ~1000 read-modify-writes on one line, performed one after another
        unsigned long long *row = rel_slots + (size_t)(blockIdx.x & (kBatchRelSlots - 1u)) * kB;
#pragma unroll
        for (int b = 0; b < kB; ++b) {
            double m = smax[0][b];
            for (int i = 1; i < kRelB / 64; ++i) m = fmax(m, smax[i][b]);
            if (m > 0.0) atomicMax(&row[b], (unsigned long long)__double_as_longlong(m));
        }
        // (ordering of the maxima against the ticket: see k_reldiff_swap_clear, oem_kernels.hip)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t ticket = atomicAdd(&st[0].blocks_arrived, 1u);
        is_last = (ticket == gridDim.x - 1);
    }
    __syncthreads();
    __shared__ unsigned long long slot_max[kB];
    if (is_last) { // (workgroup-uniform) the table's maximum per slot, the table zeroed for the next pass
        static_assert(kBatchRelSlots * kB <= kRelB && kB <= 64, "one table word per thread");
        unsigned long long bits = 0ull;
        if (threadIdx.x < kBatchRelSlots * kB) {
            bits = __hip_atomic_load(&rel_slots[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            rel_slots[threadIdx.x] = 0ull;
        }
        if (threadIdx.x < kB) slot_max[threadIdx.x] = 0ull;
        __syncthreads();
        if (bits) atomicMax(&slot_max[threadIdx.x % kB], bits); // (LDS; word w of the table belongs to slot w mod kB)
        __syncthreads();
    }
    if (is_last && threadIdx.x == 0) {
#pragma unroll
        for (int b = 0; b < kB; ++b) {
            if ((final_ >> b) & 1u) {
                st[b].n_passes += 1;
                st[b].phase = kPhaseFinished;
                continue;
            }
            if (!((running >> b) & 1u)) continue;
            const double rel_diff = __longlong_as_double((long long)slot_max[b]);
            st[b].last_rel = rel_diff;
            st[b].n_passes += 1;
            uint32_t niter = st[b].niter;
            if (rel_diff < p.conv_thresh && niter > p.min_iter_gate) { // em.rs:212
                st[b].converged = 1;
                st[b].phase = kPhaseFinal;
            } else {
                niter += 1;                                            // em.rs:218
                st[b].niter = niter;
                if (niter >= p.max_iter) st[b].phase = kPhaseFinal;    // em.rs:181
            }
        }
        st[0].blocks_arrived = 0u;
    }
}

// (re)start of ONE slot of the rolling batch: theta[t][slot] = init / avg, its counts cleared
__global__ __launch_bounds__(256) void k_reset_slot_b(double *__restrict__ theta, double *__restrict__ cnt,
                                                      const double *__restrict__ init, double avg, uint32_t n_txps,
                                                      uint32_t slot)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_txps; i += gridDim.x * blockDim.x) {
        theta[(size_t)i * kB + slot] = init ? init[i] : avg;
        cnt[(size_t)i * kB + slot] = 0.0;
    }
}

// multiplicities of ONE slot, caller order u32 [R] -> its byte column of the tile-order table
// u8 [rows][kB]; *overflow is set if a multiplicity does not fit a byte (the caller then runs that
// replicate on the one-replicate-per-pass path)
__global__ __launch_bounds__(256) void k_pack_row_w_b(const uint32_t *__restrict__ row_w,
                                                      const uint32_t *__restrict__ perm, uint64_t n_rows,
                                                      uint8_t *__restrict__ out, uint32_t slot, uint32_t *overflow)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t c = row_w[perm[i]];
        if (c > 255u) *overflow = 1u;
        out[i * kB + slot] = (uint8_t)(c & 0xffu);
    }
}

inline int grid_for(uint64_t n, int block, int max_blocks)
{
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (uint64_t)max_blocks) g = max_blocks;
    return (int)g;
}

} // namespace

int launch_batch_pass(oem_store *s, const BatchBuffers &bb)
{
    const DeviceTiled &t = s->tiled;
    if (t.n_tiles == 0) return OEM_OK;
#ifdef OEM_TESTING
    {
        static unsigned int current = 0;
        const unsigned int want = (unsigned int)knob("OEM_TILE_EXP", 0);
        if (want != current) {
            OEM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tile_e_exp), &want, sizeof(want)));
            current = want;
        }
    }
#endif
    const bool f64w = s->csr.w_is_f64;
    const uint64_t wsz = f64w ? 8 : 4;
    const uint64_t stream_bytes = (t.n_local + t.n_local / 8) * (wsz + 2) + t.n_remote * (wsz + (t.packed ? 4 : 6));
    const bool nt = stream_bytes > (192ull << 20); // beyond the Infinity Cache: stream non-temporally
    const bool fused = !f64w && t.dict_fused && t.dict && t.r_wi && t.dict_n <= kDictE; // (no weight stream: see kFused)
#define OEM_LAUNCH_TILE_E4(NT, WT, W, RW, PK, ROWS, FUSED)                                                            \
    hipLaunchKernelGGL((k_em_tile_e<NT, WT, PK, ROWS, FUSED>), dim3(t.n_tiles), dim3(kTileThreadsE), 0, bb.stream,      \
                       t.tiles, t.codes, (const WT *)W, PK ? t.r_pk : t.r_tid, (const WT *)RW, t.r_row, t.sd,            \
                       t.problem_size, bb.queue, bb.theta, bb.cnt, bb.state, bb.row_w, t.dict, t.r_wi)
#define OEM_LAUNCH_TILE_E3(NT, WT, W, RW, PK, ROWS)                                                                   \
    do {                                                                                                              \
        if (sizeof(WT) == 4 && fused) OEM_LAUNCH_TILE_E4(NT, WT, W, RW, PK, ROWS, (sizeof(WT) == 4));                  \
        else OEM_LAUNCH_TILE_E4(NT, WT, W, RW, PK, ROWS, false);                                                      \
    } while (0)
#define OEM_LAUNCH_TILE_E2(NT, WT, W, RW, PK) OEM_LAUNCH_TILE_E3(NT, WT, W, RW, PK, kTileRows)
#define OEM_LAUNCH_TILE_E(NT, WT, W, RW)                                                                              \
    do {                                                                                                              \
        if (t.packed) OEM_LAUNCH_TILE_E2(NT, WT, W, RW, true);                                                        \
        else OEM_LAUNCH_TILE_E2(NT, WT, W, RW, false);                                                                \
    } while (0)
    if (f64w) {
        if (nt) OEM_LAUNCH_TILE_E(true, double, t.w64, t.r_w64);
        else OEM_LAUNCH_TILE_E(false, double, t.w64, t.r_w64);
    } else {
        if (nt) OEM_LAUNCH_TILE_E(true, float, t.w32, t.r_w32);
        else OEM_LAUNCH_TILE_E(false, float, t.w32, t.r_w32);
    }
#undef OEM_LAUNCH_TILE_E4
#undef OEM_LAUNCH_TILE_E3
#undef OEM_LAUNCH_TILE_E2
#undef OEM_LAUNCH_TILE_E
    OEM_HIP(hipGetLastError());
    if (t.n_remote > 0) {
        // one 128 KiB workgroup per CU; every group clears and flushes a whole window, which pays from ~32 Ki entries
        uint32_t n_groups = 256 / (t.n_buckets ? t.n_buckets : 1);
        const uint64_t per_bucket = t.n_remote / (t.n_buckets ? t.n_buckets : 1) + 1;
        const uint32_t max_useful = (uint32_t)((per_bucket + 32767) / 32768);
        if (n_groups > max_useful) n_groups = max_useful;
        if (n_groups < 1) n_groups = 1;
        constexpr size_t kFoldLds = sizeof(double) * kBucket * kFS;
        static_assert(kB % kFS == 0, "the fold takes the slots four at a time");
        static_assert(kFoldLds <= 160u * 1024u, "the fold window of all slots must fit the LDS of a CU");
        // (a function attribute belongs to the device that is current: once per device of this process)
        static std::atomic<unsigned long long> attr_set{0ull};
        const unsigned long long bit = 1ull << (s->device & 63);
        if (!(attr_set.load(std::memory_order_acquire) & bit)) {
            OEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_remote_fold_b),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFoldLds));
            attr_set.fetch_or(bit, std::memory_order_release);
        }
        hipLaunchKernelGGL(k_remote_fold_b, dim3(t.n_buckets * n_groups, kB / kFS), dim3(kFoldThreadsB), kFoldLds, bb.stream,
                           t.bucket_base, bb.queue, t.q_dst, bb.theta, bb.cnt, bb.state, n_groups, s->csr.n_txps);
        OEM_HIP(hipGetLastError());
    }
    return OEM_OK;
}

int launch_batch_reldiff(oem_store *s, const BatchBuffers &bb, EmParams p)
{
    // the sweep moves kB times the bytes of k_reldiff_swap_clear: 256 workgroups (97 -> ~30 us at 200 k transcripts)
    const int grid = grid_for(p.n_txps, kRelB, 256);
    hipLaunchKernelGGL(k_reldiff_b, dim3(grid), dim3(kRelB), 0, bb.stream, bb.theta, bb.cnt, bb.out, bb.state, p, bb.rel_slots);
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

int launch_batch_reset_slot(oem_store *s, const BatchBuffers &bb, const double *d_init, double avg, uint32_t slot)
{
    // (the queue needs no reset: every pass writes every entry of every slot, zeros for the reads a resample did not draw)
    const int grid = grid_for(s->csr.n_txps, 256, 256);
    hipLaunchKernelGGL(k_reset_slot_b, dim3(grid), dim3(256), 0, bb.stream, bb.theta, bb.cnt, d_init, avg,
                       s->csr.n_txps, slot);
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

int launch_batch_pack_row_w(oem_store *s, const uint32_t *d_row_w, const BatchBuffers &bb, uint32_t slot,
                            uint32_t *d_overflow)
{
    const uint64_t n = s->tiled.n_rows;
    if (n == 0) return OEM_OK;
    hipLaunchKernelGGL(k_pack_row_w_b, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, bb.stream, d_row_w,
                       s->tiled.perm, n, bb.row_w, slot, d_overflow);
    OEM_HIP(hipGetLastError());
    return OEM_OK;
}

} // namespace oem

#ifdef OEM_TESTING
// Test hooks: route k_em_tile_e's phase stamps into a buffer (begin), fetch and release it (end).
static unsigned long long *g_probe_e_buf = nullptr;
static size_t g_probe_e_n = 0;
extern "C" int oem_debug_tile_e_probe_begin(uint64_t n_tiles)
{
    using namespace oem;
    OEM_API_BEGIN
    if (g_probe_e_buf || n_tiles == 0) return fail(OEM_ERR_STATE, "oem_debug_tile_e_probe_begin: busy / empty");
    g_probe_e_n = (size_t)n_tiles * 16;
    OEM_HIP(hipMalloc((void **)&g_probe_e_buf, g_probe_e_n * sizeof(unsigned long long)));
    OEM_HIP(hipMemset(g_probe_e_buf, 0, g_probe_e_n * sizeof(unsigned long long)));
    OEM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tile_e_probe), &g_probe_e_buf, sizeof(g_probe_e_buf)));
    return OEM_OK;
    OEM_API_END("oem_debug_tile_e_probe_begin")
}
extern "C" int oem_debug_tile_e_probe_end(unsigned long long *out, uint64_t n_out)
{
    using namespace oem;
    OEM_API_BEGIN
    if (!g_probe_e_buf || !out || n_out < g_probe_e_n) return fail(OEM_ERR_ARG, "oem_debug_tile_e_probe_end: bad argument");
    OEM_HIP(hipDeviceSynchronize());
    unsigned long long *null = nullptr;
    OEM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tile_e_probe), &null, sizeof(null)));
    OEM_HIP(hipMemcpy(out, g_probe_e_buf, g_probe_e_n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    hipFree(g_probe_e_buf);
    g_probe_e_buf = nullptr;
    return OEM_OK;
    OEM_API_END("oem_debug_tile_e_probe_end")
}
#endif
