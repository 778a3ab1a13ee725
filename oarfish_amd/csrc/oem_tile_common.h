// oem_tile_common.h -- device helpers of the tile kernel (oem_tile_kernels.hip: one workgroup per tile; round 5's
// pipelined tile walk shared them -- measured slower, removed in round 6: HISTORY.md).  SELL-64 slice registers, the
// weight codings of oem_layout_dict.hip, the per-slice fold (em.rs:97-131: denominator, then increments).
#pragma once

#include "oem_internal.h"
#include "oem_lane_runs.h"

#ifndef OEM_EXP
#define OEM_EXP(bit) false // cost-attribution switches: only the test-only build of oem_tile_kernels.hip defines them
#endif

namespace oem {

namespace {

__device__ __forceinline__ void lds_add_f64(double *p, double v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // ds_add_f64
}

// Sum over the 64 lanes of a wavefront, the same value in every lane.  On the data-parallel-primitive path of the
// vector ALU (four butterfly steps inside each row of 16 lanes, then the four row totals read as scalars): the
// __shfl_xor form is twelve ds_bpermute -- six dependent round trips through the LDS crossbar, ~0.3 us in the middle
// of a fold, in every slice whose reads share their anchor (the tiles of the highly expressed transcripts).
#ifndef OEM_WAVE_SUM_SHFL
template <int kCtrl>
__device__ __forceinline__ double dpp_f64(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, kCtrl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), kCtrl, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo));
}
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
    v += dpp_f64<0xB1>(v);  // quad_perm:[1,0,3,2]
    v += dpp_f64<0x4E>(v);  // quad_perm:[2,3,0,1]
    v += dpp_f64<0x141>(v); // row_half_mirror
    v += dpp_f64<0x140>(v); // row_mirror: every lane holds its row's total
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
#else
__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
#endif

// The matrix streams (weights, codes, remote records) are touched once per pass; theta and the
// tile descriptors are re-read all the time.  When the store is larger than the 256 MiB Infinity
// Cache, non-temporal loads keep the once-only traffic from evicting theta from the 4 MiB L2 of
// each XCD (its gathers are the L2-request-bound part of the kernel): C3 0.244 -> 0.226 ms.  A
// store that fits the Infinity Cache (C2, or one shard of an 8-GPU run) is faster with ordinary
// loads (0.0348 vs 0.0387 ms), so the policy is a template flag chosen per store.
template <bool kNT, typename T>
__device__ __forceinline__ T ld_stream(const T *p)
{
    return kNT ? __builtin_nontemporal_load(p) : *p;
}

// LDS window entries are addressed by byte offset (the 16-bit codes are stored
// pre-multiplied by 8), which saves the shift per alignment.
__device__ __forceinline__ double lds_ld(const double *base, uint32_t byte_off)
{
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ double *lds_at(double *base, uint32_t byte_off)
{
    return reinterpret_cast<double *>(reinterpret_cast<char *>(base) + byte_off);
}

// Registers of one SELL-64 slice for one lane: up to kCh local alignments
// (longer reads spill to a reload loop); kRem = remote alignments per thread whose
// theta*w stays in registers between the two remote phases.
template <typename WT, int kCh>
struct SliceRegs {
    WT w[kCh];            // the weights themselves ...
    uint32_t wi[kCh / 2]; // ... or (dictionary-coded stores, oem_layout_dict.hip) table indices: four one-byte ones per
                          // word (kWBytes: kCh / 4 words) or two 16-bit ones per word (kWWords)
    uint32_t c[kCh / 2];
};
// Weight coding of a store (oem_layout_dict.hip): 0 the f32 / f64 stream; 1 one-byte table indices in their own
// stream (129..256 distinct weights); 2 FUSED: a 7-bit index in the spare bits of the pair of 16-bit window codes an
// alignment shares a word with (up to 128 distinct weights: a code is 8 * (transcript - lo) < 4096, so bits 0..2 and
// 12..15 of either half are free; which bits: code_widx below) --
// no weight stream at all, a local alignment is its two code bytes; 3 WORDS: 16-bit indices, two per u32, stored in
// the geometry of the window codes (257..1024 distinct weights -- long reads with score gaps in the hundreds).
constexpr int kWPlain = 0, kWBytes = 1, kWFused = 2, kWWords = 3;
template <int kDict> constexpr int dict_entries() { return kDict == kWWords ? 1024 : kDict != kWPlain ? 256 : 1; }

// Register sets and launch bound (waves per SIMD the compiler must leave room for) of k_em_tile by weight coding:
// how many of a wavefront's slices sit in registers before the fold starts (see kSets in the kernel).  Measured
// (profiles/r04_notes.md; C3 pass, two sets at five workgroups per CU before): fused / byte codes, 4 / 6 registers a
// set: all five sets, still five workgroups per CU: 0.168 -> 0.150 ms; the f32 stream, 12 registers a set: five sets at
// FOUR workgroups per CU: 0.196 -> 0.177 (four sets 0.181, three 0.186); 16-bit indices, 8 a set: five sets at four
// workgroups 0.176 -> 0.166 (three sets at five: 0.170); f64 weights (coverage), 20 a set and four workgroups per CU
// either way: three sets 0.248 -> 0.239; the wide-window kernel of the per-cell batches: no difference (two).
// The defaults can be overridden per build for A/B (scripts/build_variant.sh).
#ifndef OEM_SETS_FUSED
#define OEM_SETS_FUSED 5
#endif
#ifndef OEM_SETS_BYTES
#define OEM_SETS_BYTES 5
#endif
#ifndef OEM_SETS_WORDS
#define OEM_SETS_WORDS 5
#endif
#ifndef OEM_SETS_F32
#define OEM_SETS_F32 5
#endif
#ifndef OEM_SETS_F64
#define OEM_SETS_F64 3
#endif
#ifndef OEM_SETS_WIDE
#define OEM_SETS_WIDE 2
#endif
#ifndef OEM_WAVES_WORDS
#define OEM_WAVES_WORDS 4
#endif
#ifndef OEM_WAVES_WIDE
#define OEM_WAVES_WIDE 2 // launch-bounds waves per SIMD of the wide-window kernel with f32 / coded weights
#endif
#ifndef OEM_WAVES_CODED
#define OEM_WAVES_CODED 5
#endif
#ifndef OEM_COPIES
#define OEM_COPIES 4
#endif
#ifndef OEM_CNT_ENTRIES
#define OEM_CNT_ENTRIES 0 // entries of the narrow-window count pool (0: kWin * OEM_COPIES)
#endif
#ifndef OEM_MAX_COPY_SHIFT
#define OEM_MAX_COPY_SHIFT 3
#endif
constexpr uint32_t kMaxCopyShift = OEM_MAX_COPY_SHIFT;
template <typename WT, int kDict> constexpr int tile_sets()
{
    return sizeof(WT) == 8 ? OEM_SETS_F64 : kDict == kWFused ? OEM_SETS_FUSED : kDict == kWBytes ? OEM_SETS_BYTES
                                          : kDict == kWWords ? OEM_SETS_WORDS : OEM_SETS_F32;
}
template <typename WT, int kDict> constexpr int tile_min_waves()
{
    return sizeof(WT) == 8 ? 2 : (kDict == kWFused || kDict == kWBytes) ? OEM_WAVES_CODED : kDict == kWWords ? OEM_WAVES_WORDS
                                                                           : (OEM_SETS_F32 > 2 ? 4 : 2);
}
__device__ __forceinline__ uint32_t code_half(uint32_t c, int h) { return h ? c >> 16 : c & 0xffffu; }
template <int kDict>
__device__ __forceinline__ uint32_t code_off(uint32_t half) { return kDict == kWFused ? half & 0x0ff8u : half; } // LDS byte offset
// Table index of half h of a pair of FUSED codes (oem_layout_dict.hip): its low four bits sit in bits 12..15 of the
// alignment's own half, its high three in bits 0..2 of the OTHER half of the word -- one rotation of the word puts both
// where the table offset wants them (v_alignbit_b32 + v_and_b32, against four logic operations for two fields of the
// same half).
__device__ __forceinline__ uint32_t code_widx(uint32_t c, int h)
{
    const uint32_t r = h ? ((c >> 26) | (c << 6)) : ((c >> 10) | (c << 22));
    return (r >> 2) & 0x7fu;
}
// weight of entry k of a register set: coded stores read it from the table in LDS (index 0 = 0.0: padded entries
// and entries beyond the slice's width need no masking)
template <int kDict, typename WT, int kCh>
__device__ __forceinline__ WT slice_w(const SliceRegs<WT, kCh> &r, int k, const float *dict_l)
{
    if (kDict == kWBytes) return (WT)dict_l[(r.wi[k >> 2] >> (8 * (k & 3))) & 0xffu];
    if (kDict == kWWords) return (WT)dict_l[code_half(r.wi[k >> 1], k & 1)];
    if (kDict == kWFused) return (WT)dict_l[code_widx(r.c[k >> 1], k & 1)];
    return r.w[k];
}

// Issue every load of a slice before any use.  `wbase`/`cbase`/`width` are
// wave-uniform (SGPRs), so the loads take the scalar-base + lane-offset form with
// immediate offsets, and the width tests are scalar branches: no per-alignment
// address arithmetic.  Pairs are loaded together; the second element of the last
// pair of an odd-width slice is the next slice's first alignment (the arrays are
// padded by one row) and is zeroed.
template <typename WT, int kCh, bool kNT = false, int kDict = kWPlain>
__device__ __forceinline__ void load_slice(SliceRegs<WT, kCh> &r, const WT *__restrict__ wbase,
                                           const uint32_t *__restrict__ cbase, uint32_t lane,
                                           uint32_t width, const uint32_t *__restrict__ ibase = nullptr)
{
#pragma unroll
    for (int g = 0; g < kCh / 2; ++g) {
        if ((uint32_t)(2 * g) < width) {
            if (kDict == kWBytes) {
                if ((g & 1) == 0) r.wi[g >> 1] = ld_stream<kNT>(&ibase[(g >> 1) * 64 + lane]);
            } else if (kDict == kWWords) {
                r.wi[g] = ld_stream<kNT>(&ibase[g * 64 + lane]);
            } else if (kDict == kWPlain) {
                r.w[2 * g] = ld_stream<kNT>(&wbase[(2 * g) * 64 + lane]);
                r.w[2 * g + 1] = ld_stream<kNT>(&wbase[(2 * g + 1) * 64 + lane]);
            }
            r.c[g] = ld_stream<kNT>(&cbase[g * 64 + lane]);
        } else {
            if (kDict == kWBytes) {
                if ((g & 1) == 0) r.wi[g >> 1] = 0u;
            } else if (kDict == kWWords) {
                r.wi[g] = 0u;
            } else if (kDict == kWPlain) {
                r.w[2 * g] = (WT)0;
                r.w[2 * g + 1] = (WT)0;
            }
            r.c[g] = 0u;
        }
    }
}

template <typename WT, int kCh, int kCopies, int kDict>
__device__ __forceinline__ void fold_slice(const SliceRegs<WT, kCh> &cur, uint32_t width, uint32_t s,
                                           uint32_t lane, const WT *__restrict__ wbase,
                                           const uint32_t *__restrict__ cbase, const TileDesc &td,
                                           const double *theta_l, double *cnt_l, double *den_l,
                                           const uint32_t *__restrict__ row_w_perm,
                                           const uint32_t *__restrict__ ibase, const float *dict_l, uint32_t exp_mask, uint32_t cs)
{
    // weight of alignment j >= kCh of the lane's read (reload loops)
    auto w_at = [&](uint32_t j) -> double {
        if (kDict == kWBytes) return (double)dict_l[(ibase[(j >> 2) * 64 + lane] >> (8 * (j & 3))) & 0xffu];
        if (kDict == kWWords) return (double)dict_l[code_half(ibase[(j >> 1) * 64 + lane], j & 1)];
        if (kDict == kWFused) return (double)dict_l[code_widx(cbase[(j >> 1) * 64 + lane], j & 1)];
        return (double)wbase[j * 64 + lane];
    };
    const uint32_t rl = s * 64 + lane;
    __builtin_amdgcn_sched_barrier(0);
    // Land every operand of this slice here (the loads of the NEXT slice stay in flight):
    // one counted s_waitcnt in front of the fold instead of a wait per alignment woven
    // through the LDS traffic.  Measured: 0.272 -> 0.237 ms per pass at C3.
    if (kDict == kWBytes) {
#pragma unroll
        for (int k = 0; k < kCh / 4; ++k) asm volatile("" ::"v"(cur.wi[k]));
    } else if (kDict == kWWords) {
#pragma unroll
        for (int k = 0; k < kCh / 2; ++k) asm volatile("" ::"v"(cur.wi[k]));
    } else if (kDict == kWPlain) {
#pragma unroll
        for (int k = 0; k < kCh; ++k) asm volatile("" ::"v"(cur.w[k]));
    }
#pragma unroll
    for (int k = 0; k < kCh / 2; ++k) asm volatile("" ::"v"(cur.c[k]));
    // the second element of the last pair of an odd-width slice belongs to the next row: it must
    // carry no weight.  Done once here, so the passes below need no per-alignment select.
    WT wz[kCh];
#pragma unroll
    for (int k = 0; k < kCh; ++k)
        wz[k] = (kDict == kWPlain && (k & 1) && (uint32_t)k >= width) ? (WT)0 : slice_w<kDict>(cur, k, dict_l);
    double x[kCh];
    double denom = den_l[rl];
#pragma unroll
    for (int k = 0; k < kCh; ++k) {
        const uint32_t off = code_off<kDict>(code_half(cur.c[k >> 1], k & 1));
        x[k] = lds_ld(theta_l, OEM_EXP(8u) ? lane * 8u : off) * (double)wz[k]; // em.rs:111
        denom += x[k];
    }
    for (uint32_t j = kCh; j < width; ++j) { // reads with more than kCh local alignments
        const uint32_t cc = cbase[(j >> 1) * 64 + lane];
        const uint32_t off = code_off<kDict>(code_half(cc, j & 1));
        denom += lds_ld(theta_l, off) * w_at(j);
    }
    double scale = 1.0;
    if (row_w_perm) scale = rl < td.n_rows ? (double)row_w_perm[td.row_base + rl] : 0.0;
    const double inv = denom > OEM_EM_DENOM_THRESH ? scale / denom : 0.0;  // em.rs:115
    den_l[rl] = inv;

    // The count window is kept in 1 << cs interleaved copies (entry c of copy p at
    // ((c << cs) + p) * 8): lanes of different copies that add into the same
    // transcript hit different addresses (and adjacent banks), which divides the
    // same-address serialisation of the LDS atomics by up to the number of copies.
    const uint32_t copy_off = (lane & ((1u << cs) - 1u)) * 8u;
    // k = 0 is the read's anchor.  Inside a highly expressed transcript all 64 lanes
    // share it, and 64 same-address LDS atomics would serialise: reduce across the
    // wavefront and let one lane add.
    {
        const uint32_t off0 = code_off<kDict>(code_half(cur.c[0], 0));
        const uint32_t u = __builtin_amdgcn_readfirstlane(off0);
        const double v0 = x[0] * inv;
        if (__all(off0 == u)) {
            const double sum = wave_sum_f64(v0);
            if (lane == 0 && sum != 0.0) lds_add_f64(lds_at(cnt_l, u << cs), sum);
        } else if (v0 != 0.0) {
            lds_add_f64(lds_at(cnt_l, (off0 << cs) + copy_off), v0);      // em.rs:128-129
        }
    }
#pragma unroll
    for (int k = 1; k < kCh; ++k) {
        if ((uint32_t)k < width) { // uniform
            const uint32_t off = code_off<kDict>(code_half(cur.c[k >> 1], k & 1));
            const double v = x[k] * inv;
            if (v != 0.0 && !OEM_EXP(4u)) lds_add_f64(lds_at(cnt_l, (off << cs) + copy_off), v);
        }
    }
    for (uint32_t j = kCh; j < width; ++j) {
        const uint32_t cc = cbase[(j >> 1) * 64 + lane];
        const uint32_t off = code_off<kDict>(code_half(cc, j & 1));
        const double v = lds_ld(theta_l, off) * w_at(j) * inv;
        if (v != 0.0 && !OEM_EXP(4u)) lds_add_f64(lds_at(cnt_l, (off << cs) + copy_off), v);
    }
}

// A remote record (oem_layout_pack.hip): kPacked: one u32 = (transcript - problem base) | read << 22;
// otherwise transcript u32 + read u16.  Its queue slot comes from the tile's slot table.
template <bool kPacked, bool kNT>
__device__ __forceinline__ void ld_remote(const uint32_t *__restrict__ r_a, const uint16_t *__restrict__ r_row, uint32_t o,
                                          uint32_t tid_base, uint32_t &t, uint32_t &row)
{
    if (kPacked) {
        const uint32_t pk = ld_stream<kNT>(&r_a[o]);
        t = tid_base + (pk & ((1u << kPackRowShift) - 1u));
        row = pk >> kPackRowShift;
    } else {
        t = ld_stream<kNT>(&r_a[o]);
        row = ld_stream<kNT>(&r_row[o]);
    }
}

// The FIRST slice of a wavefront is the widest of its four (a tile's reads are ordered by local-alignment
// count and dealt to the wavefronts round-robin), and at 8 alignments per read on average it is wider than the
// kCh = 8 a register set holds: its alignments 8..15 used to go through the reload loops of fold_slice -- two
// synchronous loads per alignment in the middle of the fold, each wait also draining the prefetch of the next
// slice.  In-kernel timestamps (scripts/tile_probe.py, profiles/r03_notes.md) put 7.8 us of a tile's 26 us
// there.  Both register sets are idle until the local phase begins, so the first slice's alignments 8..15 are
// loaded into the SECOND set with everything else at the top of the kernel (hidden behind the remote phases);
// the fold runs over 16 register-resident alignments, hands the first set to the next slice's prefetch as
// soon as its own scatter is done with it, and only reads with more than 16 local alignments reload.
template <typename WT, int kCh, int kCopies, bool kNT, int kDict>
__device__ __forceinline__ void fold_first(SliceRegs<WT, kCh> &lo, const SliceRegs<WT, kCh> &hi, uint32_t width, uint32_t s,
                                           uint32_t lane, const WT *__restrict__ wbase, const uint32_t *__restrict__ cbase,
                                           const TileDesc &td, const double *theta_l, double *cnt_l, double *den_l,
                                           const uint32_t *__restrict__ row_w_perm, bool prefetch_next,
                                           const WT *__restrict__ next_w, const uint32_t *__restrict__ next_c, uint32_t next_width,
                                           const uint32_t *__restrict__ ibase, const uint32_t *__restrict__ next_i,
                                           const float *dict_l, uint32_t exp_mask, uint32_t cs)
{
    auto w_at = [&](uint32_t j) -> double {
        if (kDict == kWBytes) return (double)dict_l[(ibase[(j >> 2) * 64 + lane] >> (8 * (j & 3))) & 0xffu];
        if (kDict == kWWords) return (double)dict_l[code_half(ibase[(j >> 1) * 64 + lane], j & 1)];
        if (kDict == kWFused) return (double)dict_l[code_widx(cbase[(j >> 1) * 64 + lane], j & 1)];
        return (double)wbase[j * 64 + lane];
    };
    const uint32_t rl = s * 64 + lane;
    __builtin_amdgcn_sched_barrier(0);
    if (kDict == kWBytes) {
#pragma unroll
        for (int k = 0; k < kCh / 4; ++k) asm volatile("" ::"v"(lo.wi[k]), "v"(hi.wi[k]));
    } else if (kDict == kWWords) {
#pragma unroll
        for (int k = 0; k < kCh / 2; ++k) asm volatile("" ::"v"(lo.wi[k]), "v"(hi.wi[k]));
    } else if (kDict == kWPlain) {
#pragma unroll
        for (int k = 0; k < kCh; ++k) asm volatile("" ::"v"(lo.w[k]), "v"(hi.w[k]));
    }
#pragma unroll
    for (int k = 0; k < kCh / 2; ++k) asm volatile("" ::"v"(lo.c[k]), "v"(hi.c[k]));
    // (load_slice zero-fills beyond the width; the second element of the last pair of an odd width belongs to
    // the next row and must carry no weight)
    double x[kCh];
    double denom = den_l[rl];
#pragma unroll
    for (int k = 0; k < kCh; ++k) {
        const WT wk = (kDict == kWPlain && (k & 1) && (uint32_t)k >= width) ? (WT)0 : slice_w<kDict>(lo, k, dict_l);
        const uint32_t off = code_off<kDict>(code_half(lo.c[k >> 1], k & 1));
        x[k] = lds_ld(theta_l, OEM_EXP(8u) ? lane * 8u : off) * (double)wk;      // em.rs:111
        denom += x[k];
    }
    if (width > (uint32_t)kCh) { // wave-uniform
#pragma unroll
        for (int k = 0; k < kCh; ++k) {
            const WT wk = (kDict == kWPlain && (k & 1) && (uint32_t)(k + kCh) >= width) ? (WT)0 : slice_w<kDict>(hi, k, dict_l);
            const uint32_t off = code_off<kDict>(code_half(hi.c[k >> 1], k & 1));
            denom += lds_ld(theta_l, off) * (double)wk;
        }
    }
    for (uint32_t j = 2 * kCh; j < width; ++j) { // reads with more than 16 local alignments
        const uint32_t cc = cbase[(j >> 1) * 64 + lane];
        const uint32_t off = code_off<kDict>(code_half(cc, j & 1));
        denom += lds_ld(theta_l, off) * w_at(j);
    }
    double scale = 1.0;
    if (row_w_perm) scale = rl < td.n_rows ? (double)row_w_perm[td.row_base + rl] : 0.0;
    const double inv = denom > OEM_EM_DENOM_THRESH ? scale / denom : 0.0;  // em.rs:115
    den_l[rl] = inv;
    const uint32_t copy_off = (lane & ((1u << cs) - 1u)) * 8u;
    {
        const uint32_t off0 = code_off<kDict>(code_half(lo.c[0], 0));
        const uint32_t u = __builtin_amdgcn_readfirstlane(off0);
        const double v0 = x[0] * inv;
        if (__all(off0 == u)) {
            const double sum = wave_sum_f64(v0);
            if (lane == 0 && sum != 0.0) lds_add_f64(lds_at(cnt_l, u << cs), sum);
        } else if (v0 != 0.0) {
            lds_add_f64(lds_at(cnt_l, (off0 << cs) + copy_off), v0);      // em.rs:128-129
        }
    }
#pragma unroll
    for (int k = 1; k < kCh; ++k) {
        if ((uint32_t)k < width) { // uniform
            const uint32_t off = code_off<kDict>(code_half(lo.c[k >> 1], k & 1));
            const double v = x[k] * inv;
            if (v != 0.0 && !OEM_EXP(4u)) lds_add_f64(lds_at(cnt_l, (off << cs) + copy_off), v);
        }
    }
    // the first register set is done: the next slice's loads go out now, under the rest of this fold
    if (prefetch_next) load_slice<WT, kCh, kNT, kDict>(lo, next_w, next_c, lane, next_width, next_i);
    if (width > (uint32_t)kCh) {
#pragma unroll
        for (int k = 0; k < kCh; ++k) {
            if ((uint32_t)(k + kCh) < width) { // uniform
                const uint32_t off = code_off<kDict>(code_half(hi.c[k >> 1], k & 1));
                const double v = lds_ld(theta_l, off) * (double)slice_w<kDict>(hi, k, dict_l) * inv;
                if (v != 0.0 && !OEM_EXP(4u)) lds_add_f64(lds_at(cnt_l, (off << cs) + copy_off), v);
            }
        }
    }
    for (uint32_t j = 2 * kCh; j < width; ++j) {
        const uint32_t cc = cbase[(j >> 1) * 64 + lane];
        const uint32_t off = code_off<kDict>(code_half(cc, j & 1));
        const double v = lds_ld(theta_l, off) * w_at(j) * inv;
        if (v != 0.0 && !OEM_EXP(4u)) lds_add_f64(lds_at(cnt_l, (off << cs) + copy_off), v);
    }
}

} // namespace

} // namespace oem
