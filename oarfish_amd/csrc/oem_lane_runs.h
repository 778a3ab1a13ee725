// oem_lane_runs.h -- runs of equal destinations summed across the lanes of a wavefront before they reach the LDS.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oem {

namespace {

// Queue entries arrive at a fold sorted by destination inside every (tile, bucket) run, and a wavefront takes 64
// consecutive ones: far alignments that RECUR (reads of a highly expressed gene hitting the same paralog) put the same
// destination into dozens of consecutive lanes, and 64 same-address LDS atomics take 194 clocks against 8 for 64
// different ones (profiles/r02_notes.md) -- on a store whose far hits stay inside families of three genes the fold
// took 85 us against 27 us with uniformly random far hits (profiles/r05_notes.md).  So a wavefront whose entries
// repeat sums each run of equal destinations inside its rows of 16 lanes first (a segmented scan on the vector ALU's
// data-parallel-primitive path: no LDS traffic) and only the last lane of a run adds: at most four lanes per
// destination and instruction.  Wavefronts without repeats (the uniform case: a (tile, bucket) run is ~29 entries over
// 4096 transcripts) pay one shifted compare and a ballot per trip of the fold's loop.
// kStride: lanes that belong together are kStride apart (1: a lane per entry; 2: the batched bootstrap's fold, where
// lanes 2 i and 2 i + 1 hold the two halves of entry i).
template <int kCtrl>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t v) // lanes without a source keep `old`
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, kCtrl, 0xf, 0xf, false);
}
template <int kCtrl>
__device__ __forceinline__ double dpp_f64_or(double old, double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v), o = (unsigned long long)__double_as_longlong(old);
    const uint32_t lo = dpp_u32<kCtrl>((uint32_t)o, (uint32_t)b), hi = dpp_u32<kCtrl>((uint32_t)(o >> 32), (uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
constexpr int kRowShr = 0x110, kRowShl = 0x100; // DPP controls row_shr:n / row_shl:n = base + n (n = 1 .. 15)

// do the wavefront's destinations repeat (a quarter of the lanes continue their neighbour's run)?  Wave-uniform.
template <int kStride = 1>
__device__ __forceinline__ bool keys_repeat(uint32_t d)
{
    return __popcll(__ballot(dpp_u32<kRowShr + kStride>(0xffffffffu, d) == d)) >= 16;
}

// One step of the segmented scan (Hillis-Steele with head flags): a lane whose span so far holds no run head takes the
// partial sum of the lane n down the row; `f` = a head lies inside the lane's span.  (Flags, not a comparison of the keys
// n lanes apart: where two (tile, bucket) runs meet, a destination can come back after others -- 5 9 | 5 7 -- and equal
// keys two lanes apart are then two runs.)
template <int kN, int kVals>
__device__ __forceinline__ void run_step(uint32_t &f, double (&v)[kVals])
{
    const uint32_t fn = dpp_u32<kRowShr + kN>(1u, f); // (no lane n down the row: as good as a head)
    double a[kVals];
#pragma unroll
    for (int j = 0; j < kVals; ++j) a[j] = dpp_f64_or<kRowShr + kN>(0.0, v[j]);
    if (f == 0u) {
#pragma unroll
        for (int j = 0; j < kVals; ++j) v[j] += a[j];
    }
    f |= fn;
}

// v[] of the lanes that should not add becomes 0: the last lane of each run (inside a row of 16 lanes) carries the run's
// sums.  `d` is the lane's destination (any 32-bit key); a run = consecutive lanes (kStride apart) with equal keys.
template <int kStride, int kVals>
__device__ __forceinline__ void sum_runs_of_equal_keys(uint32_t d, double (&v)[kVals])
{
    const uint32_t head = dpp_u32<kRowShr + kStride>(0xffffffffu, d) != d ? 1u : 0u; // the lane starts a run
    uint32_t f = head;
    run_step<kStride, kVals>(f, v);
    run_step<2 * kStride, kVals>(f, v);
    run_step<4 * kStride, kVals>(f, v);
    if (8 * kStride < 16) run_step<(8 * kStride < 16 ? 8 * kStride : 1), kVals>(f, v);
    const uint32_t next_head = dpp_u32<kRowShl + kStride>(1u, head); // (the last lanes of a row: no successor)
    if (next_head == 0u) {
#pragma unroll
        for (int j = 0; j < kVals; ++j) v[j] = 0.0;
    }
}

} // namespace

} // namespace oem
