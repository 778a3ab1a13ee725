// oem_layout_pack.hip -- the last step of laying a store out (after either builder): slim remote records.
//
// The builders (oem_layout.cpp on the host, oem_layout_device.hip on the GPU) describe a remote
// alignment by (transcript u32, weight, read-in-tile u16, queue slot u32): 14 bytes with f32 weights,
// streamed once per E/M pass.  Two of those fields are redundant:
//
//   * the queue slot.  A tile's remote records are sorted by transcript, hence by destination bucket,
//     and the slots of one (tile, bucket) run are consecutive: slot = slot_base(tile, bucket) + rank in
//     the run.  A small per-tile table, one u32 per bucket between the tile's first and last,
//         sd[tile.sd_begin + bucket - tile.b_min] = slot_base(tile, bucket) - index of the run's first record
//     gives slot = sd[...] + (index of the record in the tile): the 4-byte slot stream becomes a few
//     cache-resident words per tile, looked up off the critical path (the slot is needed last);
//   * half of the transcript.  Relative to the first transcript of the tile's EM problem it fits 22 bits
//     whenever a problem has fewer than 2^22 transcripts (any real transcriptome; per-cell batches are
//     many such problems), so it shares one u32 with the 10-bit read index:
//         r_pk = (transcript - problem * problem_size) | read << 22.
//
// A remote record is then 8 bytes (12 with the f64 weights of the coverage model) instead of 14 (18):
// -84 MB of the 945 MB one pass of the 10 M-read store moves.  Stores whose problems are wider than
// 2^22 transcripts keep (transcript u32, read u16) and still lose the slot stream.
#include <hipcub/hipcub.hpp>

#include "oem_internal.h"

namespace oem {

namespace {

constexpr int kPT = 256;

// table entries a tile needs: buckets between its first and its last remote record (sorted by transcript)
__global__ __launch_bounds__(kPT) void k_sd_sizes(const TileDesc *__restrict__ tiles, uint32_t n_tiles,
                                                  const uint32_t *__restrict__ r_tid, uint32_t *__restrict__ sizes)
{
    const uint32_t ti = blockIdx.x * blockDim.x + threadIdx.x;
    if (ti >= n_tiles) return;
    const TileDesc td = tiles[ti];
    uint32_t n = 0;
    if (td.remote_cnt) n = (r_tid[td.remote_begin + td.remote_cnt - 1] >> kBucketShift) - (r_tid[td.remote_begin] >> kBucketShift) + 1;
    sizes[ti] = n;
}

__global__ __launch_bounds__(kPT) void k_sd_fill(TileDesc *__restrict__ tiles, const uint32_t *__restrict__ begins,
                                                 const uint32_t *__restrict__ r_tid, const uint32_t *__restrict__ r_slot,
                                                 uint32_t *__restrict__ sd)
{
    const uint32_t ti = blockIdx.x;
    const TileDesc td = tiles[ti];
    const uint32_t begin = begins[ti];
    const uint32_t bmin = td.remote_cnt ? r_tid[td.remote_begin] >> kBucketShift : 0u;
    if (threadIdx.x == 0) {
        tiles[ti].sd_begin = begin;
        tiles[ti].b_min = bmin;
    }
    for (uint32_t i = threadIdx.x; i < td.remote_cnt; i += kPT) {
        const uint32_t o = td.remote_begin + i;
        const uint32_t b = r_tid[o] >> kBucketShift;
        if (i == 0 || (r_tid[o - 1] >> kBucketShift) != b) sd[begin + (b - bmin)] = r_slot[o] - i; // (mod 2^32)
    }
}

__global__ __launch_bounds__(kPT) void k_pack_records(const TileDesc *__restrict__ tiles, uint32_t problem_size,
                                                      const uint32_t *__restrict__ r_tid, const uint16_t *__restrict__ r_row,
                                                      uint32_t *__restrict__ r_pk)
{
    const TileDesc td = tiles[blockIdx.x];
    const uint32_t base = td.problem * problem_size;
    for (uint32_t i = threadIdx.x; i < td.remote_cnt; i += kPT) {
        const uint32_t o = td.remote_begin + i;
        r_pk[o] = (r_tid[o] - base) | ((uint32_t)r_row[o] << kPackRowShift);
    }
}

} // namespace

// s->tiled holds a complete layout in the builders' form; on return it holds the slot table, the packed
// records when they apply, and (unless `keep_unpacked`: the layout tests hash them) no slot / transcript /
// read streams any more.
int pack_remote_records(oem_store *s, uint32_t problem_size, bool keep_unpacked)
{
    DeviceTiled &t = s->tiled;
    if (!t.present || t.n_tiles == 0) return OEM_OK;
    hipStream_t st = s->stream;
    uint32_t *sizes = nullptr, *begins = nullptr;
    void *tmp = nullptr;
    int rc = OEM_OK;
    auto body = [&]() -> int {
        OEM_HIP(hipMalloc((void **)&sizes, sizeof(uint32_t) * ((size_t)t.n_tiles + 1)));
        OEM_HIP(hipMalloc((void **)&begins, sizeof(uint32_t) * ((size_t)t.n_tiles + 1)));
        OEM_HIP(hipMemsetAsync(sizes, 0, sizeof(uint32_t) * ((size_t)t.n_tiles + 1), st));
        hipLaunchKernelGGL(k_sd_sizes, dim3((t.n_tiles + kPT - 1) / kPT), dim3(kPT), 0, st, t.tiles, t.n_tiles, t.r_tid, sizes);
        OEM_HIP(hipGetLastError());
        size_t tmp_bytes = 0;   // (one element past the tiles: begins[n_tiles] = total)
        OEM_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, sizes, begins, (int)t.n_tiles + 1, st));
        OEM_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1));
        OEM_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, sizes, begins, (int)t.n_tiles + 1, st));
        uint32_t total = 0;
        OEM_HIP(hipMemcpyAsync(&total, begins + t.n_tiles, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        OEM_HIP(hipStreamSynchronize(st));
        // (a sum beyond 2^32 would need > 2^32 / n_buckets tiles: the builders' tile x bucket guard is far below)
        t.n_sd = total;
        OEM_HIP(hipMalloc((void **)&t.sd, sizeof(uint32_t) * ((size_t)total + 1)));
        s->hbm_bytes += sizeof(uint32_t) * ((size_t)total + 1);
        OEM_HIP(hipMemsetAsync(t.sd, 0, sizeof(uint32_t) * ((size_t)total + 1), st));
        hipLaunchKernelGGL(k_sd_fill, dim3(t.n_tiles), dim3(kPT), 0, st, t.tiles, begins, t.r_tid, t.r_slot, t.sd);
        OEM_HIP(hipGetLastError());
        t.problem_size = problem_size;
        const uint64_t span = problem_size ? problem_size : s->csr.n_txps;
        t.packed = span <= (1ull << kPackRowShift);
        if (t.packed) {
            OEM_HIP(hipMalloc((void **)&t.r_pk, sizeof(uint32_t) * (t.n_remote ? t.n_remote : 1)));
            s->hbm_bytes += sizeof(uint32_t) * t.n_remote;
            hipLaunchKernelGGL(k_pack_records, dim3(t.n_tiles), dim3(kPT), 0, st, t.tiles, problem_size, t.r_tid, t.r_row, t.r_pk);
            OEM_HIP(hipGetLastError());
        }
        OEM_HIP(hipStreamSynchronize(st));
        if (!keep_unpacked) {
            hipFree(t.r_slot);
            t.r_slot = nullptr;
            s->hbm_bytes -= sizeof(uint32_t) * t.n_remote;
            if (t.packed) {
                hipFree(t.r_tid);
                hipFree(t.r_row);
                t.r_tid = nullptr;
                t.r_row = nullptr;
                s->hbm_bytes -= (sizeof(uint32_t) + sizeof(uint16_t)) * t.n_remote;
            }
        }
        return OEM_OK;
    };
    rc = body();
    hipFree(sizes);
    hipFree(begins);
    hipFree(tmp);
    return rc;
}

} // namespace oem
