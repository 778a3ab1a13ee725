// oem_layout_dict.hip -- dictionary-coded weights of the local alignments (after either layout builder).
//
// The weight of an alignment is as_prob = exp((score - best score of the read) / D) in f32
// (oarfish_types.rs:1100-1114): alignment scores are integers, so a store holds as many distinct weights as it
// holds distinct score gaps -- tens to a few hundred, not 80 M.  When there are at most 256 of them the local
// weights are stored as an index into a table of the distinct f32 values: with up to 128 values FUSED into the
// spare bits of the word an alignment's 16-bit window code shares with its neighbour's (bits 0..2 and 12..15 of either
// half: a code is 8 * (transcript - lo) < 4096; k_dict_fuse below says which bits hold what)
// -- a local alignment is then its two code bytes and nothing else -- with 129..256 as BYTES, four indices per u32
// in the tiles' SELL layout (1 + 2 bytes per local alignment).  The f32 stream costs 4 + 2.  The table sits in
// 1 KiB of LDS per workgroup, and the value the kernel multiplies with is bit for bit the f32 the caller handed
// over -- lossless, no tolerance involved.  Measured at C3 (98 distinct weights): the pass
// 0.199 -> 0.176 ms (profiles/r03_notes.md): the pass follows its bytes.
//
// Round 4: 257 .. 1024 distinct weights (what long reads produce: integer score gaps up to 5 % of a best score in the
// thousands, and exp(-gap / 5) reaches 0 in f32 at a gap of ~520) take 16-bit indices, two per u32 in the geometry
// of the window codes (2 + 2 bytes per local alignment), from a 4 KiB table in LDS.
//
// Stores with more distinct weights, f64 weights (the coverage model multiplies a second factor in) or the wide
// window cap (per-cell batches: see build_weight_dictionary) keep the f32 stream; `oem_store_opts.weight_coding = 1` keeps it for any store.
//
// Layout: slice s of a tile holds words(s) = (width[s] + 3) / 4 index words per lane,
//     widx[(i_base[tile] + sum_{s' < s} words(s') + g) * 64 + lane]  = indices of alignments 4g .. 4g+3 of the lane's
// read, byte m = alignment 4g + m, 0 beyond the read's own alignments.  Index 0 is always the weight 0.0 (the
// padding of the SELL slices), so padded entries need no masking.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstring>
#include <vector>

#include "oem_internal.h"

namespace oem {

namespace {

constexpr int kDT = 256;
constexpr uint32_t kDictMax = 1024;  // entries of the largest table (16-bit indices; 4 KiB of LDS in k_em_tile, whose 16-bit
                                     // instantiation runs four workgroups per CU: all its slices register-resident)
constexpr uint32_t kSetSlots = 4096; // open addressing, <= kDictMax + 1 live keys
constexpr uint32_t kEmptyKey = 0xffffffffu; // a NaN pattern: never a weight of the store (NaN rows were dropped at upload)

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// insert `key` into an open-addressing set; returns false when the set is full of other keys
template <typename AtomicCas>
__device__ __forceinline__ bool set_insert(uint32_t *tab, uint32_t key, uint32_t *count, uint32_t limit, AtomicCas cas)
{
    uint32_t h = mix32(key) & (kSetSlots - 1);
    for (uint32_t probe = 0; probe < kSetSlots; ++probe) {
        const uint32_t old = cas(&tab[h], kEmptyKey, key);
        if (old == key) return true;
        if (old == kEmptyKey) return atomicAdd(count, 1u) < limit;
        h = (h + 1) & (kSetSlots - 1);
    }
    return false;
}

// distinct bit patterns of w[0, n): per-workgroup set in LDS, merged into a global set; *too_many is raised as
// soon as any set holds more than `limit` keys
__global__ __launch_bounds__(kDT) void k_dict_collect(const float *__restrict__ w, uint64_t n, uint32_t limit,
                                                      uint32_t *__restrict__ gtab, uint32_t *gcount, uint32_t *too_many)
{
    __shared__ uint32_t tab[kSetSlots];
    __shared__ uint32_t cnt;
    for (uint32_t i = threadIdx.x; i < kSetSlots; i += kDT) tab[i] = kEmptyKey;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    auto lds_cas = [](uint32_t *p, uint32_t expect, uint32_t v) { return atomicCAS(p, expect, v); };
    uint32_t last = kEmptyKey; // most values repeat their neighbour
    uint32_t trip = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kDT + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kDT, ++trip) {
        if ((trip & 255u) == 0 && *(volatile uint32_t *)too_many) break; // another workgroup gave up: so do we
        const uint32_t key = __float_as_uint(w[i]);
        if (key == last) continue;
        last = key;
        if (key & 0x80000000u) { // a negative weight, -0.0 or a negative NaN: not ordered like its bits -- keep the f32 stream
            *too_many = 1u;
            break;
        }
        // (almost every value is in the set already: a plain read of its slot -- a broadcast when the lanes agree --
        // before the compare-and-swap that would serialise them)
        if (((volatile uint32_t *)tab)[mix32(key) & (kSetSlots - 1)] == key) continue;
        if (!set_insert(tab, key, &cnt, limit, lds_cas)) {
            *too_many = 1u;
            break;
        }
    }
    __syncthreads();
    if (*(volatile uint32_t *)too_many) return;
    auto glb_cas = [](uint32_t *p, uint32_t expect, uint32_t v) { return atomicCAS(p, expect, v); };
    for (uint32_t i = threadIdx.x; i < kSetSlots; i += kDT) {
        const uint32_t key = tab[i];
        if (key != kEmptyKey && !set_insert(gtab, key, gcount, limit, glb_cas)) *too_many = 1u;
    }
}

// index words a tile needs
__global__ __launch_bounds__(kDT) void k_dict_sizes(const TileDesc *__restrict__ tiles, uint32_t n_tiles,
                                                    uint32_t *__restrict__ sizes)
{
    const uint32_t ti = blockIdx.x * blockDim.x + threadIdx.x;
    if (ti >= n_tiles) return;
    uint32_t n = 0;
    for (uint32_t s = 0; s < kTileSlices; ++s) n += (tiles[ti].width[s] + 3u) >> 2;
    sizes[ti] = n;
}

// one workgroup per tile: the tile's weights -> index words
__global__ __launch_bounds__(kDT) void k_dict_encode(const TileDesc *__restrict__ tiles, const uint32_t *__restrict__ i_base,
                                                     const float *__restrict__ w, const float *__restrict__ dict,
                                                     uint32_t n_dict, uint32_t *__restrict__ widx, uint32_t *bad)
{
    __shared__ uint32_t keys[256];
    const TileDesc td = tiles[blockIdx.x];
    keys[threadIdx.x] = threadIdx.x < n_dict ? __float_as_uint(dict[threadIdx.x]) : 0x7f800000u; // +inf beyond the table
    __syncthreads();
    uint32_t woff = td.w_base, ioff = i_base[blockIdx.x];
    for (uint32_t s = 0; s < kTileSlices; ++s) {
        const uint32_t width = td.width[s], words = (width + 3u) >> 2;
        for (uint32_t e = threadIdx.x; e < words * 64; e += kDT) {
            const uint32_t g = e >> 6, lane = e & 63u;
            uint32_t word = 0;
            for (uint32_t m = 0; m < 4; ++m) {
                const uint32_t j = 4 * g + m;
                if (j >= width) break;
                const uint32_t key = __float_as_uint(w[(size_t)(woff + j) * 64 + lane]);
                // non-negative floats order like their bit patterns: binary search in the ascending table
                uint32_t a = 0, b = n_dict;
                while (a < b) {
                    const uint32_t mid = (a + b) >> 1;
                    if (keys[mid] < key) a = mid + 1;
                    else b = mid;
                }
                if (a >= n_dict || keys[a] != key) { *bad = 1u; a = 0; }
                word |= a << (8 * m);
            }
            widx[(size_t)(ioff + g) * 64 + lane] = word;
        }
        woff += width;
        ioff += words;
    }
}

// 257 .. kDictMax distinct weights: 16-bit indices, two per u32, in the geometry of the window codes (the index
// word of alignments 2g, 2g+1 of a lane's read sits where their code word sits: widx16[(c_base + ...) * 64 + lane]),
// so the kernels address it with the code offsets they already have.  2 + 2 bytes per local alignment.
__global__ __launch_bounds__(kDT) void k_dict_encode16(const TileDesc *__restrict__ tiles, const float *__restrict__ w,
                                                       const float *__restrict__ dict, uint32_t n_dict,
                                                       uint32_t *__restrict__ widx16, uint32_t *bad)
{
    __shared__ uint32_t keys[kDictMax];
    const TileDesc td = tiles[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < kDictMax; i += kDT) keys[i] = i < n_dict ? __float_as_uint(dict[i]) : 0x7f800000u;
    __syncthreads();
    uint32_t woff = td.w_base, coff = td.c_base;
    for (uint32_t s = 0; s < kTileSlices; ++s) {
        const uint32_t width = td.width[s], pairs = (width + 1u) >> 1;
        for (uint32_t e = threadIdx.x; e < pairs * 64; e += kDT) {
            const uint32_t g = e >> 6, lane = e & 63u;
            uint32_t word = 0;
            for (uint32_t m = 0; m < 2; ++m) {
                const uint32_t j = 2 * g + m;
                if (j >= width) break;
                const uint32_t key = __float_as_uint(w[(size_t)(woff + j) * 64 + lane]);
                uint32_t a = 0, b = n_dict;
                while (a < b) {
                    const uint32_t mid = (a + b) >> 1;
                    if (keys[mid] < key) a = mid + 1;
                    else b = mid;
                }
                if (a >= n_dict || keys[a] != key) { *bad = 1u; a = 0; }
                word |= a << (16 * m);
            }
            widx16[(size_t)(coff + g) * 64 + lane] = word;
        }
        woff += width;
        coff += pairs;
    }
}

// the remote records' weights -> one table index byte each
__global__ __launch_bounds__(kDT) void k_dict_encode_remote(const float *__restrict__ r_w, uint64_t n, const float *__restrict__ dict,
                                                            uint32_t n_dict, uint8_t *__restrict__ r_wi, uint32_t *bad)
{
    __shared__ uint32_t keys[256];
    keys[threadIdx.x] = threadIdx.x < n_dict ? __float_as_uint(dict[threadIdx.x]) : 0x7f800000u;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * kDT + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kDT) {
        const uint32_t key = __float_as_uint(r_w[i]);
        uint32_t a = 0, b = n_dict;
        while (a < b) {
            const uint32_t mid = (a + b) >> 1;
            if (keys[mid] < key) a = mid + 1;
            else b = mid;
        }
        if (a >= n_dict || keys[a] != key) { *bad = 1u; a = 0; }
        r_wi[i] = (uint8_t)a;
    }
}

// <= 128 distinct weights: the index goes into the spare bits of the pair of window codes an alignment shares a word
// with (a code is 8 * (transcript - lo) < 4096: bits 0..2 and 12..15 of either half are free): its low four bits into
// bits 12..15 of its own half, its high three into bits 0..2 of the other half; no index stream
// kCheckOnly: nothing is written -- the codes are changed in place, so the pass that changes them runs only after
// this one has found every code and every weight to fit (a store that does not keeps its untouched codes and takes
// the next coding down)
template <bool kCheckOnly>
__global__ __launch_bounds__(kDT) void k_dict_fuse(const TileDesc *__restrict__ tiles, const float *__restrict__ w,
                                                   const float *__restrict__ dict, uint32_t n_dict,
                                                   uint32_t *__restrict__ codes, uint32_t *bad)
{
    __shared__ uint32_t keys[256];
    const TileDesc td = tiles[blockIdx.x];
    keys[threadIdx.x] = threadIdx.x < n_dict ? __float_as_uint(dict[threadIdx.x]) : 0x7f800000u;
    __syncthreads();
    uint32_t woff = td.w_base, coff = td.c_base;
    for (uint32_t s = 0; s < kTileSlices; ++s) {
        const uint32_t width = td.width[s], pairs = (width + 1u) >> 1;
        for (uint32_t e = threadIdx.x; e < pairs * 64; e += kDT) {
            const uint32_t g = e >> 6, lane = e & 63u;
            const uint32_t word0 = codes[(size_t)(coff + g) * 64 + lane];
            uint32_t word = word0;
            for (uint32_t m = 0; m < 2; ++m) {
                const uint32_t j = 2 * g + m;
                if (j >= width) break;
                const uint32_t key = __float_as_uint(w[(size_t)(woff + j) * 64 + lane]);
                uint32_t a = 0, b = n_dict;
                while (a < b) {
                    const uint32_t mid = (a + b) >> 1;
                    if (keys[mid] < key) a = mid + 1;
                    else b = mid;
                }
                if (a >= n_dict || keys[a] != key || a > 127u) { *bad = 1u; a = 0; }
                const uint32_t half = (word0 >> (16 * m)) & 0xffffu;
                if (half & 0xf007u) *bad = 1u; // (a code of the narrow window has these bits clear)
                // low four bits of the index into the alignment's own half, the high three into the low bits of the
                // pair's other half (code_widx, oem_tile_common.h: one rotation of the word decodes it)
                word |= ((a & 15u) << (12 + 16 * m)) | (((a >> 4) & 7u) << (16 * (1 - m)));
            }
            if (!kCheckOnly) codes[(size_t)(coff + g) * 64 + lane] = word;
        }
        woff += width;
        coff += pairs;
    }
}

} // namespace

// s->tiled holds a complete layout with f32 weights; on return it also holds the coded weights when the store
// has at most 256 distinct ones (t.dict_n > 0).  The f32 stream stays: the batched bootstrap kernel reads it.
int build_weight_dictionary(oem_store *s)
{
    DeviceTiled &t = s->tiled;
    if (!t.present || t.n_tiles == 0 || s->csr.w_is_f64 || !t.w32) return OEM_OK;
    // The wide-window instantiation of k_em_tile holds exactly four 40 KiB workgroups per CU with the 1 KiB table of
    // the BYTE coding (its tiles are cut for 1984 transcripts, oem_layout.h: round 3 had 2048 and the table made it
    // three workgroups, 5 % slower).  Its codes have no spare bits for a fused index and the CU no room for the
    // 4 KiB table of 16-bit indices: a wide store is byte-coded (<= 256 distinct weights) or keeps the f32 stream.
    const bool wide = t.win_cap > kWin;
    hipStream_t st = s->stream;
    uint32_t *gtab = nullptr, *small = nullptr, *sizes = nullptr, *begins = nullptr;
    void *tmp = nullptr;
    auto body = [&]() -> int {
        // number of weight rows: the last tile's base + its slices
        TileDesc last;
        OEM_HIP(hipMemcpyAsync(&last, t.tiles + (t.n_tiles - 1), sizeof(TileDesc), hipMemcpyDeviceToHost, st));
        OEM_HIP(hipStreamSynchronize(st));
        uint64_t w_slots = last.w_base;
        for (uint32_t i = 0; i < kTileSlices; ++i) w_slots += last.width[i];
        const uint64_t n = w_slots * 64;
        if (n == 0) return OEM_OK;
        OEM_HIP(hipMalloc((void **)&gtab, sizeof(uint32_t) * kSetSlots));
        OEM_HIP(hipMalloc((void **)&small, sizeof(uint32_t) * 4));
        OEM_HIP(hipMemsetAsync(gtab, 0xff, sizeof(uint32_t) * kSetSlots, st));
        OEM_HIP(hipMemsetAsync(small, 0, sizeof(uint32_t) * 4, st));
        const uint32_t limit = wide ? 256u : kDictMax; // (0.0 is one of them: the SELL padding; a store without padding gets it added below)
        uint64_t g = (n + kDT - 1) / kDT;
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(k_dict_collect, dim3((uint32_t)g), dim3(kDT), 0, st, t.w32, n, limit, gtab, small, small + 1);
        if (t.n_remote) { // the remote records' weights come from the same table
            uint64_t gr = (t.n_remote + kDT - 1) / kDT;
            if (gr > 2048) gr = 2048;
            hipLaunchKernelGGL(k_dict_collect, dim3((uint32_t)gr), dim3(kDT), 0, st, t.r_w32, t.n_remote, limit, gtab, small,
                               small + 1);
        }
        OEM_HIP(hipGetLastError());
        uint32_t h_small[4];
        std::vector<uint32_t> h_tab(kSetSlots);
        OEM_HIP(hipMemcpyAsync(h_small, small, sizeof(h_small), hipMemcpyDeviceToHost, st));
        OEM_HIP(hipMemcpyAsync(h_tab.data(), gtab, sizeof(uint32_t) * kSetSlots, hipMemcpyDeviceToHost, st));
        OEM_HIP(hipStreamSynchronize(st));
        if (h_small[1]) return OEM_OK; // more than kDictMax distinct weights: the f32 stream stays the only one
        std::vector<uint32_t> keys;
        for (uint32_t k : h_tab)
            if (k != kEmptyKey) keys.push_back(k);
        if (std::find(keys.begin(), keys.end(), 0u) == keys.end()) keys.push_back(0u); // index 0 = weight 0.0
        for (uint32_t k : keys)
            if (k & 0x80000000u) return OEM_OK; // a negative (or -0.0) weight: not ordered like its bits; keep f32
        if (keys.size() > (wide ? 256u : kDictMax)) return OEM_OK;
        std::sort(keys.begin(), keys.end());
        std::vector<float> dict(kDictMax, 0.0f);
        for (size_t i = 0; i < keys.size(); ++i) std::memcpy(&dict[i], &keys[i], sizeof(float));
        OEM_HIP(hipMalloc((void **)&t.dict, sizeof(float) * kDictMax));
        OEM_HIP(hipMemcpyAsync(t.dict, dict.data(), sizeof(float) * kDictMax, hipMemcpyHostToDevice, st));
        if (keys.size() > 256) {
            // 257 .. 1024 distinct weights (long reads: integer score gaps up to 5 % of a best score in the thousands):
            // 16-bit indices beside the codes; the remote records keep their f32 weights
            uint32_t c_rows = last.c_base;
            for (uint32_t i = 0; i < kTileSlices; ++i) c_rows += (last.width[i] + 1u) >> 1;
            OEM_HIP(hipMalloc((void **)&t.widx, sizeof(uint32_t) * ((size_t)c_rows + 1) * 64));
            OEM_HIP(hipMemsetAsync(t.widx + (size_t)c_rows * 64, 0, sizeof(uint32_t) * 64, st));
            hipLaunchKernelGGL(k_dict_encode16, dim3(t.n_tiles), dim3(kDT), 0, st, t.tiles, t.w32, t.dict, (uint32_t)keys.size(),
                               t.widx, small + 2);
            OEM_HIP(hipGetLastError());
            OEM_HIP(hipMemcpyAsync(h_small, small, sizeof(h_small), hipMemcpyDeviceToHost, st));
            OEM_HIP(hipStreamSynchronize(st));
            if (h_small[2]) { // (cannot happen: every weight was collected) -- the f32 stream stays the only one
                hipFree(t.widx);
                hipFree(t.dict);
                t.widx = nullptr;
                t.dict = nullptr;
                return OEM_OK;
            }
            s->hbm_bytes += sizeof(uint32_t) * ((size_t)c_rows + 1) * 64 + sizeof(float) * kDictMax;
            t.dict_words = true;
            t.dict_n = (uint32_t)keys.size();
            return OEM_OK;
        }
        {   // remote records: one index byte instead of the f32 (kept for the batched kernel)
            OEM_HIP(hipMalloc((void **)&t.r_wi, t.n_remote ? t.n_remote : 1));
            s->hbm_bytes += t.n_remote;
            if (t.n_remote) {
                uint64_t gr = (t.n_remote + kDT - 1) / kDT;
                if (gr > 4096) gr = 4096;
                hipLaunchKernelGGL(k_dict_encode_remote, dim3((uint32_t)gr), dim3(kDT), 0, st, t.r_w32, t.n_remote, t.dict,
                                   (uint32_t)keys.size(), t.r_wi, small + 2);
                OEM_HIP(hipGetLastError());
            }
        }
        if (keys.size() <= 128 && !wide && knob("OEM_DICT_NO_FUSE", 0) == 0) { // (knob: testing build, reaches the byte-stream coding)
            // the index fits the spare bits of the window codes: no stream of its own
            // (checked first, written second: a store that does not fit keeps its codes as they are and takes the
            // byte stream below)
            hipLaunchKernelGGL((k_dict_fuse<true>), dim3(t.n_tiles), dim3(kDT), 0, st, t.tiles, t.w32, t.dict, (uint32_t)keys.size(),
                               t.codes, small + 3);
            OEM_HIP(hipGetLastError());
            OEM_HIP(hipMemcpyAsync(h_small, small, sizeof(h_small), hipMemcpyDeviceToHost, st));
            OEM_HIP(hipStreamSynchronize(st));
            if (!h_small[3] && !h_small[2]) {
                hipLaunchKernelGGL((k_dict_fuse<false>), dim3(t.n_tiles), dim3(kDT), 0, st, t.tiles, t.w32, t.dict,
                                   (uint32_t)keys.size(), t.codes, small + 3);
                OEM_HIP(hipGetLastError());
                OEM_HIP(hipStreamSynchronize(st));
                t.dict_fused = true;
                t.dict_n = (uint32_t)keys.size();
                return OEM_OK;
            }
            if (h_small[2]) { // a remote weight that is not in the table (cannot happen): the f32 stream stays the only one
                hipFree(t.dict);
                hipFree(t.r_wi);
                t.dict = nullptr;
                t.r_wi = nullptr;
                return OEM_OK;
            }
        }
        // per-tile bases of the index words
        OEM_HIP(hipMalloc((void **)&sizes, sizeof(uint32_t) * ((size_t)t.n_tiles + 1)));
        OEM_HIP(hipMalloc((void **)&begins, sizeof(uint32_t) * ((size_t)t.n_tiles + 1)));
        OEM_HIP(hipMemsetAsync(sizes, 0, sizeof(uint32_t) * ((size_t)t.n_tiles + 1), st));
        hipLaunchKernelGGL(k_dict_sizes, dim3((t.n_tiles + kDT - 1) / kDT), dim3(kDT), 0, st, t.tiles, t.n_tiles, sizes);
        OEM_HIP(hipGetLastError());
        size_t tmp_bytes = 0;
        OEM_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, sizes, begins, (int)t.n_tiles + 1, st));
        OEM_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1));
        OEM_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, sizes, begins, (int)t.n_tiles + 1, st));
        uint32_t total = 0;
        OEM_HIP(hipMemcpyAsync(&total, begins + t.n_tiles, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        OEM_HIP(hipStreamSynchronize(st));
        // (one slack row, as the other slice arrays have: a wavefront may prefetch the word behind its last one)
        OEM_HIP(hipMalloc((void **)&t.widx, sizeof(uint32_t) * ((size_t)total + 1) * 64));
        OEM_HIP(hipMemsetAsync(t.widx + (size_t)total * 64, 0, sizeof(uint32_t) * 64, st));
        s->hbm_bytes += sizeof(uint32_t) * ((size_t)total + 1) * 64 + sizeof(uint32_t) * ((size_t)t.n_tiles + 1) + 1024;
        hipLaunchKernelGGL(k_dict_encode, dim3(t.n_tiles), dim3(kDT), 0, st, t.tiles, begins, t.w32, t.dict,
                           (uint32_t)keys.size(), t.widx, small + 2);
        OEM_HIP(hipGetLastError());
        OEM_HIP(hipMemcpyAsync(h_small, small, sizeof(h_small), hipMemcpyDeviceToHost, st));
        OEM_HIP(hipStreamSynchronize(st));
        if (h_small[2]) { // (cannot happen: every weight was collected) -- fall back to the f32 stream
            hipFree(t.widx);
            hipFree(t.dict);
            hipFree(t.r_wi);
            t.widx = nullptr;
            t.dict = nullptr;
            t.r_wi = nullptr;
            return OEM_OK;
        }
        t.i_base = begins;
        begins = nullptr;
        t.dict_n = (uint32_t)keys.size();
        return OEM_OK;
    };
    const int rc = body();
    hipFree(gtab);
    hipFree(small);
    hipFree(sizes);
    hipFree(begins);
    hipFree(tmp);
    return rc;
}

} // namespace oem
