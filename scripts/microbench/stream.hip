// Streaming-read bandwidth vs access width and loads in flight per lane (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 stream.hip -o stream
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename V> __device__ __forceinline__ uint32_t fold(V v);
template <> __device__ __forceinline__ uint32_t fold<uint32_t>(uint32_t v) { return v; }
template <> __device__ __forceinline__ uint32_t fold<u32x2>(u32x2 v) { return v.x + v.y; }
template <> __device__ __forceinline__ uint32_t fold<u32x3>(u32x3 v) { return v.x + v.y + v.z; }
template <> __device__ __forceinline__ uint32_t fold<u32x4>(u32x4 v) { return v.x + v.y + v.z + v.w; }

// each wave reads blocks of UNROLL consecutive "rows" of 64 lanes (like a SELL slice)
template <typename V, int UNROLL>
__global__ __launch_bounds__(256) void k_stream(const V *__restrict__ in, uint64_t n_rows64, uint32_t *sink)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (uint64_t r = wave * UNROLL; r + UNROLL <= n_rows64; r += n_waves * UNROLL) {
        V v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) v[k] = in[(r + k) * 64 + lane];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) acc += fold<V>(v[k]);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename V, int UNROLL>
static void run(const char *name, const void *buf, uint64_t bytes, uint32_t *sink, hipEvent_t e0, hipEvent_t e1)
{
    for (int g : {256 * 4, 256 * 8, 256 * 16}) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_stream<V, UNROLL>), dim3(g), dim3(256), 0, 0, (const V *)buf, bytes / sizeof(V) / 64, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("  %-10s unroll=%2d grid=%5d  %7.3f ms  %7.1f GB/s\n", name, UNROLL, g, best, bytes / best * 1e-6);
    }
}

int main()
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint64_t bytes = 3ull << 29; // 1.5 GiB (divisible by 12 * 64)
    void *buf;
    uint32_t *sink;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, bytes));
    run<uint32_t, 1>("4B/lane", buf, bytes, sink, e0, e1);
    run<uint32_t, 4>("4B/lane", buf, bytes, sink, e0, e1);
    run<uint32_t, 16>("4B/lane", buf, bytes, sink, e0, e1);
    run<u32x2, 1>("8B/lane", buf, bytes, sink, e0, e1);
    run<u32x2, 8>("8B/lane", buf, bytes, sink, e0, e1);
    run<u32x3, 1>("12B/lane", buf, bytes, sink, e0, e1);
    run<u32x3, 4>("12B/lane", buf, bytes, sink, e0, e1);
    run<u32x3, 8>("12B/lane", buf, bytes, sink, e0, e1);
    run<u32x4, 1>("16B/lane", buf, bytes, sink, e0, e1);
    run<u32x4, 4>("16B/lane", buf, bytes, sink, e0, e1);
    return 0;
}
