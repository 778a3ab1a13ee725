// What the count-window copies of k_em_tile buy per LDS atomic on gfx950: f64 atomic adds of a wavefront into a window of
// E entries kept in 2^cs interleaved copies (entry c of copy p at ((c << cs) + p) * 8, p = lane mod copies), the entries
// drawn uniformly or from a few hot ones -- addresses precomputed in registers, the loop issues LDS instructions only.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_scatter.hip -o lds_scatter
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

constexpr int kOps = 16;

template <bool kRead>
__global__ __launch_bounds__(256) void k(double *out, uint32_t W, uint32_t iters, uint32_t E, uint32_t cs, uint32_t hot, uint32_t active)
{
    extern __shared__ double lds[];
    for (uint32_t i = threadIdx.x; i < W; i += blockDim.x) lds[i] = 0.0;
    __syncthreads();
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t off[kOps];
#pragma unroll
    for (int j = 0; j < kOps; ++j) {
        // hot > 0: the entry is one of `hot` (a slice's anchors: runs of lanes share one); else uniform over the window
        const uint32_t c = hot ? (hash32(j * 131u + (gid >> 2) * 7u) % hot) * (E / hot) : hash32(gid * 977u + j) % E;
        off[j] = ((c << cs) + (lane & ((1u << cs) - 1u))) * 8u;
    }
    const bool on = lane < active; // lanes whose increment is zero do not add (padding, undrawn reads)
    double acc = 0;
    char *base = reinterpret_cast<char *>(lds);
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < kOps; ++j) {
            double *p = reinterpret_cast<double *>(base + off[j]);
            if (kRead) acc += *(volatile double *)p;
            else if (on) __hip_atomic_fetch_add(p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    double s = acc;
    for (uint32_t i = threadIdx.x; i < W; i += blockDim.x) s += lds[i];
    if (s == -1.0) out[gid] = s;
}

template <bool kRead> float run(double *out, uint32_t W, uint32_t iters, uint32_t E, uint32_t cs, uint32_t hot, uint32_t active, int blocks)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<kRead>, dim3(blocks), dim3(256), W * 8, 0, out, W, iters, E, cs, hot, active);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    const int blocks = 2048;
    double *out; CK(hipMalloc(&out, sizeof(double) * 256 * blocks));
    const uint32_t iters = 256, W = 4096; // 32 KiB: five workgroups per CU, as k_em_tile
    printf("%-34s %10s %12s %s\n", "window / copies / entries / lanes", "ms", "clk/wave-op", "(lanes/clk/CU)");
    for (uint32_t E : {150u, 250u})
        for (uint32_t hot : {0u, 20u})
            for (uint32_t active : {64u, 57u})
                for (uint32_t cs = 0; cs <= 4; ++cs) {
                    if ((E << cs) > W) continue;
                    const float ms = run<false>(out, W, iters, E, cs, hot, active, blocks);
                    const double waveops = (double)blocks * 4 * iters * kOps;
                    // clocks of one CU's LDS per wavefront-instruction: 256 CUs work in parallel
                    printf("E=%3u copies=%2u %-8s lanes=%2u   %10.3f %12.1f (%.2f)\n", E, 1u << cs, hot ? "20 hot" : "uniform", active, ms,
                           ms * 1e-3 * 2.4e9 / (waveops / 256.0), waveops * active / (ms * 1e-3) / 256 / 2.4e9);
                }
    const float mr = run<true>(out, W, iters, 150, 3, 0, 64, blocks);
    printf("b64 reads, E=150 x 8 copies uniform: %.3f ms, %.1f clk/wave-op\n", mr, mr * 1e-3 * 2.4e9 / ((double)blocks * 4 * iters * kOps / 256.0));
    return 0;
}
