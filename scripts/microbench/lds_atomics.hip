// LDS throughput by operation on gfx950, with the addresses precomputed in registers so that the loop
// issues LDS instructions and nothing else.  (The LDS section of atomics.hip computes a hash and an integer
// modulo per operation and is bound by those ~30 VALU instructions, not by the LDS: every operation type,
// plain reads included, lands at 2.2-2.8 lanes/clk/CU there.)
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomics.hip -o lds_atomics
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

constexpr int kOps = 16; // addresses per lane, all issued per loop trip

// MODE: 0 f64 atomic add, 1 u64 atomic add, 2 u32 atomic add, 3 b64 read, 4 b64 write, 5 f64 add with return,
//       6 u64 atomic add of an f64 value converted to 2^-44 fixed point in the loop (what a fixed-point count
//         window would cost: conversion included), 7 f64 atomic add of the same varying value
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, uint32_t W, uint32_t iters, int pat)
{
    extern __shared__ double lds[];
    for (uint32_t i = threadIdx.x; i < W; i += blockDim.x) lds[i] = 0.0;
    __syncthreads();
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t off[kOps];
#pragma unroll
    for (int j = 0; j < kOps; ++j) {
        uint32_t t;
        if (pat == 0) t = hash32(gid * 977u + j) % W;                                   // uniform random
        else if (pat == 1) t = (threadIdx.x + j * 67) % W;                              // conflict-free (consecutive lanes)
        else if (pat == 2) t = (hash32(j * 131u + (threadIdx.x >> 6)) % 16) * 4 + (threadIdx.x & 3); // 16 hot entries x 4 copies
        else if (pat == 3) t = (hash32(j * 131u + (threadIdx.x >> 6)) % 16) * 4;        // 16 hot entries, one copy (64 lanes, one address)
        else t = (hash32(gid * 977u + (j >> 2)) % (W / 4)) * 4 + ((threadIdx.x + j) & 3); // [entry][4 slots], lanes rotate over the slots (k_em_tile_e)
        off[j] = t * 8;
    }
    double acc = 0;
    double vv = 1.0 / (double)(1 + (gid & 15));
    char *base = reinterpret_cast<char *>(lds);
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < kOps; ++j) {
            double *p = reinterpret_cast<double *>(base + off[j]);
            if (MODE == 0) __hip_atomic_fetch_add(p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 1) __hip_atomic_fetch_add((unsigned long long *)p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 2) __hip_atomic_fetch_add((uint32_t *)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 3) acc += *(volatile double *)p;
            else if (MODE == 4) *(volatile double *)p = (double)it;
            else if (MODE == 5) acc += __hip_atomic_fetch_add(p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 6) {
                const double v = vv * (double)(j + 1);
                __hip_atomic_fetch_add((unsigned long long *)p, (unsigned long long)(v * 0x1p44), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                const double v = vv * (double)(j + 1);
                __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        vv += 0.125;
    }
    __syncthreads();
    double s = acc;
    for (uint32_t i = threadIdx.x; i < W; i += blockDim.x) s += lds[i];
    if (s == -1.0) out[gid] = s;
}

template <int MODE> float run(double *out, uint32_t W, uint32_t iters, int pat, int blocks)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), W * 8, 0, out, W, iters, pat);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    const int blocks = 2048;
    double *out; CK(hipMalloc(&out, sizeof(double) * 256 * blocks));
    const uint32_t iters = 256, W = 2048;
    const char *mn[] = {"f64 atomic add", "u64 atomic add", "u32 atomic add", "b64 read", "b64 write", "f64 add, returning",
                        "u64 add of cvt(f64)", "f64 add, varying"};
    const char *pn[] = {"uniform random", "conflict-free", "16 hot x 4 copies", "16 hot, 1 copy", "[entry][4], rotated"};
    for (int pat = 0; pat < 5; ++pat)
        for (int m = 0; m < 8; ++m) {
            float ms = 0;
            switch (m) {
            case 0: ms = run<0>(out, W, iters, pat, blocks); break; case 1: ms = run<1>(out, W, iters, pat, blocks); break;
            case 2: ms = run<2>(out, W, iters, pat, blocks); break; case 3: ms = run<3>(out, W, iters, pat, blocks); break;
            case 4: ms = run<4>(out, W, iters, pat, blocks); break; case 5: ms = run<5>(out, W, iters, pat, blocks); break;
            case 6: ms = run<6>(out, W, iters, pat, blocks); break; default: ms = run<7>(out, W, iters, pat, blocks); break;
            }
            const double n = (double)blocks * 256 * iters * kOps;
            printf("%-18s %-20s %8.3f ms  %9.1f Gop/s  (%.2f lanes/clk/CU @2.4GHz)\n", pn[pat], mn[m], ms, n / ms * 1e-6, n / (ms * 1e-3) / 256 / 2.4e9);
        }
    return 0;
}
