// Microbenchmarks that decide the layout of the E/M kernel on gfx950:
//   A. global f64 atomic add throughput (agent scope vs workgroup scope on XCD-private copies)
//   B. hot-address serialisation
//   C. LDS f64 atomic add throughput
//   D. streaming read bandwidth at 8 and 16 B/lane
//   E. random 8-byte gathers from an L2-resident table
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomics.hip -o atomics
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xf;
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

enum Pattern { UNIFORM = 0, HOT1 = 1, HOT64 = 2, SEQ = 3, SKEW = 4 };

__device__ __forceinline__ uint32_t make_idx(uint64_t i, uint32_t T, int pat)
{
    switch (pat) {
    case UNIFORM: return hash32((uint32_t)i) % T;
    case HOT1: return 0;
    case HOT64: return hash32((uint32_t)i) & 63;
    case SEQ: return (uint32_t)(i % T);
    default: {
        // skew: u^3 concentrates mass on low ids (top id ~1.7 % at T=200k)
        float u = (hash32((uint32_t)i) >> 8) * (1.0f / 16777216.0f);
        return (uint32_t)(u * u * u * (float)T) % T;
    }
    }
}

template <int SCOPE> // 0 = agent, 1 = workgroup on XCD-private copy
__global__ __launch_bounds__(256) void k_global_atomics(double *cnt, uint32_t T, uint64_t n, int pat)
{
    double *base = cnt;
    if (SCOPE == 1) base = cnt + (size_t)xcc_id() * T;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t t = make_idx(i, T, pat);
        if (SCOPE == 0) __hip_atomic_fetch_add(&base[t], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(&base[t], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

__global__ __launch_bounds__(256) void k_lds_atomics(double *out, uint32_t W, uint32_t iters, int pat)
{
    extern __shared__ double lds[];
    for (uint32_t i = threadIdx.x; i < W; i += blockDim.x) lds[i] = 0.0;
    __syncthreads();
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = 0; k < iters; ++k) {
        uint32_t t;
        if (pat == UNIFORM) t = hash32(gid * 977u + k) % W;
        else if (pat == HOT1) t = 0;
        else if (pat == SEQ) t = (threadIdx.x + k * 7) % W;
        else { float u = (hash32(gid * 977u + k) >> 8) * (1.0f / 16777216.0f); t = (uint32_t)(u * u * u * (float)W) % W; }
        __hip_atomic_fetch_add(&lds[t], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    double s = 0;
    for (uint32_t i = threadIdx.x; i < W; i += blockDim.x) s += lds[i];
    if (s == -1.0) out[gid] = s;
    if (threadIdx.x == 0) atomicAdd(&out[0], s);
}

template <typename V>
__global__ __launch_bounds__(256) void k_stream(const V *in, uint64_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        V v = __builtin_nontemporal_load(&in[i]);
        acc ^= ((const uint32_t *)&v)[0] ^ ((const uint32_t *)&v)[sizeof(V) / 4 - 1];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_gather(const double *theta, uint32_t T, uint64_t n, int pat, double *sink)
{
    double acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        acc += theta[make_idx(i, T, pat)];
    if (acc == -1.0) sink[0] = acc;
}

static float time_it(hipEvent_t e0, hipEvent_t e1)
{
    float ms;
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main()
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint32_t T = 200000;
    const uint64_t N = 64ull << 20;
    double *cnt;
    CK(hipMalloc(&cnt, sizeof(double) * T * 8));
    const char *pn[] = {"uniform", "hot1", "hot64", "seq", "skew"};
    const int grid = 256 * 8;

    printf("== A/B. global f64 atomic add, N=%llu ops over T=%u\n", (unsigned long long)N, T);
    for (int scope = 0; scope < 2; ++scope) {
        for (int pat = 0; pat < 5; ++pat) {
            const uint64_t n = (pat == HOT1) ? (N >> 4) : (pat == HOT64 ? N >> 2 : N);
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(cnt, 0, sizeof(double) * T * 8));
                CK(hipEventRecord(e0));
                if (scope == 0) hipLaunchKernelGGL(k_global_atomics<0>, dim3(grid), dim3(256), 0, 0, cnt, T, n, pat);
                else hipLaunchKernelGGL(k_global_atomics<1>, dim3(grid), dim3(256), 0, 0, cnt, T, n, pat);
                CK(hipEventRecord(e1));
                float ms = time_it(e0, e1);
                if (ms < best) best = ms;
            }
            std::vector<double> h((size_t)T * 8);
            CK(hipMemcpy(h.data(), cnt, sizeof(double) * T * 8, hipMemcpyDeviceToHost));
            double sum = 0;
            for (double v : h) sum += v;
            printf("  scope=%-9s %-8s %8.3f ms  %7.2f Gatom/s  sum_ok=%d\n", scope ? "workgroup" : "agent", pn[pat], best,
                   n / best * 1e-6, sum == (double)n);
        }
    }

    printf("== C. LDS f64 atomic add (256 thr/block, 2048 blocks, 4096 adds/thread)\n");
    {
        double *out;
        CK(hipMalloc(&out, sizeof(double) * 256 * 2048));
        const uint32_t iters = 4096;
        for (uint32_t W : {2048u, 8192u}) {
            for (int pat : {UNIFORM, HOT1, SEQ, SKEW}) {
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemset(out, 0, 8));
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(k_lds_atomics, dim3(2048), dim3(256), W * 8, 0, out, W, iters, pat);
                    CK(hipEventRecord(e1));
                    float ms = time_it(e0, e1);
                    if (ms < best) best = ms;
                }
                double tot;
                CK(hipMemcpy(&tot, out, 8, hipMemcpyDeviceToHost));
                const double n = 2048.0 * 256 * iters;
                printf("  W=%-5u %-8s %8.3f ms  %8.2f Gatom/s  (%.2f atom/clk/CU @2.4GHz) sum_ok=%d\n", W, pn[pat], best,
                       n / best * 1e-6, n / (best * 1e-3) / 256 / 2.4e9, tot == n);
            }
        }
        CK(hipFree(out));
    }

    printf("== D. streaming read, 1 GiB\n");
    {
        const uint64_t bytes = 1ull << 30;
        void *buf;
        uint32_t *sink;
        CK(hipMalloc(&buf, bytes));
        CK(hipMalloc(&sink, 4));
        CK(hipMemset(buf, 1, bytes));
        for (int w = 0; w < 3; ++w) {
            for (int g : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(e0));
                    if (w == 0) hipLaunchKernelGGL(k_stream<uint32_t>, dim3(g), dim3(256), 0, 0, (const uint32_t *)buf, bytes / 4, sink);
                    else if (w == 1) hipLaunchKernelGGL(k_stream<u32x2>, dim3(g), dim3(256), 0, 0, (const u32x2 *)buf, bytes / 8, sink);
                    else hipLaunchKernelGGL(k_stream<u32x4>, dim3(g), dim3(256), 0, 0, (const u32x4 *)buf, bytes / 16, sink);
                    CK(hipEventRecord(e1));
                    float ms = time_it(e0, e1);
                    if (ms < best) best = ms;
                }
                printf("  %2d B/lane grid=%5d  %7.3f ms  %7.1f GB/s\n", 4 << w, g, best, bytes / best * 1e-6);
            }
        }
        CK(hipFree(buf));
        CK(hipFree(sink));
    }

    printf("== E. random 8-byte gathers from a %u-entry f64 table, N=%llu\n", T, (unsigned long long)N);
    {
        double *sink;
        CK(hipMalloc(&sink, 8));
        for (int pat : {UNIFORM, SEQ, SKEW}) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, 0, cnt, T, N, pat, sink);
                CK(hipEventRecord(e1));
                float ms = time_it(e0, e1);
                if (ms < best) best = ms;
            }
            printf("  %-8s %8.3f ms  %7.2f Ggather/s\n", pn[pat], best, N / best * 1e-6);
        }
        CK(hipFree(sink));
    }
    CK(hipFree(cnt));
    return 0;
}
