// oem_coverage_device.hip -- the coverage model on the device.
//
// Same arithmetic, in f64, as the host restatement in oem_builder.cpp (which follows
// TranscriptInfo::add_interval, src/util/oarfish_types.rs:496-538; logistic_prob,
// src/util/logistic_probability.rs:7-79; binomial_continuous_prob, src/util/binomial_probability.rs:7-224;
// normalize_read_probs, src/util/normalize_probability.rs:5-74):
//
//   k_cov_bins       one thread per alignment: overlap fraction of every coverage bin it spans,
//                    added with f64 atomics (the host adds in store order: sums agree to ~1e-16)
//   k_cov_bin_probs  one thread per transcript: min coverage, f32 counts, logistic or binomial
//                    bin probabilities (sequential over the transcript's bins, as on the host)
//   k_cov_reads      one thread per read: per-alignment coverage probability, normalised per read
//
// The host version manages 8 M alignments/s on one core; this one is bound by the upload of the
// alignment coordinates.
#include "oem_internal.h"

namespace oem {

namespace {

constexpr int kCT = 256;
enum : uint32_t { kErrInterval = 1, kErrOlfrac = 2, kErrNoBins = 4, kErrDegenerate = 8, kErrNonFinite = 16 };

__global__ __launch_bounds__(kCT) void k_cov_bin_counts(const uint64_t *__restrict__ txp_len, uint32_t n_txps,
                                                        uint32_t bin_width, uint32_t *__restrict__ n_bins)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_txps) return;
    n_bins[t] = (uint32_t)ceil((double)txp_len[t] / (double)bin_width); // with_len_and_bin_width (:460-468)
}

// exclusive prefix sum of n_bins (one workgroup; T <= 2^32 but this is O(T / 1024) per thread)
__global__ __launch_bounds__(1024) void k_cov_bin_offsets(const uint32_t *__restrict__ n_bins, uint32_t n_txps,
                                                          unsigned long long *__restrict__ off /* [T + 1] */)
{
    __shared__ unsigned long long part[1024];
    const uint32_t per = (n_txps + blockDim.x - 1) / blockDim.x;
    const uint32_t b = threadIdx.x * per, e = min(n_txps, b + per);
    unsigned long long s = 0;
    for (uint32_t i = b; i < e; ++i) s += n_bins[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long acc = 0;
        for (uint32_t i = 0; i < blockDim.x; ++i) { const unsigned long long v = part[i]; part[i] = acc; acc += v; }
        off[n_txps] = acc;
    }
    __syncthreads();
    s = part[threadIdx.x];
    for (uint32_t i = b; i < e; ++i) { off[i] = s; s += n_bins[i]; }
}

__global__ __launch_bounds__(kCT) void k_cov_bins(const uint32_t *__restrict__ tid, const uint32_t *__restrict__ aln_start,
                                                  const uint32_t *__restrict__ aln_end, const uint64_t *__restrict__ txp_len,
                                                  const uint32_t *__restrict__ n_bins,
                                                  const unsigned long long *__restrict__ off, uint64_t nnz,
                                                  double *__restrict__ bins, uint32_t *__restrict__ total_weight,
                                                  uint32_t *err)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nnz) return;
    const uint32_t t = tid[j];
    const uint32_t num_intervals = n_bins[t];
    const double nf = (double)num_intervals, tlen_f = (double)txp_len[t];
    const double bw = round(tlen_f / nf);                                          // :501
    uint32_t start = aln_start[j], stop = aln_end[j];
    start = min(start, stop);                                                      // :502
    stop = max(start, stop);                                                       // :503
    const uint64_t start_bin = (uint64_t)floor(((double)start / tlen_f) * nf);     // :504
    const uint64_t end_bin = (uint64_t)floor(((double)stop / tlen_f) * nf);        // :505
    if (start_bin > end_bin || end_bin > num_intervals) { atomicOr(err, kErrInterval); return; }
    double *tb = bins + off[t];
    for (uint64_t bi = start_bin; bi < end_bin; ++bi) {                            // :515-536
        const double bidxf = (double)bi;
        const uint32_t cbs = (uint32_t)(bidxf * bw);
        const uint32_t cbe = (uint32_t)fmin((bidxf + 1.0) * bw, tlen_f);
        const uint32_t olap = start <= cbe ? min(stop, cbe) - max(start, cbs) : 0u; // :507-513 (u32)
        const double olfrac = (double)olap / (double)(uint32_t)(cbe - cbs);
        if (olfrac > 1.0 + 2.220446049250313e-16) atomicOr(err, kErrOlfrac);       // :524-535: the reference panics
        unsafeAtomicAdd(&tb[bi], olfrac);
    }
    atomicAdd(&total_weight[t], 1u);                                               // :537 (weight 1.0, :727)
}

__device__ double binomial_bins(const double *tb, uint32_t n, float bwf, float lenf32, double *prob, uint32_t *err)
{
    // binomial_continuous_prob + binomial_probability (binomial_probability.rs:7-224); tb already holds
    // bins + min_cov.  Two sweeps over the bins recompute the f32 counts rather than store them.
    const double kZero = 1e-20, kMaxScale = 709.0;
    float count_sum = 0.0f, max_val = 0.0f;
    double distinct_rate = 0.0;
    for (uint32_t i = 0; i < n; ++i) {
        const float c = (float)tb[i];
        const float len = fminf(((float)i + 1.0f) * bwf, lenf32) - (float)i * bwf;
        count_sum += c;                                                            // :14
        max_val = i == 0 ? c : fmaxf(max_val, c);                                  // :50
        distinct_rate += (double)c / (double)len;                                  // :184-188
    }
    if (count_sum == 0.0f || distinct_rate == 0.0) {                               // :19-25
        for (uint32_t i = 0; i < n; ++i) prob[i] = 0.0;
        return 0.0;
    }
    float sum_vec = 0.0f;
    for (uint32_t i = 0; i < n; ++i) {                                             // :61-72
        const float c = (float)tb[i];
        sum_vec += c == max_val ? (float)kMaxScale : (float)(((double)c * kMaxScale) / (double)max_val);
    }
    const double ln1 = lgamma((double)sum_vec + 1.0);                              // :75
    double total = 0.0;
    for (uint32_t i = 0; i < n; ++i) {
        const float c = (float)tb[i];
        const float len = fminf(((float)i + 1.0f) * bwf, lenf32) - (float)i * bwf;
        const float m = c == max_val ? (float)kMaxScale : (float)(((double)c * kMaxScale) / (double)max_val);
        const double p = (c == 0.0f || len == 0.0f) ? 0.0 : (double)c / ((double)len * distinct_rate); // :27-43
        const double denom = lgamma((double)m + 1.0) + lgamma((double)(sum_vec - m) + 1.0);              // :76-79
        const double num2 = (p > kZero ? log(p) : log(kZero)) * (double)m;                               // :82
        const double q = 1.0 - p;
        const double num3 = (q > kZero ? log(q) : log(kZero)) * (double)(sum_vec - m);                   // :89
        const double res = exp(ln1 - denom + num2 + num3);                                               // :101
        if (isnan(num2) || isinf(num2) || isnan(num3) || isinf(num3) || isnan(res) || isinf(res))
            atomicOr(err, kErrNonFinite);                                          // the reference panics (:83-112)
        prob[i] = res;
        total += res;                                                              // :120
    }
    for (uint32_t i = 0; i < n; ++i) {
        prob[i] /= total;                                                          // :124
        if (isnan(prob[i])) atomicOr(err, kErrNonFinite);
    }
    return total;
}

__global__ __launch_bounds__(kCT) void k_cov_bin_probs(const uint64_t *__restrict__ txp_len, const uint32_t *__restrict__ n_bins,
                                                       const unsigned long long *__restrict__ off,
                                                       const uint32_t *__restrict__ total_weight, uint32_t n_txps,
                                                       int model, double growth_rate, double *__restrict__ bins,
                                                       double *__restrict__ prob, uint32_t *err)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_txps) return;
    const uint32_t n = n_bins[t];
    if (n == 0) { atomicOr(err, kErrNoBins); return; }                             // assert (logistic_probability.rs:54)
    double *tb = bins + off[t], *tp = prob + off[t];
    const double lenf = (double)txp_len[t];
    const double min_cov = (double)total_weight[t] / 100.;                         // :55 / binomial :180
    for (uint32_t i = 0; i < n; ++i) tb[i] += min_cov;                             // :56
    // get_normalized_counts_and_lengths (oarfish_types.rs:471-493): f32 counts and bin widths
    const float bwf = (float)round(lenf / (double)n), lenf32 = (float)lenf;
    for (uint32_t i = 0; i < n; ++i) {
        const float bs = (float)i * bwf, be = fminf(((float)i + 1.0f) * bwf, lenf32);
        if (!(be > bs)) { atomicOr(err, kErrDegenerate); return; }                 // assert (:490)
    }
    if (model == 1) {
        binomial_bins(tb, n, bwf, lenf32, tp, err);
        return;
    }
    double count_sum = 0.0;                                                        // logstic_function (:13-39)
    for (uint32_t i = 0; i < n; ++i) count_sum += (double)(float)tb[i];
    if (count_sum <= 1e-8) {                                                       // :21-23
        for (uint32_t i = 0; i < n; ++i) tp[i] = 0.0;
        return;
    }
    const double expected = count_sum / (double)n;                                 // :27
    for (uint32_t i = 0; i < n; ++i) {
        const double diff = (expected - (double)(float)tb[i]) / expected;          // :32
        double r = 1.0 / (1.0 + exp(-growth_rate * diff));                         // logistic (:7-10)
        r = r < 1e-8 ? 1e-8 : (r > 0.99999 ? 0.99999 : r);
        tp[i] = r;
    }
}

__global__ __launch_bounds__(kCT) void k_cov_reads(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ tid,
                                                   const uint32_t *__restrict__ aln_start, const uint32_t *__restrict__ aln_end,
                                                   const uint64_t *__restrict__ txp_len, const uint32_t *__restrict__ n_bins,
                                                   const unsigned long long *__restrict__ off,
                                                   const double *__restrict__ prob, uint64_t n_reads, double bin_length,
                                                   double *__restrict__ out, uint32_t *err)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    double nprob_sum = 0.0;                                                        // normalize_probability.rs:5-74
    for (uint32_t j = row_ptr[r]; j < row_ptr[r + 1]; ++j) {
        const uint32_t t = tid[j];
        const double *tp = prob + off[t];
        const double start_aln = (double)aln_start[j], end_aln = (double)aln_end[j], tlen = (double)txp_len[t];
        const uint64_t start_bin = (uint64_t)(start_aln / bin_length);             // :25
        uint64_t end_bin = (uint64_t)(end_aln / bin_length);                       // :26-27
        if (end_bin > (uint64_t)n_bins[t] - 1) end_bin = (uint64_t)n_bins[t] - 1;
        double total_weight = 0.0, cov_prob = 0.0;
        if (start_bin == end_bin) {                                                // :33-35
            const double w = (end_aln - start_aln) / bin_length;
            total_weight = w;
            cov_prob = w * tp[start_bin];
        } else {
            for (uint64_t i = start_bin; i < end_bin; ++i) {                       // :37-46
                const double w = i == start_bin ? (fmin(bin_length * (double)i + bin_length, tlen) - start_aln) / bin_length : 1.0;
                total_weight += w;
                cov_prob += w * tp[i];
            }
        }
        const double expected = cov_prob / total_weight;                           // :58
        if (isnan(cov_prob) || isinf(cov_prob)) atomicOr(err, kErrNonFinite); // :49-57 (a 0/0 expected value is not an error there)
        out[j] = expected;
        nprob_sum += expected;
    }
    const double denom = nprob_sum > 0.0 ? nprob_sum : 1.0;                        // :62
    for (uint32_t j = row_ptr[r]; j < row_ptr[r + 1]; ++j) out[j] /= denom;        // :65-69
}

struct Bufs {
    std::vector<void *> p;
    template <typename T> int get(T **q, size_t n)
    {
        *q = nullptr;
        hipError_t e = hipMalloc((void **)q, (n ? n : 1) * sizeof(T));
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? OEM_ERR_OOM : OEM_ERR_HIP, "coverage model: %s", hipGetErrorString(e));
        p.push_back(*q);
        return OEM_OK;
    }
    ~Bufs() { for (void *q : p) hipFree(q); }
};

} // namespace
} // namespace oem

using namespace oem;

extern "C" int oem_coverage_probs_device(const uint64_t *row_ptr, const uint32_t *tid, const uint32_t *aln_start,
                                         const uint32_t *aln_end, const uint64_t *txp_len, uint64_t n_reads,
                                         uint64_t nnz, uint32_t n_txps, uint32_t bin_width, int model,
                                         double growth_rate, int device, double *out_cov_prob)
{
    OEM_API_BEGIN
    if (!row_ptr || !txp_len || (nnz && (!tid || !aln_start || !aln_end || !out_cov_prob)))
        return fail(OEM_ERR_ARG, "oem_coverage_probs_device: NULL argument");
    if (bin_width == 0)
        return fail(OEM_ERR_ARG, "coverage model with 0 bin width is not implemented (logistic_probability.rs:59, binomial_probability.rs:192)");
    if (model != 0 && model != 1) return fail(OEM_ERR_ARG, "oem_coverage_probs_device: model must be 0 (logistic) or 1 (binomial)");
    if (n_txps == 0) return fail(OEM_ERR_ARG, "oem_coverage_probs_device: n_txps is 0");
    if (nnz >= (1ull << 32)) return fail(OEM_ERR_ARG, "oem_coverage_probs_device: needs nnz < 2^32");
    if (row_ptr[0] != 0 || row_ptr[n_reads] != nnz) return fail(OEM_ERR_ARG, "oem_coverage_probs_device: row_ptr must span [0, nnz]");
    for (uint64_t j = 0; j < nnz; ++j)
        if (tid[j] >= n_txps) return fail(OEM_ERR_ARG, "tid[%llu]=%u is not below n_txps=%u", (unsigned long long)j, tid[j], n_txps);
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev)
        return fail(OEM_ERR_NO_DEVICE, "oem_coverage_probs_device: no HIP device %d", device);
    OEM_HIP(hipSetDevice(device));
    if (nnz == 0) return OEM_OK;

    Bufs bufs;
    std::vector<uint32_t> rp32(n_reads + 1);
    for (uint64_t i = 0; i <= n_reads; ++i) rp32[i] = (uint32_t)row_ptr[i];
    uint32_t *d_rp, *d_tid, *d_start, *d_end, *d_nbins, *d_tw, *d_err;
    uint64_t *d_len;
    unsigned long long *d_off;
    double *d_out;
    OEM_TRY(bufs.get(&d_rp, n_reads + 1));
    OEM_TRY(bufs.get(&d_tid, nnz));
    OEM_TRY(bufs.get(&d_start, nnz));
    OEM_TRY(bufs.get(&d_end, nnz));
    OEM_TRY(bufs.get(&d_len, n_txps));
    OEM_TRY(bufs.get(&d_nbins, n_txps));
    OEM_TRY(bufs.get(&d_off, (size_t)n_txps + 1));
    OEM_TRY(bufs.get(&d_tw, n_txps));
    OEM_TRY(bufs.get(&d_err, 1));
    OEM_TRY(bufs.get(&d_out, nnz));
    OEM_HIP(hipMemcpy(d_rp, rp32.data(), sizeof(uint32_t) * (n_reads + 1), hipMemcpyHostToDevice));
    OEM_HIP(hipMemcpy(d_tid, tid, sizeof(uint32_t) * nnz, hipMemcpyHostToDevice));
    OEM_HIP(hipMemcpy(d_start, aln_start, sizeof(uint32_t) * nnz, hipMemcpyHostToDevice));
    OEM_HIP(hipMemcpy(d_end, aln_end, sizeof(uint32_t) * nnz, hipMemcpyHostToDevice));
    OEM_HIP(hipMemcpy(d_len, txp_len, sizeof(uint64_t) * n_txps, hipMemcpyHostToDevice));
    OEM_HIP(hipMemset(d_tw, 0, sizeof(uint32_t) * n_txps));
    OEM_HIP(hipMemset(d_err, 0, sizeof(uint32_t)));

    const uint32_t tg = (n_txps + kCT - 1) / kCT;
    hipLaunchKernelGGL(k_cov_bin_counts, dim3(tg), dim3(kCT), 0, 0, d_len, n_txps, bin_width, d_nbins);
    hipLaunchKernelGGL(k_cov_bin_offsets, dim3(1), dim3(1024), 0, 0, d_nbins, n_txps, d_off);
    unsigned long long total_bins = 0;
    OEM_HIP(hipMemcpy(&total_bins, d_off + n_txps, sizeof(total_bins), hipMemcpyDeviceToHost));
    double *d_bins, *d_prob;
    OEM_TRY(bufs.get(&d_bins, total_bins));
    OEM_TRY(bufs.get(&d_prob, total_bins));
    OEM_HIP(hipMemset(d_bins, 0, sizeof(double) * (total_bins ? total_bins : 1)));
    hipLaunchKernelGGL(k_cov_bins, dim3((uint32_t)((nnz + kCT - 1) / kCT)), dim3(kCT), 0, 0, d_tid, d_start, d_end, d_len,
                       d_nbins, d_off, nnz, d_bins, d_tw, d_err);
    hipLaunchKernelGGL(k_cov_bin_probs, dim3(tg), dim3(kCT), 0, 0, d_len, d_nbins, d_off, d_tw, n_txps, model, growth_rate,
                       d_bins, d_prob, d_err);
    hipLaunchKernelGGL(k_cov_reads, dim3((uint32_t)((n_reads + kCT - 1) / kCT)), dim3(kCT), 0, 0, d_rp, d_tid, d_start, d_end,
                       d_len, d_nbins, d_off, d_prob, n_reads, (double)bin_width, d_out, d_err);
    OEM_HIP(hipGetLastError());
    uint32_t h_err = 0;
    OEM_HIP(hipMemcpy(&h_err, d_err, sizeof(h_err), hipMemcpyDeviceToHost));
    if (h_err & kErrInterval) return fail(OEM_ERR_STATE, "add_interval: an alignment lies outside its transcript");
    if (h_err & kErrOlfrac) return fail(OEM_ERR_STATE, "coverage computation error: overlap fraction above 1");
    if (h_err & kErrNoBins) return fail(OEM_ERR_STATE, "a transcript has no coverage bins");
    if (h_err & kErrDegenerate) return fail(OEM_ERR_STATE, "degenerate coverage bin (assert, oarfish_types.rs:490)");
    if (h_err & kErrNonFinite) return fail(OEM_ERR_STATE, "coverage model: non-finite probability");
    OEM_HIP(hipMemcpy(out_cov_prob, d_out, sizeof(double) * nnz, hipMemcpyDeviceToHost));
    return OEM_OK;
    OEM_API_END("oem_coverage_probs_device")
}
