// Throughput of the host-side steps right before the EM, through the C ABI:
// oem_builder_add_group (filters + as_prob) and the two coverage models.  No GPU needed.
// build: g++ -O2 -std=c++17 -I include scripts/native/builder_bench.cpp -L oarfish_amd -loarfish_em -Wl,-rpath,$PWD/oarfish_amd -o /tmp/builder_bench
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "oarfish_em.h"

int main(int argc, char **argv)
{
    const uint32_t R = argc > 1 ? atoi(argv[1]) : 1000000, T = argc > 2 ? atoi(argv[2]) : 60000;
    std::mt19937_64 rng(7);
    std::vector<uint64_t> len(T);
    for (auto &l : len) l = 600 + rng() % 4000;
    oem_filters F{0, 0, 0.95f, 0.5f, 50, 0, 5.0f, 0};
    F.five_prime_clip = 0xffffffffu;
    F.three_prime_clip = INT64_MAX;
    oem_builder *b = nullptr;
    if (oem_builder_create(&F, len.data(), T, &b)) { printf("create: %s\n", oem_last_error()); return 1; }
    std::vector<oem_aln_record> recs;
    std::vector<std::vector<oem_aln_record>> groups(R);
    uint64_t n_rec = 0;
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t k = 1 + rng() % 15, t0 = rng() % T;
        for (uint32_t j = 0; j < k; ++j) {
            oem_aln_record a{};
            a.ref_id = (t0 + j) % T;
            const uint32_t L = (uint32_t)len[a.ref_id];
            a.aln_start = rng() % (L - 500);
            a.aln_end = a.aln_start + 450 + rng() % 50;
            a.aln_span = a.aln_end - a.aln_start;
            a.score = 2000 - (j ? (int64_t)(rng() % 60) : 0);
            a.seq_len = 520;
            a.flags = OEM_REC_HAS_SCORE | ((rng() & 1) ? OEM_REC_REVERSE : 0);
            groups[r].push_back(a);
        }
        n_rec += k;
    }
    auto t0 = std::chrono::steady_clock::now();
    uint32_t kept = 0;
    for (uint32_t r = 0; r < R; ++r) oem_builder_add_group(b, groups[r].data(), (uint32_t)groups[r].size(), &kept);
    auto t1 = std::chrono::steady_clock::now();
    uint64_t nr = 0, nnz = 0;
    oem_builder_dims(b, &nr, &nnz);
    const double s_add = std::chrono::duration<double>(t1 - t0).count();
    printf("add_group: %u reads, %llu records -> %llu kept alignments in %.3f s = %.1f M records/s\n", R,
           (unsigned long long)n_rec, (unsigned long long)nnz, s_add, n_rec / s_add * 1e-6);
    std::vector<double> cov(nnz);
    t0 = std::chrono::steady_clock::now();
    if (oem_builder_coverage_probs(b, 100, 2.0, cov.data())) { printf("coverage: %s\n", oem_last_error()); return 1; }
    t1 = std::chrono::steady_clock::now();
    const double s_log = std::chrono::duration<double>(t1 - t0).count();
    t0 = std::chrono::steady_clock::now();
    if (oem_builder_coverage_probs_binomial(b, 100, cov.data())) { printf("binomial: %s\n", oem_last_error()); return 1; }
    t1 = std::chrono::steady_clock::now();
    const double s_bin = std::chrono::duration<double>(t1 - t0).count();
    printf("coverage_probs (logistic): %.3f s = %.1f M alignments/s; binomial: %.3f s = %.1f M alignments/s\n", s_log,
           nnz / s_log * 1e-6, s_bin, nnz / s_bin * 1e-6);
    oem_builder_destroy(b);
    return 0;
}
