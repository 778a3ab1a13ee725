/*
 * oem_oracle.c -- CPU restatement of oarfish's EM / bootstrap hot path.
 * TEST INFRASTRUCTURE ONLY (see oem_oracle.h).  Parity unpinned by the
 * reference: pinned by closed forms, invariants and oracle/oracle_np.py.
 *
 * Citations are to the reference checkout (COMBINE-lab/oarfish v0.10.3).
 * The KDE factor (em.rs:109,173-178) is the constant 1.0 here: the model lives
 * in the un-vendored, un-pinned `kders` crate and is only reachable through
 * the hidden --use-kde flag (SURVEY.md section 8a note 4).
 */
#include "oem_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* em.rs:87-133  m_step                                                      */
/* ------------------------------------------------------------------------ */
static inline void m_step_row(const oracle_store *s, uint64_t row, double scale,
                              const double *prev, double *curr)
{
    const uint64_t b = s->row_ptr[row], e = s->row_ptr[row + 1];
    const int model_coverage = s->cov_prob != NULL; /* em.rs:108 */
    double denom = 0.0;                             /* em.rs:98 */
    for (uint64_t j = b; j < e; ++j) {              /* em.rs:99-112 */
        const uint32_t target_id = s->tid[j];
        const double prob = (double)s->as_prob[j];  /* em.rs:107 */
        const double cov = model_coverage ? s->cov_prob[j] : 1.0;
        const double dens = 1.0;                    /* em.rs:109 with kde_model=None */
        denom += prev[target_id] * prob * cov * dens; /* em.rs:111 */
    }
    if (denom > ORACLE_EM_DENOM_THRESH) {           /* em.rs:115 */
        for (uint64_t j = b; j < e; ++j) {          /* em.rs:119-130 */
            const uint32_t target_id = s->tid[j];
            const double prob = (double)s->as_prob[j];
            const double cov = model_coverage ? s->cov_prob[j] : 1.0;
            const double dens = 1.0;
            const double inc = (prev[target_id] * prob * cov * dens) / denom; /* em.rs:128 */
            curr[target_id] += scale * inc;         /* em.rs:129; scale==1 on the reference path */
        }
    }
}

void oracle_m_step(const oracle_store *s, const uint64_t *inds, uint64_t n_inds,
                   const uint32_t *row_w, const double *prev, double *curr)
{
    if (inds) {
        /* random_sampling_iter: visit rows in the order of the (sorted) index
         * list, repeats included (oarfish_types.rs:576-592). */
        for (uint64_t k = 0; k < n_inds; ++k)
            m_step_row(s, inds[k], 1.0, prev, curr);
    } else if (row_w) {
        for (uint64_t i = 0; i < s->n_reads; ++i)
            if (row_w[i])
                m_step_row(s, i, (double)row_w[i], prev, curr);
    } else {
        for (uint64_t i = 0; i < s->n_reads; ++i)   /* em.rs:97 */
            m_step_row(s, i, 1.0, prev, curr);
    }
}

/* ------------------------------------------------------------------------ */
/* em.rs:144-255  do_em                                                      */
/* ------------------------------------------------------------------------ */
int oracle_do_em(const oracle_store *s, const double *init, uint32_t max_iter,
                 double conv_thresh, uint32_t min_iter_gate, const uint64_t *inds,
                 uint64_t n_inds, const uint32_t *row_w, double *out_counts,
                 oracle_run_info *info)
{
    const uint32_t T = s->n_txps;
    /* em.rs:154: total_weight is the number of reads in the store, also for a
     * bootstrap replicate (do_bootstrap passes the same EMInfo, em.rs:287-289) */
    const double total_weight = (double)s->n_reads;
    double *prev = (double *)malloc(sizeof(double) * (T ? T : 1));
    double *curr = (double *)calloc(T ? T : 1, sizeof(double)); /* em.rs:158 */
    if (!prev || !curr) { free(prev); free(curr); return 1; }

    if (init) {
        memcpy(prev, init, sizeof(double) * T);     /* em.rs:160-162 */
    } else {
        const double avg = total_weight / (double)T; /* em.rs:165 */
        for (uint32_t i = 0; i < T; ++i) prev[i] = avg; /* em.rs:166 */
    }

    double rel_diff = 0.0;                          /* em.rs:169 */
    double last_rel = 0.0;
    uint32_t niter = 0;                             /* em.rs:170 */
    uint32_t n_passes = 0, converged = 0;

    while (niter < max_iter) {                      /* em.rs:181 */
        oracle_m_step(s, inds, n_inds, row_w, prev, curr); /* em.rs:183-190 */
        ++n_passes;
        for (uint32_t i = 0; i < T; ++i) {          /* em.rs:194-201 */
            if (prev[i] > ORACLE_MIN_READ_THRESH) {
                const double cc = curr[i], pc = prev[i];
                const double rd = (cc - pc) / pc;   /* signed */
                rel_diff = fmax(rel_diff, rd);      /* f64::max */
            }
        }
        { double *t = prev; prev = curr; curr = t; } /* em.rs:204 */
        memset(curr, 0, sizeof(double) * T);        /* em.rs:207 */
        last_rel = rel_diff;
        if (rel_diff < conv_thresh && niter > min_iter_gate) { /* em.rs:212 (gate 50) / :399 (gate 1) */
            converged = 1;
            break;
        }
        niter += 1;                                 /* em.rs:218 */
        rel_diff = 0.0;                             /* em.rs:234 */
    }

    for (uint32_t i = 0; i < T; ++i)                /* em.rs:238-242 */
        if (prev[i] < ORACLE_MIN_READ_THRESH) prev[i] = 0.0;
    oracle_m_step(s, inds, n_inds, row_w, prev, curr); /* em.rs:245-252 */
    ++n_passes;
    memcpy(out_counts, curr, sizeof(double) * T);   /* em.rs:254 */

    if (info) {
        info->niter = niter;
        info->n_passes = n_passes;
        info->converged = converged;
        info->rel_diff = last_rel;
    }
    free(prev);
    free(curr);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* em.rs:22-79 m_step_par + em.rs:320-447 em_par                             */
/* ------------------------------------------------------------------------ */
static inline void atomic_add_f64(double *addr, double v)
{
    /* atomic_float 1.1.0 AtomicF64::fetch_add is a compare-exchange loop on the
     * bit pattern (Cargo.toml:52; call site em.rs:74). */
    uint64_t *p = (uint64_t *)addr;
    uint64_t old = __atomic_load_n(p, __ATOMIC_RELAXED);
    for (;;) {
        double d;
        memcpy(&d, &old, 8);
        d += v;
        uint64_t nw;
        memcpy(&nw, &d, 8);
        if (__atomic_compare_exchange_n(p, &old, nw, 1, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED))
            break;
    }
}

static void m_step_par(const oracle_store *s, const double *prev, double *curr, int nthreads)
{
    const int model_coverage = s->cov_prob != NULL;
    const int64_t R = (int64_t)s->n_reads;
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 4096) num_threads(nthreads)
    for (int64_t i = 0; i < R; ++i) {               /* em.rs:33 par_iter */
        const uint64_t b = s->row_ptr[i], e = s->row_ptr[i + 1];
        double denom = 0.0;
        for (uint64_t j = b; j < e; ++j) {          /* em.rs:37-52 */
            const double prob = (double)s->as_prob[j];
            const double cov = model_coverage ? s->cov_prob[j] : 1.0;
            denom += prev[s->tid[j]] * prob * cov * 1.0;
        }
        if (denom > ORACLE_EM_DENOM_THRESH) {       /* em.rs:55 */
            for (uint64_t j = b; j < e; ++j) {      /* em.rs:59-75 */
                const double prob = (double)s->as_prob[j];
                const double cov = model_coverage ? s->cov_prob[j] : 1.0;
                const double inc = (prev[s->tid[j]] * prob * cov * 1.0) / denom;
                atomic_add_f64(&curr[s->tid[j]], inc); /* em.rs:74 */
            }
        }
    }
}

int oracle_em_par(const oracle_store *s, const double *init, uint32_t max_iter,
                  double conv_thresh, uint32_t min_iter_gate, int nthreads,
                  double *out_counts, oracle_run_info *info)
{
    const uint32_t T = s->n_txps;
    const double total_weight = (double)s->n_reads; /* em.rs:334 */
    double *prev = (double *)malloc(sizeof(double) * (T ? T : 1));
    double *curr = (double *)calloc(T ? T : 1, sizeof(double));
    if (!prev || !curr) { free(prev); free(curr); return 1; }
    if (nthreads < 1) nthreads = 1;
    if (init) memcpy(prev, init, sizeof(double) * T); /* em.rs:343-345 */
    else { const double avg = total_weight / (double)T; for (uint32_t i = 0; i < T; ++i) prev[i] = avg; }

    double rel_diff = 0.0, last_rel = 0.0;
    uint32_t niter = 0, n_passes = 0, converged = 0;
    while (niter < max_iter) {                      /* em.rs:366 */
        m_step_par(s, prev, curr, nthreads);        /* em.rs:368-375 */
        ++n_passes;
        for (uint32_t i = 0; i < T; ++i) {          /* em.rs:379-386, serial */
            if (prev[i] > ORACLE_MIN_READ_THRESH) {
                const double rd = (curr[i] - prev[i]) / prev[i];
                rel_diff = fmax(rel_diff, rd);
            }
        }
        { double *t = prev; prev = curr; curr = t; } /* em.rs:389 */
        memset(curr, 0, sizeof(double) * T);        /* em.rs:392-394 */
        last_rel = rel_diff;
        if (rel_diff < conv_thresh && niter > min_iter_gate) { converged = 1; break; } /* em.rs:399 */
        niter += 1;                                 /* em.rs:405 */
        rel_diff = 0.0;                             /* em.rs:421 */
    }
    for (uint32_t i = 0; i < T; ++i)                /* em.rs:425-429 */
        if (prev[i] < ORACLE_MIN_READ_THRESH) prev[i] = 0.0;
    m_step_par(s, prev, curr, nthreads);            /* em.rs:433-440 */
    ++n_passes;
    memcpy(out_counts, curr, sizeof(double) * T);   /* em.rs:443-446 */
    if (info) { info->niter = niter; info->n_passes = n_passes; info->converged = converged; info->rel_diff = last_rel; }
    free(prev); free(curr);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* bootstrap.rs:7-16  get_sample_inds                                        */
/* ------------------------------------------------------------------------ */
typedef struct { uint64_t s[4]; } xoshiro;

static inline uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t xoshiro_next(xoshiro *g)
{
    const uint64_t r = rotl64(g->s[1] * 5, 7) * 9;
    const uint64_t t = g->s[1] << 17;
    g->s[2] ^= g->s[0]; g->s[3] ^= g->s[1]; g->s[1] ^= g->s[2]; g->s[0] ^= g->s[3];
    g->s[2] ^= t; g->s[3] = rotl64(g->s[3], 45);
    return r;
}
/* unbiased Uniform{0..n-1} (Lemire's multiply-and-reject), the distribution
 * rand 0.10's Uniform::new(0, n) samples (bootstrap.rs:8). */
static inline uint64_t uniform_below(xoshiro *g, uint64_t n)
{
    uint64_t x = xoshiro_next(g);
    __uint128_t m = (__uint128_t)x * (__uint128_t)n;
    uint64_t l = (uint64_t)m;
    if (l < n) {
        const uint64_t t = (0 - n) % n;
        while (l < t) {
            x = xoshiro_next(g);
            m = (__uint128_t)x * (__uint128_t)n;
            l = (uint64_t)m;
        }
    }
    return (uint64_t)(m >> 64);
}

void oracle_get_sample_inds(uint64_t n, uint64_t seed, uint64_t *out_inds)
{
    xoshiro g;
    uint64_t sm = seed;
    for (int i = 0; i < 4; ++i) g.s[i] = splitmix64(&sm);
    /* bootstrap.rs:9-13: n iid draws; bootstrap.rs:14: sort_unstable.  The
     * sorted list is produced here by a counting pass, which yields exactly
     * the multiset sort_unstable would, in the same order. */
    uint32_t *cnt = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
    for (uint64_t k = 0; k < n; ++k) cnt[uniform_below(&g, n)]++;
    uint64_t o = 0;
    for (uint64_t i = 0; i < n; ++i)
        for (uint32_t c = 0; c < cnt[i]; ++c) out_inds[o++] = i;
    free(cnt);
}

void oracle_inds_to_weights(const uint64_t *inds, uint64_t n_inds, uint64_t n_reads,
                            uint32_t *row_w)
{
    memset(row_w, 0, sizeof(uint32_t) * n_reads);
    for (uint64_t k = 0; k < n_inds; ++k) row_w[inds[k]]++;
}

/* ------------------------------------------------------------------------ */
/* em.rs:273-314  do_bootstrap / bootstrap                                   */
/* ------------------------------------------------------------------------ */
int oracle_bootstrap(const oracle_store *s, const double *init, uint32_t n_boot, uint64_t seed,
                     const uint32_t *row_w_all, uint32_t max_iter, double conv_thresh,
                     int nthreads, double *out, oracle_run_info *infos)
{
    int rc = 0;
    if (nthreads < 1) nthreads = 1;
    /* em.rs:303-312: rayon pool over replicates, each a serial do_em with the
     * niter>50 gate (do_em, em.rs:212) and logging off (em.rs:289). */
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int64_t b = 0; b < (int64_t)n_boot; ++b) {
        oracle_run_info ri;
        int r;
        if (row_w_all) {
            r = oracle_do_em(s, init, max_iter, conv_thresh, 50, NULL, 0,
                             row_w_all + (uint64_t)b * s->n_reads,
                             out + (uint64_t)b * s->n_txps, &ri);
        } else {
            uint64_t *inds = (uint64_t *)malloc(sizeof(uint64_t) * (s->n_reads ? s->n_reads : 1));
            oracle_get_sample_inds(s->n_reads, seed + (uint64_t)b, inds); /* em.rs:275-276 */
            r = oracle_do_em(s, init, max_iter, conv_thresh, 50, inds, s->n_reads, NULL,
                             out + (uint64_t)b * s->n_txps, &ri); /* em.rs:287-289 */
            free(inds);
        }
        if (infos) infos[b] = ri;
        if (r) {
#pragma omp critical
            rc = r;
        }
    }
    return rc;
}

/* ------------------------------------------------------------------------ */
/* aux_counts.rs:23-50                                                       */
/* ------------------------------------------------------------------------ */
void oracle_aux_counts(const oracle_store *s, uint32_t *unique_count, uint32_t *total_count)
{
    memset(unique_count, 0, sizeof(uint32_t) * s->n_txps);
    memset(total_count, 0, sizeof(uint32_t) * s->n_txps);
    for (uint64_t i = 0; i < s->n_reads; ++i) {
        const uint64_t b = s->row_ptr[i], e = s->row_ptr[i + 1];
        const int is_unique = (e - b) == 1;         /* aux_counts.rs:35 */
        for (uint64_t j = b; j < e; ++j) {
            const uint32_t t = s->tid[j];
            if (t < s->n_txps) {                    /* aux_counts.rs:41 get_mut */
                total_count[t] += 1;
                if (is_unique) unique_count[t] += 1;
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* write_function.rs:283-318                                                 */
/* ------------------------------------------------------------------------ */
void oracle_assignment_probs(const oracle_store *s, const double *counts, double display_thresh,
                             double *out_prob)
{
    const int model_coverage = s->cov_prob != NULL;           /* write_function.rs:270 */
    for (uint64_t i = 0; i < s->n_reads; ++i) {
        const uint64_t b = s->row_ptr[i], e = s->row_ptr[i + 1];
        double denom = 0.0;                                   /* :284 */
        for (uint64_t j = b; j < e; ++j) {                    /* :286-291 */
            const double prob = (double)s->as_prob[j];
            const double cov = model_coverage ? s->cov_prob[j] : 1.0;
            denom += counts[s->tid[j]] * prob * cov;
        }
        double denom2 = 0.0;                                  /* :301 */
        for (uint64_t j = b; j < e; ++j) {                    /* :303-314 */
            const double prob = (double)s->as_prob[j];
            const double cov = model_coverage ? s->cov_prob[j] : 1.0;
            double nprob = (counts[s->tid[j]] * prob * cov) / denom;
            /* f64::clamp(0.0, 1.0): NaN stays NaN */
            if (nprob < 0.0) nprob = 0.0;
            if (nprob > 1.0) nprob = 1.0;
            if (nprob >= display_thresh) {                    /* :309 (false for NaN) */
                out_prob[j] = nprob;
                denom2 += nprob;
            } else {
                out_prob[j] = -1.0;
            }
        }
        for (uint64_t j = b; j < e; ++j)                      /* :316-318 */
            if (out_prob[j] >= 0.0) out_prob[j] /= denom2;
    }
}

/* ------------------------------------------------------------------------ */
/* Reference-faithful 36 B/nnz layout (CPU baseline only)                    */
_Static_assert(sizeof(oracle_alninfo) == 24, "AlnInfo is 24 bytes");
/* ------------------------------------------------------------------------ */
static inline void m_step_row_aos(const uint64_t *row_ptr, const oracle_alninfo *alns,
                                  const float *as_prob, const double *cov_prob,
                                  int model_coverage, uint64_t row, const double *prev,
                                  double *curr, int atomic)
{
    const uint64_t b = row_ptr[row], e = row_ptr[row + 1];
    double denom = 0.0;
    for (uint64_t j = b; j < e; ++j) {
        const double prob = (double)as_prob[j];
        const double cov = model_coverage ? cov_prob[j] : 1.0;
        denom += prev[alns[j].ref_id] * prob * cov * 1.0;
    }
    if (denom > ORACLE_EM_DENOM_THRESH) {
        for (uint64_t j = b; j < e; ++j) {
            const double prob = (double)as_prob[j];
            const double cov = model_coverage ? cov_prob[j] : 1.0;
            const double inc = (prev[alns[j].ref_id] * prob * cov * 1.0) / denom;
            if (atomic) atomic_add_f64(&curr[alns[j].ref_id], inc);
            else curr[alns[j].ref_id] += inc;
        }
    }
}

static int em_aos_impl(uint64_t n_reads, uint32_t T, const uint64_t *row_ptr,
                       const oracle_alninfo *alns, const float *as_prob, const double *cov_prob,
                       int model_coverage, const double *init, uint32_t max_iter,
                       double conv_thresh, uint32_t min_iter_gate, int nthreads, int parallel,
                       double *out_counts, oracle_run_info *info)
{
    double *prev = (double *)malloc(sizeof(double) * (T ? T : 1));
    double *curr = (double *)calloc(T ? T : 1, sizeof(double));
    if (!prev || !curr) { free(prev); free(curr); return 1; }
    if (nthreads < 1) nthreads = 1;
    if (init) memcpy(prev, init, sizeof(double) * T);
    else { const double avg = (double)n_reads / (double)T; for (uint32_t i = 0; i < T; ++i) prev[i] = avg; }
    double rel_diff = 0.0, last_rel = 0.0;
    uint32_t niter = 0, n_passes = 0, converged = 0;
    const int64_t R = (int64_t)n_reads;
    for (int final_pass = 0; final_pass < 2; ++final_pass) {
        while (final_pass || niter < max_iter) {
            if (parallel) {
#pragma omp parallel for schedule(dynamic, 4096) num_threads(nthreads)
                for (int64_t i = 0; i < R; ++i)
                    m_step_row_aos(row_ptr, alns, as_prob, cov_prob, model_coverage, (uint64_t)i, prev, curr, 1);
            } else {
                for (int64_t i = 0; i < R; ++i)
                    m_step_row_aos(row_ptr, alns, as_prob, cov_prob, model_coverage, (uint64_t)i, prev, curr, 0);
            }
            ++n_passes;
            if (final_pass) break;
            for (uint32_t i = 0; i < T; ++i)
                if (prev[i] > ORACLE_MIN_READ_THRESH)
                    rel_diff = fmax(rel_diff, (curr[i] - prev[i]) / prev[i]);
            { double *t = prev; prev = curr; curr = t; }
            memset(curr, 0, sizeof(double) * T);
            last_rel = rel_diff;
            if (rel_diff < conv_thresh && niter > min_iter_gate) { converged = 1; break; }
            niter += 1;
            rel_diff = 0.0;
        }
        if (!final_pass)
            for (uint32_t i = 0; i < T; ++i)
                if (prev[i] < ORACLE_MIN_READ_THRESH) prev[i] = 0.0;
    }
    memcpy(out_counts, curr, sizeof(double) * T);
    if (info) { info->niter = niter; info->n_passes = n_passes; info->converged = converged; info->rel_diff = last_rel; }
    free(prev); free(curr);
    return 0;
}

int oracle_do_em_aos(uint64_t n_reads, uint32_t n_txps, const uint64_t *row_ptr,
                     const oracle_alninfo *alns, const float *as_prob, const double *cov_prob,
                     int model_coverage, const double *init, uint32_t max_iter,
                     double conv_thresh, uint32_t min_iter_gate, double *out_counts,
                     oracle_run_info *info)
{
    return em_aos_impl(n_reads, n_txps, row_ptr, alns, as_prob, cov_prob, model_coverage, init,
                       max_iter, conv_thresh, min_iter_gate, 1, 0, out_counts, info);
}

int oracle_em_par_aos(uint64_t n_reads, uint32_t n_txps, const uint64_t *row_ptr,
                      const oracle_alninfo *alns, const float *as_prob, const double *cov_prob,
                      int model_coverage, const double *init, uint32_t max_iter,
                      double conv_thresh, uint32_t min_iter_gate, int nthreads,
                      double *out_counts, oracle_run_info *info)
{
    return em_aos_impl(n_reads, n_txps, row_ptr, alns, as_prob, cov_prob, model_coverage, init,
                       max_iter, conv_thresh, min_iter_gate, nthreads, 1, out_counts, info);
}
