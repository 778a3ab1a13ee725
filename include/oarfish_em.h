/*
 * oarfish_em.h -- C ABI of the MI355X-native EM quantification engine.
 *
 * This is the drop-in boundary for oarfish's abundance-estimation hot path
 * (COMBINE-lab/oarfish v0.10.3, src/em.rs + src/bootstrap.rs).  The reference
 * has no FFI layer; the seam it would bind is three Rust functions
 *     em::em      (src/em.rs:262)   -> oem_em_run(..., min_iter_gate = 50)
 *     em::em_par  (src/em.rs:320)   -> oem_em_run(..., min_iter_gate = 1)
 *     em::bootstrap (src/em.rs:292) -> oem_bootstrap
 * over an `EMInfo` (src/util/oarfish_types.rs:408-428), whose `eq_map`
 * (`InMemoryAlignmentStore`, :548-558) becomes an `oem_store` handle that keeps
 * the sparse read x transcript conditional-probability matrix resident in HBM.
 * INTEGRATION.md shows the Rust `extern "C"` block + shim a maintainer adds.
 *
 * Conventions
 *   - plain pointers and sizes only; all host buffers are caller-owned and are
 *     never retained after the call returns (the store is copied to HBM);
 *   - every entry point returns an `oem_status` (0 = OK) and never aborts or
 *     throws across the boundary (the reference's EM is infallible and built
 *     with panic=abort, Cargo.toml:118; a GPU library cannot be);
 *   - `oem_last_error()` gives the thread-local message of the last failure;
 *   - calls on distinct handles are thread-safe (single_cell.rs:96-150 calls
 *     em::em concurrently from N workers); calls on one handle are serialised.
 *   - there is NO CPU fallback: without a HIP device every compute entry point
 *     fails with OEM_ERR_NO_DEVICE.
 */
#ifndef OARFISH_EM_H
#define OARFISH_EM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: oem_time_bootstrap_passes, oem_store_opts.layout_build and .weight_coding (were reserved words: zero = the
 *    default, as before), the peer-to-peer communicator entry points (oem_comm_p2p_*), oem_store_info; version 1
 *    callers keep working (additions only). */
#define OEM_ABI_VERSION 2

typedef enum {
    OEM_OK = 0,
    OEM_ERR_ARG = 1,       /* NULL / inconsistent argument (bad row_ptr, tid >= n_txps ...) */
    OEM_ERR_OOM = 2,       /* host or device allocation failed */
    OEM_ERR_HIP = 3,       /* HIP runtime error */
    OEM_ERR_RCCL = 4,      /* RCCL error / librccl not loadable */
    OEM_ERR_NO_DEVICE = 5, /* no usable HIP device */
    OEM_ERR_STATE = 6      /* handle used in a state that does not allow the call */
} oem_status;

/* src/util/constants.rs:1-2 */
#define OEM_MIN_READ_THRESH 1e-5
#define OEM_EM_DENOM_THRESH 1e-30

/* Opaque handles. */
typedef struct oem_store oem_store; /* InMemoryAlignmentStore resident on one GPU (one row shard) */
typedef struct oem_comm oem_comm;   /* RCCL communicator over the row shards of one node */

/* What do_em / em_par leave behind besides the counts. */
typedef struct {
    uint32_t niter;     /* value of `niter` at loop exit (em.rs:170,218 / :354,405) */
    uint32_t n_passes;  /* E/M passes executed, including the final one (em.rs:245 / :433) */
    uint32_t converged; /* 1 if the loop left through `break` (em.rs:212-214 / :399-401) */
    uint32_t reserved;
    double rel_diff;    /* rel_diff of the last loop pass (em.rs:194-201), as logged at :219-233 */
} oem_run_info;

/* Layout / tuning knobs of a store (all optional; zero = default). */
typedef struct {
    uint32_t reorder_rows; /* 0 = default (locality reorder on), 1 = keep caller order, 2 = force reorder */
    uint32_t problem_size; /* 0 = one EM problem; > 0: the transcript space is the concatenation of
                              independent problems of this many transcripts (tiles never mix them);
                              set by oem_em_run_cells for its per-cell batches */
    uint32_t window_cap;   /* transcripts per tile window: 0 = chosen from the store (large and sparse -- at least
                              1 M reads, fewer than 2 per transcript: 2048; else 512), or 512 / 2048 to force it */
    uint32_t layout_build; /* 0 = build the tiled layout on the device (the host builder takes the stores the
                              device builder declines); 1 = always the host builder (oem_layout.cpp, the
                              specification the device builder is tested against) */
    uint32_t weight_coding; /* 0 = a store with at most 1024 distinct f32 weights (as_prob is exp of an integer score
                              gap over a constant: tens to hundreds of values) keeps its weights as indices into a
                              table of them -- in the spare bits of the window codes up to 128 values, a byte each up
                              to 256, 16 bits each up to 1024 (wide-window stores: up to 256, a byte each); lossless,
                              oem_layout_dict.hip; 1 = always the f32 stream (was reserved[0]); 2 = opt-in for stores
                              with a coverage column: the static weight (p as f64) * cov (em.rs:107-111) is rounded
                              once to f32 (relative error <= 6e-8 per weight, against the 1e-4 the abundances are held
                              to) and the store streams 8 B per alignment through the f32 kernels instead of 12 through
                              the f64 ones; all arithmetic of the EM stays f64.  Without cov_prob: the same as 0 */
    uint32_t reserved[3];
} oem_store_opts;

/* --------------------------------------------------------------------- */
/* library                                                                */
/* --------------------------------------------------------------------- */
int oem_abi_version(void);
const char *oem_last_error(void);
int oem_device_count(int *out_count);

/* --------------------------------------------------------------------- */
/* alignment store                                                        */
/* --------------------------------------------------------------------- */

/* Replaces InMemoryAlignmentStore as em.rs reads it (oarfish_types.rs:548-558
 * via iter(), :651-656): row_ptr == boundaries (n_reads+1 entries, row_ptr[0]==0,
 * strictly increasing: the reference never stores an empty read, :724,735-737;
 * empty rows are tolerated and contribute nothing), tid == alignments[].ref_id,
 * as_prob == as_probabilities (f32, :552), cov_prob == coverage_probabilities
 * (f64) or NULL when filter_opts.model_coverage is false (em.rs:108).
 * `device` is the HIP device ordinal.  The arrays are copied to HBM, laid out
 * for the E/M kernels, and may be freed by the caller on return. */
int oem_store_create(const uint64_t *row_ptr, const uint32_t *tid, const float *as_prob,
                     const double *cov_prob, uint64_t n_reads, uint64_t nnz, uint32_t n_txps,
                     int device, const oem_store_opts *opts, oem_store **out);
void oem_store_destroy(oem_store *store);

/* Tuning switches of a store (not part of the reference's semantics; results are
 * unchanged up to floating-point summation order). */
typedef enum {
    OEM_OPT_BATCH_BOOTSTRAP = 1, /* value 1 (default): oem_bootstrap runs its replicates in batches that share
                                    each pass over the matrix (4 per pass, two such chains side by side on their
                                    own streams) when it can (narrow window cap, multiplicities < 256); 0: one per pass */
    OEM_OPT_BOOTSTRAP_FIRST_REPLICA = 2 /* value b0 (default 0): replicate k of the next oem_bootstrap calls
                                    draws the device resample of global replica b0 + k.  Lets N processes
                                    that each hold the whole store split one set of replicates with no
                                    collective (the reference's replicates are independent, em.rs:303-309). */
} oem_option;
int oem_store_set_option(oem_store *store, uint32_t option, uint64_t value);

/* store.len() / num_aligned_reads() (oarfish_types.rs:562-564,746-748),
 * total_len() (:741-743), txp_info.len(). */
int oem_store_dims(const oem_store *store, uint64_t *n_reads, uint64_t *nnz, uint32_t *n_txps);

/* Bytes of HBM the store occupies and the algorithmic bytes of one E/M pass
 * (nnz*(4+4|8) + (R+1)*4 + 2*T*8). */
int oem_store_bytes(const oem_store *store, uint64_t *hbm_bytes, uint64_t *algorithmic_bytes_per_pass);

/* Facts about the resident layout.  OEM_INFO_WEIGHT_DICT_ENTRIES: entries of the weight table when the local
 * weights are dictionary-coded (oem_store_opts.weight_coding), 0 when the store streams f32 / f64 weights;
 * OEM_INFO_TILES, OEM_INFO_REMOTE_ALIGNMENTS: tiles of the layout and alignments outside their tile's window. */
typedef enum { OEM_INFO_WEIGHT_DICT_ENTRIES = 1, OEM_INFO_TILES = 2, OEM_INFO_REMOTE_ALIGNMENTS = 3 } oem_store_info_key;
int oem_store_info(const oem_store *store, uint32_t key, uint64_t *value);

/* --------------------------------------------------------------------- */
/* store builder: the step right before the EM (host side; only the *_device entry points use the GPU) */
/* --------------------------------------------------------------------- */

/* AlignmentFilters (src/util/oarfish_types.rs:763-806), the fields filter() reads. */
typedef struct {
    uint32_t five_prime_clip;    /* :768 */
    int64_t three_prime_clip;    /* :772 */
    float score_threshold;       /* :778 */
    float min_aligned_fraction;  /* :782 */
    uint32_t min_aligned_len;    /* :786 */
    int32_t which_strand;        /* :789  0 = Unknown (both), 1 = Forward only, 2 = Reverse only */
    float score_prob_denom;      /* :804  D in exp((score - best) / D), default 5.0 */
    uint32_t reserved;
} oem_filters;

/* One alignment record as AlnRecordLike exposes it (src/util/oarfish_types.rs:180-202, :264-326). */
typedef struct {
    uint32_t ref_id;     /* ref_id() */
    uint32_t aln_start;  /* aln_start() */
    uint32_t aln_end;    /* aln_end() */
    uint32_t aln_span;   /* aln_span() */
    int64_t score;       /* aln_score(); ignored unless OEM_REC_HAS_SCORE */
    int64_t seq_len;     /* opt_sequence_len(); < 0 = None */
    uint32_t flags;      /* OEM_REC_* */
    uint32_t reserved;
} oem_aln_record;
#define OEM_REC_UNMAPPED 1u      /* is_unmapped() */
#define OEM_REC_REVERSE 2u       /* is_reverse_complemented() */
#define OEM_REC_SUPPLEMENTARY 4u /* is_supp() */
#define OEM_REC_HAS_SCORE 8u     /* aln_score() is Some */

/* DiscardTable (src/util/oarfish_types.rs:811-857). */
typedef struct {
    uint64_t discard_5p, discard_3p, discard_score, discard_aln_frac, discard_aln_len, discard_ori,
        discard_supp, valid_best_aln, no_mapping, no_valid_aln;
} oem_discard_table;

typedef struct oem_builder oem_builder;

/* InMemoryAlignmentStore::new + the transcript lengths filter() needs (TranscriptInfo.len). */
int oem_builder_create(const oem_filters *filters, const uint64_t *txp_len, uint32_t n_txps,
                       oem_builder **out);
void oem_builder_destroy(oem_builder *b);
/* InMemoryAlignmentStore::add_group (src/util/oarfish_types.rs:672-685): AlignmentFilters::filter
 * (:955-1130: strand / supplementary / length / 3' / 5' filters, best-score tracking, aligned-
 * fraction test, score threshold, as_prob = expf((score - best) / D) in f32, :1107-1113) followed by
 * add_filtered_group (:718-738).  *out_kept = alignments appended (0: the read was dropped).
 * The coverage intervals add_filtered_group also updates (:725-728) are recomputed from the retained
 * alignments by the coverage entry points below. */
int oem_builder_add_group(oem_builder *b, const oem_aln_record *records, uint32_t n_records,
                          uint32_t *out_kept);
int oem_builder_dims(const oem_builder *b, uint64_t *n_reads, uint64_t *nnz);
int oem_builder_discard_table(const oem_builder *b, oem_discard_table *out);
/* Copies the store out: row_ptr[n_reads+1], and per alignment tid / as_prob / start / end / strand
 * (0 forward, 1 reverse); any output pointer may be NULL. */
int oem_builder_export(const oem_builder *b, uint64_t *row_ptr, uint32_t *tid, float *as_prob,
                       uint32_t *start, uint32_t *end, uint8_t *strand);
/* The bulk coverage model: per-transcript binned coverage of the
 * retained alignments (TranscriptInfo::with_len_and_bin_width + add_interval,
 * src/util/oarfish_types.rs:460-468, :496-538), the clamped logistic bin probabilities
 * (logistic_prob, src/util/logistic_probability.rs:7-79; min coverage total_weight/100, f32 bin
 * counts) and the per-alignment coverage probability normalised to sum 1 per read
 * (normalize_read_probs, src/util/normalize_probability.rs:5-74).  out_cov_prob: nnz f64 in the
 * builder's alignment order -- the `cov_prob` column of oem_store_create.  bin_width: --bin-width
 * (prog_opts.rs:555, default 100); growth_rate: --growth-rate (prog_opts.rs:502, default 2.0).
 * Where the reference would panic (degenerate bins, non-finite probability) this returns
 * OEM_ERR_STATE. */
int oem_builder_coverage_probs(const oem_builder *b, uint32_t bin_width, double growth_rate,
                               double *out_cov_prob);
/* The single-cell coverage model: same bins and the same per-read normalisation, but the bin
 * probabilities are binomial_continuous_prob's (src/util/binomial_probability.rs:7-224; called per
 * cell at src/single_cell.rs:132-137): Binomial pmf of each bin's count, counts rescaled so the
 * largest is 709, normalised over the transcript's bins.  ln_gamma is libm's lgamma (the reference
 * uses statrs' Lanczos evaluation of the same function). */
int oem_builder_coverage_probs_binomial(const oem_builder *b, uint32_t bin_width, double *out_cov_prob);
/* Both coverage models on the device (oem_coverage_device.hip): the same f64 arithmetic, one thread per
 * alignment / transcript / read; the bins are summed with f64 atomics, so results agree with the host
 * functions above to the last few bits (and to ~1e-7 where a bin count sits on an f32 rounding boundary,
 * the reference's f32 truncation of the counts, oarfish_types.rs:478).  model: 0 = logistic
 * (growth_rate used), 1 = binomial.  Arrays are the caller's (host) CSR with the alignment coordinates
 * AlnInfo carries (start / end, oarfish_types.rs:330-337); out_cov_prob: nnz f64. */
int oem_coverage_probs_device(const uint64_t *row_ptr, const uint32_t *tid, const uint32_t *aln_start,
                              const uint32_t *aln_end, const uint64_t *txp_len, uint64_t n_reads, uint64_t nnz,
                              uint32_t n_txps, uint32_t bin_width, int model, double growth_rate, int device,
                              double *out_cov_prob);
int oem_builder_coverage_probs_device(const oem_builder *b, uint32_t bin_width, int model, double growth_rate,
                                      int device, double *out_cov_prob);
/* Uploads the built store (oem_store_create on the builder's arrays). */
int oem_builder_store_create(const oem_builder *b, const double *cov_prob, int device,
                             const oem_store_opts *opts, oem_store **out);

/* --------------------------------------------------------------------- */
/* EM                                                                     */
/* --------------------------------------------------------------------- */

/* One E/M pass (em.rs:87-133 m_step): out_counts[t] = sum over reads i of
 * row_w[i] * theta[t]*w_it / sum_j theta[t_j]*w_ij, reads with denominator
 * <= 1e-30 dropped (em.rs:115).  row_w == NULL means all ones.  theta and
 * out_counts are host arrays of n_txps f64.  For step-level parity tests and
 * external loop drivers; the fused drivers below never leave the device. */
int oem_m_step(oem_store *store, const double *theta, const uint32_t *row_w, double *out_counts);

/* The EM driver (em.rs:144-255 do_em / :320-447 em_par).
 *   init_abundances : n_txps f64 or NULL => uniform n_reads/n_txps (em.rs:160-167)
 *   max_iter        : EMInfo.max_iter (prog_opts.rs:532, default 1000)
 *   conv_thresh     : EMInfo.convergence_thresh (prog_opts.rs:536, default 1e-3)
 *   min_iter_gate   : 50 reproduces em::em (em.rs:212), 1 reproduces em::em_par (em.rs:399)
 *   out_counts      : n_txps f64, un-normalised expected read counts (em.rs:254)
 *   info            : optional
 * The stopping iteration is the reference's: the loop state is frozen on the
 * device at the first pass that satisfies the gate. */
int oem_em_run(oem_store *store, const double *init_abundances, uint32_t max_iter,
               double conv_thresh, uint32_t min_iter_gate, double *out_counts,
               oem_run_info *info);

/* --------------------------------------------------------------------- */
/* the steps right after the EM, on the same resident store                */
/* --------------------------------------------------------------------- */

/* aux_counts::get_aux_counts (src/util/aux_counts.rs:23-50): per transcript, the number of
 * alignments (total) and the number of single-alignment reads (unique); n_txps u32 each. */
int oem_aux_counts(oem_store *store, uint32_t *out_unique, uint32_t *out_total);

/* The E-step of write_function::write_out_prob (src/util/write_function.rs:283-318): per read,
 * nprob_j = clamp(counts[t_j]*p_j*cov_j / sum_j(...), 0, 1); alignments with nprob >=
 * display_thresh are kept and renormalised by their sum.  out_prob is nnz f64 in the caller's
 * alignment order: the probability the reference prints, or -1 for an alignment it omits. */
int oem_assignment_probs(oem_store *store, const double *counts, double display_thresh,
                         double *out_prob);

/* --------------------------------------------------------------------- */
/* bootstrap                                                              */
/* --------------------------------------------------------------------- */

/* Device-side draw of one bootstrap resample in multiplicity form: the
 * Multinomial(n_reads; 1/n_reads ...) count vector of bootstrap.rs:7-16
 * (n draws from Uniform[0,n) with replacement; sorting is immaterial once
 * expressed as counts).  Counter-based (Philox4x32-10) stream keyed by
 * (seed, replica).  out_row_w: n_reads u32 on the host. */
int oem_bootstrap_weights(oem_store *store, uint64_t seed, uint32_t replica, uint32_t *out_row_w);

/* em::bootstrap (em.rs:292-314): n_boot resampled EMs, each do_bootstrap
 * (em.rs:273-290) = do_em over random_sampling_iter with gate niter>50.
 *   row_w_all : optional n_boot x n_reads u32 (row-major) to inject the
 *               resamples (parity tests); NULL => drawn on the device as
 *               oem_bootstrap_weights(seed, b).
 *   out       : n_boot x n_txps f64, row-major (replicate-major, as the
 *               Vec<Vec<f64>> of em.rs:292 / columns bootstrap.{i} of bulk.rs:181-193)
 *   infos     : optional, n_boot entries. */
int oem_bootstrap(oem_store *store, uint32_t n_boot, uint64_t seed, const uint32_t *row_w_all,
                  const double *init_abundances, uint32_t max_iter, double conv_thresh,
                  double *out, oem_run_info *infos);

/* --------------------------------------------------------------------- */
/* single-cell batch                                                      */
/* --------------------------------------------------------------------- */

/* The per-cell contract of single_cell.rs:139-160: every cell is an
 * independent em::em(&emi, 1) with init_abundances None, gate 50, over its
 * own reads.  Cells are given as one concatenated CSR (arrays as in
 * oem_store_create) plus cell_row_off[n_cells+1] (read offsets per cell);
 * out is n_cells x n_txps f64 row-major (the caller keeps entries > 0 as
 * (col u32, val f32) triplets, single_cell.rs:155-160). */
int oem_em_run_cells(const uint64_t *cell_row_off, uint32_t n_cells, const uint64_t *row_ptr,
                     const uint32_t *tid, const float *as_prob, const double *cov_prob,
                     uint64_t n_reads, uint64_t nnz, uint32_t n_txps, int device,
                     uint32_t max_iter, double conv_thresh, double *out, oem_run_info *infos);

/* --------------------------------------------------------------------- */
/* multi-GPU (row shards + one RCCL all-reduce of the count vector / pass) */
/* --------------------------------------------------------------------- */

#define OEM_UNIQUE_ID_BYTES 128
/* Rank 0 obtains an id and hands it to the other ranks through whatever the
 * host uses (torch.distributed broadcast, MPI, a file ...). */
int oem_comm_unique_id(void *out_id /* OEM_UNIQUE_ID_BYTES */);
int oem_comm_create(const void *unique_id, int rank, int n_ranks, int device, oem_comm **out);
void oem_comm_destroy(oem_comm *comm);

/* The one-shot peer-to-peer exchange (oem_p2p.hip): every rank publishes its partial count vector in a
 * buffer its peers have mapped (hipIpc memory handles between processes; plain pointers between ranks
 * that are threads of one process) and sums the N partials itself, in rank order -- bit-identical on
 * every rank, no ring, no RCCL.  It serves vectors of up to 4 MB (the 1.6 MB count vector of a
 * 200 k-transcript store is latency-bound; the reference's analogue is the shared Vec<AtomicF64> of
 * em.rs:338-341); larger exchanges stay with RCCL when the communicator has one.
 *   oem_comm_create(NULL, rank, n_ranks > 1, ...) makes a communicator without RCCL;
 *   oem_comm_p2p_export: allocates this rank's exchange buffer for vectors of up to `capacity` doubles
 *     (n_txps, or n_txps * 4 to cover the batched bootstrap of a row-sharded store) and writes its
 *     handle (OEM_P2P_HANDLE_BYTES bytes), which the host gathers over whatever it has (as the unique id);
 *   oem_comm_p2p_connect: `all_handles` = the n_ranks handles in rank order; maps the peers' buffers.
 * All ranks must export the same capacity.  Needs HSA_ENABLE_IPC_MODE_LEGACY=0 where the driver only
 * supports dmabuf IPC.  A rank that waits more than 8 s for a peer fails with OEM_ERR_STATE. */
#define OEM_P2P_HANDLE_BYTES 128
int oem_comm_p2p_export(oem_comm *comm, uint64_t capacity, void *out_handle /* OEM_P2P_HANDLE_BYTES */);
int oem_comm_p2p_connect(oem_comm *comm, const void *all_handles /* n_ranks x OEM_P2P_HANDLE_BYTES */);

/* Collective (every rank, same value).  OEM_COMM_OPT_P2P_MAX_BYTES: largest vector, in bytes, that takes
 * the peer-to-peer exchange when the communicator also has RCCL (default 4 MB; 0 = always RCCL).
 * OEM_COMM_OPT_P2P_SHAPE: 0 (default) = by the number of ranks and the vector's size, 1 = one-shot (every rank reads every
 * partial whole), 2 = two-phase (rank r sums slice r, then every rank reads the reduced slices from their
 * owners: a quarter of the bytes per xGMI link at 8 ranks for one more flag round); oem_p2p.hip.
 * OEM_COMM_OPT_P2P_TIMEOUT_MS: bound of one wait for a peer inside an exchange kernel, in milliseconds (default
 * 8000; a wait that gives up ends the run with OEM_ERR_STATE instead of hanging the GPU).
 * OEM_COMM_OPT_P2P_SELF_CHECK (set BEFORE oem_comm_p2p_connect): 1 = connect ends with a checked exchange in both
 * shapes against a closed-form sum -- also the ranks' rendezvous, with a long wait (120 s), so a peer that is still
 * building its store does not time the EM loop's first exchange out.  Needs every rank inside connect at the same
 * time (ranks = processes); a failure leaves the peer-to-peer backend disconnected (RCCL, if any, carries on).
 * Value 2, AFTER oem_comm_p2p_connect: run that checked exchange now -- for hosts that first make sure every rank has
 * mapped its peers (oarfish_amd.dist gathers one byte per rank), so that a rank whose connect failed does not leave the
 * others waiting in a kernel for the rendezvous bound (the longer of OEM_COMM_OPT_P2P_TIMEOUT_MS and 120 s). */
typedef enum { OEM_COMM_OPT_P2P_MAX_BYTES = 1, OEM_COMM_OPT_P2P_SHAPE = 2, OEM_COMM_OPT_P2P_TIMEOUT_MS = 3,
               OEM_COMM_OPT_P2P_SELF_CHECK = 4 } oem_comm_option;
int oem_comm_set_option(oem_comm *comm, uint32_t option, uint64_t value);

/* What a communicator is made of, for the host's records (bench.py's config.exchange): OEM_COMM_INFO_RANKS = n_ranks
 * as created; OEM_COMM_INFO_RCCL_RANKS = the number of ranks RCCL itself reports for its communicator
 * (ncclCommCount; 0 without RCCL) -- the first multi-GPU run says from the library's own mouth how many ranks the
 * collective spanned; OEM_COMM_INFO_P2P_CONNECTED = 1 when the peer-to-peer exchange is mapped on this rank. */
typedef enum { OEM_COMM_INFO_RANKS = 1, OEM_COMM_INFO_RCCL_RANKS = 2, OEM_COMM_INFO_P2P_CONNECTED = 3 } oem_comm_info_key;
int oem_comm_info(const oem_comm *comm, uint32_t key, uint64_t *out);

/* Declare `store` to be rank-local row shard of a store with
 * `global_n_reads` reads in total (needed for the uniform init, em.rs:154,165).
 * After this, oem_em_run / oem_bootstrap on the shard are collective calls:
 * every rank must make them with the same arguments; each pass all-reduces
 * (sum, f64) the n_txps partial counts over `comm`, after which all ranks take
 * the identical convergence decision. */
int oem_store_attach_comm(oem_store *store, oem_comm *comm, uint64_t global_n_reads,
                          uint64_t global_row_offset);

/* --------------------------------------------------------------------- */
/* measurement                                                            */
/* --------------------------------------------------------------------- */

/* Launch the E/M kernel `n_launches` times back to back on the store's
 * stream, bracketed by HIP events on that stream; returns the average
 * launch duration in milliseconds (bench.py's roofline.achieved). */
int oem_time_m_step(oem_store *store, uint32_t n_launches, float *out_avg_ms);

/* Run exactly `n_iters` loop iterations (E/M pass + rel-diff + swap/clear,
 * em.rs:181-207) from the uniform init with no convergence exit, timed with
 * HIP events on the store's stream; out_ms = total milliseconds. */
int oem_time_em_iters(oem_store *store, uint32_t n_iters, float *out_ms);

/* Run `n_passes` batched bootstrap passes (tile + fold + rel-diff kernels of oem_bootstrap's
 * rolling batch) with every slot running its own device-drawn resample and no slot ever stopping,
 * timed with HIP events on the store's stream.  out_avg_ms = milliseconds per batched pass;
 * out_slots = replicates served by one pass; out_algorithmic_bytes = SURVEY.md section 8d's bytes
 * of one batched pass (matrix once, row weights + theta/counts per replicate).  OEM_ERR_STATE when
 * the store runs its bootstraps one per pass (wide windows, no tiled layout). */
int oem_time_bootstrap_passes(oem_store *store, uint32_t n_passes, float *out_avg_ms, uint32_t *out_slots,
                              uint64_t *out_algorithmic_bytes);

/* Collective on a store with an attached communicator: `n_calls` all-reduces of the n_txps count vector
 * back to back on the store's stream, between HIP events; *out_avg_us = microseconds per all-reduce
 * (the exchange by itself: peer to peer = publish + reduce kernels, RCCL = ncclAllReduce). */
int oem_time_allreduce(oem_store *store, uint32_t n_calls, float *out_avg_us);

/* Device time of the batched EM loops of this thread's LAST oem_em_run_cells call: milliseconds between
 * HIP events recorded on the group's stream right before the first and right after the last pass of
 * every batched group (upload, layout build and read-back excluded) -- groups that ran side by side on the
 * device (two workers) count the time they shared once: the length of the union of the groups' loops -- and
 * the batched passes launched, summed over the groups.
 * Together with the per-cell n_passes of `infos` this gives bench.py the roofline of the per-cell leg:
 * bytes = sum over cells of n_passes * (nnz_c * 8 + (R_c + 1) * 4 + 2 * T * 8).  Zero when every group
 * took the cell-by-cell fallback. */
int oem_cells_last_timing(float *out_loop_ms, uint64_t *out_batched_passes);

#ifdef __cplusplus
}
#endif
#endif /* OARFISH_EM_H */
