/*
 * mendeliht_hip.h -- C ABI of the MI355X-native IHT hot path.
 *
 * Drop-in boundary for MendelIHT.jl's IHT inner loop.  The reference has no
 * FFI layer; its seam is Julia dispatch on the design-matrix type
 * (`fit_iht(y, x::AbstractMatrix{T}, z)` src/fit.jl:60-63,
 *  `cv_iht` src/cross_validation.jl:60-63, `IHTVariable{T,M}`
 *  src/data_structures.jl:4).  Each entry point below names the reference
 * method(s) it replaces; INTEGRATION.md shows the `ccall` glue a maintainer
 * adds on the Julia side (julia/MendelIHTHip.jl).
 *
 * Conventions: every function returns an `int` status (MIH_OK = 0); all
 * pointers are HOST pointers unless the name says `_dev`; matrices are
 * column-major (Julia layout); indices crossing the boundary are 0-based
 * unless stated; the library owns device memory behind opaque handles and the
 * caller owns every host buffer.  A handle is immutable after creation and may
 * be shared by host threads; each fit call builds its own workspace + stream.
 */
#ifndef MENDELIHT_HIP_H
#define MENDELIHT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes <-> Julia exceptions (src/fit.jl:87-101,259-260;
 * src/data_structures.jl:63-85; src/utilities.jl:554,975-993) */
enum {
    MIH_OK = 0,
    MIH_BAD_DIM = 1,        /* DimensionMismatch */
    MIH_BAD_ARG = 2,        /* ArgumentError / DomainError / AssertionError */
    MIH_NOT_CENTERED = 3,   /* "x is not centered!" fit.jl:98 */
    MIH_NAN_LOGL = 4,       /* "Loglikelihood function is NaN" fit.jl:259 */
    MIH_INF_LOGL = 5,       /* "Loglikelihood function is Inf" fit.jl:260 */
    MIH_HIP_ERROR = 6,
    MIH_OOM = 7,
    MIH_NO_DEVICE = 8
};

/* Distributions / links of the GLM (GLM.jl, Distributions.jl; fit.jl:66-67) */
enum { MIH_NORMAL = 0, MIH_BERNOULLI = 1, MIH_POISSON = 2, MIH_NEGBIN = 3,
       MIH_GAMMA = 4, MIH_INVGAUSS = 5 };   /* the distributions loglik_obs covers (src/utilities.jl:32-43) */
enum { MIH_IDENTITY = 0, MIH_LOGIT = 1, MIH_LOG = 2, MIH_PROBIT = 3, MIH_CLOGLOG = 4, MIH_CAUCHIT = 5,
       MIH_INVERSE = 6, MIH_INVSQUARE = 7, MIH_SQRT = 8 };   /* GLM.jl Link types accepted as `l` */
enum { MIH_ESTR_NONE = 0, MIH_ESTR_MM = 1, MIH_ESTR_NEWTON = 2 };

typedef struct mih_mat mih_mat;     /* device-resident design matrix */

/* ---- library / device ------------------------------------------------------ */
int mih_device_count(int *count);
/* thread-local message of the last failing call on this host thread */
int mih_last_error(char *buf, size_t len);
int mih_version(int *major, int *minor);     /* 0.4: round 4 of this header (cv_threads, mih_cv_allgather, column-sharded lock-step drivers) */
/* sizeof(mih_fit_params), sizeof(mih_fit_result), sizeof(mih_mv_result), sizeof(mih_comm): lets a binding
 * check its struct mirrors against the library it loaded. */
int mih_abi_sizes(int64_t *sizes, int32_t n);

/* ---- design matrix: replaces SnpArrays.SnpLinAlg{T}(::SnpArray; model=ADDITIVE_MODEL,
 *      center, scale, impute) as constructed at src/wrapper.jl:68-69 and test/L0_reg_test.jl:11.
 * bed_cols: p columns of PLINK .bed body (after the 3-byte header), each
 * col_stride_bytes >= ceil(n/4) bytes, codes 00->0, 01->missing, 10->1, 11->2.
 * dtype is the caller's element type T, 64 or 32 (src/MendelIHT.jl:39: Float = Union{Float64, Float32}; MIH_BAD_ARG otherwise).  It
 * changes nothing on the device -- the 2-bit matrix has no element type there and every dot product is exact fixed point
 * recombined in Float64 -- so a SnpLinAlg{Float32} caller gets results at least as accurate as the reference's all-Float32 run:
 * the binding converts y, z to Float64 on the way in and the model to Float32 on the way out.  Uploads once; computes mu_j (mean of non-missing
 * dosages) and sinv_j = 1/sqrt(mu_j(1-mu_j/2)) (1 when that sqrt is 0) on device. */
int mih_snp_create(const uint8_t *bed_cols, int64_t n, int64_t p, int64_t col_stride_bytes,
                   int center, int scale, int impute, int dtype, int device, mih_mat **out);
/* On-device synthetic SnpArray for benchmarks: maf_j ~ U(0,0.5), g_ij ~ Binomial(2, maf_j),
 * each entry missing with probability missing_rate (distributions of
 * src/simulate_utilities.jl:33-47,85-101; counter-based hash RNG, seed-reproducible). */
int mih_snp_create_synthetic(int64_t n, int64_t p, uint64_t seed, double missing_rate,
                             int center, int scale, int impute, int device, mih_mat **out);
/* Columns [col_offset, col_offset + p) of the same synthetic matrix (the generator is keyed by
 * (seed, global column)): one shard of a column-sharded fit. */
int mih_snp_create_synthetic_shard(int64_t n, int64_t p, int64_t col_offset, uint64_t seed,
                                   double missing_rate, int center, int scale, int impute,
                                   int device, mih_mat **out);
/* Dense Float64 design matrix (the reference's `x::Matrix{Float64}` path), n x p column-major. */
int mih_dense_create(const double *x, int64_t n, int64_t p, int device, mih_mat **out);
/* `x::Matrix{Float32}` (test/L0_reg_test.jl:245, test/cv_iht_test.jl:41): the matrix is STORED in Float32 (half
 * the HBM traffic of X'r); y, the model and every sum stay Float64, so results are at least as accurate as the
 * reference's all-Float32 run. */
int mih_dense_create_f32(const float *x, int64_t n, int64_t p, int device, mih_mat **out);
int mih_dense_create_synthetic(int64_t n, int64_t p, uint64_t seed, int device, mih_mat **out);
/* Releases the matrix.  A mih_session must not be stepped after its matrix is gone, but it may be destroyed later: the
 * reserve of device memory a large matrix keeps for its fits lives until its last user. */
int mih_mat_destroy(mih_mat *h);
/* The reserve of device memory a matrix keeps for the fits that run on it (lock-step workspaces, IHTVariable blocks: no fit then
 * calls hipMalloc / hipFree).  A 2-bit matrix of 4 GiB or more gets one when it is created; this call gives one to ANY 2-bit
 * matrix -- bytes = 0: sized from its dimensions by the library's rule, bytes > 0: that size -- or releases it (bytes < 0).
 * Not to be called while a fit is running on the matrix.  MIH_OOM if the device has less than four times the size free. */
int mih_mat_reserve(mih_mat *h, int64_t bytes);
int mih_mat_dims(const mih_mat *h, int64_t *n, int64_t *p);
/* `x.μ`, `x.σinv` of the SnpLinAlg */
int mih_snp_mu_sigma(const mih_mat *h, double *mu, double *sinv);
/* Re-encode the device matrix as PLINK .bed columns (ceil(n/4) bytes each): lets a
 * synthetic matrix be handed to any other PLINK consumer. */
int mih_snp_export_bed(const mih_mat *h, uint8_t *bed_cols_out);
/* naive_impute(x::SnpArray, destination) -- src/utilities.jl:862-899: the PLINK columns of the matrix with every
 * missing entry (0x01) replaced by the most frequent genotype of its SNP (ties: 0x02, then 0x03, then 0x00, the order
 * of the reference's if / elseif chain); non-missing entries are unchanged.  The caller writes the 3-byte .bed header. */
int mih_snp_naive_impute(const mih_mat *h, uint8_t *bed_cols_out);

/* ---- genotype linear algebra ---------------------------------------------- */
/* mul!(out, Transpose(x), r)  -- call site src/utilities.jl:133 (score!) */
int mih_xtv(const mih_mat *h, const double *r, double *out);
/* SnpArrays.mul!(p_by_r, Transpose(sla), n_by_r) -- call site src/multivariate.jl:85;
 * R is n x m column-major, OUT is p x m column-major. */
int mih_xtv_batched(const mih_mat *h, const double *R, int m, double *OUT);
/* The same with an explicit fixed-point format of the residuals.  A residual is scaled by a power of two, rounded to an
 * integer R and written as digits that FP6 / FP4 represent exactly; each digit plane is one column of the MFMA B operand,
 * the dot products with the dosages are exact, and the only rounding is that of the residual.  digits = base * 100 + digits:
 *   0 = library default: 4910 for every multi-residual context (this call, cv_iht, model paths, multivariate fits,
 *        init_beta) whatever the number of residuals, 428 for the workspace of a single univariate fit.
 *   4910 10 base-49 digits d/8 in FP6 (e2m3), |R| < 2^54: rounding 2^-54 max|r|, below that of an n-term f64 sum.  The digit
 *        columns of a pass's residuals are packed back to back over its 32-column operands (19 residuals in six operands).  Row slices of 2^18 rows; above 2^22 rows the default steps down to 1316.
 *   1316 16 base-13 digits d/2 in FP4 (e2m1), |R| < 2^57, two residuals per operand (2^20-row slices; above 2^24
 *        rows the default steps down to 428).
 *   428  28 base-4 digits {-2..1}/2 in FP4, |R| < 2^54, one residual per operand (the first format; cross-check).
 *   4908 8 base-49 FP6 digits, |R| < 2^43, FOUR residuals per operand: opt-in fast mode for fused multi-RHS
 *        passes (cv_iht, multivariate, init_beta); relative error of X'r about 2^-43 max|r| / |r|_rms ~ 1e-12.
 *   1308 8 base-13 FP4 digits, |R| < 2^27, four per operand (~15 % faster than 4908, error ~1e-7).
 * A result never depends on which other residuals share a pass or on the kernel shape, only on the format.
 * What the fixed point costs: the quantum is 2^-54 of the LARGEST entry the fixed point carries (2^-53 at worst), and the error of
 * a column's X'r is at most sum_i g_ij times that quantum (plus 64 ulp of the result for the f64 recombination) -- for residuals
 * without heavy tails tighter than an n-term f64 dot product.  Heavy tails (round 6, csrc/peel.h): rows whose |r_i| towers over the
 * rest -- max|r| > 64 x the lower quartile of the maxima of the 64 strided 256-row blocks, at most 64 such rows -- leave the fixed
 * point: their digits are zero, the scale is set by the largest of the REST, and k_xtv_finalize adds their terms g_ij r_i in f64
 * (m rows of the 2-bit matrix per column, only when the guard fires).  With one entry 1e8 or 1e12 x the rest every column, with or
 * without that row, is within 2 x 2^-53 sum_i g_ij |r_i| of the exact rational value (numpy's pairwise sum: 8 x) and within 1e-13 of
 * its own value unless that value has cancelled; a residual without such rows does not move a bit.  What is left: more than 64 rows
 * above the guard's threshold (a heavy TAIL rather than a few outliers) keep the plain scale, i.e. entry r_i keeps
 * 54 + log2(|r_i| / max|r|) bits -- 200 rows 1e6 x the rest: ~1e-7 on the columns that carry none of them.  A Poisson fit with a
 * planted count of 500 among counts of ~1 keeps the CPU restatement's loglikelihood trace to 1e-12 over 172 steps.
 * A residual with a NaN or +-Inf entry gives NaN in EVERY column of its row of OUT (the reference's floating-point mul! gives NaN or
 * +-Inf in every column that touches the entry -- all of them for a centered matrix); the other residuals of the call are untouched.
 * (tests/test_gpu_parity.py: test_xtv_fixed_point_under_adversarial_dynamic_range, test_peeled_rows_in_fused_passes_with_missing_genotypes,
 * test_poisson_fit_with_a_planted_count_outlier) */
int mih_xtv_batched_fmt(const mih_mat *h, const double *R, int m, int digits, double *OUT);
/* out = sum_t x[:, idx[t]] * val[t]  -- the column loops of update_xb!
 * (src/utilities.jl:98-106) and iht_stepsize! (:731-739); idx 0-based. */
int mih_xv_sparse(const mih_mat *h, const int64_t *idx, const double *val, int64_t nnz, double *out);

/* ---- projections ----------------------------------------------------------- */
/* project_k!(x, k) src/utilities.jl:553-559: zero every |x_i| < |k-th largest|;
 * ties at the threshold are kept; *n_kept = number of non-zeros left. */
int mih_project_topk(double *x, int64_t len, int64_t k, int64_t *n_kept);
/* project_group_sparse!(y, group, J, k) src/utilities.jl:613-679; group labels 1..G. */
int mih_project_group_sparse(double *y, const int64_t *group, int64_t len, int64_t J,
                             const int64_t *k, int k_is_vector);

/* ---- fit_iht ---------------------------------------------------------------- */
/* Column-sharded single fit (one process per GPU): every process owns a contiguous block
 * of the SNP columns as its own mih_mat; y, z and all n-vectors are replicated.  The reference has no
 * multi-device fit, so there is no line to cite for the exchange itself: the library calls back into
 * the host at the points where iht_stepsize!/update_xb! (src/utilities.jl:722-764, 93-118) sum over
 * the support columns (all-reduce of n+1 / n doubles on the device) and where project_k! (utilities.jl:553-559)
 * needs the global k-th largest entry (all-gather of 1 + 2K doubles per rank: the shards' candidates as
 * (global index, value) pairs, from which every rank also rebuilds the whole k-sparse model, so that _choose!
 * and check_convergence, utilities.jl:444-458, 953-957, need no exchange; with prior weights or ties beyond the
 * message size two scalar reductions remain).  The host implements the
 * two collectives with its own communicator (RCCL through torch.distributed in the Python mirror,
 * MPI.Allreduce in the Julia glue).  Both must return 0 on success, on every rank, in the same order. */
typedef struct mih_comm {
    int32_t rank, world;
    int64_t col_offset;       /* global 0-based index of this shard's first column */
    int64_t p_global;         /* total number of SNP columns over all shards */
    /* in-place reduction of `count` doubles over the ranks; op 0 = sum, 1 = max.  on_device = 1: buf is
     * device memory on the fit's GPU and everything queued on the fit's stream has completed; the
     * callback returns after the reduced values are visible to any stream. */
    int (*allreduce)(void *user, double *buf, int64_t count, int32_t op, int32_t on_device);
    /* recv[r*count .. (r+1)*count) = `send` of rank r; host memory */
    int (*allgather)(void *user, const double *send, int64_t count, double *recv);
    void *user;
} mih_comm;

/* A native exchange for the column-sharded fit: an RCCL communicator owned by the library, whose two callbacks run
 * ncclAllReduce / ncclAllGather on a private stream (xGMI between the GPUs of a node) without re-entering the host
 * language.  One process per GPU.  Rank 0 calls mih_rccl_unique_id and hands the 128 bytes to the other ranks by whatever
 * the launcher offers (MPI_Bcast, torch.distributed.broadcast, a file); every rank then calls mih_comm_create_rccl and
 * passes the returned mih_comm in mih_fit_params::comm (and frees it with mih_comm_destroy_rccl).  librccl is loaded on
 * first use (dlopen; MENDELIHT_RCCL_LIB overrides the name).  No reference counterpart: the reference has no multi-device fit. */
int mih_rccl_unique_id(void *id128);
int mih_comm_create_rccl(const void *id128, int32_t rank, int32_t world, int32_t device, int64_t col_offset,
                         int64_t p_global, mih_comm **out);
/* diagnostics: the rank count RCCL itself reports for the communicator (ncclCommCount; -1 if unavailable) and the librccl file that
 * was loaded (a process may hold two: comm.hip picks the one beside its own libamdhip64) */
int mih_comm_info(const mih_comm *c, int32_t *ranks_seen, char *librccl_path, int64_t cap);
int mih_comm_destroy_rccl(mih_comm *c);
/* The ONE exchange of a cross-validation run by one process per GPU (the reference combines its threads' losses in the shared
 * `mses` vector, cross_validation.jl:99,113,124-127): every rank passes the nfolds * npath losses mih_cv_iht / mih_cv_mv /
 * mih_fit_iht_path left it (zeros for the combinations of other ranks); on return every rank holds their sum over the ranks
 * = the complete matrix, ready for mih_cv_meanloss.  One all-gather through `c` (ncclAllGather over xGMI for a communicator
 * made by mih_comm_create_rccl -- col_offset / p_global play no role here, pass 0 and 1 -- or the caller's own callbacks),
 * summed in rank order.  No MPI, no host-language collective. */
int mih_cv_allgather(const mih_comm *c, double *mses_raw, int64_t count);

/* keyword arguments of fit_iht (src/fit.jl:64-81) */
typedef struct mih_fit_params {
    int64_t  k;               /* sparsity (ignored when ks != NULL) */
    int64_t  J;               /* max groups */
    int32_t  dist, link;      /* MIH_NORMAL.., MIH_IDENTITY.. */
    double   nb_r;            /* NegativeBinomial r (d.r) */
    double   tol;
    int32_t  max_iter, min_iter, max_step;
    int32_t  est_r;           /* MIH_ESTR_* */
    const uint8_t *zkeep;     /* q flags or NULL = trues(q) */
    const double  *weight;    /* p prior weights or NULL */
    const int64_t *group;     /* p group labels (1-based) or NULL */
    const int64_t *ks;        /* per-group sparsity (k::Vector{Int}) or NULL */
    int64_t  nks;
    /* per-iteration callback = the `verbose` line of fit.jl:194-196; may be NULL */
    void (*progress)(void *user, int iter, double logl, int backtracks, double tol);
    void    *progress_user;
    int32_t  init_beta;       /* fit.jl:80 init_beta: start from the p univariate regressions
                                 (initialize_beta!, src/utilities.jl:776-812; Normal only) */
    const mih_comm *comm;     /* NULL = single process; else this process's shard of a column-sharded
                                 fit: h, weight, res->beta (mih_mv_result::B) cover the LOCAL columns only (mih_fit_iht,
                                 mih_session_* and mih_fit_mv; no cross-validation; init_beta for the univariate fit only.
                                 Round 6, univariate fit: debias -- the support's n x k panel is summed over the shards, one
                                 all-reduce of n * k doubles per refit, and every shard runs the same refit -- and the group
                                 projection: `group` holds the labels 1..G of the LOCAL columns (G, J, k and a vector k are
                                 the whole matrix's), every shard sends its own k_g largest of each group -- two all-gathers
                                 per projection -- and walks the union by the reference's rule) */
    int32_t  debias;          /* fit.jl:73,188 debias: after a step (iter >= 5) that kept the support, refit the
                                 support columns by GLM (debias!, src/utilities.jl:1014-1020; the reference needs
                                 memory_efficient=false for it, the device builds the n x k panel on the fly) */
    int32_t  xtv_digits;      /* fixed-point format of the residual in this call's X'r passes (no reference counterpart):
                                 0 = library default; 4910, 4908, 1316, 1308, 428 -- see mih_xtv_batched_fmt.  A property
                                 of the CALL: concurrent fits on one matrix may use different formats.
                                 -1 = auto, for the lock-step drivers (mih_cv_iht, mih_fit_iht_path) with a GLM link
                                 (Bernoulli / Poisson / NegBin / Gamma / InverseGaussian: the reference tolerance there is
                                 1e-4): every residual is looked at on its own -- if max |r| <= 128 rms(r) it is scored in the
                                 43-bit format (4908: four residuals per operand, ~19 % less pass time at configs[3]; its
                                 rounding then stays ~1e-11 of a column's X'r), otherwise in the 54-bit format.  The choice
                                 depends on that residual alone, so a fit still gives the same bits whatever its company.
                                 mih_fit_mv: per X'R pass, the 43-bit format when EVERY trait's row of T1 = Gamma * resid passes
                                 that test (three operands instead of four at r = 10), else the 54-bit one; one more small
                                 readback per iteration.  Univariate Normal fits, single fits and mih_cv_mv: as 0. */
    /* The reference's RANDOM tie-break, _choose! (src/utilities.jl:444-458, src/multivariate.jl:310-351): when a projection
     * leaves more than k non-zero effects (exact ties in |b|) the reference removes the excess at random with the caller's
     * RNG.  NULL: the library removes the smallest |b| (ties: highest index) and raises choose_fired in the result.
     * Otherwise the library asks the caller, so a Julia binding that answers with the reference's own two calls reproduces
     * the reference's draw from the same RNG state (the only RNG use on the path besides the default `folds`):
     *   kind MIH_CHOOSE_SAMPLE    `sample(non_zero_idx, excess, replace=false)` (utilities.jl:453): list = the n positions of
     *                             non-zero SNP effects (findall order, 0-based); write the `excess` positions to zero to out.
     *   kind MIH_CHOOSE_SHUFFLE_B `shuffle!(B_nz_idx)` (multivariate.jl:336): list = the n linear indices (trait + r * SNP,
     *                             eachindex order) of the non-zero entries of B; write ALL n in shuffled order to out.
     *   kind MIH_CHOOSE_SHUFFLE_C `shuffle!(C_nz_idx)` (multivariate.jl:337), called right after _B: the non-zero entries of
     *                             the covariates NOT in zkeep (trait + r * covariate); write all n in shuffled order to out.
     * An EMPTY list is not handed over (shuffle! of an empty vector draws nothing from the RNG).
     * The library then zeroes what the reference's loop zeroes (multivariate.jl:338-348).  Called on the thread that called
     * mih_fit_iht / mih_session_* / mih_fit_mv (a session keeps the pointer: it must stay valid until mih_session_destroy);
     * a non-zero return aborts the fit with MIH_BAD_ARG.  Ignored by the lock-step
     * drivers (cv_iht, model paths: their fits run on the library's threads -- the reference's own threaded loop draws from
     * task-local RNGs there) and by column-sharded fits (comm != NULL). */
    int (*choose)(void *user, int32_t kind, const int64_t *list, int64_t n, int64_t excess, int64_t *out);
    void    *choose_user;
    /* mih_cv_iht with est_r != MIH_ESTR_NONE only.  The reference keeps one IHTVariable per Julia thread and re-uses it for every
     * (fold, k) combination `Threads.@threads :static` hands that thread (cross_validation.jl:91,100-110): the NegBin r left by
     * one fit is where the thread's next fit starts, so the losses depend on JULIA_NUM_THREADS.  cv_threads = that number: the
     * fold-major combinations are cut into cv_threads contiguous blocks (the first total % cv_threads one longer, as :static
     * does), each block is a chain of fits handing r on, and the chains advance in lock-step.  0 or 1 = ONE chain over the whole
     * grid: what the reference does at its default Threads.nthreads() == 1, and what both bindings (and the tests' CPU checker) mean by the
     * same value; nfolds = one chain per fold (five times faster at configs[3]'s shape: an explicit choice, because
     * the losses differ from a default reference run).  With world > 1 chain c is
     * evaluated by rank c mod world (mih_cv_assignment does not apply).  Ignored without est_r: those fits are independent. */
    int32_t  cv_threads;
    /* How iht_one_step! (fit.jl:213-263) is driven.  0 = the step is resident on the device wherever the fit allows it (a plain 2-bit
     * univariate fit: no group projection, est_r or debias; a column shard only over the library's own communicator,
     * mih_comm_create_rccl, whose collectives are then queued inside the chain): the k-sparse iterate, the exact finish of project_k!,
     * the backtracking decision and the stopping rule live in device memory, the host queues the kernels of a step without
     * waiting and reads one record per series of attempts, one step behind.  Exact ties that need _choose! hand the step back to the host-driven
     * path.  Since round 6 also the fits of the lock-step drivers (mih_cv_iht, mih_fit_iht_path: a fit queues its step behind its lane's
     * fused pass and reads one record per step) and models of up to ~8000 effects (k_res_select_big; ~2000 before).
     * 1 = every step host-driven (rounds 1-4: 26 launches, three waits).  Same results either way, bit for bit. */
    int32_t  step_mode;
} mih_fit_params;
enum { MIH_CHOOSE_SAMPLE = 0, MIH_CHOOSE_SHUFFLE_B = 1, MIH_CHOOSE_SHUFFLE_C = 2 };

/* IHTResult (src/data_structures.jl:245-256) + the per-iteration log */
typedef struct mih_fit_result {
    double   time;            /* seconds inside the fit loop (fit.jl:157,200) */
    double   logl;            /* best loglikelihood */
    int64_t  iter;
    double   pve;             /* sigma_g (src/pve.jl:32) */
    double   nb_r;            /* final NegBin r */
    int32_t  choose_fired;    /* 1 if the reference's RNG tie-break _choose! would have run */
    int32_t  n_trace;
    double  *beta;            /* caller-allocated p  (best_b) */
    double  *c;               /* caller-allocated q  (best_c) */
    double  *logl_trace;      /* caller-allocated max_iter doubles, or NULL */
    double  *tol_trace;       /* "                                      */
    int32_t *bt_trace;        /* caller-allocated max_iter int32,  or NULL */
    double  *mu;              /* caller-allocated n (final v.mu), or NULL */
} mih_fit_result;

/* fit_iht(y, x, z; ...) src/fit.jl:60-118 (always the memory_efficient=true formulation: no dense
 * n x k copy of the support is kept).  z is n x q column-major (first column all ones);
 * train is NULL (all samples) or n flags = cv_train_idx. */
int mih_fit_iht(const mih_mat *h, const mih_fit_params *prm, const double *y,
                const double *z, int64_t q, const uint8_t *train, mih_fit_result *res);

/* cv_iht(y, x, z; path, q, folds, ...) src/cross_validation.jl:60-131.
 * folds: n labels in 1..nfolds.  The (fold,k) combinations are enumerated
 * fold-major (cross_validation.jl:217-223); this call evaluates those that
 * mih_cv_assignment gives to `rank` and writes their held-out deviance sums into
 * mses_raw[nfolds*npath] (others left 0), so ranks combine with ONE sum-reduce /
 * gather.  mih_cv_meanloss then applies meanloss (:304-320). */
int mih_cv_iht(const mih_mat *h, const mih_fit_params *prm, const double *y,
               const double *z, int64_t q, const int32_t *folds, int32_t nfolds,
               const int64_t *path, int64_t npath, int32_t rank, int32_t world,
               double *mses_raw);
/* The sharding rule of mih_cv_iht / mih_cv_mv (the reference's pmap / @threads hands combinations to workers as they come,
 * cross_validation.jl:98-121): rank_of[fold * npath + ik] = the rank of `world` that evaluates combination (fold, path[ik]).
 * Combinations are dealt out round-robin in the order (k descending, fold ascending), so every rank gets a stratified sample
 * of the model sizes (a fit's iteration count depends mostly on k).  No device needed. */
int mih_cv_assignment(const int64_t *path, int64_t npath, int32_t nfolds, int32_t world, int32_t *rank_of);
int mih_cv_meanloss(const double *mses_raw, const int32_t *folds, int64_t n, int32_t nfolds,
                    int64_t npath, double *mse_out);
/* iht_run_many_models(y, x, z; path, ...) src/cross_validation.jl:232-273: fit_iht on the FULL data for every
 * model size in path (no hold-out).  The fits advance in lock-step like the cross-validation fits (one fused
 * multi-RHS X'r pass per round), est_r included (every fit starts from prm->nb_r, as each fit_iht call of the reference builds
 * its own IHTVariable, :254-258).  Entry i is fitted by rank mih_cv_assignment(path, npath, 1, world)[i] (largest models first,
 * round-robin); logl_out[npath] (others 0), and optionally iter_out[npath], beta_out[npath*p], c_out[npath*q] (may be NULL). */
int mih_fit_iht_path(const mih_mat *h, const mih_fit_params *prm, const double *y, const double *z,
                     int64_t q, const int64_t *path, int64_t npath, int32_t rank, int32_t world,
                     double *logl_out, int64_t *iter_out, double *beta_out, double *c_out);
/* The same cross-validation driven from ONE process over several GPUs, the way the reference drives it
 * from several threads of one Julia process (Threads.@threads over the (fold,k) combinations on a shared x,
 * cross_validation.jl:100-112): hs[g] is a replica of the matrix on GPU g (they may also share a device),
 * one host thread per replica runs mih_cv_iht(hs[g], ..., rank = g, world = nrep) -- the combinations mih_cv_assignment gives
 * to g, or with est_r the chains c with c mod nrep == g -- and mses_raw receives the complete nfolds x npath matrix (no
 * separate reduction step). */
int mih_cv_iht_multi(const mih_mat *const *hs, int32_t nrep, const mih_fit_params *prm, const double *y,
                     const double *z, int64_t q, const int32_t *folds, int32_t nfolds,
                     const int64_t *path, int64_t npath, double *mses_raw);

/* ---- multivariate Gaussian IHT (src/multivariate.jl) ------------------------ */
typedef struct mih_mv_result {
    double   time, logl;
    int64_t  iter;
    int32_t  choose_fired, n_trace;
    double  *B;               /* r x p */
    double  *C;               /* r x q */
    double  *Sigma;           /* r x r = inv(Gamma) */
    double  *pve;             /* r */
    double  *logl_trace, *tol_trace;
    int32_t *bt_trace;
} mih_mv_result;
/* fit_iht(Y, Transpose(x), Z; d=MvNormal) : Y r x n, Z q x n column-major.
 * prm->comm != NULL: this process's shard of a column-sharded multivariate fit (round 5) -- h and B (r x local p) cover the local
 * columns, Y, Z, C, Sigma, the traces are replicated; per iteration one all-reduce of n r + 1 doubles (X_S df_S of iht_stepsize!,
 * multivariate.jl:220-254), one of n r doubles per update_xb! (:21-31) and one all-gather of 1 + 2K doubles per project_k! (the
 * shards' top-K entries of vec(B) as (global linear index, value) pairs).  init_beta and more than K exact ties at one shard's
 * threshold are refused. */
int mih_fit_mv(const mih_mat *h, const mih_fit_params *prm, const double *Y, int64_t r,
               const double *Z, int64_t q, const uint8_t *train, mih_mv_result *res);
int mih_cv_mv(const mih_mat *h, const mih_fit_params *prm, const double *Y, int64_t r,
              const double *Z, int64_t q, const int32_t *folds, int32_t nfolds,
              const int64_t *path, int64_t npath, int32_t rank, int32_t world, double *mses_raw);

/* ---- stepping session: one IHTVariable kept alive across calls ---------------- */
/* `initialize(...)` (src/data_structures.jl:117-135) + repeated `iht_one_step!`
 * (src/fit.jl:213-263) under the caller's control; used by bench.py to time exactly K
 * iterations and by callers that want their own convergence logic. */
typedef struct mih_session mih_session;
int mih_session_create(const mih_mat *h, const mih_fit_params *prm, const double *y,
                       const double *z, int64_t q, const uint8_t *train, mih_session **out);
/* save_prev! + iht_one_step! + check_convergence (fit.jl:182-193) */
int mih_session_step(mih_session *s, double *logl, int32_t *backtracks, double *tol);
/* `nsteps` such steps in one call -- the loop of fit_iht! (fit.jl:182-205) without its convergence test; logl and tol are
 * those of the last step, *backtracks the total */
int mih_session_run(mih_session *s, int64_t nsteps, double *logl, int64_t *backtracks, double *tol);
/* current model (v.b, v.c), not the best-so-far copy */
int mih_session_model(mih_session *s, double *beta, double *c);
int mih_session_destroy(mih_session *s);

/* ---- measurement hooks (bench.py; no reference counterpart) ----------------- */
/* The hook hangs on the matrix handle (no process-wide state).  While enabled, every launch of the dominant X'r kernel on
 * this matrix is bracketed by HIP events on the stream it is launched on and recorded with the name of the kernel that was
 * dispatched and the number of residuals it scored; the lock-step drivers (cv_iht, model paths) count what they did. */
typedef struct mih_pass_record {
    double  start_ms;         /* kernel start, relative to the mih_profile_enable(h, 1) call */
    double  ms;               /* kernel duration (HIP events on its own stream) */
    int32_t residuals;        /* residual vectors scored by this launch */
    int32_t operands;         /* B operands of the launch (32 digit columns each; a residual takes 10 in the default format) */
    int32_t stream_tag;       /* 0 = the handle's / a single fit's stream; 1, 2 = lock-step lanes */
    int32_t reserved;
    char    kernel[48];       /* e.g. "k_xtv_dma<1,2,4,8,fp4>" or "k_xtv_dma16<5,2,8,4,half>" */
} mih_pass_record;
enum { MIH_CNT_LANES = 0,          /* lock-step lanes started (summed over calls) */
       MIH_CNT_MAX_IN_FLIGHT = 1,  /* most fits in flight at once over all lanes */
       MIH_CNT_HANDOVERS = 2,      /* tail hand-overs (lane 1's fits adopted by lane 0) */
       MIH_CNT_SHARED_INIT = 3,    /* fits that started from another fit's initial score */
       MIH_CNT_ROUNDS = 4,         /* lock-step rounds */
       MIH_CNT_FITS = 5,           /* fits completed by the lock-step drivers */
       MIH_CNT_SCORES = 6,         /* IHT iterations of those fits as fit.jl counts them (one score ends each -- except a fit's last, see [14]) */
       MIH_CNT_MAX_LANE_SLOTS = 7, /* most fits in flight on ONE lane */
       MIH_CNT_INIT_SCORES = 8,    /* initial scores (init_iht_indices!, one per fit): rode a pass or were served by a copy */
       MIH_CNT_RESIDENT_STEPS = 9, /* iht_one_step! calls that ran resident on the device (step_mode 0; single fits, sessions and -- round 6 -- the lock-step drivers' fits) */
       MIH_CNT_RESIDENT_ATTEMPTS = 10, /* series of attempt slots that ended with their step still backtracking (it went on in the next series) */
       MIH_CNT_RESIDENT_HANDBACKS = 11, /* steps the device handed back to the host-driven path (_choose! ties, lists beyond its buffers) */
       MIH_CNT_RESIDENT_DIRECT = 12, /* attempts whose projection was queued as a direct gather (threshold forecast, verified) */
       MIH_CNT_RESIDENT_REDOS = 13,  /* ... of which the forecast failed: re-queued with the two histogram sweeps */
       MIH_CNT_SKIPPED_LAST_SCORES = 14, /* lock-step fits that converged: their last step's score (which the reference computes and never reads) was not computed */
       MIH_CNT_RESIDUALS_43BIT = 15, /* xtv_digits = -1: residuals the lock-step drivers scored in the 43-bit format (the rest: 54-bit) */
       MIH_CNT_PEELED_RESIDUALS = 16, /* residuals whose outlier guard fired: up to 64 rows towering over the rest left the fixed point and
                                       * rode the f64 side channel of k_xtv_finalize (csrc/peel.h); counted by mih_xtv_batched_fmt, mih_fit_iht, mih_fit_mv
                                       * and the lock-step drivers when they end */
       MIH_PROFILE_NCOUNTERS = 17 };
int mih_profile_enable(const mih_mat *h, int on);
/* synchronises the recorded launches; totals since the last reset */
int mih_profile_read(const mih_mat *h, double *xtv_kernel_ms, int64_t *xtv_launches, int reset);
/* the launches themselves, oldest first: up to cap records into out (may be NULL to ask for the count), *n = available */
int mih_profile_passes(const mih_mat *h, mih_pass_record *out, int64_t cap, int64_t *n, int reset);
/* exchanges of the column-sharded fits run on h while the hook was on, by kind: [0] all-reduce of n + 1 doubles (X_S g_S of
 * iht_stepsize! with the shards' |df_S|^2 riding along), [1] all-reduce of n doubles (X_S b_S of update_xb!), [2] all-gather of the
 * projection's candidates, [3] scalar exchanges on the host.  ms4 = summed duration (HIP events on the fit's stream for the
 * collectives queued there, the host clock for those the host waits for), count4 = how many. */
int mih_profile_exchange(const mih_mat *h, double *ms4, int64_t *count4, int reset);
int mih_profile_counters(const mih_mat *h, int64_t *out /* [MIH_PROFILE_NCOUNTERS] */, int reset);
/* Runs `iters` X'r passes of m residuals back to back on the handle's stream with R resident in HBM, bracketed by HIP
 * events; *ms_per_pass = average time of the whole chain (statistics, digit planes, pass, finalize).  m = 1 uses the
 * workspace of a single univariate fit, m > 1 the fused multi-residual one; digits as in mih_xtv_batched_fmt. */
int mih_bench_xtv(const mih_mat *h, int digits, int m, int iters, int warmup, uint64_t seed,
                  float *ms_per_pass, double *checksum);
/* Algorithmic bytes of one X'r pass: p*ceil(n/4) + 8*m*(n+p) + 16*p. */
int mih_xtv_algorithmic_bytes(const mih_mat *h, int m, double *bytes);

#ifdef __cplusplus
}
#endif
#endif /* MENDELIHT_HIP_H */
