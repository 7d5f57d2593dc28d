/*
 * iht_oracle.h -- CPU restatement ("oracle") of MendelIHT.jl's IHT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under mendeliht.jl_amd/ (the product)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / CPU baseline.
 *
 * Parity status: PINNED for Normal/Identity on a SnpLinAlg by the reference's
 * recorded run on its shipped data (docs/src/man/examples.md:230-267, golden
 * G1 in tests/golden/golden_normal_k7.json).  Bernoulli/Poisson/NegBin, cv_iht,
 * multivariate and the group projection are pinned only through the
 * reference's invariants (test/utilities_test.jl, test/L0_reg_test.jl) because
 * the reference (Julia) cannot run in this image and records no RNG-free
 * outputs for them.
 *
 * The 2-bit mat-vec arithmetic the reference calls lives in SnpArrays.jl
 * (v0.3.14/0.3.15, `SnpLinAlg`, linalg_direct.jl), which is not vendored in
 * /root/reference; its published semantics are restated in orc_snp_* below.
 */
#ifndef IHT_ORACLE_H
#define IHT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_NORMAL = 0, ORC_BERNOULLI = 1, ORC_POISSON = 2, ORC_NEGBIN = 3, ORC_GAMMA = 4, ORC_INVGAUSS = 5 };
enum { ORC_IDENTITY = 0, ORC_LOGIT = 1, ORC_LOG = 2, ORC_PROBIT = 3, ORC_CLOGLOG = 4, ORC_CAUCHIT = 5,
       ORC_INVERSE = 6, ORC_INVSQUARE = 7, ORC_SQRT = 8 };
enum { ORC_OK = 0, ORC_BAD_DIM = 1, ORC_BAD_ARG = 2, ORC_NOT_CENTERED = 3,
       ORC_NAN_LOGL = 4, ORC_INF_LOGL = 5 };

/* Design matrix: either a PLINK 2-bit SnpLinAlg (kind 0) or a dense
 * column-major double matrix (kind 1; the reference's Matrix{Float64}). */
typedef struct orc_mat {
    int      kind;
    int64_t  n, p;
    /* kind 0 */
    const uint8_t *cols;      /* p columns of `stride` bytes, PLINK .bed body */
    int64_t  stride;
    int      center, scale, impute;
    double  *mu, *sinv;       /* owned, length p */
    uint8_t *owned;           /* private NUMA-local copy of the columns */
    /* kind 1 */
    const double *dense;      /* n x p column-major, not owned */
} orc_mat;

typedef struct orc_params {
    int64_t  k;               /* fit.jl:64 */
    int64_t  J;               /* fit.jl:65 */
    int      dist, link;      /* fit.jl:66-67 */
    double   nb_r;            /* NegativeBinomial r */
    double   tol;             /* fit.jl:75 */
    int32_t  max_iter, min_iter, max_step; /* fit.jl:76-78 */
    int32_t  est_r;           /* 0 :None, 1 :MM, 2 :Newton (fit.jl:71) */
    const uint8_t *zkeep;     /* q flags or NULL = trues (fit.jl:70) */
    const double  *weight;    /* p or NULL (fit.jl:69) */
    const int64_t *group;     /* p (1-based labels) or NULL (fit.jl:68) */
    const int64_t *ks;        /* per-group sparsity or NULL (k::Vector{Int}) */
    int64_t  nks;
    int32_t  init_beta;       /* fit.jl:80: start from univariate regression estimates (Normal only) */
    int32_t  debias;          /* fit.jl:73,188: refit the support by GLM after a step that kept it */
    /* _choose! draws from the caller's RNG in the reference (utilities.jl:453 `sample`, multivariate.jl:336-337 `shuffle!`): a test
     * hands the restatement and the library the same draw through this callback (kind 0: write the `excess` sampled entries of
     * list to out; kind 1 / 2: write the whole list, shuffled, to out).  NULL = deterministic rule of choose().  Single fits only. */
    int (*choose)(void *user, int32_t kind, const int64_t *list, int64_t n, int64_t excess, int64_t *out);
    void    *choose_user;
    /* cv_iht: Threads.nthreads() of the reference run being restated.  V[threadid()] is re-used for every combination a
     * thread gets from `Threads.@threads :static` (cross_validation.jl:91,100-110), so with est_r the NegBin r of one fit is
     * where the thread's next fit starts.  0 or 1 = one thread (one chain over all combinations). */
    int32_t  cv_threads;
} orc_params;

typedef struct orc_result {
    double   logl;            /* best loglikelihood (fit.jl:206) */
    int64_t  iter;
    double   pve;             /* pve.jl:32 */
    double   nb_r;            /* final NegBin r */
    int32_t  choose_fired;    /* _choose! would have sampled (utilities.jl:444) */
    int32_t  n_trace;
    double  *beta;            /* caller-allocated p */
    double  *c;               /* caller-allocated q */
    double  *logl_trace;      /* caller-allocated max_iter or NULL */
    double  *tol_trace;
    int32_t *bt_trace;
    double  *mu;              /* caller-allocated n or NULL: final v.mu */
    double   eta_cond;        /* diagnostic, not in the reference: the smallest, over the steps, of
                                 (|df[idx]|^2 + |df2[idc]|^2) / (|df|^2 + |df2|^2), the share of the squared score that lies on
                                 the support.  iht_stepsize! divides exactly that numerator by |X_S df_S + Z df2|^2
                                 (utilities.jl:754-757).  A share of ~1e-20 means the score on the support is a rounding residue
                                 (the previous step was an exact line search on this support; or the support is empty -- initial
                                 support with a vector k, utilities.jl:427-429 -- and the intercept has just been solved for).
                                 With two or more active directions eta is then a ratio of two residues; with one direction
                                 it is well defined unless the residue happens to round to exactly 0, when the guard of
                                 utilities.jl:760-761 makes the step 1e-8 instead (seed 4036 of tools/fuzz_parity.py: the sum of
                                 y - mu came out as 0 on the GPU and as 1e-13 here).  Either way two floating-point
                                 implementations take different steps; the sweeps set such a trajectory aside. */
    double   ib_cond;         /* diagnostic, not in the reference: with init_beta, the smallest over the univariate regressions of
                                 |sum x^2 - (sum x)^2 / N| / sum x^2, the relative second pivot of linreg!'s 2 x 2 Cholesky
                                 (utilities.jl:823-842); 1 without init_beta.  A predictor that is CONSTANT over the training rows
                                 (a SNP monomorphic in a fold) makes that pivot a rounding residue: if it comes out <= 0 the
                                 reference's `catch` returns the unsolved right-hand side (beta = x'y, clamped to +-2), if it comes
                                 out as +1e-14 the "solution" is a ratio of two residues -- Julia's pairwise sums, this file's
                                 running sums and the device's exact integer counts each land somewhere else (seed 9568 of
                                 tools/fuzz_parity.py).  The sweeps set such a fit aside. */
    double   bt_cond;         /* diagnostic, not in the reference: the smallest, over the evaluations of _iht_backtrack_'s
                                 `old_logl > new_logl` (utilities.jl:484) that had another halving left to decide, of
                                 |old_logl - new_logl| / |old_logl|; 1 when none was evaluated.  A converging fit proposes a step whose
                                 loglikelihood equals the previous one to the last bit or two (seed 9878 of tools/fuzz_parity.py:
                                 -2492.3159915021474 against ...480): whether that counts as "lower" -- two halvings of the
                                 step here, none on the device, estimates 7e-7 apart -- is decided by the order of the n terms
                                 of the loglikelihood sum.  The sweeps set a fit that DIFFERS and has bt_cond below 1e-13 aside. */
    double   db_minstep;      /* diagnostic, not in the reference: the smallest step factor any IRLS iteration of any debias! refit of
                                 this fit was halved to (GLM.jl's step halving: f = 1/2, 1/4, ... while the deviance rises; 1 = no
                                 refit ever halved).  Written even when the fit ends in an error (the refit's own "step-halving
                                 failed" / "did not converge" is such an error).  A well-posed refit takes full Newton steps; one
                                 that halves crawls along a non-convex deviance -- debias! fits WITHOUT intercept or covariates
                                 (utilities.jl:1014-1020), so with the sqrt link eta = X_S b changes sign over the samples, mu =
                                 eta^2 folds, the working residual (y - eta^2) / 2 eta blows up where eta passes zero -- and no
                                 two implementations follow the same crawl: seed 16330 of tools/fuzz_parity.py, where this file
                                 (under every order of the rows) and a numpy restatement run out of their 30 iterations at
                                 deviances 10 688 .. 11 291, and the device's 28th step, halved twice, happens to lower the
                                 deviance by less than the tolerance and counts as converged.  The sweeps set a fit with debias
                                 that DIFFERS and has db_minstep below 1 aside. */
} orc_result;

/* ---- SnpLinAlg restatement (SnpArrays.jl linalg_direct.jl) ------------- */
orc_mat *orc_snp_create(const uint8_t *cols, int64_t n, int64_t p, int64_t stride,
                        int center, int scale, int impute);
orc_mat *orc_dense_create(const double *x, int64_t n, int64_t p);
void     orc_mat_destroy(orc_mat *m);
void     orc_mat_mu_sinv(const orc_mat *m, double *mu, double *sinv);
double   orc_getindex(const orc_mat *m, int64_t i, int64_t j);
void     orc_xtv(const orc_mat *m, const double *r, double *out);   /* mul!(out, Transpose(x), r) */
void     orc_xtv_colwise(const orc_mat *m, const double *r, double *out);   /* same sums, one column at a time */
void     orc_xtv_multi(const orc_mat *m, const double *R, int64_t nrhs, double *OUT);
/* out = sum_{j: idx[j]!=0} x[:,j]*coef[j]   (utilities.jl:98-106, 731-739) */
void     orc_xv_masked(const orc_mat *m, const uint8_t *idx, const double *coef, double *out);
void     orc_set_threads(int nthreads);
int      orc_get_threads(void);

/* ---- projections (utilities.jl:553-559, 613-679) ------------------------ */
int  orc_project_k(double *x, int64_t len, int64_t k);
int  orc_project_group_sparse(double *y, const int64_t *group, int64_t len,
                              int64_t J, const int64_t *k, int k_is_vector);

/* ---- GLM scalars (GLM.jl / Distributions.jl closed forms) --------------- */
double orc_linkinv(int link, double eta);
double orc_mueta(int link, double eta);
double orc_glmvar(int dist, double mu, double nb_r);
double orc_devresid(int dist, double y, double mu, double nb_r);
double orc_loglik_obs(int dist, double y, double mu, double wt, double phi, double nb_r);
double orc_loglikelihood(int dist, double nb_r, const double *y, const double *mu,
                         const double *wts, int64_t n);
double orc_deviance(int dist, double nb_r, const double *y, const double *mu,
                    const double *wts, int64_t n);

/* debias! (utilities.jl:1014-1020) on its own: the GLM refit of y on the columns with idx[j] != 0 (no intercept, unit
 * weights); b[j] receives the coefficients.  For pinning the restated GLM.jl IRLS against an independent solver. */
int orc_debias_glm(const orc_mat *x, const uint8_t *idx, const double *y, int dist, int link, double nb_r, double *b);

/* mle_for_r (utilities.jl:141-247) on its own: method 1 = one :MM update, 2 = :Newton to its fixed point; wts may be NULL. */
double orc_mle_for_r(const double *y, const double *mu, const double *wts, int64_t n, double r0, int method);

/* ---- drivers ------------------------------------------------------------ */
/* fit_iht (fit.jl:60-118); z is n x q column-major; train = NULL or n flags. */
int orc_fit_iht(const orc_mat *x, const orc_params *prm, const double *y,
                const double *z, int64_t q, const uint8_t *train, orc_result *res);
/* cv_iht (cross_validation.jl:60-131); folds 1-based; mses_raw q*npath
 * (fold-major, cross_validation.jl:217-223) or NULL; mse_out npath. */
int orc_cv_iht(const orc_mat *x, const orc_params *prm, const double *y,
               const double *z, int64_t q, const int32_t *folds, int32_t nfolds,
               const int64_t *path, int64_t npath, double *mses_raw, double *mse_out);

/* multivariate Gaussian IHT (multivariate.jl); Y is r x n column-major (traits
 * of a sample contiguous), Z is q x n column-major.  B_out r x p, C_out r x q,
 * Sigma_out r x r. */
typedef struct orc_mv_result {
    double   logl;
    int64_t  iter;
    int32_t  n_trace;
    int32_t  choose_fired;
    double  *B, *C, *Sigma, *pve;      /* caller-allocated r*p, r*q, r*r, r */
    double  *logl_trace, *tol_trace;   /* max_iter or NULL */
    int32_t *bt_trace;
} orc_mv_result;
int orc_fit_mv(const orc_mat *x, const orc_params *prm, const double *Y, int64_t r,
               const double *Z, int64_t q, const uint8_t *train, orc_mv_result *res);
int orc_cv_mv(const orc_mat *x, const orc_params *prm, const double *Y, int64_t r,
              const double *Z, int64_t q, const int32_t *folds, int32_t nfolds,
              const int64_t *path, int64_t npath, double *mses_raw, double *mse_out);

#ifdef __cplusplus
}
#endif
#endif
