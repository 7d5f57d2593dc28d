/*
 * iht_oracle.c -- CPU restatement of MendelIHT.jl's IHT hot path (see header).
 * TEST INFRASTRUCTURE ONLY: never linked into the product library.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * it follows.  SnpLinAlg arithmetic follows SnpArrays.jl 0.3.14/0.3.15
 * linalg_direct.jl (not vendored in the reference; semantics in SURVEY.md 8c).
 */
#include "iht_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;
void orc_set_threads(int t) { g_threads = t < 1 ? 1 : t; }
int  orc_get_threads(void) { return g_threads; }

/* ------------------------------------------------------------------------ */
/* PLINK .bed decoding.  Sample i of SNP j sits in bits 2*(i%4) of byte i/4.  */
/* Codes (simulate_utilities.jl:88-99, utilities.jl:871-893):                */
/*   00 -> 0, 01 -> missing, 10 -> 1, 11 -> 2 (ADDITIVE_MODEL).               */
/* ------------------------------------------------------------------------ */
static double  LUT_G[256][4];   /* genotype value, missing -> 0 */
static double  LUT_M[256][4];   /* 1 where missing              */
static uint8_t LUT_HASM[256];
static int     lut_ready = 0;

static void lut_init(void)
{
    if (lut_ready) return;
    for (int b = 0; b < 256; ++b) {
        LUT_HASM[b] = 0;
        for (int t = 0; t < 4; ++t) {
            int c = (b >> (2 * t)) & 3;
            LUT_G[b][t] = (c == 2) ? 1.0 : (c == 3) ? 2.0 : 0.0;
            LUT_M[b][t] = (c == 1) ? 1.0 : 0.0;
            if (c == 1) LUT_HASM[b] = 1;
        }
    }
    lut_ready = 1;
}

static inline int bed_code(const orc_mat *m, int64_t i, int64_t j)
{
    return (m->cols[j * m->stride + (i >> 2)] >> (2 * (i & 3))) & 3;
}

/* SnpLinAlg constructor: mu = mean of non-missing dosages; sinv = 1/sqrt(mu(1-mu/2))
 * when that sqrt is > 0, else 1 (same formula in-repo at wrapper.jl:414-419). */
orc_mat *orc_snp_create(const uint8_t *cols, int64_t n, int64_t p, int64_t stride,
                        int center, int scale, int impute)
{
    lut_init();
    orc_mat *m = (orc_mat *)calloc(1, sizeof(orc_mat));
    m->kind = 0; m->n = n; m->p = p; m->stride = stride;
    m->center = center; m->scale = scale; m->impute = impute;
    /* private copy, first-touched by the thread that will stream it (NUMA-local pages for the
     * OpenMP X'r used as the CPU baseline; the static column schedule below is the same everywhere) */
    m->owned = (uint8_t *)malloc((size_t)p * (size_t)stride);
    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t j = 0; j < p; ++j) memcpy(m->owned + j * stride, cols + j * stride, (size_t)stride);
    m->cols = m->owned;
    m->mu = (double *)malloc(sizeof(double) * (size_t)p);
    m->sinv = (double *)malloc(sizeof(double) * (size_t)p);
    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t j = 0; j < p; ++j) {
        int64_t cnt[4] = {0, 0, 0, 0};
        for (int64_t i = 0; i < n; ++i) cnt[bed_code(m, i, j)]++;
        int64_t nn = n - cnt[1];
        double mu = (double)(cnt[2] + 2 * cnt[3]) / (double)nn;
        m->mu[j] = mu;
        double s = sqrt(mu * (1.0 - mu / 2.0));
        m->sinv[j] = (s > 0.0) ? 1.0 / s : 1.0;
    }
    return m;
}

orc_mat *orc_dense_create(const double *x, int64_t n, int64_t p)
{
    orc_mat *m = (orc_mat *)calloc(1, sizeof(orc_mat));
    m->kind = 1; m->n = n; m->p = p; m->dense = x;
    return m;
}

void orc_mat_destroy(orc_mat *m)
{
    if (!m) return;
    free(m->mu); free(m->sinv); free(m->owned); free(m);
}

void orc_mat_mu_sinv(const orc_mat *m, double *mu, double *sinv)
{
    if (m->kind != 0) return;
    memcpy(mu, m->mu, sizeof(double) * (size_t)m->p);
    memcpy(sinv, m->sinv, sizeof(double) * (size_t)m->p);
}

/* x[i,j] of a SnpLinAlg (call sites utilities.jl:102,735): dosage (missing ->
 * mu_j when impute, else 0), minus mu_j if center, times sinv_j if scale. */
double orc_getindex(const orc_mat *m, int64_t i, int64_t j)
{
    if (m->kind == 1) return m->dense[j * m->n + i];
    int c = bed_code(m, i, j);
    double g = (c == 2) ? 1.0 : (c == 3) ? 2.0 : 0.0;
    if (c == 1 && m->impute) g = m->mu[j];
    if (m->center) g -= m->mu[j];
    if (m->scale) g *= m->sinv[j];
    return g;
}

static void snp_col_dot(const orc_mat *m, int64_t j, const double *r,
                        double *sg, double *sm)
{
    const uint8_t *col = m->cols + j * m->stride;
    int64_t nfull = m->n >> 2;
    double s[4] = {0, 0, 0, 0}, ms[4] = {0, 0, 0, 0};
    for (int64_t b = 0; b < nfull; ++b) {
        uint8_t byte = col[b];
        const double *rr = r + 4 * b;
        const double *g = LUT_G[byte];
        s[0] += g[0] * rr[0]; s[1] += g[1] * rr[1];
        s[2] += g[2] * rr[2]; s[3] += g[3] * rr[3];
        if (LUT_HASM[byte]) {
            const double *mm = LUT_M[byte];
            ms[0] += mm[0] * rr[0]; ms[1] += mm[1] * rr[1];
            ms[2] += mm[2] * rr[2]; ms[3] += mm[3] * rr[3];
        }
    }
    for (int64_t i = nfull * 4; i < m->n; ++i) {   /* ragged tail: pad bits ignored */
        int c = bed_code(m, i, j);
        if (c == 2) s[0] += r[i];
        else if (c == 3) s[0] += 2.0 * r[i];
        else if (c == 1) ms[0] += r[i];
    }
    *sg = (s[0] + s[1]) + (s[2] + s[3]);
    *sm = (ms[0] + ms[1]) + (ms[2] + ms[3]);
}

/* mul!(out, Transpose(x::SnpLinAlg), r) (call site utilities.jl:133):
 * out_j = sinv_j * ( sum_i g_ij r_i [+ mu_j * sum_{i missing} r_i] - mu_j * sum_i r_i ). */
void orc_xtv(const orc_mat *m, const double *r, double *out)
{
    int64_t n = m->n, p = m->p;
    if (m->kind == 1) {
        #pragma omp parallel for num_threads(g_threads) schedule(static)
        for (int64_t j = 0; j < p; ++j) {
            const double *xc = m->dense + j * n;
            double s = 0.0;
            for (int64_t i = 0; i < n; ++i) s += xc[i] * r[i];
            out[j] = s;
        }
        return;
    }
    double sumr = 0.0;
    for (int64_t i = 0; i < n; ++i) sumr += r[i];
    /* Loop order only: blocks of XTV_NB columns walk the rows in tiles of XTV_TB bytes (= 4*XTV_TB samples), so a
     * tile of r is reused from cache by the whole block.  Every column still adds its rows in ascending order
     * into the same four lane accumulators as snp_col_dot, so the results are bit-identical to the
     * column-at-a-time loop (checked by tests/test_oracle_golden.py). */
    enum { XTV_NB = 32, XTV_TB = 2048 };
    const int64_t nfull = n >> 2;
    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t j0 = 0; j0 < p; j0 += XTV_NB) {
        const int64_t nbk = (p - j0 < XTV_NB) ? p - j0 : XTV_NB;
        double s[XTV_NB][4], ms[XTV_NB][4];
        memset(s, 0, sizeof(s)); memset(ms, 0, sizeof(ms));
        for (int64_t b0 = 0; b0 < nfull; b0 += XTV_TB) {
            const int64_t b1 = (b0 + XTV_TB < nfull) ? b0 + XTV_TB : nfull;
            for (int64_t c = 0; c < nbk; ++c) {
                const uint8_t *col = m->cols + (j0 + c) * m->stride;
                double s0 = s[c][0], s1 = s[c][1], s2 = s[c][2], s3 = s[c][3];
                for (int64_t b = b0; b < b1; ++b) {
                    const uint8_t byte = col[b];
                    const double *rr = r + 4 * b;
                    const double *g = LUT_G[byte];
                    s0 += g[0] * rr[0]; s1 += g[1] * rr[1];
                    s2 += g[2] * rr[2]; s3 += g[3] * rr[3];
                    if (LUT_HASM[byte]) {
                        const double *mm = LUT_M[byte];
                        ms[c][0] += mm[0] * rr[0]; ms[c][1] += mm[1] * rr[1];
                        ms[c][2] += mm[2] * rr[2]; ms[c][3] += mm[3] * rr[3];
                    }
                }
                s[c][0] = s0; s[c][1] = s1; s[c][2] = s2; s[c][3] = s3;
            }
        }
        for (int64_t c = 0; c < nbk; ++c) {
            const int64_t j = j0 + c;
            for (int64_t i = nfull * 4; i < n; ++i) {   /* ragged tail: pad bits ignored */
                int cd = bed_code(m, i, j);
                if (cd == 2) s[c][0] += r[i];
                else if (cd == 3) s[c][0] += 2.0 * r[i];
                else if (cd == 1) ms[c][0] += r[i];
            }
            double o = (s[c][0] + s[c][1]) + (s[c][2] + s[c][3]);
            const double sm = (ms[c][0] + ms[c][1]) + (ms[c][2] + ms[c][3]);
            if (m->impute) o += m->mu[j] * sm;
            if (m->center) o -= m->mu[j] * sumr;
            if (m->scale) o *= m->sinv[j];
            out[j] = o;
        }
    }
}

/* the column-at-a-time form of the same sums (reference loop order; kept for the equivalence test) */
void orc_xtv_colwise(const orc_mat *m, const double *r, double *out)
{
    int64_t n = m->n, p = m->p;
    if (m->kind == 1) { orc_xtv(m, r, out); return; }
    double sumr = 0.0;
    for (int64_t i = 0; i < n; ++i) sumr += r[i];
    for (int64_t j = 0; j < p; ++j) {
        double sg, sm;
        snp_col_dot(m, j, r, &sg, &sm);
        double o = sg;
        if (m->impute) o += m->mu[j] * sm;
        if (m->center) o -= m->mu[j] * sumr;
        if (m->scale) o *= m->sinv[j];
        out[j] = o;
    }
}

/* SnpArrays.mul!(p_by_r, Transpose(sla), n_by_r) (call site multivariate.jl:85):
 * R is n x nrhs column-major, OUT is p x nrhs column-major. */
void orc_xtv_multi(const orc_mat *m, const double *R, int64_t nrhs, double *OUT)
{
    for (int64_t t = 0; t < nrhs; ++t) orc_xtv(m, R + t * m->n, OUT + t * m->p);
}

/* X[:, idx] * coef[idx] by the memory-efficient column loop
 * (utilities.jl:98-106 update_xb!, :731-739 iht_stepsize!). */
/* out += x[:,j] * bj, with x[i,j] as orc_getindex defines it */
static void axpy_col(const orc_mat *m, int64_t j, double bj, double *out)
{
    int64_t n = m->n;
    if (m->kind == 1) {
        const double *xc = m->dense + j * n;
        for (int64_t i = 0; i < n; ++i) out[i] += xc[i] * bj;
        return;
    }
    double v[4];
    double mu = m->mu[j], sinv = m->sinv[j];
    for (int c = 0; c < 4; ++c) {
        double g = (c == 2) ? 1.0 : (c == 3) ? 2.0 : 0.0;
        if (c == 1 && m->impute) g = mu;
        if (m->center) g -= mu;
        if (m->scale) g *= sinv;
        v[c] = g;
    }
    const uint8_t *col = m->cols + j * m->stride;
    for (int64_t i = 0; i < n; ++i) {
        int c = (col[i >> 2] >> (2 * (i & 3))) & 3;
        out[i] += v[c] * bj;
    }
}

void orc_xv_masked(const orc_mat *m, const uint8_t *idx, const double *coef, double *out)
{
    memset(out, 0, sizeof(double) * (size_t)m->n);
    for (int64_t j = 0; j < m->p; ++j)
        if (idx[j]) axpy_col(m, j, coef[j], out);
}

/* ------------------------------------------------------------------------ */
/* GLM closed forms (GLM.jl 1.x glmtools.jl / Distributions.jl)              */
/* ------------------------------------------------------------------------ */
double orc_linkinv(int link, double eta)
{
    switch (link) {
    case ORC_LOGIT: return 1.0 / (1.0 + exp(-eta));
    case ORC_LOG:   return exp(eta);
    /* GLM.jl glmtools.jl linkinv for the remaining Link types */
    case ORC_PROBIT:    return 0.5 * erfc(-eta / 1.4142135623730951);
    case ORC_CLOGLOG:   return -expm1(-exp(eta));
    case ORC_CAUCHIT:   return 0.5 + atan(eta) / 3.141592653589793;
    case ORC_INVERSE:   return 1.0 / eta;
    case ORC_INVSQUARE: return 1.0 / sqrt(eta);
    case ORC_SQRT:      return eta * eta;
    default:        return eta;
    }
}

double orc_mueta(int link, double eta)
{
    switch (link) {
    case ORC_LOGIT: { double e = exp(-fabs(eta)); double f = 1.0 + e; return e / (f * f); }
    case ORC_LOG:   return exp(eta);
    case ORC_PROBIT:    return exp(-0.5 * eta * eta) / 2.5066282746310002;
    case ORC_CLOGLOG:   return exp(eta) * exp(-exp(eta));
    case ORC_CAUCHIT:   return 1.0 / (3.141592653589793 * (1.0 + eta * eta));
    case ORC_INVERSE:   return -1.0 / (eta * eta);
    case ORC_INVSQUARE: { double m = 1.0 / sqrt(eta); return -m * m * m / 2.0; }
    case ORC_SQRT:      return 2.0 * eta;
    default:        return 1.0;
    }
}

double orc_glmvar(int dist, double mu, double nb_r)
{
    switch (dist) {
    case ORC_BERNOULLI: return mu * (1.0 - mu);
    case ORC_POISSON:   return mu;
    case ORC_NEGBIN:    return mu * (1.0 + mu / nb_r);
    case ORC_GAMMA:     return mu * mu;
    case ORC_INVGAUSS:  return mu * mu * mu;
    default:            return 1.0;
    }
}

static inline double xlogy(double x, double y) { return (x == 0.0) ? 0.0 : x * log(y); }

double orc_devresid(int dist, double y, double mu, double nb_r)
{
    switch (dist) {
    case ORC_BERNOULLI:
        return (y == 1.0) ? -2.0 * log(mu) : -2.0 * log1p(-mu);
    case ORC_POISSON:
        return 2.0 * (xlogy(y, y / mu) - (y - mu));
    case ORC_NEGBIN: {
        double v = 2.0 * (xlogy(y, y / mu) + xlogy(y + nb_r, (mu + nb_r) / (y + nb_r)));
        return (mu == 0.0) ? NAN : v;
    }
    case ORC_GAMMA:    return -2.0 * (log(y / mu) - (y - mu) / mu);
    case ORC_INVGAUSS: { double d = y - mu; return d * d / (y * mu * mu); }
    default: { double d = y - mu; return d * d; }
    }
}

/* utilities.jl:32-43 */
double orc_loglik_obs(int dist, double y, double mu, double wt, double phi, double nb_r)
{
    switch (dist) {
    case ORC_BERNOULLI:
        return wt * ((y == 1.0) ? log(mu) : log(1.0 - mu));
    case ORC_POISSON:
        return wt * (xlogy(y, mu) - mu - lgamma(y + 1.0));
    case ORC_NEGBIN: {
        double pp = nb_r / (mu + nb_r);
        double v = lgamma(nb_r + y) - lgamma(nb_r) - lgamma(y + 1.0)
                 + nb_r * log(pp) + xlogy(y, 1.0 - pp);
        return wt * v;
    }
    case ORC_GAMMA: {          /* wt*logpdf(Gamma(inv(phi), mu*phi), y) */
        double a = 1.0 / phi, th = mu * phi;
        return wt * (-lgamma(a) - a * log(th) + (a - 1.0) * log(y) - y / th);
    }
    case ORC_INVGAUSS: {       /* wt*logpdf(InverseGaussian(mu, inv(phi)), y) */
        double lam = 1.0 / phi, d = y - mu;
        return wt * (0.5 * log(lam / (6.283185307179586 * y * y * y)) - lam * d * d / (2.0 * mu * mu * y));
    }
    default: {
        double sd = sqrt(phi);
        double z = (y - mu) / sd;
        return wt * (-(z * z + 1.8378770664093454835606594728112) / 2.0 - log(sd));
    }
    }
}

/* utilities.jl:52-59 */
double orc_deviance(int dist, double nb_r, const double *y, const double *mu,
                    const double *wts, int64_t n)
{
    double dev = 0.0;
    for (int64_t i = 0; i < n; ++i) dev += wts[i] * orc_devresid(dist, y[i], mu[i], nb_r);
    return dev;
}

/* utilities.jl:9-20: phi divides by length(y), not the number of training samples */
double orc_loglikelihood(int dist, double nb_r, const double *y, const double *mu,
                         const double *wts, int64_t n)
{
    double phi = orc_deviance(dist, nb_r, y, mu, wts, n) / (double)n;
    double logl = 0.0;
    for (int64_t i = 0; i < n; ++i)
        logl += orc_loglik_obs(dist, y[i], mu[i], wts[i], phi, nb_r);
    return logl;
}

/* ------------------------------------------------------------------------ */
/* projections                                                               */
/* ------------------------------------------------------------------------ */
/* k-th largest of a[0..n) (destroys a); quickselect, O(n) expected */
static double kth_largest(double *a, int64_t n, int64_t k)
{
    int64_t lo = 0, hi = n - 1, target = k - 1;  /* index in descending order */
    while (lo < hi) {
        int64_t mid = lo + (hi - lo) / 2;
        double x = a[lo], y = a[mid], z = a[hi], piv;
        if ((x >= y && y >= z) || (z >= y && y >= x)) piv = y;
        else if ((y >= x && x >= z) || (z >= x && x >= y)) piv = x;
        else piv = z;
        int64_t i = lo, j = hi;
        while (i <= j) {
            while (a[i] > piv) ++i;
            while (a[j] < piv) --j;
            if (i <= j) { double t = a[i]; a[i] = a[j]; a[j] = t; ++i; --j; }
        }
        if (target <= j) hi = j;
        else if (target >= i) lo = i;
        else return a[target];
    }
    return a[target];
}

/* project_k!(x, k) utilities.jl:553-559: a = |k-th largest by abs|; zero every
 * |x_i| < a (ties at a are KEPT).  k<0 DomainError; k==0 or k>len BoundsError. */
int orc_project_k(double *x, int64_t len, int64_t k)
{
    if (k < 0 || k == 0 || k > len) return ORC_BAD_ARG;
    double *tmp = (double *)malloc(sizeof(double) * (size_t)len);
    for (int64_t i = 0; i < len; ++i) tmp[i] = fabs(x[i]);
    double a = kth_largest(tmp, len, k);
    free(tmp);
    for (int64_t i = 0; i < len; ++i) if (fabs(x[i]) < a) x[i] = 0.0;
    return ORC_OK;
}

typedef struct { double key; int64_t idx; } keyidx;
static int cmp_desc_stable(const void *pa, const void *pb)
{
    const keyidx *a = (const keyidx *)pa, *b = (const keyidx *)pb;
    if (a->key > b->key) return -1;
    if (a->key < b->key) return 1;
    return (a->idx < b->idx) ? -1 : (a->idx > b->idx);
}

/* project_group_sparse! utilities.jl:613-645 (k Int) and :647-679 (k Vector).
 * sortperm!(by=abs, rev=true) breaks ties by ascending index (Base.Order.Perm). */
int orc_project_group_sparse(double *y, const int64_t *group, int64_t len,
                             int64_t J, const int64_t *k, int k_is_vector)
{
    int64_t groups = 0;
    for (int64_t i = 0; i < len; ++i) if (group[i] > groups) groups = group[i];
    for (int64_t i = 0; i < len; ++i) if (group[i] < 1) return ORC_BAD_ARG;
    keyidx *perm = (keyidx *)malloc(sizeof(keyidx) * (size_t)len);
    for (int64_t i = 0; i < len; ++i) { perm[i].key = fabs(y[i]); perm[i].idx = i; }
    qsort(perm, (size_t)len, sizeof(keyidx), cmp_desc_stable);
    int64_t *gcount = (int64_t *)calloc((size_t)groups, sizeof(int64_t));
    keyidx *gnorm = (keyidx *)malloc(sizeof(keyidx) * (size_t)groups);
    for (int64_t g = 0; g < groups; ++g) { gnorm[g].key = 0.0; gnorm[g].idx = g; }
    for (int64_t i = 0; i < len; ++i) {
        int64_t j = perm[i].idx, g = group[j] - 1;
        int64_t kg = k_is_vector ? k[g] : k[0];
        if (gcount[g] < kg) { gnorm[g].key += y[j] * y[j]; gcount[g]++; }
    }
    qsort(gnorm, (size_t)groups, sizeof(keyidx), cmp_desc_stable);
    int64_t *grank = (int64_t *)malloc(sizeof(int64_t) * (size_t)groups);
    for (int64_t rnk = 0; rnk < groups; ++rnk) grank[gnorm[rnk].idx] = rnk + 1;
    for (int64_t g = 0; g < groups; ++g) gcount[g] = 1;
    for (int64_t i = 0; i < len; ++i) {
        int64_t j = perm[i].idx, g = group[j] - 1;
        int64_t kg = k_is_vector ? k[g] : k[0];
        if (grank[g] > J || gcount[g] > kg) y[j] = 0.0;
        else gcount[g]++;
    }
    free(perm); free(gcount); free(gnorm); free(grank);
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* univariate IHT state (data_structures.jl:4-43)                            */
/* ------------------------------------------------------------------------ */
typedef struct {
    const orc_mat *x; const double *y, *z; int64_t n, p, q;
    int64_t k, J; const int64_t *ks; int64_t nks;
    int dist, link; double nb_r; int est_r;
    double *b, *b0, *best_b, *xb, *xgk, *r, *df, *df2, *c, *c0, *best_c, *zc, *zdf2,
           *mu, *cv_wts, *full_b;
    uint8_t *idx, *idx0, *idc, *idc0, *zkeep;
    int64_t zkeepn;
    const int64_t *group; const double *weight;
    int choose_fired;
    double eta_cond;      /* orc_result.eta_cond */
    double ib_cond;       /* orc_result.ib_cond */
    double bt_cond;       /* orc_result.bt_cond */
    double db_minstep;    /* orc_result.db_minstep */
    int (*choose_cb)(void *, int32_t, const int64_t *, int64_t, int64_t, int64_t *);
    void *choose_user;
    int init_beta;
} ihtvar;

static double *dalloc(int64_t n) { return (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }
static uint8_t *balloc(int64_t n) { return (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1); }

static ihtvar *iv_create(const orc_mat *x, const orc_params *prm, const double *y,
                         const double *z, int64_t q)
{
    ihtvar *v = (ihtvar *)calloc(1, sizeof(ihtvar));
    int64_t n = x->n, p = x->p;
    v->x = x; v->y = y; v->z = z; v->n = n; v->p = p; v->q = q;
    v->k = prm->k; v->J = prm->J; v->ks = prm->ks; v->nks = prm->ks ? prm->nks : 0;
    if (v->nks > 0) v->k = 0;                         /* data_structures.jl:75-81 */
    v->dist = prm->dist; v->link = prm->link; v->nb_r = prm->nb_r; v->est_r = prm->est_r;
    v->group = prm->group; v->weight = prm->weight; v->init_beta = prm->init_beta;
    v->choose_cb = prm->choose; v->choose_user = prm->choose_user;
    v->eta_cond = 1.0; v->ib_cond = 1.0; v->bt_cond = 1.0; v->db_minstep = 1.0;
    v->b = dalloc(p); v->b0 = dalloc(p); v->best_b = dalloc(p); v->df = dalloc(p);
    v->xb = dalloc(n); v->xgk = dalloc(n); v->r = dalloc(n); v->zc = dalloc(n);
    v->zdf2 = dalloc(n); v->mu = dalloc(n); v->cv_wts = dalloc(n);
    v->df2 = dalloc(q); v->c = dalloc(q); v->c0 = dalloc(q); v->best_c = dalloc(q);
    v->full_b = dalloc(p + q);
    v->idx = balloc(p); v->idx0 = balloc(p); v->idc = balloc(q); v->idc0 = balloc(q);
    v->zkeep = balloc(q);
    v->zkeepn = 0;
    for (int64_t i = 0; i < q; ++i) {
        v->zkeep[i] = prm->zkeep ? (prm->zkeep[i] != 0) : 1;
        v->zkeepn += v->zkeep[i];
    }
    return v;
}

static void iv_destroy(ihtvar *v)
{
    free(v->b); free(v->b0); free(v->best_b); free(v->df); free(v->xb); free(v->xgk);
    free(v->r); free(v->zc); free(v->zdf2); free(v->mu); free(v->cv_wts); free(v->df2);
    free(v->c); free(v->c0); free(v->best_c); free(v->full_b); free(v->idx); free(v->idx0);
    free(v->idc); free(v->idc0); free(v->zkeep); free(v);
}

/* zc = Z c (utilities.jl:113) */
static void zmulc(const ihtvar *v, const double *c, double *out)
{
    for (int64_t i = 0; i < v->n; ++i) out[i] = 0.0;
    for (int64_t j = 0; j < v->q; ++j) {
        const double *zj = v->z + j * v->n; double cj = c[j];
        for (int64_t i = 0; i < v->n; ++i) out[i] += zj[i] * cj;
    }
}

/* update_mu! utilities.jl:74-82 */
static void update_mu(ihtvar *v)
{
    for (int64_t i = 0; i < v->n; ++i) v->mu[i] = orc_linkinv(v->link, v->xb[i] + v->zc[i]);
}

/* update_xb! utilities.jl:93-118 */
static void update_xb(ihtvar *v)
{
    orc_xv_masked(v->x, v->idx, v->b, v->xb);
    zmulc(v, v->c, v->zc);
    if (v->dist != ORC_NORMAL) {
        for (int64_t i = 0; i < v->n; ++i) {
            if (v->xb[i] < -20.0) v->xb[i] = -20.0; else if (v->xb[i] > 20.0) v->xb[i] = 20.0;
            if (v->zc[i] < -20.0) v->zc[i] = -20.0; else if (v->zc[i] > 20.0) v->zc[i] = 20.0;
        }
    }
}

/* score! utilities.jl:126-135 */
static void score(ihtvar *v)
{
    for (int64_t i = 0; i < v->n; ++i) {
        double eta = v->xb[i] + v->zc[i];
        double w = orc_mueta(v->link, eta) / orc_glmvar(v->dist, v->mu[i], v->nb_r);
        v->r[i] = w * (v->y[i] - v->mu[i]) * v->cv_wts[i];
    }
    orc_xtv(v->x, v->r, v->df);
    for (int64_t j = 0; j < v->q; ++j) {
        const double *zj = v->z + j * v->n; double s = 0.0;
        for (int64_t i = 0; i < v->n; ++i) s += zj[i] * v->r[i];
        v->df2[j] = s;
    }
}

static double loglik(const ihtvar *v)
{
    return orc_loglikelihood(v->dist, v->nb_r, v->y, v->mu, v->cv_wts, v->n);
}

/* vectorize! utilities.jl:291-315.  With a prior weight the reference scales
 * the covariate tail by weight[i>p] (an out-of-bounds read, SURVEY 8a note 12);
 * that is not emulated: the tail is copied unscaled. */
static void vectorize(const ihtvar *v, double *a, const double *b, const double *c)
{
    int64_t p = v->p, q = v->q;
    if (!v->weight) memcpy(a, b, sizeof(double) * (size_t)p);
    else for (int64_t i = 0; i < p; ++i) a[i] = b[i] * v->weight[i];
    for (int64_t i = 0; i < q; ++i) a[p + i] = v->zkeep[i] ? INFINITY : c[i];
}

/* unvectorize! utilities.jl:326-354 */
static void unvectorize(const ihtvar *v, const double *a, double *b, double *c)
{
    int64_t p = v->p, q = v->q;
    if (!v->weight) memcpy(b, a, sizeof(double) * (size_t)p);
    else for (int64_t i = 0; i < p; ++i) b[i] = a[i] / v->weight[i];
    for (int64_t i = 0; i < q; ++i) if (!v->zkeep[i]) c[i] = a[p + i];
}

/* _choose! utilities.jl:444-458.  The reference removes `excess` RANDOM
 * non-zero SNPs (StatsBase.sample).  With a callback (orc_params.choose) the caller
 * makes that draw: `for pos in sample(non_zero_idx, excess, replace=false)`
 * (utilities.jl:453).  Without one -- RNG parity with Julia is impossible -- the
 * restatement removes the smallest-|b| ones (ties: highest index first).  Either
 * way choose_fired tells a caller that the reference would have sampled. */
static int choose(ihtvar *v)
{
    int64_t sparsity = v->k + v->zkeepn;
    int64_t groups = (v->J == 0) ? 1 : v->J;
    int64_t nz = -v->zkeepn, nzb = 0;
    for (int64_t j = 0; j < v->p; ++j) nzb += v->idx[j];
    for (int64_t j = 0; j < v->q; ++j) nz += v->idc[j];
    nz += nzb;
    if (nz <= groups * sparsity) return ORC_OK;
    int64_t excess = nz - groups * sparsity;
    v->choose_fired = 1;
    if (v->choose_cb) {
        if (excess > nzb) return ORC_BAD_ARG;
        int64_t *list = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nzb + excess + 1)), *out = list + nzb, m = 0;
        for (int64_t j = 0; j < v->p; ++j) if (v->idx[j]) list[m++] = j;      /* findall(!iszero, v.idx) */
        int rc = v->choose_cb(v->choose_user, 0, list, nzb, excess, out);
        for (int64_t t = 0; t < excess && !rc; ++t) {
            int64_t pos = out[t];
            if (pos < 0 || pos >= v->p || !v->idx[pos]) { rc = 1; break; }     /* not in the list, or drawn twice */
            v->b[pos] = 0.0; v->idx[pos] = 0;
        }
        free(list);
        return rc ? ORC_BAD_ARG : ORC_OK;
    }
    keyidx *cand = (keyidx *)malloc(sizeof(keyidx) * (size_t)nzb);
    int64_t m = 0;
    for (int64_t j = 0; j < v->p; ++j)
        if (v->idx[j]) { cand[m].key = -fabs(v->b[j]); cand[m].idx = -j; ++m; }
    qsort(cand, (size_t)m, sizeof(keyidx), cmp_desc_stable); /* |b| asc, index desc */
    for (int64_t t = 0; t < excess && t < m; ++t) {
        int64_t pos = -cand[t].idx;
        v->b[pos] = 0.0; v->idx[pos] = 0;
    }
    free(cand);
    return ORC_OK;
}

/* _iht_gradstep! utilities.jl:252-280 */
static int gradstep(ihtvar *v, double eta)
{
    /* BLAS.axpy! (utilities.jl:258-259): OpenBLAS' daxpy kernels are FMA-based */
    for (int64_t j = 0; j < v->p; ++j) v->b[j] = fma(eta, v->df[j], v->b[j]);
    for (int64_t j = 0; j < v->q; ++j) v->c[j] = fma(eta, v->df2[j], v->c[j]);
    int k_is_int = (v->nks == 0);
    if (!v->group) {
        vectorize(v, v->full_b, v->b, v->c);
        int rc = orc_project_k(v->full_b, v->p + v->q, v->k + v->zkeepn);
        if (rc) return rc;
        unvectorize(v, v->full_b, v->b, v->c);
    } else {
        int rc = orc_project_group_sparse(v->b, v->group, v->p, v->J,
                                          k_is_int ? &v->k : v->ks, !k_is_int);
        if (rc) return rc;
    }
    for (int64_t j = 0; j < v->p; ++j) v->idx[j] = (v->b[j] != 0.0);
    for (int64_t j = 0; j < v->q; ++j) v->idc[j] = (v->c[j] != 0.0);
    if (k_is_int) return choose(v);
    return ORC_OK;
}

/* linreg! utilities.jl:823-842: regress y on [1 x]; on a Cholesky failure (e.g. a constant x) the
 * reference's `catch` returns xty_store UNSOLVED, i.e. (sum y, x'y) */
static void linreg(const double *x, const double *y, int64_t N, double *b0, double *b1, double *cond)
{
    double sx = 0.0, sxx = 0.0, sy = 0.0, sxy = 0.0;
    for (int64_t i = 0; i < N; ++i) { sx += x[i]; sxx += x[i] * x[i]; sy += y[i]; sxy += x[i] * y[i]; }
    double u11 = sqrt((double)N), u12 = sx / u11, d = sxx - u12 * u12;
    if (cond && sxx > 0.0) { double c = fabs(d) / sxx; if (c < *cond) *cond = c; }     /* diagnostic only (orc_result.ib_cond) */
    if (!(N > 0) || !(d > 0.0)) { *b0 = sy; *b1 = sxy; return; }
    double u22 = sqrt(d);
    double w1 = sy / u11, w2 = (sxy - u12 * w1) / u22;
    *b1 = w2 / u22;
    *b0 = (w1 - u12 * *b1) / u11;
}

/* initialize_beta! utilities.jl:776-812 followed by project_k!(v) utilities.jl:561-573 */
static int initialize_beta(ihtvar *v)
{
    int64_t n = v->n, p = v->p, q = v->q, nt = 0;
    for (int64_t i = 0; i < n; ++i) nt += (v->cv_wts[i] != 0.0);
    double *xs = dalloc(nt), *ys = dalloc(nt), *col = dalloc(n);
    int64_t t = 0;
    for (int64_t i = 0; i < n; ++i) if (v->cv_wts[i] != 0.0) ys[t++] = v->y[i];
    double c0 = 0.0, b0, b1;
    for (int64_t j = 0; j < p; ++j) {
        memset(col, 0, sizeof(double) * (size_t)n);
        axpy_col(v->x, j, 1.0, col);
        t = 0;
        for (int64_t i = 0; i < n; ++i) if (v->cv_wts[i] != 0.0) xs[t++] = col[i];
        linreg(xs, ys, nt, &b0, &b1, &v->ib_cond);
        c0 += b0; v->b[j] = b1;
    }
    for (int64_t l = 1; l < q; ++l) {
        t = 0;
        for (int64_t i = 0; i < n; ++i) if (v->cv_wts[i] != 0.0) xs[t++] = v->z[l * n + i];
        linreg(xs, ys, nt, &b0, &b1, &v->ib_cond);
        c0 += b0; v->c[l] = b1;
    }
    v->c[0] = c0 / (double)(p + q - 1);
    for (int64_t j = 0; j < p; ++j) v->b[j] = v->b[j] < -2.0 ? -2.0 : (v->b[j] > 2.0 ? 2.0 : v->b[j]);
    for (int64_t l = 0; l < q; ++l) v->c[l] = v->c[l] < -2.0 ? -2.0 : (v->c[l] > 2.0 ? 2.0 : v->c[l]);
    memcpy(v->b0, v->b, sizeof(double) * (size_t)p); memcpy(v->c0, v->c, sizeof(double) * (size_t)q);
    free(xs); free(ys); free(col);
    /* project_k!(v) */
    vectorize(v, v->full_b, v->b, v->c);
    int rc = orc_project_k(v->full_b, p + q, v->k + v->zkeepn);
    if (rc) return rc;
    unvectorize(v, v->full_b, v->b, v->c);
    for (int64_t j = 0; j < p; ++j) v->idx[j] = (v->b[j] != 0.0);
    for (int64_t l = 0; l < q; ++l) v->idc[l] = (v->c[l] != 0.0);
    return ORC_OK;
}

/* init_iht_indices! utilities.jl:366-438 */
static int init_iht_indices(ihtvar *v, const uint8_t *train)
{
    int64_t n = v->n, p = v->p, q = v->q;
    memset(v->b, 0, sizeof(double) * p); memset(v->b0, 0, sizeof(double) * p);
    memset(v->best_b, 0, sizeof(double) * p); memset(v->df, 0, sizeof(double) * p);
    memset(v->xb, 0, sizeof(double) * n); memset(v->xgk, 0, sizeof(double) * n);
    memset(v->r, 0, sizeof(double) * n); memset(v->zc, 0, sizeof(double) * n);
    memset(v->zdf2, 0, sizeof(double) * n); memset(v->mu, 0, sizeof(double) * n);
    memset(v->df2, 0, sizeof(double) * q); memset(v->c, 0, sizeof(double) * q);
    memset(v->c0, 0, sizeof(double) * q); memset(v->best_c, 0, sizeof(double) * q);
    memset(v->full_b, 0, sizeof(double) * (p + q));
    memset(v->idx, 0, p); memset(v->idx0, 0, p);
    memcpy(v->idc, v->zkeep, q); memcpy(v->idc0, v->zkeep, q);
    int64_t ntrain = 0;
    for (int64_t i = 0; i < n; ++i) { v->cv_wts[i] = (!train || train[i]) ? 1.0 : 0.0; ntrain += (v->cv_wts[i] != 0.0); }

    /* intercept by <=20 clamped Newton steps (utilities.jl:394-405) */
    double ybar = 0.0;
    for (int64_t i = 0; i < n; ++i) ybar += v->y[i] * v->cv_wts[i];
    ybar /= (double)ntrain;
    for (int it = 0; it < 20; ++it) {
        double g1 = orc_linkinv(v->link, v->c[0]);
        double g2 = orc_mueta(v->link, v->c[0]);
        double step = (g1 - ybar) / g2;
        if (step < -1.0) step = -1.0; else if (step > 1.0) step = 1.0;
        v->c[0] -= step;
        if (fabs(g1 - ybar) < 1e-10) break;
    }
    zmulc(v, v->c, v->zc);
    update_mu(v);
    score(v);

    if (v->init_beta) {
        if (v->dist != ORC_NORMAL) return ORC_BAD_ARG;       /* utilities.jl:391-392 */
        return initialize_beta(v);
    }
    /* initial support from the largest gradient entries; df is overwritten by
     * its own projection (utilities.jl:417-425) */
    vectorize(v, v->full_b, v->df, v->df2);
    if (v->nks == 0) {
        int rc = orc_project_k(v->full_b, p + q, v->k + v->zkeepn);
        if (rc) return rc;
        unvectorize(v, v->full_b, v->df, v->df2);
        for (int64_t j = 0; j < p; ++j) v->idx[j] = (v->df[j] != 0.0);
        memcpy(v->idc, v->zkeep, q);
        rc = choose(v);
        if (rc) return rc;
    } else {
        /* utilities.jl:427-429: idx is taken from b (all zero) -> empty support */
        int rc = orc_project_group_sparse(v->df, v->group, p, v->J, v->ks, 1);
        if (rc) return rc;
        for (int64_t j = 0; j < p; ++j) v->idx[j] = (v->b[j] != 0.0);
        memset(v->idc, 1, q);
    }
    return ORC_OK;
}

/* iht_stepsize! utilities.jl:722-764 */
static double stepsize(ihtvar *v)
{
    int64_t n = v->n;
    orc_xv_masked(v->x, v->idx, v->df, v->xgk);
    for (int64_t i = 0; i < n; ++i) v->zdf2[i] = 0.0;
    for (int64_t j = 0; j < v->q; ++j) {
        if (!v->idc[j]) continue;
        const double *zj = v->z + j * n; double dj = v->df2[j];
        for (int64_t i = 0; i < n; ++i) v->zdf2[i] += zj[i] * dj;
    }
    for (int64_t i = 0; i < n; ++i) v->xgk[i] += v->zdf2[i];
    for (int64_t i = 0; i < n; ++i) {
        double me = orc_mueta(v->link, v->xb[i] + v->zc[i]);
        v->zdf2[i] = sqrt(me * me / orc_glmvar(v->dist, v->mu[i], v->nb_r)) * v->cv_wts[i];
    }
    double numer = 0.0, denom = 0.0;
    for (int64_t i = 0; i < n; ++i) { v->xgk[i] *= v->zdf2[i]; denom += v->xgk[i] * v->xgk[i]; }
    for (int64_t j = 0; j < v->p; ++j) if (v->idx[j]) numer += v->df[j] * v->df[j];
    for (int64_t j = 0; j < v->q; ++j) if (v->idc[j]) numer += v->df2[j] * v->df2[j];
    double eta = numer / denom;
    if (isinf(eta) || isnan(eta)) eta = 1e-8;
    {   /* diagnostic only (orc_result.eta_cond) */
        double total = 0.0;
        for (int64_t j = 0; j < v->p; ++j) total += v->df[j] * v->df[j];
        for (int64_t j = 0; j < v->q; ++j) total += v->df2[j] * v->df2[j];
        double share = total > 0.0 ? numer / total : 1.0;
        if (share < v->eta_cond) v->eta_cond = share;
    }
    return eta;
}

/* save_prev! utilities.jl:702-712 */
static double save_prev(ihtvar *v, double cur, double best)
{
    memcpy(v->b0, v->b, sizeof(double) * v->p); memcpy(v->idx0, v->idx, v->p);
    memcpy(v->idc0, v->idc, v->q); memcpy(v->c0, v->c, sizeof(double) * v->q);
    if (cur > best) {
        memcpy(v->best_b, v->b, sizeof(double) * v->p);
        memcpy(v->best_c, v->c, sizeof(double) * v->q);
    }
    return cur > best ? cur : best;
}

/* save_best_model! utilities.jl:995-1006: mu = linkinv(xb), genetic part only */
static void save_best_model(ihtvar *v)
{
    memcpy(v->b, v->best_b, sizeof(double) * v->p); memcpy(v->c, v->best_c, sizeof(double) * v->q);
    for (int64_t j = 0; j < v->p; ++j) v->idx[j] = (v->b[j] != 0.0);
    for (int64_t j = 0; j < v->q; ++j) v->idc[j] = (v->c[j] != 0.0);
    update_xb(v);
    for (int64_t i = 0; i < v->n; ++i) v->mu[i] = orc_linkinv(v->link, v->xb[i]);
}

/* check_convergence utilities.jl:953-957 */
static double check_convergence(const ihtvar *v)
{
    double d = 0.0, nb = 0.0;
    for (int64_t j = 0; j < v->p; ++j) {
        double a = fabs(v->b[j] - v->b0[j]); if (a > d) d = a;
        double m = fabs(v->b0[j]); if (m > nb) nb = m;
    }
    for (int64_t j = 0; j < v->q; ++j) {
        double a = fabs(v->c[j] - v->c0[j]); if (a > d) d = a;
        double m = fabs(v->c0[j]); if (m > nb) nb = m;
    }
    return d / (nb + 1.0);
}

/* NegBin nuisance parameter: update_r_MM utilities.jl:158-173 */
static double digamma_(double x)
{
    double r = 0.0;
    while (x < 6.0) { r -= 1.0 / x; x += 1.0; }
    double f = 1.0 / (x * x);
    return r + log(x) - 0.5 / x
         - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132)))));
}
static double trigamma_(double x)
{
    double r = 0.0;
    while (x < 6.0) { r += 1.0 / (x * x); x += 1.0; }
    double f = 1.0 / (x * x);
    return r + 1.0 / x + f / 2
         + f / x * (1.0 / 6 - f * (1.0 / 30 - f * (1.0 / 42 - f * (1.0 / 30 - f * (5.0 / 66)))));
}

static double update_r_mm(const ihtvar *v)
{
    double r = v->nb_r, num = 0.0, den = 0.0;
    for (int64_t i = 0; i < v->n; ++i) {
        for (int64_t j = 0; j <= (int64_t)v->y[i] - 1; ++j) num += r / (r + (double)j);
        den += log(r / (r + v->mu[i]));
    }
    return -num / den;
}

/* update_r_newton utilities.jl:180-247 */
static double update_r_newton(ihtvar *v)
{
    double r = v->nb_r, new_r = 1.0, stepsz = 1.0, saved = v->nb_r;
    for (int it = 0; it < 100; ++it) {
        double dx = 0.0, dx2 = 0.0;
        for (int64_t i = 0; i < v->n; ++i) {
            double yi = v->y[i], mi = v->mu[i];
            dx += -(yi + r) / (mi + r) - log(mi + r) + 1.0 + log(r) + digamma_(r + yi) - digamma_(r);
            dx2 += (yi + r) / ((mi + r) * (mi + r)) - 2.0 / (mi + r) + 1.0 / r
                 + trigamma_(r + yi) - trigamma_(r);
        }
        double inc = (dx2 < 0.0) ? dx / dx2 : dx;
        new_r = r - stepsz * inc;
        v->nb_r = r; double old_logl = loglik(v);
        for (int j = 0; j < 20; ++j) {
            if (new_r <= 0.0) { stepsz /= 2; new_r = r - stepsz * inc; }
            else {
                v->nb_r = new_r; double new_logl = loglik(v);
                if (old_logl >= new_logl) { stepsz /= 2; new_r = r - stepsz * inc; }
                else break;
            }
        }
        if (fabs(r - new_r) <= 1e-6) { v->nb_r = saved; return new_r; }
        r = new_r;
    }
    v->nb_r = saved;
    return r;
}

static void mle_for_r(ihtvar *v)
{
    if (v->est_r == 1) v->nb_r = update_r_mm(v);
    else if (v->est_r == 2) v->nb_r = update_r_newton(v);
}

/* mle_for_r on its own (method 1 = :MM one update, 2 = :Newton to its fixed point): for pinning the restated digamma / trigamma
 * updates against an independent maximiser (tests/test_oracle_golden.py).  wts may be NULL (all ones). */
double orc_mle_for_r(const double *y, const double *mu, const double *wts, int64_t n, double r0, int method)
{
    ihtvar v;
    memset(&v, 0, sizeof(v));
    double *w = NULL;
    if (!wts) { w = dalloc(n); for (int64_t i = 0; i < n; ++i) w[i] = 1.0; }
    v.y = y; v.mu = (double *)mu; v.cv_wts = wts ? (double *)wts : w; v.n = n; v.dist = ORC_NEGBIN; v.link = ORC_LOG; v.nb_r = r0; v.est_r = method;
    mle_for_r(&v);
    free(w);
    return v.nb_r;
}

/* backtrack! utilities.jl:959-973 */
static int backtrack(ihtvar *v, double eta, double *logl)
{
    memcpy(v->b, v->b0, sizeof(double) * v->p); memcpy(v->c, v->c0, sizeof(double) * v->q);
    int rc = gradstep(v, eta); if (rc) return rc;
    update_xb(v); update_mu(v);
    if (v->est_r) mle_for_r(v);
    *logl = loglik(v);
    return ORC_OK;
}

/* diagnostic only (orc_result.bt_cond): how far apart the two loglikelihoods of a backtracking decision are */
static void bt_margin(ihtvar *v, double old_logl, double new_logl)
{
    if (!isfinite(old_logl) || !isfinite(new_logl)) return;
    double scale = fabs(old_logl) > 1e-300 ? fabs(old_logl) : 1e-300;
    double m = fabs(old_logl - new_logl) / scale;
    if (m < v->bt_cond) v->bt_cond = m;
}

/* iht_one_step! fit.jl:213-263 */
static int one_step(ihtvar *v, double old_logl, int nstep, int *bt, double *new_logl_out)
{
    double eta = stepsize(v);
    int rc = gradstep(v, eta); if (rc) return rc;
    update_xb(v); update_mu(v);
    if (v->est_r) mle_for_r(v);
    double new_logl = loglik(v);
    int eta_step = 0;
    bt_margin(v, old_logl, new_logl);
    while (old_logl > new_logl && eta_step < nstep) {   /* _iht_backtrack_ utilities.jl:484 */
        eta /= 2;
        rc = backtrack(v, eta, &new_logl); if (rc) return rc;
        eta_step++;
        if (eta_step < nstep) bt_margin(v, old_logl, new_logl);
    }
    score(v);
    if (isnan(new_logl)) return ORC_NAN_LOGL;
    if (isinf(new_logl)) return ORC_INF_LOGL;
    *bt = eta_step; *new_logl_out = new_logl;
    return ORC_OK;
}

/* fit_iht! fit.jl:145-207 */
/* ------------------------------------------------------------------------ */
/* debias! (utilities.jl:1014-1020): fit(GeneralizedLinearModel, xk, y, d, l) on the support columns --  */
/* no intercept, no covariates, ALL n samples with unit weights (cv_wts is not passed) -- and           */
/* b[idx] = beta0 of that fit.  GLM.jl 1.x is not vendored in the reference; its IRLS is restated from  */
/* its published algorithm (glmfit.jl _fit!: mustart, first WLS on the working response, then           */
/* delbeta!/step-halving until devold - dev < max(rtol*devold, atol), rtol = atol = 1e-6, <= 30 steps,   */
/* minstepfac = 1e-3; Cholesky of X'WX without pivoting).  Parity of this block is unpinned.            */
/* ------------------------------------------------------------------------ */
static double glm_mustart(int dist, double y)
{
    switch (dist) {
    case ORC_BERNOULLI: return (y + 0.5) / 2.0;
    case ORC_POISSON:   return y + 0.1;
    case ORC_NEGBIN:    return y + (y == 0.0 ? 1.0 / 6.0 : 0.0);
    case ORC_GAMMA:     return y == 0.0 ? 0.1 : y;
    default:            return y;
    }
}
static int glm_linkfun(int link, double mu, double *eta)
{
    switch (link) {
    case ORC_IDENTITY:  *eta = mu; return 0;
    case ORC_LOGIT:     *eta = log(mu / (1.0 - mu)); return 0;
    case ORC_LOG:       *eta = log(mu); return 0;
    case ORC_CLOGLOG:   *eta = log(-log1p(-mu)); return 0;
    case ORC_CAUCHIT:   *eta = tan(3.141592653589793 * (mu - 0.5)); return 0;
    case ORC_INVERSE:   *eta = 1.0 / mu; return 0;
    case ORC_INVSQUARE: *eta = 1.0 / (mu * mu); return 0;
    case ORC_SQRT:      *eta = sqrt(mu); return 0;
    default: return 1;                       /* ProbitLink: no closed-form quantile here */
    }
}
/* eta -> mu, working residual, working weight; returns the deviance (NaN/domain failures -> +Inf) */
static double glm_update_mu(const ihtvar *v, const double *eta, double *mu, double *wres, double *wwt)
{
    double dev = 0.0;
    for (int64_t i = 0; i < v->n; ++i) {
        double m = orc_linkinv(v->link, eta[i]), me = orc_mueta(v->link, eta[i]);
        mu[i] = m;
        wres[i] = (v->y[i] - m) / me;
        wwt[i] = me * me / orc_glmvar(v->dist, m, v->nb_r);
        dev += orc_devresid(v->dist, v->y[i], m, v->nb_r);
    }
    return isnan(dev) ? INFINITY : dev;
}
/* delbeta = (X'WX)^-1 X'W r by an unpivoted Cholesky; X is n x k column-major */
static int glm_delbeta(const double *X, int64_t n, int64_t k, const double *w, const double *r, double *A, double *out)
{
    for (int64_t a = 0; a < k; ++a) {
        for (int64_t b = a; b < k; ++b) {
            double s = 0.0;
            for (int64_t i = 0; i < n; ++i) s += X[i + n * a] * w[i] * X[i + n * b];
            A[a + k * b] = s;
        }
        double g = 0.0;
        for (int64_t i = 0; i < n; ++i) g += X[i + n * a] * w[i] * r[i];
        out[a] = g;
    }
    for (int64_t j = 0; j < k; ++j) {                   /* A = U'U in the upper triangle */
        double d = A[j + k * j];
        for (int64_t l = 0; l < j; ++l) d -= A[l + k * j] * A[l + k * j];
        if (!(d > 0.0)) return 1;
        d = sqrt(d); A[j + k * j] = d;
        for (int64_t i = j + 1; i < k; ++i) {
            double s = A[j + k * i];
            for (int64_t l = 0; l < j; ++l) s -= A[l + k * j] * A[l + k * i];
            A[j + k * i] = s / d;
        }
    }
    for (int64_t j = 0; j < k; ++j) {                   /* U' z = g */
        double s = out[j];
        for (int64_t l = 0; l < j; ++l) s -= A[l + k * j] * out[l];
        out[j] = s / A[j + k * j];
    }
    for (int64_t j = k - 1; j >= 0; --j) {              /* U x = z */
        double s = out[j];
        for (int64_t l = j + 1; l < k; ++l) s -= A[j + k * l] * out[l];
        out[j] = s / A[j + k * j];
    }
    return 0;
}
static int debias(ihtvar *v)
{
    int64_t n = v->n, p = v->p, k = 0;
    for (int64_t j = 0; j < p; ++j) k += v->idx[j];
    if (k == 0) return ORC_OK;
    double *X = dalloc(n * k), *eta = dalloc(n), *mu = dalloc(n), *wres = dalloc(n), *wwt = dalloc(n), *t = dalloc(n);
    double *A = dalloc(k * k), *beta0 = dalloc(k), *del = dalloc(k), *lp = dalloc(n);
    int64_t c = 0;
    for (int64_t j = 0; j < p; ++j) if (v->idx[j]) { axpy_col(v->x, j, 1.0, X + n * c); ++c; }   /* dalloc zero-fills */
    int rc = ORC_OK;
    for (int64_t i = 0; i < n; ++i) {                   /* GlmResp: mu = mustart, eta = linkfun(mu) */
        double m = glm_mustart(v->dist, v->y[i]);
        if (glm_linkfun(v->link, m, &eta[i])) { rc = ORC_BAD_ARG; goto done; }
    }
    glm_update_mu(v, eta, mu, wres, wwt);
    for (int64_t i = 0; i < n; ++i) t[i] = eta[i] + wres[i];              /* wrkresp */
    if (glm_delbeta(X, n, k, wwt, t, A, del)) { rc = ORC_BAD_ARG; goto done; }
    for (int64_t a = 0; a < k; ++a) beta0[a] = del[a];                    /* installbeta!(p) from beta0 = 0 */
    for (int64_t i = 0; i < n; ++i) { double s = 0.0; for (int64_t a = 0; a < k; ++a) s += X[i + n * a] * beta0[a]; lp[i] = s; }
    double devold = glm_update_mu(v, lp, mu, wres, wwt);
    int cvg = 0;
    for (int it = 1; it <= 30; ++it) {
        if (glm_delbeta(X, n, k, wwt, wres, A, del)) { rc = ORC_BAD_ARG; goto done; }
        double f = 1.0, dev;
        for (int64_t i = 0; i < n; ++i) { double s = 0.0; for (int64_t a = 0; a < k; ++a) s += X[i + n * a] * (beta0[a] + del[a]); lp[i] = s; }
        dev = glm_update_mu(v, lp, mu, wres, wwt);
        while (dev > devold + 1e-6 * dev) {              /* step halving */
            f /= 2.0;
            if (f < v->db_minstep) v->db_minstep = f;      /* diagnostic only (orc_result.db_minstep) */
            if (!(f > 0.001)) { rc = ORC_BAD_ARG; goto done; }
            for (int64_t i = 0; i < n; ++i) { double s = 0.0; for (int64_t a = 0; a < k; ++a) s += X[i + n * a] * (beta0[a] + f * del[a]); lp[i] = s; }
            dev = glm_update_mu(v, lp, mu, wres, wwt);
        }
        for (int64_t a = 0; a < k; ++a) beta0[a] += f * del[a];
        if (devold - dev < fmax(1e-6 * devold, 1e-6)) { cvg = 1; break; }
        if (!isfinite(dev)) { rc = ORC_BAD_ARG; goto done; }
        devold = dev;
    }
    if (!cvg) { rc = ORC_BAD_ARG; goto done; }
    c = 0;
    for (int64_t j = 0; j < p; ++j) if (v->idx[j]) v->b[j] = beta0[c++];
done:
    free(X); free(eta); free(mu); free(wres); free(wwt); free(t); free(A); free(beta0); free(del); free(lp);
    return rc;
}

/* The GLM refit of debias! on its own: b[j] for idx[j] != 0 <- the coefficients of fit(GeneralizedLinearModel, x[:, idx], y, d, l).
 * Exposed so that tests/test_oracle_golden.py can pin this restated IRLS against an independent GLM solver (scikit-learn). */
int orc_debias_glm(const orc_mat *x, const uint8_t *idx, const double *y, int dist, int link, double nb_r, double *b)
{
    ihtvar v;
    memset(&v, 0, sizeof(v));
    v.db_minstep = 1.0;
    v.x = x; v.y = y; v.n = x->n; v.p = x->p; v.dist = dist; v.link = link; v.nb_r = nb_r;
    v.idx = (uint8_t *)idx; v.b = b;
    return debias(&v);
}

static int fit_loop(ihtvar *v, const orc_params *prm, double *best_logl_out, int64_t *iter_out,
                    double *lt, double *tt, int32_t *bt, int32_t *ntrace)
{
    double next_logl = -INFINITY, best_logl = -INFINITY;
    int64_t mm_iter = 0;
    int32_t nt = 0;
    for (int iter = 1; iter <= prm->max_iter; ++iter) {
        if (iter >= prm->max_iter) {               /* fit.jl:170: max_iter=N takes N-1 steps */
            best_logl = save_prev(v, next_logl, best_logl);
            save_best_model(v);
            mm_iter = iter;
            break;
        }
        best_logl = save_prev(v, next_logl, best_logl);
        int nbt = 0;
        int rc = one_step(v, next_logl, prm->max_step, &nbt, &next_logl);
        if (rc) return rc;
        if (prm->debias && iter >= 5 && memcmp(v->idx, v->idx0, (size_t)v->p) == 0) {   /* fit.jl:188 */
            rc = debias(v);
            if (rc) return rc;
        }
        double scaled = check_convergence(v);
        if (lt) lt[nt] = next_logl;
        if (tt) tt[nt] = scaled;
        if (bt) bt[nt] = nbt;
        nt++;
        if (iter >= prm->min_iter && scaled < prm->tol) {
            best_logl = save_prev(v, next_logl, best_logl);
            save_best_model(v);
            mm_iter = iter;
            break;
        }
    }
    *best_logl_out = best_logl; *iter_out = mm_iter;
    if (ntrace) *ntrace = nt;
    return ORC_OK;
}

static double sample_var(const double *a, int64_t n)
{
    double m = 0.0; for (int64_t i = 0; i < n; ++i) m += a[i]; m /= (double)n;
    double s = 0.0; for (int64_t i = 0; i < n; ++i) s += (a[i] - m) * (a[i] - m);
    return s / (double)(n - 1);
}

static int check_params(const orc_mat *x, const orc_params *prm)
{
    if (prm->J < 0 || prm->max_iter < 0 || prm->max_step < 0) return ORC_BAD_ARG;
    if (!(prm->tol > DBL_EPSILON)) return ORC_BAD_ARG;
    if (x->kind == 0 && !x->center) return ORC_NOT_CENTERED;      /* fit.jl:98 */
    if (prm->est_r && prm->dist != ORC_NEGBIN) return ORC_BAD_ARG; /* fit.jl:93 */
    if (!prm->ks && prm->k < 0) return ORC_BAD_ARG;
    return ORC_OK;
}

int orc_fit_iht(const orc_mat *x, const orc_params *prm, const double *y,
                const double *z, int64_t q, const uint8_t *train, orc_result *res)
{
    int rc = check_params(x, prm); if (rc) return rc;
    ihtvar *v = iv_create(x, prm, y, z, q);
    rc = init_iht_indices(v, train);
    if (!rc) rc = fit_loop(v, prm, &res->logl, &res->iter, res->logl_trace, res->tol_trace,
                           res->bt_trace, &res->n_trace);
    res->db_minstep = v->db_minstep;                 /* (also when the fit ends in an error: debias!'s refit may be that error) */
    if (!rc) {
        res->pve = sample_var(v->mu, v->n) / sample_var(v->y, v->n);  /* pve.jl:22,32 */
        res->nb_r = v->nb_r; res->choose_fired = v->choose_fired; res->eta_cond = v->eta_cond; res->ib_cond = v->ib_cond; res->bt_cond = v->bt_cond;
        if (res->beta) memcpy(res->beta, v->best_b, sizeof(double) * v->p);
        if (res->c) memcpy(res->c, v->best_c, sizeof(double) * v->q);
        if (res->mu) memcpy(res->mu, v->mu, sizeof(double) * v->n);
    }
    iv_destroy(v);
    return rc;
}

/* meanloss cross_validation.jl:304-320 */
static void meanloss(const double *fitloss, int32_t q, int64_t npath, const int32_t *folds,
                     int64_t n, double *loss)
{
    int64_t *ninfold = (int64_t *)calloc((size_t)q, sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) ninfold[folds[i] - 1]++;
    for (int64_t i = 0; i < npath; ++i) loss[i] = 0.0;
    for (int32_t j = 0; j < q; ++j) {
        double w = (double)ninfold[j] / (double)n;
        for (int64_t i = 0; i < npath; ++i) loss[i] += fitloss[i + j * npath] * w;
    }
    free(ninfold);
}

/* cv_iht cross_validation.jl:60-131; predict! :279-286 */
int orc_cv_iht(const orc_mat *x, const orc_params *prm, const double *y,
               const double *z, int64_t q, const int32_t *folds, int32_t nfolds,
               const int64_t *path, int64_t npath, double *mses_raw, double *mse_out)
{
    int rc = check_params(x, prm); if (rc) return rc;
    int64_t n = x->n, kmax = 0;
    for (int64_t i = 0; i < npath; ++i) if (path[i] > kmax) kmax = path[i];
    if (kmax > x->p) return ORC_BAD_ARG;                 /* cross_validation.jl:84 */
    for (int64_t i = 0; i < n; ++i) if (folds[i] < 1 || folds[i] > nfolds) return ORC_BAD_ARG;
    double *mses = (double *)malloc(sizeof(double) * (size_t)(nfolds * npath));
    uint8_t *train = balloc(n);
    orc_params pr = *prm;
    pr.choose = NULL;                                 /* the callback is for single fits (iht_oracle.h) */
    ihtvar *v = iv_create(x, &pr, y, z, q);
    /* Threads.@threads :static (cross_validation.jl:100): thread t of T takes the contiguous block of len (+1 for the first
     * rem threads) fold-major combinations, len, rem = divrem(total, T), with its own V[t] (cross_validation.jl:91): at the
     * start of a block the variable is a fresh one -- init_iht_indices! resets everything else, so only v.d has to be */
    int64_t total = (int64_t)nfolds * npath, T = prm->cv_threads > 1 ? prm->cv_threads : 1;
    int64_t blen = total / T, brem = total % T;
    for (int32_t fold = 1; fold <= nfolds && !rc; ++fold) {
        for (int64_t ik = 0; ik < npath && !rc; ++ik) {
            int64_t combo = (int64_t)(fold - 1) * npath + ik;
            for (int64_t t = 0; t < T; ++t)
                if (combo == t * blen + (t < brem ? t : brem)) { v->nb_r = prm->nb_r; break; }
            for (int64_t i = 0; i < n; ++i) train[i] = (folds[i] != fold);
            v->k = path[ik];
            rc = init_iht_indices(v, train); if (rc) break;
            double bl; int64_t it;
            rc = fit_loop(v, &pr, &bl, &it, NULL, NULL, NULL, NULL); if (rc) break;
            for (int64_t i = 0; i < n; ++i) v->cv_wts[i] = train[i] ? 0.0 : 1.0;
            update_xb(v); update_mu(v);
            mses[(fold - 1) * npath + ik] =
                orc_deviance(v->dist, v->nb_r, v->y, v->mu, v->cv_wts, n);
        }
    }
    if (!rc) {
        if (mses_raw) memcpy(mses_raw, mses, sizeof(double) * (size_t)(nfolds * npath));
        meanloss(mses, nfolds, npath, folds, n, mse_out);
    }
    iv_destroy(v); free(train); free(mses);
    return rc;
}

#include "iht_oracle_mv.inc"
