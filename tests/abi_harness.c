/* abi_harness.c -- the drop-in boundary exercised from plain C: include/mendeliht_hip.h is compiled as C
 * (gcc -std=c99 -Wall -I include), the library is dlopen'ed, every declared entry point is resolved, and -- when a GPU
 * is present -- the reference's recorded run (docs/src/man/examples.md:230-267: iht("normal", 7, Normal,
 * covariates="covariates.txt", phenotypes=6)) is reproduced through mih_snp_create / mih_fit_iht / mih_cv_iht exactly
 * as a Julia `ccall` binding would drive them (julia/MendelIHTHip.jl), without the ctypes mirrors of the test suite.
 *
 * usage: abi_harness LIB.so FIXTURE_DIR [symbols-only]
 * exit code 0 = pass; 77 = symbols verified but no GPU (the fit was skipped). */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mendeliht_hip.h"

#define N 1000
#define P 10000
#define Q 2

static void *lib;
static void *sym(const char *name)
{
    void *f = dlsym(lib, name);
    if (!f) { fprintf(stderr, "missing symbol %s\n", name); exit(2); }
    return f;
}

/* every `int mih_*(` declaration of the header (kept in step by tests/test_abi_cpu.py::test_c_harness) */
static const char *const kSymbols[] = {
#include "abi_symbols.inc"
};

static int read_doubles(const char *path, double *out, int rows, int cols, char sep)
{
    FILE *f = fopen(path, "r");
    if (!f) { perror(path); return -1; }
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j) {
            if (fscanf(f, "%lf", &out[(size_t)j * rows + i]) != 1) { fclose(f); return -1; }   /* column-major */
            if (j + 1 < cols) { int ch = fgetc(f); if (ch != sep && ch != ' ') ungetc(ch, f); }
        }
    fclose(f);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s LIB.so FIXTURE_DIR [symbols-only]\n", argv[0]); return 2; }
    lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    for (size_t i = 0; i < sizeof(kSymbols) / sizeof(kSymbols[0]); ++i) (void)sym(kSymbols[i]);
    printf("%zu entry points resolved\n", sizeof(kSymbols) / sizeof(kSymbols[0]));

    int (*abi_sizes)(int64_t *, int32_t) = (int (*)(int64_t *, int32_t))sym("mih_abi_sizes");
    int64_t sz[4];
    if (abi_sizes(sz, 4) != MIH_OK) return 3;
    if (sz[0] != (int64_t)sizeof(mih_fit_params) || sz[1] != (int64_t)sizeof(mih_fit_result) ||
        sz[2] != (int64_t)sizeof(mih_mv_result) || sz[3] != (int64_t)sizeof(mih_comm)) {
        fprintf(stderr, "struct sizes differ between header (as C) and library: %lld %lld %lld %lld\n",
                (long long)sz[0], (long long)sz[1], (long long)sz[2], (long long)sz[3]);
        return 3;
    }
    int (*device_count)(int *) = (int (*)(int *))sym("mih_device_count");
    int ndev = 0;
    (void)device_count(&ndev);
    if (argc > 3 || ndev < 1) { printf("no GPU (or symbols-only): fit skipped\n"); return ndev < 1 && argc <= 3 ? 77 : 0; }

    /* ---- the G1 fit ---------------------------------------------------------------------------------------------- */
    char path[1024];
    const size_t stride = (N + 3) / 4;
    uint8_t *bed = (uint8_t *)malloc(stride * P);
    snprintf(path, sizeof path, "%s/normal.bed", argv[2]);
    FILE *f = fopen(path, "rb");
    unsigned char magic[3];
    if (!f || fread(magic, 1, 3, f) != 3 || magic[0] != 0x6c || magic[1] != 0x1b || magic[2] != 0x01 ||
        fread(bed, 1, stride * P, f) != stride * P) { fprintf(stderr, "cannot read %s\n", path); return 4; }
    fclose(f);
    double *y = (double *)malloc(sizeof(double) * N), *z = (double *)malloc(sizeof(double) * N * Q);
    snprintf(path, sizeof path, "%s/normal_y_fam6.txt", argv[2]);
    if (read_doubles(path, y, N, 1, ' ')) return 4;
    snprintf(path, sizeof path, "%s/covariates.txt", argv[2]);
    if (read_doubles(path, z, N, Q, ',')) return 4;
    /* standardize!(@view z[:, 2:end]) (src/utilities.jl:494-530; wrapper.jl:245): sample s.d. */
    double m = 0, s2 = 0;
    for (int i = 0; i < N; ++i) m += z[N + i];
    m /= N;
    for (int i = 0; i < N; ++i) s2 += (z[N + i] - m) * (z[N + i] - m);
    const double sd = sqrt(s2 / (N - 1));
    for (int i = 0; i < N; ++i) z[N + i] = (z[N + i] - m) / sd;

    int (*snp_create)(const uint8_t *, int64_t, int64_t, int64_t, int, int, int, int, int, mih_mat **) =
        (int (*)(const uint8_t *, int64_t, int64_t, int64_t, int, int, int, int, int, mih_mat **))sym("mih_snp_create");
    int (*fit_iht)(const mih_mat *, const mih_fit_params *, const double *, const double *, int64_t, const uint8_t *, mih_fit_result *) =
        (int (*)(const mih_mat *, const mih_fit_params *, const double *, const double *, int64_t, const uint8_t *, mih_fit_result *))sym("mih_fit_iht");
    int (*cv_iht)(const mih_mat *, const mih_fit_params *, const double *, const double *, int64_t, const int32_t *, int32_t,
                  const int64_t *, int64_t, int32_t, int32_t, double *) =
        (int (*)(const mih_mat *, const mih_fit_params *, const double *, const double *, int64_t, const int32_t *, int32_t,
                 const int64_t *, int64_t, int32_t, int32_t, double *))sym("mih_cv_iht");
    int (*meanloss)(const double *, const int32_t *, int64_t, int32_t, int64_t, double *) =
        (int (*)(const double *, const int32_t *, int64_t, int32_t, int64_t, double *))sym("mih_cv_meanloss");
    int (*mat_destroy)(mih_mat *) = (int (*)(mih_mat *))sym("mih_mat_destroy");
    int (*last_error)(char *, size_t) = (int (*)(char *, size_t))sym("mih_last_error");

    mih_mat *x = NULL;
    char err[512];
    if (snp_create(bed, N, P, (int64_t)stride, 1, 1, 1, 64, 0, &x) != MIH_OK) { last_error(err, sizeof err); fprintf(stderr, "mih_snp_create: %s\n", err); return 5; }
    mih_fit_params prm;
    memset(&prm, 0, sizeof prm);
    prm.k = 7; prm.J = 1; prm.dist = MIH_NORMAL; prm.link = MIH_IDENTITY; prm.nb_r = 1.0; prm.tol = 1e-4;
    prm.max_iter = 200; prm.min_iter = 5; prm.max_step = 3; prm.est_r = MIH_ESTR_NONE;
    double *beta = (double *)calloc(P, sizeof(double)), c[Q], lt[201], tt[201];
    int32_t bt[201];
    mih_fit_result res;
    memset(&res, 0, sizeof res);
    res.beta = beta; res.c = c; res.logl_trace = lt; res.tol_trace = tt; res.bt_trace = bt;
    if (fit_iht(x, &prm, y, z, Q, NULL, &res) != MIH_OK) { last_error(err, sizeof err); fprintf(stderr, "mih_fit_iht: %s\n", err); return 5; }

    static const double g_logl[5] = {-1403.6085154464329, -1397.922430744325, -1397.8812223841496, -1397.8807476657355, -1397.8807416751808};
    static const int g_pos[7] = {3137, 4246, 4717, 6290, 7755, 8375, 9415};
    static const double g_beta[7] = {0.424376, 0.52343, 0.922857, -0.677832, -0.542983, -0.792813, -2.17998};
    static const double g_c[2] = {1.65223, 0.749865};
    int bad = 0;
    if (res.iter != 5 || res.n_trace != 5) { fprintf(stderr, "iterations %lld (expected 5)\n", (long long)res.iter); bad = 1; }
    for (int i = 0; i < 5 && i < res.n_trace; ++i)
        if (fabs(lt[i] - g_logl[i]) > 1e-11 * fabs(g_logl[i]) || bt[i] != 0) { fprintf(stderr, "logl[%d] = %.15g\n", i, lt[i]); bad = 1; }
    int nnz = 0;
    for (int j = 0; j < P; ++j) if (beta[j] != 0.0) {
        if (nnz >= 7 || j + 1 != g_pos[nnz] || fabs(beta[j] - g_beta[nnz]) > 5e-6 * fabs(g_beta[nnz])) { fprintf(stderr, "beta[%d] = %g unexpected\n", j + 1, beta[j]); bad = 1; }
        nnz++;
    }
    if (nnz != 7) bad = 1;
    for (int l = 0; l < Q; ++l) if (fabs(c[l] - g_c[l]) > 5e-6 * fabs(g_c[l])) { fprintf(stderr, "c[%d] = %g\n", l, c[l]); bad = 1; }
    if (fabs(res.pve - 0.8343751445053728) > 1e-9) { fprintf(stderr, "pve = %.12g\n", res.pve); bad = 1; }
    printf("fit_iht through the C ABI: %lld iterations, logl %.13f, %d non-zero SNPs, pve %.10f\n", (long long)res.iter, res.logl, nnz, res.pve);

    /* a small cross-validation with explicit folds: the two-rank split adds up to the one-rank grid bit for bit */
    int32_t *folds = (int32_t *)malloc(sizeof(int32_t) * N);
    for (int i = 0; i < N; ++i) folds[i] = 1 + (int32_t)((i * 2654435761u >> 7) % 3);
    const int64_t pathv[4] = {3, 5, 7, 9};
    double raw[12], r0[12], r1[12], mse[4];
    prm.k = 1; prm.max_iter = 100;
    if (cv_iht(x, &prm, y, z, Q, folds, 3, pathv, 4, 0, 1, raw) || cv_iht(x, &prm, y, z, Q, folds, 3, pathv, 4, 0, 2, r0) ||
        cv_iht(x, &prm, y, z, Q, folds, 3, pathv, 4, 1, 2, r1) || meanloss(raw, folds, N, 3, 4, mse)) {
        last_error(err, sizeof err); fprintf(stderr, "mih_cv_iht: %s\n", err); return 5;
    }
    for (int i = 0; i < 12; ++i) if (r0[i] + r1[i] != raw[i] || !(raw[i] > 0.0)) { fprintf(stderr, "cv grid entry %d: %g + %g vs %g\n", i, r0[i], r1[i], raw[i]); bad = 1; }
    int best = 0;
    for (int i = 1; i < 4; ++i) if (mse[i] < mse[best]) best = i;
    printf("cv_iht through the C ABI: losses %.6f %.6f %.6f %.6f, best k = %lld\n", mse[0], mse[1], mse[2], mse[3], (long long)pathv[best]);
    if (pathv[best] != 7) { fprintf(stderr, "cross-validation should pick the true model size 7\n"); bad = 1; }
    mat_destroy(x);
    free(bed); free(y); free(z); free(beta); free(folds);
    printf(bad ? "FAIL\n" : "PASS\n");
    return bad ? 1 : 0;
}
