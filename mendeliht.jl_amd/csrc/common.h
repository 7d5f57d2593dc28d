// common.h -- internal declarations shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/mendeliht_hip.h"

namespace mih {

void set_error(const char *fmt, ...);
int  hip_fail(hipError_t e, const char *what, const char *file, int line);

#define MIH_HIP(expr)                                                         \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) return ::mih::hip_fail(e_, #expr, __FILE__, __LINE__); \
    } while (0)

#define MIH_TRY(expr)                      \
    do {                                   \
        int rc_ = (expr);                  \
        if (rc_ != MIH_OK) return rc_;     \
    } while (0)

// Rows per r-tile chunk: one wave-load of 64 dwords = 64 lanes x 16 genotypes.
constexpr int kChunkRows = 1024;

inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// RAII device buffer
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    int alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e != hipSuccess) { set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e)); p = nullptr; return e == hipErrorOutOfMemory ? MIH_OOM : MIH_HIP_ERROR; }
        n = count;
        return MIH_OK;
    }
};

}  // namespace mih

// Device-resident design matrix.
//
// kind 0 (SnpLinAlg): 2-bit dosage codes, column-major, `stride_dw` dwords per SNP
// column (a multiple of 64 dwords = 256 B so every wave-load is one aligned
// 256-B segment).  Device code is the DOSAGE itself (00->0, 01->1, 10->2; the
// PLINK code is remapped once at upload) and missing entries are stored as 0
// with their row numbers kept in a per-column CSR side list, so the inner
// loop of X'r decodes with one bit-field op and never branches on missingness.
// Pad rows beyond n are 0.
struct mih_mat {
    int       kind = 0;            // 0 snp, 1 dense
    int       device = 0;
    int64_t   n = 0, p = 0;
    int       center = 1, scale = 1, impute = 1;
    int64_t   stride_dw = 0;       // dwords per column
    int64_t   n_pad = 0;           // rows covered by stride (= stride_dw*16)
    uint32_t *X = nullptr;         // p * stride_dw dwords
    double   *mu = nullptr, *sinv = nullptr;   // p
    int64_t  *miss_ptr = nullptr;  // p+1
    int32_t  *miss_row = nullptr;  // total_missing
    int64_t   total_missing = 0;
    double   *D = nullptr;         // dense n x p
    hipStream_t stream = nullptr;  // for the stand-alone linear-algebra entry points
};

namespace mih {

// ---- X'r ---------------------------------------------------------------------
struct XtvWork {            // scratch for one in-flight X'r
    DevBuf<double> rperm;   // m * n_perm  (tile-permuted residuals)
    DevBuf<double> partial; // splits * m * p raw dots
    DevBuf<double> sums;    // m * 2 : sum(r) (and spare)
    int64_t n_perm = 0;
    int m_cap = 0, splits_cap = 0;
};
int  xtv_work_init(const mih_mat *h, XtvWork &w, int m);
// r_dev: m vectors of length n (natural order, column-major n x m) on device; out_dev p x m.
int  xtv_device(const mih_mat *h, XtvWork &w, const double *r_dev, int m, double *out_dev, hipStream_t s);
// Same but r already permuted into w.rperm and sums filled (fused producer path).
int  xtv_device_preperm(const mih_mat *h, XtvWork &w, const double *r_dev, int m, double *out_dev, hipStream_t s);
// device-side helper: position of row i in the tile-permuted residual vector
// (a lane of the X'r kernel holds `lw` consecutive dwords = 16*lw rows of a 1024*lw-row superchunk)
__host__ __device__ inline int64_t rperm_pos(int64_t i, int lw)
{
    int64_t sc = i / (1024 * lw);
    int w = (int)(i - sc * (1024 * lw));
    int l = w / (16 * lw);          // lane
    int d = (w / 16) % lw;          // dword within the lane's load
    int s = w & 15;                 // slot within the dword
    return (((sc * lw + d) * 8 + (s >> 1)) * 64 + l) * 2 + (s & 1);
}
int  xtv_current_lw();
int  xtv_num_variants();
extern int g_xtv_variant;

// ---- X[:,S] v -----------------------------------------------------------------
struct XvWork {
    DevBuf<double> partial;   // groups * n
    DevBuf<double> coefA, coefB;  // per support column: sinv*val, -mu*sinv*val
    int64_t cap = 0; int groups = 0;
};
int  xv_work_init(const mih_mat *h, XvWork &w, int64_t max_nnz);
// out[i] = sum_t x[i, idx[t]] * val[t]; idx/val on device; clamp20 applies clamp!(out,-20,20)
int  xv_sparse_device(const mih_mat *h, XvWork &w, const int64_t *idx_dev, const double *val_dev,
                      int64_t nnz, double *out_dev, int clamp20, hipStream_t s);

// ---- top-k --------------------------------------------------------------------
struct TopkWork {
    DevBuf<uint32_t> hist;     // 256 bins
    DevBuf<uint64_t> state;    // [0]=prefix, [1]=remaining k, [2]=threshold bits, [3]=count_ge
    DevBuf<int64_t>  sel_idx;  // compacted survivors
    DevBuf<double>   sel_val;
    DevBuf<uint32_t> sel_cnt;
    int64_t cap = 0;
};
int  topk_work_init(TopkWork &w, int64_t max_keep);
// In-place project_k! on a device vector; returns threshold and survivors (sorted by index) on host.
int  topk_project_device(double *x_dev, int64_t len, int64_t k, TopkWork &w, hipStream_t s,
                         std::vector<int64_t> &idx_out, std::vector<double> &val_out);

}  // namespace mih
