/*
 * mendeliht_hip_probes.h -- extra entry points of the MEASUREMENT build (libmendeliht_hip_probes.so = the product's
 * sources compiled with -DMIH_PROBES).  Not part of the drop-in boundary and not exported by libmendeliht_hip.so:
 * tools/ and the "this switch changes nothing" tests use them to sweep launch shapes, to cross-check the product's
 * kernels bit for bit against the round-1 kernel families, and to run timing probes.  The measurement build also reads
 * the MENDELIHT_* A/B environment switches (XTV_MAX_OPS, XTV_SLICES, XTV_NO_HALF, CV_LANES, CV_NO_MERGE,
 * CV_NO_INIT_SHARE, CV_NO_COOP, COOP_SPIN_US, CV_ASSIGN, CV_TRACE, INGEST_TRACE, TRACE_ETA, NO_SPIN, NO_ARENA, TOPK_RADIX8, XV_MULTI,
 * and the round-6 ones: RES_FORCE_ABORT_ES, LANE_BATCHED, CV_PASS_ORDER, WORKER_PRIORITY, TOPK_HOST_FINISH, CV_ORDER, LANE_CU_RESERVE, DEBIAS_TRACE = the IRLS
 * iterates of debias!'s GLM refit on stderr); the product reads none
 * of them.
 * These knobs are process-wide on purpose (one measurement at a time).
 */
#ifndef MENDELIHT_HIP_PROBES_H
#define MENDELIHT_HIP_PROBES_H
#include "mendeliht_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* single-operand X'r kernel of every subsequent call: -1 = product default, 0..15 = round 1's per-wave-load shapes */
int mih_probe_set_xtv_variant(int variant);
/* launch shape of the LDS-shared / ring kernels: 0 = product defaults; 1..15 = round-1 register-staged shapes (6 / 9: its
 * defaults); 20.. = 32x32x64 ring shapes; 40.. = 16x16x128 ring shapes; 7, 8, 15, 30..33, 49 = timing probes whose output
 * is NOT X'r */
int mih_probe_set_xtv_multi_variant(int variant);
/* B operands fused per pass of the register-staged kernels (1, 2 or 4) */
int mih_probe_set_max_fused(int max_nr);
/* X'r of the first ms[i] columns of R (n x mcap, column-major) for i = 0 .. nms-1, one call after the other on ONE fused-pass
 * workspace sized for mcap residuals -- what a lock-step lane does from round to round.  OUT: the results back to back
 * (p * ms[0] doubles, then p * ms[1], ...).  For the test that a pass ignores what an earlier pass with another residual
 * count left in the unused digit columns of its last operand (flat packing, csrc/xtv.hip). */
int mih_probe_xtv_sequence(const mih_mat *h, const double *R, int mcap, const int *ms, int nms, int digits, double *OUT);
#ifdef __cplusplus
}
#endif
#endif
