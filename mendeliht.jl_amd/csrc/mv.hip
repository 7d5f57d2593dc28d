// placeholder, replaced below
#include "common.h"
using namespace mih;
extern "C" int mih_fit_mv(const mih_mat *, const mih_fit_params *, const double *, int64_t, const double *, int64_t, const uint8_t *, mih_mv_result *) { set_error("not implemented"); return MIH_BAD_ARG; }
extern "C" int mih_cv_mv(const mih_mat *, const mih_fit_params *, const double *, int64_t, const double *, int64_t, const int32_t *, int32_t, const int64_t *, int64_t, int32_t, int32_t, double *) { set_error("not implemented"); return MIH_BAD_ARG; }
