#include "common.h"
using namespace mih;
extern "C" int mih_project_group_sparse(double *, const int64_t *, int64_t, int64_t, const int64_t *, int) { set_error("not implemented"); return MIH_BAD_ARG; }
