#!/bin/bash
# bit-identity of multi-trait fits between two builds (tools/check_xv_multi.py), then the multivariate A/B
MENDELIHT_HIP_LIB=$PWD/tools/_old.so python tools/check_xv_multi.py /tmp/xvm_a.npz
MENDELIHT_HIP_LIB=$PWD/mendeliht.jl_amd/libmendeliht_hip.so python tools/check_xv_multi.py /tmp/xvm_b.npz
python - <<'PY'
import numpy as np
a, b = np.load("/tmp/xvm_a.npz"), np.load("/tmp/xvm_b.npz")
print("check_xv_multi (r = 10, 7, 3, 12):", {k: bool(np.array_equal(a[k].view(np.uint64), b[k].view(np.uint64))) for k in a.files})
PY
python tools/ab_mv.py tools/_mid.so mendeliht.jl_amd/libmendeliht_hip.so 2>&1 | tail -1 | cut -c1-400
