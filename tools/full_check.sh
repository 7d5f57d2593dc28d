#!/bin/bash
# The round's acceptance run on one GPU box: the whole -m gpu suite, the driver's bench command, and bench.py --gpus 2 through the
# stand-in exchange (both ranks on the one device: a functional check of the N > 1 path and its diagnostics, not a scaling number).
# usage: tools/full_check.sh TAG
tag=$1
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -14 | tee gpurun_out/${tag}_gpu_suite_tail.txt
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 300 gpurun_out/${tag}_bench.err
gcc -O2 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/fake_rccl.c -o tests/libfake_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt -lpthread -ldl
/opt/rocm/bin/hipcc --genco --offload-arch=gfx950 -O2 tests/fake_rccl_kernels.hip -o tests/fake_rccl_kernels.hsaco
MIH_FAKE_RCCL_SLOT_MB=64 MIH_BENCH_BACKEND=gloo MIH_BENCH_ONE_DEVICE=1 MENDELIHT_RCCL_LIB=$PWD/tests/libfake_rccl.so timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cv > gpurun_out/${tag}_bench_n2_standin.json 2> gpurun_out/${tag}_bench_n2_standin.err
tail -c 300 gpurun_out/${tag}_bench_n2_standin.err
python - <<PY
import json
for f in ("gpurun_out/${tag}_bench.json", "gpurun_out/${tag}_bench_n2_standin.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    c = d["config"]
    print(f, d["value"], d["ms_per_step"], {k: c.get(k) for k in ("host_small_kernels_and_exchange_ms_per_step", "chain_ms_per_step", "exchange_ms_per_step", "resident_steps", "rccl_ranks_seen", "librccl", "collectives_rank0")})
    print("  A/B", c.get("host_driven_steps_same_box"))
    print("  mv", d.get("mv"))
    print("  cv", {k: d.get("cv_iht", {}).get(k) for k in ("cv_iht_s", "fused_passes", "residuals_scored_by_passes", "best_k")})
    print("  cpu", {k: (d.get("cpu_baseline") or {}).get(k) for k in ("value", "cores", "sample")})
    print("  failed", d.get("failed"))
PY
