#!/bin/bash
# Counter passes over the fused 12-residual pass (separate rocprofv3 --pmc runs; FETCH_SIZE and WRITE_SIZE apart).
# usage: tools/pmc_fused.sh TAG MULTI_VARIANT [RESIDUALS]  -> gpurun_out/pmc_TAG_<set>/pmc_results.db
tag=$1; mv=$2; mm=${3:-12}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  out=$R/gpurun_out/pmc_${tag}_$i
  rm -rf $out
  rocprofv3 --pmc $set --kernel-trace -d $out -o pmc -- python3 $R/tools/pmc_fused.py $mv $mm 2 > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "set $i ($set): rc=$? $(grep 'ms/pass' $R/gpurun_out/pmc_${tag}_$i.log)"
done
