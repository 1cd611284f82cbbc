#!/bin/bash
# PMC passes over the single-kernel micro-benchmark (tools/bench_conv.py); usage: tools/pmc_conv.sh <which> [bench_conv args]
# writes gpurun_out/pmc_<which>_<k>/ ; summarise with tools/pmc_table.py
which=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
k=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $R/gpurun_out/pmc_${which}_$k
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_${which}_$k -o p -- python $R/tools/bench_conv.py --which $which --iters 3 "$@" > $R/gpurun_out/pmc_${which}_$k.log 2>&1
  k=$((k+1))
done
