#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in 0 1; do
  export NPPC_CONV_DMA=$mode
  echo "=== NPPC_CONV_DMA=$mode"
  python3 $R/tools/diag/conv_pmc.py 2>&1 | grep NPPC_CONV
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" \
             "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/conv_pmc_$i -o r -- python3 $R/tools/diag/conv_pmc.py > $R/gpurun_out/conv_pmc_$i.log 2>&1 || { tail -3 $R/gpurun_out/conv_pmc_$i.log; echo "pass $i ($set) failed"; rm -rf $R/gpurun_out/conv_pmc_$i; continue; }
    f=$(find $R/gpurun_out/conv_pmc_$i -name "*counter_collection.csv" | head -1)
    python3 $R/tools/diag/conv_pmc.py --summarize $f
    grep NPPC_CONV $R/gpurun_out/conv_pmc_$i.log | sed 's/^/      (profiled run: /; s/$/)/'
    rm -rf $R/gpurun_out/conv_pmc_$i
  done
done
