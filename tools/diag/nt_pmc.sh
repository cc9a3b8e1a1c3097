#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/nt_pmc_$i -o r -- python3 $R/tools/diag/nt_pmc.py > $R/gpurun_out/nt_pmc_$i.log 2>&1 || { tail -3 $R/gpurun_out/nt_pmc_$i.log; echo "pass $i ($set) failed"; rm -rf $R/gpurun_out/nt_pmc_$i; continue; }
  f=$(find $R/gpurun_out/nt_pmc_$i -name "*counter_collection.csv" | head -1)
  python3 $R/tools/diag/nt_pmc.py --summarize $f
  rm -rf $R/gpurun_out/nt_pmc_$i
done
