#!/bin/bash
# counter passes over tools/diag/lstm_pmc.py (GPU box); results under gpurun_out/lstm_pmc_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/lstm_pmc_$i -o r -- python3 $R/tools/diag/lstm_pmc.py > $R/gpurun_out/lstm_pmc_$i.log 2>&1 || { tail -5 $R/gpurun_out/lstm_pmc_$i.log; echo "pass $i failed"; continue; }
  f=$(find $R/gpurun_out/lstm_pmc_$i -name "*counter_collection.csv" | head -1)
  python3 $R/tools/diag/lstm_pmc.py --summarize $f > $R/gpurun_out/lstm_pmc_$i.txt
  rm -rf $R/gpurun_out/lstm_pmc_$i
  cat $R/gpurun_out/lstm_pmc_$i.txt
done
