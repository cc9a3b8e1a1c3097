#!/bin/bash
# one counter pass (LDS) over tools/diag/lstm_pmc.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT --output-format csv -d $R/gpurun_out/lstm_pmc_l -o r -- python3 $R/tools/diag/lstm_pmc.py > $R/gpurun_out/lstm_pmc_l.log 2>&1 || { tail -5 $R/gpurun_out/lstm_pmc_l.log; exit 1; }
f=$(find $R/gpurun_out/lstm_pmc_l -name "*counter_collection.csv" | head -1)
python3 $R/tools/diag/lstm_pmc.py --summarize $f > $R/gpurun_out/lstm_pmc_l.txt
rm -rf $R/gpurun_out/lstm_pmc_l
cat $R/gpurun_out/lstm_pmc_l.txt
