#!/bin/bash
# kernel trace of bench.py -> tools/diag/window.py listing: prof_window.sh <tag> [bench flags...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_prof -o r -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32 --no-families "$@" > $R/gpurun_out/${tag}_prof.log 2>&1 || { tail -n 5 $R/gpurun_out/${tag}_prof.log; exit 1; }
cd $R
f=$(find gpurun_out/${tag}_prof -name "*kernel_trace.csv" | head -1)
python tools/diag/window.py $f > gpurun_out/${tag}_window.txt 2>&1
rm -f $f
cat gpurun_out/${tag}_window.txt
