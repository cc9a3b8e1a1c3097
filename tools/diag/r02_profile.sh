#!/bin/bash
# round-2 profile pass: quick C2 bench, then rocprofv3 kernel trace (+ timeline of one steady-state step)
# usage: r02_profile.sh <tag> [pmc]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1
mkdir -p $R/gpurun_out
cd $R && python bench.py --no-cpu-baseline --no-fp32 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.log || { tail -n 20 gpurun_out/${tag}_bench.log; exit 1; }
python - <<PY
import json
d = json.load(open("$R/gpurun_out/${tag}_bench.json")); r = d["roofline"]
print("c2", round(d["ms_per_step"], 2), "ms/step | dominant", r["kernel"], r.get("frac"), "| step frac", r["step"]["frac"])
print("   ranked:", r.get("ranked_ms_per_step"))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -o r -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32 --no-families > $R/gpurun_out/${tag}_prof.log 2>&1 || { tail -n 5 $R/gpurun_out/${tag}_prof.log; exit 1; }
cd $R
f=$(find gpurun_out/${tag}_prof -name "*kernel_trace.csv" | head -1)
python tools/diag/timeline.py $f 40 > gpurun_out/${tag}_timeline.txt 2>&1
head -60 gpurun_out/${tag}_timeline.txt
s=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1)
cp $s gpurun_out/${tag}_kernel_stats.csv
rm -f $f
if [ "$2" = "pmc" ]; then
  cd /tmp
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_pmcf -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32 --no-families > $R/gpurun_out/${tag}_pmcf.log 2>&1 && \
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_pmcw -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32 --no-families > $R/gpurun_out/${tag}_pmcw.log 2>&1
  cd $R
  python tools/summarize_pmc.py $(find gpurun_out/${tag}_pmcf -name "*counter_collection.csv" | head -1) $(find gpurun_out/${tag}_pmcw -name "*counter_collection.csv" | head -1) 40 > gpurun_out/${tag}_pmc_traffic.csv
  rm -rf gpurun_out/${tag}_pmcf gpurun_out/${tag}_pmcw
  head -12 gpurun_out/${tag}_pmc_traffic.csv
fi
echo PROFILE_DONE
