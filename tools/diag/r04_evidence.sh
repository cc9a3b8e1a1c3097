#!/bin/bash
# round-4 evidence pass (one gpurun call per configuration): bench + rocprofv3 kernel trace (stats, timeline, windows) + the two
# PMC passes.  Outputs under gpurun_out/r04_*; the summaries to be judged are copied into profiles/ by tools/diag/r04_collect.py
# (run on the box, so that the bench line that follows reports roofline.traffic from the passes just made, and again in the
# build container to commit the copies).
# usage: r04_evidence.sh <c2|c5|c3> <git head> [tests]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=r04
cd $R && mkdir -p gpurun_out
cfg=$1; head=$2
if [ "$3" = "tests" ]; then
  python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/${tag}_gpu_tests.log 2>&1; rc=$?
  tail -n 4 gpurun_out/${tag}_gpu_tests.log
  [ $rc -eq 0 ] || exit $rc
  cp gpurun_out/parity_errors.json gpurun_out/${tag}_parity_errors.json
fi
prof() {  # <name> <bench args...>
  name=$1; shift
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_${name}_prof -o r -- python3 $R/bench.py "$@" --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${tag}_${name}_prof.log 2>&1 || { tail -n 5 $R/gpurun_out/${tag}_${name}_prof.log; return 1; }
  timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_${name}_pmcf -o r -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_${name}_pmcf.log 2>&1 || return 1
  timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_${name}_pmcw -o r -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_${name}_pmcw.log 2>&1 || return 1
  cd $R
  f=$(find gpurun_out/${tag}_${name}_prof -name "*kernel_trace.csv" | head -1)
  [ "$name" = c2 ] && python tools/diag/timeline.py $f 45 > gpurun_out/${tag}_${name}_timeline.txt 2>&1
  [ "$name" = c2 ] && python tools/diag/window.py $f > gpurun_out/${tag}_${name}_windows.txt 2>&1 || true
  cp $(find gpurun_out/${tag}_${name}_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_${name}_kernel_stats.csv
  python tools/summarize_pmc.py $(find gpurun_out/${tag}_${name}_pmcf -name "*counter_collection.csv" | head -1) $(find gpurun_out/${tag}_${name}_pmcw -name "*counter_collection.csv" | head -1) 40 > gpurun_out/${tag}_${name}_pmc_traffic.csv
  rm -rf gpurun_out/${tag}_${name}_prof gpurun_out/${tag}_${name}_pmcf gpurun_out/${tag}_${name}_pmcw
  head -6 gpurun_out/${tag}_${name}_pmc_traffic.csv
}
case "$cfg" in
c2)
  prof c2 --no-fp32 --no-families || exit 1
  python tools/diag/r04_collect.py c2 $head
  # the restorer's staged input at 64 columns (round 3) for comparison with the packed 40 columns: LSTM launch + staging family
  NPPC_X_PACKED=0 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32 > gpurun_out/${tag}_bench_c2_xpacked0.json 2> gpurun_out/${tag}_bench_c2_xpacked0.log
  timeout -k 10 900 python bench.py > gpurun_out/${tag}_bench_c2_bf16.json 2> gpurun_out/${tag}_bench_c2_bf16.log || { tail -n 5 gpurun_out/${tag}_bench_c2_bf16.log; exit 1; }
  cut -c1-600 gpurun_out/${tag}_bench_c2_bf16.json ;;
c5)
  prof c5 --config c5 --no-fp32 --no-families || exit 1
  python tools/diag/r04_collect.py c5 $head
  timeout -k 10 500 python bench.py --config c5 --no-fp32 --no-cpu-baseline > gpurun_out/${tag}_bench_c5_bf16.json 2> gpurun_out/${tag}_bench_c5_bf16.log || { tail -n 5 gpurun_out/${tag}_bench_c5_bf16.log; exit 1; }
  cut -c1-400 gpurun_out/${tag}_bench_c5_bf16.json ;;
c3)
  prof c3 --config c3 || exit 1
  python tools/diag/r04_collect.py c3 $head
  timeout -k 10 400 python bench.py --config c3 > gpurun_out/${tag}_bench_c3_bf16.json 2> gpurun_out/${tag}_bench_c3_bf16.log || { tail -n 5 gpurun_out/${tag}_bench_c3_bf16.log; exit 1; }
  cut -c1-400 gpurun_out/${tag}_bench_c3_bf16.json
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_fp32_prof -o r -- python3 $R/bench.py --precision fp32 --steps 2 --warmup 1 --no-cpu-baseline --no-families > $R/gpurun_out/${tag}_fp32_prof.log 2>&1
  cd $R
  cp $(find gpurun_out/${tag}_fp32_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_c2_fp32_kernel_stats.csv && rm -rf gpurun_out/${tag}_fp32_prof ;;
esac
echo EVIDENCE_DONE
