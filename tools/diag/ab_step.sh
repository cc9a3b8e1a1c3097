#!/bin/bash
# A/B of env / library variants inside the C2 train step on ONE box: alternating short bench runs (ms/step of each)
# usage: ab_step.sh "<env A>" "<env B>" [rounds]      e.g.  ab_step.sh "NPPC_MID_FINISH_DEFER=0" "NPPC_MID_FINISH_DEFER=1" 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
A="$1"; B="$2"; n=${3:-3}
for i in $(seq 1 $n); do
  for v in "$A" "$B"; do
    ms=$(env $v python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-families 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $i  [$v]  $ms ms/step"
  done
done
