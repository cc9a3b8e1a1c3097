#!/bin/bash
# kernel-trace averages of selected kernels under two environments (GPU box): kstat_ab.sh "<env A>" "<env B>" <kernel name regex>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$1" "$2"; do
  tag=$(echo "$v" | tr -c 'A-Za-z0-9' '_')
  export $v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$tag -o r -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp32 --no-families > $R/gpurun_out/ks_$tag.log 2>&1 || { tail -3 $R/gpurun_out/ks_$tag.log; exit 1; }
  unset ${v%%=*}
  f=$(find $R/gpurun_out/ks_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $v"
  python3 - "$f" "$3" <<'PY'
import csv,re,sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r['Name']):
        print(f"{r['Name'][28:100]:72s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
  rm -rf $R/gpurun_out/ks_$tag
done
