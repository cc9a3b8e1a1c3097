#!/bin/bash
# usage: r02_run.sh <tag> <pytest selection or "none"> [bench configs...]   (GPU box; outputs under gpurun_out/)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
tag=$1; sel=$2; shift 2
if [ "$sel" != "none" ]; then
  python -m pytest $sel -m gpu -x -q -s --durations=12 > gpurun_out/${tag}_tests.log 2>&1
  rc=$?
  tail -n 25 gpurun_out/${tag}_tests.log
  [ $rc -eq 0 ] || exit $rc
fi
for cfg in "$@"; do
  extra=""
  [ "$cfg" = "c5" ] && extra="--steps 5 --warmup 2 --no-cpu-baseline --no-fp32"
  [ "$cfg" = "c2q" ] && { extra="--no-cpu-baseline --no-fp32"; cfgarg="c2"; } || cfgarg=$cfg
  python bench.py --config $cfgarg $extra > gpurun_out/${tag}_bench_${cfg}.json 2> gpurun_out/${tag}_bench_${cfg}.log || { tail -n 20 gpurun_out/${tag}_bench_${cfg}.log; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench_${cfg}.json"))
r = d["roofline"]
print("${cfg}", round(d["ms_per_step"], 2), "ms/step", round(d["value"]), "frames/s | dominant", r["kernel"], r.get("frac"), "| step frac", (r.get("step") or {}).get("frac"))
print("   ranked:", r.get("ranked_ms_per_step"))
print("   families:", {k: (round(v["ms_per_step"], 3), round(v["achieved_GBps"])) for k, v in (r.get("hbm_families") or {}).items()})
print("   fp32:", d.get("fp32"), "cpu:", d.get("cpu_baseline"))
PY
done
echo RUN_DONE
