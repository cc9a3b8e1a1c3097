#!/bin/bash
# round-2 evidence pass: full -m gpu suite, C2 bench + rocprof kernel trace + PMC passes, LSTM backward phase stamps,
# C5 / C3 bench lines.  Outputs under gpurun_out/<tag>_*; copy what is to be judged into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; sel=${2:-tests}
cd $R && mkdir -p gpurun_out
python -m pytest $sel -m gpu -x -q --durations=8 > gpurun_out/${tag}_gpu_tests.log 2>&1; rc=$?
tail -n 14 gpurun_out/${tag}_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/diag/r02_profile.sh $tag pmc || exit 1
python tools/diag/stamp_bwd2.py > gpurun_out/${tag}_lstm_bwd_phase_stamps.txt 2>&1; grep -v amdgpu gpurun_out/${tag}_lstm_bwd_phase_stamps.txt | head -16
python bench.py --config c2 > gpurun_out/${tag}_bench_c2_full.json 2> gpurun_out/${tag}_bench_c2_full.log || exit 1
python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-fp32 > gpurun_out/${tag}_bench_c5.json 2> gpurun_out/${tag}_bench_c5.log || exit 1
python bench.py --config c3 > gpurun_out/${tag}_bench_c3.json 2> gpurun_out/${tag}_bench_c3.log || exit 1
python - <<PY
import json
for c in ("c2_full", "c5", "c3"):
    d = json.load(open("gpurun_out/${tag}_bench_%s.json" % c))
    print(c, round(d["ms_per_step"], 2), "ms/step", round(d["value"]), "frames/s", d["roofline"]["kernel"], d["roofline"].get("frac"))
PY
echo EVIDENCE_DONE
