#!/bin/bash
# round-2 GPU pass 1: the -m gpu suite, then the three driver-visible bench lines (C2 headline, C5, C3)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s --durations=15 > gpurun_out/r02_gpu_tests.log 2>&1
rc=$?
tail -n 30 gpurun_out/r02_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 10 --warmup 3 > gpurun_out/r02_bench_c2_a.json 2> gpurun_out/r02_bench_c2_a.log || { tail -n 20 gpurun_out/r02_bench_c2_a.log; exit 1; }
python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02_bench_c5_a.json 2> gpurun_out/r02_bench_c5_a.log || { tail -n 20 gpurun_out/r02_bench_c5_a.log; exit 1; }
python bench.py --config c3 --steps 10 --warmup 3 > gpurun_out/r02_bench_c3_a.json 2> gpurun_out/r02_bench_c3_a.log || { tail -n 20 gpurun_out/r02_bench_c3_a.log; exit 1; }
echo BENCHES_DONE
