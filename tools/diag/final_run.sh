export R=$GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests -m gpu -x -q --timeout 400 > gpurun_out/t_all4.log 2>&1; tail -2 gpurun_out/t_all4.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_v13f -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_v13f.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_v13w -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_v13w.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v13 -o r -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_v13.log 2>&1
cd $R && python tools/summarize_pmc.py gpurun_out/pmc_v13f/r_counter_collection.csv gpurun_out/pmc_v13w/r_counter_collection.csv 14 > profiles/r01_bench_c2_bf16_pmc_traffic.csv && cp profiles/r01_bench_c2_bf16_pmc_traffic.csv gpurun_out/pmc_traffic_v13.csv && python bench.py > gpurun_out/bench_v13.json 2> gpurun_out/bench_v13.log
rm -f gpurun_out/prof_v13/r_kernel_trace.csv
grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/bench_v13.json | head -1
