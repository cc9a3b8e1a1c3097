export R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_fullsize_properties_gpu.py tests/test_lstm_gpu.py -m gpu -x -q --timeout 250 > gpurun_out/t_fin.log 2>&1; tail -2 gpurun_out/t_fin.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_v14f -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_v14f.log 2>&1 && \
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_v14w -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_v14w.log 2>&1 && \
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v14 -o r -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_v14.log 2>&1
cd $R && python tools/summarize_pmc.py gpurun_out/pmc_v14f/r_counter_collection.csv gpurun_out/pmc_v14w/r_counter_collection.csv 14 > profiles/r01_bench_c2_bf16_pmc_traffic.csv && cp profiles/r01_bench_c2_bf16_pmc_traffic.csv gpurun_out/pmc_traffic_v14.csv && python bench.py > gpurun_out/bench_v14.json 2> gpurun_out/bench_v14.log
rm -f gpurun_out/prof_v14/r_kernel_trace.csv
grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/bench_v14.json | head -1
