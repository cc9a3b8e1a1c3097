export R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { # name, env...
  name=$1; shift
  env_str="$*"
  for kv in $env_str; do export $kv; done
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sw_$name -o r -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/sw_$name.log 2>&1 || return 1
  for kv in $env_str; do unset ${kv%%=*}; done
  echo "== $name" >> $R/gpurun_out/sweep.txt
  python3 $R/tools/diag/kstat.py $R/gpurun_out/sw_$name/r_kernel_stats.csv $KERNELS >> $R/gpurun_out/sweep.txt
  (grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/sw_$name.log || true) >> $R/gpurun_out/sweep.txt
  rm -f $R/gpurun_out/sw_$name/r_kernel_trace.csv
}
rm -f $R/gpurun_out/sweep.txt
