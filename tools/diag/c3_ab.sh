#!/bin/bash
# inpainting unit tests, then alternating C3 bench runs with / without the LDS-DMA convolution kernel (GPU box)
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_inpaint_gpu.py -x -q > gpurun_out/r04_t_i.log 2>&1
tail -4 gpurun_out/r04_t_i.log | cut -c1-300
grep -q failed gpurun_out/r04_t_i.log && exit 1
grep -q passed gpurun_out/r04_t_i.log || exit 1
for v in ${@:-1 0 1 0}; do
  NPPC_CONV_DMA=$v timeout -k 10 200 python bench.py --config c3 --no-cpu-baseline > gpurun_out/c3_dma$v.json 2> gpurun_out/c3_dma$v.err || { tail -5 gpurun_out/c3_dma$v.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/c3_dma$v.json')); print('NPPC_CONV_DMA=$v', d['ms_per_step'], d['value'])"
done
