#!/bin/bash
# per-layer convolution times of sizing builds of conv_tiled_kernel (tools/diag/libconv_<V>.so: -DCONV_DIAG_<V>) next to the product
cd $GRAFT_REPO_ROOT
for v in "$@"; do NPPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/diag/libconv_$v.so timeout -k 10 200 python tools/diag/c3_layers.py > gpurun_out/c3_layers_$v.txt 2>&1; done
timeout -k 10 200 python tools/diag/c3_layers.py > gpurun_out/c3_layers_base.txt 2>&1
for v in "$@" base; do echo -n "$v "; grep "^{" gpurun_out/c3_layers_$v.txt; done
