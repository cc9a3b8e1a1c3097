source tools/diag/sweep_tcn.sh
export KERNELS="dwconv_bwd gn_bwd_reduce gn_prelu_bwd colsum head_bwd_w"
run base && run rpb32 NPPC_TCN_RPB=32
cat $R/gpurun_out/sweep.txt
