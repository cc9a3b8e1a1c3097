#!/bin/bash
# stand-alone time of the fused TCN middle backward (C2 shape) -- tools/diag/bench_tcn_mid.py -- plus its unit tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
python $R/tools/diag/bench_tcn_mid.py
