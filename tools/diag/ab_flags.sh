#!/bin/bash
# A/B: bench.py ms/step under command-line flag sets given as arguments ("base" = none; commas separate flags of one run)
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "$@"; do
  flags=$(echo "$cfg" | tr ',' ' ')
  [ "$cfg" = "base" ] && flags=""
  ms=$(python bench.py --no-cpu-baseline --no-fp32 --no-families $flags 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_step'],3), d['lstm_handoff_timeouts'], round(d['config']['objective_last'],6))")
  echo "$cfg: $ms"
done
