#!/bin/bash
# A/B: bench.py ms/step under environment settings given as arguments "NAME=VAL[,NAME=VAL]" (one run each, same box)
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "$@"; do
  envs=$(echo "$cfg" | tr ',' ' ')
  [ "$cfg" = "base" ] && envs=""
  ms=$(env $envs python bench.py --no-cpu-baseline --no-fp32 --no-families 2>/dev/null | python -c "import json,sys; print(round(json.load(sys.stdin)['ms_per_step'],3))")
  echo "$cfg: $ms ms/step"
done
