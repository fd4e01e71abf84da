#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in 0 1; do
  if [ $f = 1 ]; then export RH_FORCE_SELECT=1; fi
  echo "== force select: $f"
  timeout -k 10 200 python3 tools/pipeline_breakdown.py 2>&1 | grep "streams=4" || exit 1
  timeout -k 10 200 python3 tools/bench_stages.py --cfg 2 --infer --iters 200 2>&1 | grep -E "^(sort|all)" || exit 1
done
