#!/bin/bash
# usage (GPU box): tools/kernel_gaps.sh [bench]  -> idle time between consecutive launches on one stream
# (with "bench": of bench.py's replayed training steps instead of the tool's own launch trains)
d=$GRAFT_REPO_ROOT/gpurun_out/gaps; rm -rf $d; mkdir -p $d
cd /tmp && export TMPDIR=/tmp
if [ "$1" = "bench" ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --repeats 10 --no-cpu-baseline > $d/run.log 2>&1 || { tail -5 $d/run.log; exit 1; }
else
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/kernel_gaps.py run > $d/run.log 2>&1 || { tail -5 $d/run.log; exit 1; }
fi
python3 $GRAFT_REPO_ROOT/tools/kernel_gaps.py gaps $d
rm -rf $d/*/
