#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/rocprof_bench.sh r2_benchprof --steps 20 --warmup 5 --no-cpu-baseline --repeats 10
