#!/bin/bash
# first GPU call of the round: the new tests, the whole GPU suite, a kernel trace of the bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c1
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread -m gpu > gpurun_out/r2c1/noalloc.log 2>&1 || { tail -30 gpurun_out/r2c1/noalloc.log; exit 1; }
tail -3 gpurun_out/r2c1/noalloc.log
timeout -k 10 600 python -u -m pytest tests -x -q --timeout 240 --timeout-method thread -m gpu > gpurun_out/r2c1/tests.log 2>&1 || { tail -40 gpurun_out/r2c1/tests.log; exit 1; }
tail -3 gpurun_out/r2c1/tests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2c1/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2c1/bench_trace.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/r2c1/bench_trace.log; exit 1; }
python3 $GRAFT_REPO_ROOT/tools/trace_gaps.py $GRAFT_REPO_ROOT/gpurun_out/r2c1/trace 50 > $GRAFT_REPO_ROOT/gpurun_out/r2c1/gaps.txt
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r2c1/trace
head -60 $GRAFT_REPO_ROOT/gpurun_out/r2c1/gaps.txt
