#!/bin/bash
# usage (on the GPU box): tools/r3_final.sh a|b
# a: GPU test suite (writes gpurun_out/tolerance_floors.json), three driver-style bench runs, the
#    rocprofv3 kernel statistics of bench.py and of the stage tool.
# b: PMC traffic passes, the configuration table, the scalar-load reproducer, the in-kernel timelines.
cd $GRAFT_REPO_ROOT
o=gpurun_out
if [ "$1" = a ]; then
  timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $o/r03_gpu_tests.txt 2>&1 || { tail -20 $o/r03_gpu_tests.txt; exit 1; }
  tail -1 $o/r03_gpu_tests.txt
  for i in 1 2 3; do
    timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $o/r03_bench_line_run$i.json 2> $o/r03_bench_run$i.err || { tail -5 $o/r03_bench_run$i.err; exit 1; }
    echo "bench run $i done"
  done
  tools/collect_profiles.sh r03 > $o/collect_r03.txt 2>&1 || { tail -5 $o/collect_r03.txt; exit 1; }
  echo "profiles collected"
else
  timeout -k 10 300 python bench.py > $o/r03_bench_line_default.json 2> $o/r03_bench_default.err || { tail -5 $o/r03_bench_default.err; exit 1; }
  echo "default bench run done"
  timeout -k 10 400 tools/pmc_traffic.sh > $o/pmc_r03.txt 2>&1 || { tail -5 $o/pmc_r03.txt; exit 1; }
  echo "pmc done"
  timeout -k 10 900 tools/config_table.sh > $o/configs.txt 2> $o/configs.err || { tail -5 $o/configs.err; exit 1; }
  echo "config table done"
  timeout -k 10 200 tools/ubench/sload_boundary > $o/r03_sload_boundary.txt 2>&1 || { tail -5 $o/r03_sload_boundary.txt; exit 1; }
  echo "reproducer done"
  timeout -k 10 120 python3 tools/assign_timeline.py 16,608,80,20 > $o/r03_assign_timeline.txt 2>/dev/null && \
  timeout -k 10 120 python3 tools/assign_timeline.py 4,1024,80,256 >> $o/r03_assign_timeline.txt 2>/dev/null && \
  timeout -k 10 120 python3 tools/nms_timeline.py > $o/r03_nms_timeline.txt 2>/dev/null
  echo "timelines done"
fi
