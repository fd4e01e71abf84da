#!/bin/bash
# oracle check + tests for the in-tree build, then the pipeline breakdown and stage averages for
# builds with 1, 2 (in-tree) and 4 keys per lane in the rank sort
cd $GRAFT_REPO_ROOT
timeout -k 5 90 python3 -u tools/dbg_nms.py > gpurun_out/r2_dbg.txt 2>&1 || { tail gpurun_out/r2_dbg.txt; exit 1; }
echo "oracle check: $(grep -c ' ok' gpurun_out/r2_dbg.txt) ok, $(grep 'cap=4096' gpurun_out/r2_dbg.txt | grep -c DIFF) DIFF at capacity 4096, $(grep '^nms\|^[0-9]' gpurun_out/r2_dbg.txt | grep -c DIFF) DIFF plain"
timeout -k 10 500 python -u -m pytest tests -x -q --timeout 120 --timeout-method thread -m gpu > gpurun_out/r2_tests.log 2>&1 || { tail -40 gpurun_out/r2_tests.log; exit 1; }
tail -1 gpurun_out/r2_tests.log
for v in kpl1 in-tree kpl4; do
  lib=pytorch-retinanet_amd/csrc/build/libretinahead_$v.so
  [ $v = in-tree ] && lib=pytorch-retinanet_amd/retinahead/libretinahead.so
  echo "== $v"
  timeout -k 10 200 python3 tools/try_lib.py $lib tools/pipeline_breakdown.py 2>&1 | grep "streams=4" || exit 1
  timeout -k 10 200 python3 tools/try_lib.py $lib tools/bench_stages.py --cfg 2 --infer --iters 200 2>&1 | grep -E "^(sort|all)" || exit 1
done
