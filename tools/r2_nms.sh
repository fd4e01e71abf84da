#!/bin/bash
# oracle check on small inputs, GPU suite, in-kernel timeline, stages under rocprofv3, multi-stream pipeline
cd $GRAFT_REPO_ROOT
timeout -k 5 90 python3 -u tools/dbg_nms.py > gpurun_out/r2_dbg.txt 2>&1 || { tail gpurun_out/r2_dbg.txt; exit 1; }
echo "oracle check: $(grep -c ' ok' gpurun_out/r2_dbg.txt) ok, $(grep 'cap=4096' gpurun_out/r2_dbg.txt | grep -c DIFF) DIFF at capacity 4096, $(grep '^nms\|^[0-9]' gpurun_out/r2_dbg.txt | grep -c DIFF) DIFF plain"
timeout -k 10 500 python -u -m pytest tests -x -q --timeout 120 --timeout-method thread -m gpu > gpurun_out/r2_tests.log 2>&1 || { tail -40 gpurun_out/r2_tests.log; exit 1; }
tail -2 gpurun_out/r2_tests.log
timeout -k 10 120 python3 tools/nms_timeline.py > gpurun_out/r2_tl.txt 2>&1 || { tail gpurun_out/r2_tl.txt; exit 1; }
cat gpurun_out/r2_tl.txt
bash tools/rocprof_stages.sh r2_inf --cfg 2 --infer --iters 100 > gpurun_out/r2_inf_stats.txt 2>&1 || { tail -20 gpurun_out/r2_inf/run.log; exit 1; }
head -4 gpurun_out/r2_inf_stats.txt; grep -E "^(decode|sort|nms|all|counts)" gpurun_out/r2_inf/run.log
rm -rf gpurun_out/r2_inf/vm* gpurun_out/r2_inf/*/
timeout -k 10 200 python3 tools/bench_pipeline.py > gpurun_out/r2_pipeline.txt 2>&1 || { tail -20 gpurun_out/r2_pipeline.txt; exit 1; }
cat gpurun_out/r2_pipeline.txt
