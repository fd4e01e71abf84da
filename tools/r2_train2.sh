#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -u -m pytest tests/test_gpu_prefetch.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -1
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench_x.json 2> gpurun_out/r2_bench_1.err || { tail -20 gpurun_out/r2_bench_1.err; exit 1; }
python - <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r2_bench_x.json").read().strip().splitlines()[-1])
print("ms/step %.4f p10/p90 %s stream %.4f alone %.4f" % (d["ms_per_step"], [round(x,4) for x in d["ms_per_step_p10_p90"]], d["roofline"]["kernel_ms_mean"], d["roofline"]["stream_alone_kernel_ms_mean"]))
PY
