#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -u -m pytest tests/test_gpu_pipeline.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench_x.json 2> gpurun_out/r2_bench_1.err || { tail -20 gpurun_out/r2_bench_1.err; exit 1; }
python - <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r2_bench_x.json").read().strip().splitlines()[-1])
dn=d["decode_nms"]
print("ms/step %.4f | decode %.1f us, pipeline class %.1f us, single %.1f us" % (d["ms_per_step"], dn["ms_per_batch"]*1e3, dn["pipeline_class_ms_per_batch"]*1e3, dn["single_stream_ms_per_batch"]*1e3))
PY
