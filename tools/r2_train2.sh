#!/bin/bash
cd $GRAFT_REPO_ROOT
for per in 1 4; do
cp pytorch-retinanet_amd/retinahead/libretinahead.so /tmp/keep.so
cp pytorch-retinanet_amd/csrc/build/libretinahead_per$per.so pytorch-retinanet_amd/retinahead/libretinahead.so
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench_per$per.json 2> gpurun_out/r2_bench_1.err || { tail -20 gpurun_out/r2_bench_1.err; exit 1; }
cp /tmp/keep.so pytorch-retinanet_amd/retinahead/libretinahead.so
python - per$per <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r2_bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], "ms/step %.4f p10/p90 %s stream %.4f assign %.4f" % (d["ms_per_step"], d["ms_per_step_p10_p90"], d["roofline"]["kernel_ms_mean"], d["kernels_ms"]["assign_stage"]))
PY
done
