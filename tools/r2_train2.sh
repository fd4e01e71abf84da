#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in in-tree prio in-tree prio; do
  lib=pytorch-retinanet_amd/csrc/build/libretinahead_$v.so
  [ $v = in-tree ] && lib=pytorch-retinanet_amd/retinahead/libretinahead.so
  timeout -k 10 200 python tools/try_lib.py $lib bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench_x.json 2> gpurun_out/r2_bench_1.err || { tail -20 gpurun_out/r2_bench_1.err; exit 1; }
python - $v <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r2_bench_x.json").read().strip().splitlines()[-1])
dn=d["decode_nms"]
print(sys.argv[1], "ms/step %.4f p10/p90 %s | stream %.4f alone %.4f | decode %.1f %s" % (d["ms_per_step"], [round(x,4) for x in d["ms_per_step_p10_p90"]], d["roofline"]["kernel_ms_mean"], d["roofline"]["stream_alone_kernel_ms_mean"], dn["ms_per_batch"]*1e3, [round(x*1e3,1) for x in dn["stream_set_trials_ms_per_batch"]]))
PY
done
