#!/bin/bash
# usage (on the GPU box): tools/b1_profile.sh
# rocprofv3 kernel averages of the post-processing stages for ONE 608 px image per call.
cd /tmp && export TMPDIR=/tmp
d=$GRAFT_REPO_ROOT/gpurun_out/b1prof; rm -rf $d; mkdir -p $d
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/bench_stages.py --shape 1,608,80,20 --infer --iters 200 > $d/run.log 2>&1 || { tail -3 $d/run.log; exit 1; }
grep -E "^(decode|sort|nms|all|counts)" $d/run.log
python3 - $d <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "rh::" in r["Name"]: print(r["Name"].split("(")[0][-40:], r["Calls"], round(float(r["AverageNs"])/1e3,2), round(float(r["MinNs"])/1e3,2))
PY
