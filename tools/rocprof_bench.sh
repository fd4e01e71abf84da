#!/bin/bash
# usage: tools/rocprof_bench.sh <outdir-under-gpurun_out> <bench.py args...>
# rocprofv3 kernel-trace + stats of bench.py itself; run on the GPU box.
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $out/run.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, shutil
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
shutil.copy(f, sys.argv[1] + '/kernel_stats.csv')
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} min={float(r['MinNs'])/1e3:9.2f} max={float(r['MaxNs'])/1e3:9.2f}")
PY
rm -rf $out/*/
