#!/bin/bash
# usage: tools/rocprof_pmc.sh <outdir-under-gpurun_out> "<counters>" <bench_stages args...>
# One rocprofv3 --pmc pass (kernel-trace only) over tools/bench_stages.py; prints per-kernel means.
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
ctr="$1"; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/bench_stages.py "$@" > $out/run.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, {c: (sum(v) / len(v), len(v)) for c, v in d.items()})
PY
