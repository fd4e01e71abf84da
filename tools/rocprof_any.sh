#!/bin/bash
# usage: tools/rocprof_any.sh <outdir-under-gpurun_out> <python script + args>   (kernel trace, per-call listing)
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 "$@" > $out/run.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
t = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r['Start_Timestamp']))
for r in rows:
    n = r['Kernel_Name']
    if n.startswith('rh::') or 'rh::' in n[:12]:
        print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.2f} us  grid {r['Grid_Size_X']:>8s} lds {r['LDS_Block_Size']:>6s}  {n[:50]}")
PY
