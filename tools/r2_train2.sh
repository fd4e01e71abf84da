#!/bin/bash
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r2_fin
mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/time_finalize.py > $out/run.log 2>&1 || { tail $out/run.log; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} min={float(r['MinNs'])/1e3:9.2f} max={float(r['MaxNs'])/1e3:9.2f}")
PY
rm -rf $out/*/
