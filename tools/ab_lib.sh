#!/bin/bash
# usage (on the GPU box): tools/ab_lib.sh <other/libretinahead.so> <bench_stages args...>
# Kernel averages (rocprofv3 --kernel-trace --stats) of tools/bench_stages.py for another build of
# the library and for the in-tree one, alternating twice, so that both are seen on the same box.
other=$(readlink -f "$1"); shift
mine=$GRAFT_REPO_ROOT/pytorch-retinanet_amd/retinahead/libretinahead.so
cd /tmp && export TMPDIR=/tmp
for v in other in-tree other in-tree; do
  if [ $v = other ]; then lib=$other; else lib=$mine; fi
  d=$GRAFT_REPO_ROOT/gpurun_out/ab_$v; rm -rf $d; mkdir -p $d
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- \
    python3 $GRAFT_REPO_ROOT/tools/try_lib.py $lib $GRAFT_REPO_ROOT/tools/bench_stages.py "$@" > $d/run.log 2>&1 || exit 1
  python3 - $v $d <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[2] + "/*/*kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "rh::" in r["Name"] and "anchors_level" not in r["Name"]]
print(f"{sys.argv[1]:8s}", " | ".join(f"{r['Name'].split('(')[0][-22:]} {float(r['AverageNs'])/1e3:.2f} (min {float(r['MinNs'])/1e3:.2f})" for r in rows))
PY
done
