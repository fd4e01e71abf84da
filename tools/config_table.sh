#!/bin/bash
# usage (on the GPU box): tools/config_table.sh > gpurun_out/configs.txt
# Kernel averages (rocprofv3 --kernel-trace --stats) of the loss and post-processing stages for the
# benchmark configurations: B,size,C,M = 2,512,600,10 | 16,608,80,20 | 8,800,601,8 | 4,1024,80,256.
cd /tmp && export TMPDIR=/tmp
for shape in 2,512,600,10 16,608,80,20 8,800,601,8 4,1024,80,256; do
  for mode in "" "--infer"; do
    d=$GRAFT_REPO_ROOT/gpurun_out/cfgtab; rm -rf $d; mkdir -p $d
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- \
      python3 $GRAFT_REPO_ROOT/tools/bench_stages.py --shape $shape --iters 40 $mode > $d/run.log 2>&1 || exit 1
    python3 - "$shape" "$mode" $d <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[3] + "/*/*kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "rh::" in r["Name"] and "anchors_level" not in r["Name"]]
print(f"B,size,C,M = {sys.argv[1]:16s} {'post-processing' if sys.argv[2] else 'loss':16s}",
      " | ".join(f"{r['Name'].split('(')[0].split('rh::')[-1][:28]} {float(r['AverageNs'])/1e3:.1f} us" for r in rows))
PY
  done
done
