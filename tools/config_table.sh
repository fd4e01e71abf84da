#!/bin/bash
# usage (on the GPU box): tools/config_table.sh > gpurun_out/configs.txt
# Kernel averages (rocprofv3 --kernel-trace --stats) of the loss and post-processing stages for the
# benchmark configurations: B,size,C,M = 2,512,600,10 | 16,608,80,20 | 8,800,601,8 | 4,1024,80,256
# (the last with its top-1000 pre-NMS selection and room for its ~5600 candidates per image).
cd /tmp && export TMPDIR=/tmp
# The 600-class shapes are post-processed twice: with the inference logits of the other shapes
# (mean -7: with 600 classes ~9 600 / ~23 500 anchors per image pass 0.05, the lists are cut at 4 096
# and the NMS runs on dense random boxes: a saturated case) and with mean -8 (~2 000 candidates per
# image, the rate the GPU tests use for these shapes).
for shape in 2,512,600,10 16,608,80,20 8,800,601,8 4,1024,80,256; do
  for mode in "" "--infer" "--infer --mu -8"; do
    extra=""
    if [ "$mode" = "--infer --mu -8" ]; then case $shape in *,600,*|*,601,*) ;; *) continue;; esac; fi
    if [ "$mode" = "--infer" ] && [ "$shape" = "4,1024,80,256" ]; then extra="--topk 1000 --cap 8192"; fi
    d=$GRAFT_REPO_ROOT/gpurun_out/cfgtab; rm -rf $d; mkdir -p $d
    # whole call by HIP events in a run of its own: under the profiler the host enqueues slower
    # than the GPU works through these few short kernels
    timeout -k 10 120 python3 $GRAFT_REPO_ROOT/tools/bench_stages.py --shape $shape --iters 200 $mode $extra > $d/plain.log 2>&1 || exit 1
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- \
      python3 $GRAFT_REPO_ROOT/tools/bench_stages.py --shape $shape --iters 40 $mode $extra > $d/run.log 2>&1 || exit 1
    python3 - "$shape" "$mode" $d <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[3] + "/*/*kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "rh::" in r["Name"] and "anchors_level" not in r["Name"]]
import re
log = open(sys.argv[3] + "/plain.log").read()
m = re.search(r"^all\s+mean\s+([0-9.]+) us\s+min\s+([0-9.]+) us", log, re.M)
total = sum(float(r["AverageNs"]) for r in rows) / 1e3
whole = f" | kernels together {total:.1f} us | whole call, one stream, HIP events: mean {m.group(1)} min {m.group(2)} us" if m else ""
st = re.search(r"^step, look-ahead, one launch: mean\s+([0-9.]+) us.*= ([0-9.]+) of the HBM peak", log, re.M)
if st:
    whole += f" | step, look-ahead, one launch (trains of 20): {st.group(1)} us = {st.group(2)} of the HBM peak"
print(f"B,size,C,M = {sys.argv[1]:16s} {('post-processing ' + sys.argv[2].replace('--infer', '').strip()).strip() if sys.argv[2] else 'loss':24s}",
      " | ".join(f"{r['Name'].split('(')[0].split('rh::')[-1][:28]} {float(r['AverageNs'])/1e3:.1f} us" for r in rows) + whole)
PY
  done
done
