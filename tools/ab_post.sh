#!/bin/bash
# usage (on the GPU box): tools/ab_post.sh <variant> ...
# Post-processing timings (tools/try_variant.py) of library variants built by tools/build_variant.sh
# ("base" = the regular build) and the rocprofv3 kernel averages of the stage tool for each.
# AB_ARGS: the stage tool's workload (default "--cfg 2"; e.g. "--shape 2,512,600,10"), AB_QUICK=1: kernel
# averages only.
root=$GRAFT_REPO_ROOT
for lib in "$@"; do
  if [ "$lib" = base ]; then so=$root/pytorch-retinanet_amd/retinahead/libretinahead.so; else so=$root/pytorch-retinanet_amd/csrc/build/lib_$lib.so; fi
  echo "== $lib"
  if [ -z "$AB_QUICK" ]; then timeout -k 10 200 python3 $root/tools/try_variant.py $so 2>/dev/null | grep -v amdgpu || exit 1; fi
  d=$root/gpurun_out/abpost; rm -rf $d; mkdir -p $d
  ( cd /tmp && export TMPDIR=/tmp && RH_LIB=$so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $root/tools/bench_stages.py ${AB_ARGS:---cfg 2} --infer --iters 60 > $d/run.log 2>&1 ) || { tail -3 $d/run.log; exit 1; }
  python3 - $d <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "rh::" in r["Name"] and "anchors_level" not in r["Name"]]
print("   kernels:", " | ".join(f"{r['Name'].split('(')[0].split('rh::')[-1][:22]} {float(r['AverageNs'])/1e3:.2f}" for r in rows))
PY
done
