#!/bin/bash
# usage (on the GPU box): tools/ab_finalize.sh <variant> ... -- <B,size,C,M> ...
# rocprofv3 kernel average of finalize_kernel for library variants (tools/build_variant.sh; "base" = regular build)
libs=(); shapes=(); seen=0
for a in "$@"; do if [ "$a" = "--" ]; then seen=1; elif [ $seen = 0 ]; then libs+=($a); else shapes+=($a); fi; done
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "${libs[@]}"; do
  for shape in "${shapes[@]}"; do
    if [ "$lib" != base ]; then export RH_LIB=$root/pytorch-retinanet_amd/csrc/build/lib_$lib.so; else unset RH_LIB; fi
    d=$root/gpurun_out/abfin; rm -rf $d; mkdir -p $d
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $root/tools/time_finalize.py $shape > $d/run.log 2>&1 || { tail -3 $d/run.log; exit 1; }
    echo "lib=$lib shape=$shape: $(grep finalize_kernel $d/*/*kernel_stats.csv | awk -F, '{printf "finalize_kernel avg %.2f us over %d launches", $(NF-4)/1000, $(NF-6)}')"
  done
done
