#!/bin/bash
# usage (on the GPU box): tools/ab_assign.sh <variant> ... -- <B,size,C,M> ...
# Loss stage timings (tools/bench_stages.py) of library variants built by tools/build_variant.sh
# ("base" = the regular build) at the given shapes.
libs=(); shapes=(); seen=0
for a in "$@"; do if [ "$a" = "--" ]; then seen=1; elif [ $seen = 0 ]; then libs+=($a); else shapes+=($a); fi; done
for lib in "${libs[@]}"; do
  for shape in "${shapes[@]}"; do
    echo "== lib=$lib shape=$shape"
    if [ "$lib" != base ]; then export RH_LIB=pytorch-retinanet_amd/csrc/build/lib_$lib.so; else unset RH_LIB; fi
    timeout -k 10 120 python3 tools/bench_stages.py --shape $shape --iters 100 | grep -E "^(assign|stream|all|step|assignment)" || exit 1
  done
done
