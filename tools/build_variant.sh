#!/bin/bash
# usage: tools/build_variant.sh <name> [-DFLAG ...]   ->  pytorch-retinanet_amd/csrc/build/lib_<name>.so
# The library with rh_infer.hip and rh_loss.hip compiled with extra flags (for A/B timings with
# tools/try_variant.py); the other objects come from the regular build.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
c=$root/pytorch-retinanet_amd/csrc
make -C $c >/dev/null
mkdir -p $c/build/var_$name
for f in rh_infer rh_loss; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I$root/include -I$c "$@" \
    -c $c/$f.hip -o $c/build/var_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $c/build/rh_api.o $c/build/rh_anchors.o $c/build/rh_generic.o \
  $c/build/var_$name/rh_infer.o $c/build/var_$name/rh_loss.o -o $c/build/lib_$name.so
echo $c/build/lib_$name.so
