#!/bin/bash
# build an experimental variant of the library for A/B runs: tools/build_variant.sh <name> [-DMACRO=value ...]
# -> pbml_mantle_convection_amd/build/lib_<name>.so (select with MANTLE_LIB=...; see tools/ab.sh)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/pbml_mantle_convection_amd/csrc
out=$root/pbml_mantle_convection_amd/build/var_$name
mkdir -p $out
for f in elementwise conv_api conv_f32 conv_bf16 conv_rr_bf16 loss optim; do
  extra=""; [ $f = conv_rr_bf16 ] && extra="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off $extra "$@" -c $src/$f.hip -o $out/$f.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $out/*.o -o $root/pbml_mantle_convection_amd/build/lib_$name.so
echo $root/pbml_mantle_convection_amd/build/lib_$name.so
