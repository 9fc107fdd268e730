#!/bin/bash
# Build a variant of the library with extra compiler flags:  bash tools/build_variant.sh <name> "<flags>"
# -> event_3dgs_amd/lib_<name>.so  (for same-box A/B runs with tools/ab_bench3.sh)
NAME=$1; FLAGS=$2
D=/tmp/e3v_$NAME; mkdir -p $D
for f in capi forward backward scan_sort aux densify; do
  [ -f event_3dgs_amd/csrc/$f.hip ] || continue
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC $FLAGS -c event_3dgs_amd/csrc/$f.hip -o $D/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o event_3dgs_amd/lib_$NAME.so $D/*.o && echo built event_3dgs_amd/lib_$NAME.so
