#!/bin/bash
# Build libpvamd with one source file swapped (and/or extra -D flags) for A/B timing:
#   tools/build_variant.sh NAME SRC.hip [FLAGS...]     SRC's basename (mesh.hip, composed.hip, ...) says which object it replaces
# -> tools/variants/libpvamd_NAME.so ; run a tool against it with PVAMD_LIB=tools/variants/libpvamd_NAME.so
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
make -s -C pytorch_volumetric_amd/csrc
mkdir -p tools/variants
C=pytorch_volumetric_amd/csrc
which=$(basename "$src" .hip); which=${which%%_*}
extra=""; { [ "$which" = composed ] || [ "$which" = mesh ]; } && extra="-fno-slp-vectorize"   # as csrc/Makefile (FLAGS_*)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -Wno-unused-value -I$C -Iinclude $extra "-DPVAMD_VARIANT=\"$name: $*\"" "$@" -c "$src" -o tools/variants/${which}_$name.o
objs=""
for o in api cached composed mesh chamfer_grid xform fk voxelgrid sample sort; do
  if [ $o = $which ]; then objs="$objs tools/variants/${which}_$name.o"; else objs="$objs $C/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tools/variants/libpvamd_$name.so $objs
echo tools/variants/libpvamd_$name.so
