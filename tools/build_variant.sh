#!/bin/bash
# Build libpvamd with a different mesh.hip (and/or extra -D flags) for A/B timing:  tools/build_variant.sh NAME MESH_SRC [FLAGS...]
# -> tools/variants/libpvamd_NAME.so ; run a tool against it with PVAMD_LIB=tools/variants/libpvamd_NAME.so
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
make -s -C pytorch_volumetric_amd/csrc
mkdir -p tools/variants
C=pytorch_volumetric_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -Wno-unused-value -I$C -Iinclude "$@" -c "$src" -o tools/variants/mesh_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tools/variants/libpvamd_$name.so $C/api.o $C/cached.o $C/composed.o tools/variants/mesh_$name.o $C/chamfer_grid.o $C/xform.o $C/fk.o
echo tools/variants/libpvamd_$name.so
