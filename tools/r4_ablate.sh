#!/bin/bash
# Where does the time of the wave-tile composed kernel go on C4?  Builds with parts of the leaf visit compiled out (WRONG
# results, timing only): 1 = no in-range look-ups, 2 = no out-of-range candidates, 3 = neither (affine + range test + loop only),
# 4 = the affine replaced by one add per coordinate (tools/build_variant.sh abl4 pytorch_volumetric_amd/csrc/composed.hip -DPVAMD_ABLATE=4)
for v in "" tools/variants/libpvamd_abl1.so tools/variants/libpvamd_abl2.so tools/variants/libpvamd_abl3.so; do
  echo "== PVAMD_LIB=$v"
  PVAMD_LIB=$v timeout 600 python tools/composed_ab.py c4 2>&1 | grep "^C4"
done
