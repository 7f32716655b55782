#!/bin/bash
# round 5, composed kernel A/B: waves-per-SIMD caps (scratch vs occupancy), the med3 range test, one point per lane per pass
export TMPDIR=/tmp
O=gpurun_out/r5composed; mkdir -p $O
for v in "" mw8 mw7 mw6 mw5 med3 ppp1; do
  lib=""; [ -n "$v" ] && lib=tools/variants/libpvamd_$v.so
  echo "== variant ${v:-shipped}"
  PVAMD_LIB=$lib timeout 300 python tools/composed_ab.py c3 c4 2>&1 | grep "^C3\|^C4\|README"
done > $O/variants1.txt 2>&1
cat $O/variants1.txt
