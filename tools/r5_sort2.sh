#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5sort; mkdir -p $O
python tools/sort_sizes.py > $O/sizes_default.txt 2>&1; cat $O/sizes_default.txt
PVAMD_LIB=tools/variants/libpvamd_radix16k.so PVAMD_RADIX_FROM=16385 python tools/sort_sizes.py > $O/sizes_radix_all.txt 2>&1; cat $O/sizes_radix_all.txt
