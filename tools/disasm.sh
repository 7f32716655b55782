#!/bin/bash
# Disassemble the gfx950 code of one csrc/*.hip translation unit (no GPU needed):
#   tools/disasm.sh composed [extra hipcc flags]  ->  /tmp/pvamd_<name>.s  + per-kernel resource usage on stdout
set -e
name=$1; shift
src=/root/repo/pytorch_volumetric_amd/csrc/$name.hip
extra=""
case $name in composed|mesh) extra="-fno-slp-vectorize";; esac
cd /tmp && rm -f pvamd_$name.o pvamd_$name.o.*
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off $extra "$@" -c $src -o /tmp/pvamd_$name.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" | paste - - - - - \
  | sed -e 's/[a-z_]*\.hip:[0-9]*:[0-9]*: remark: //g' -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g' -e 's/Function Name: //'
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading pvamd_$name.o >/dev/null
/opt/rocm/lib/llvm/bin/llvm-objdump -d pvamd_$name.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 > /tmp/pvamd_$name.s
grep -n "^[0-9a-f]* <" /tmp/pvamd_$name.s | awk -F'[<>]' '{print $1 $2}' | c++filt | cut -c1-140
