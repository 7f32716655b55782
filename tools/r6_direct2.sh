mkdir -p gpurun_out/r06b
export PVAMD_ALLOW_VARIANT=1 CQ_MARGINS="0.05,-0.001,9"
{
for rep in 1 2; do
  for v in "" $VARIANTS; do
    for m in 0.05 -0.001; do
      if [ -z "$v" ]; then CQ_MARGINS=$m CQ_P=${CQ_P:-262144,1048576,4194304,8388608} python tools/cq_sweep.py; else CQ_MARGINS=$m CQ_P=${CQ_P:-262144,1048576,4194304,8388608} PVAMD_LIB=tools/variants/libpvamd_$v.so python tools/cq_sweep.py; fi
    done
  done
done
} 2>&1 | grep -v "amdgpu.ids\|A/B build" | tee gpurun_out/r06b/direct2.txt
