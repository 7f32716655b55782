mkdir -p gpurun_out/r06b
export PVAMD_ALLOW_VARIANT=1
for rep in 1 2; do
  python tools/cq_sweep.py
  for v in s8l2 s16l2 s32l2 s4l4 s8l4 s16l4; do PVAMD_LIB=tools/variants/libpvamd_cq_$v.so python tools/cq_sweep.py; done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06b/stagger.txt
