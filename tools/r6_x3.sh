mkdir -p gpurun_out/r06b
export PVAMD_ALLOW_VARIANT=1 CQ_LOGP=${CQ_LOGP:-20,22,23}
for rep in 1 2; do
  python tools/cq_sweep.py
  for v in ${X3_VARIANTS:-x3w4 x3w8 x3w16 x3ntw4 x3ntw16}; do PVAMD_LIB=tools/variants/libpvamd_cq_$v.so python tools/cq_sweep.py; done
done 2>&1 | grep -v "amdgpu.ids\|A/B build" | tee gpurun_out/r06b/x3.txt
