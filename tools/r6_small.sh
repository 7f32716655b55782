mkdir -p gpurun_out/r06b
export PVAMD_ALLOW_VARIANT=1
{
for rep in 1 2; do
  CQ_P=256,1024,4096,8192,15251,16383 python tools/cq_sweep.py
  CQ_P=256,1024,4096,8192,15251,16383 PVAMD_LIB=tools/variants/libpvamd_cq_min64.so python tools/cq_sweep.py
done
for rep in 1 2; do
  python tools/readme_probe.py
  PVAMD_LIB=tools/variants/libpvamd_scalar_nt.so python tools/readme_probe.py
done
} 2>&1 | grep -v "amdgpu.ids\|A/B build" | tee gpurun_out/r06b/small.txt
