mkdir -p gpurun_out/r06i
export PVAMD_ALLOW_VARIANT=1
{ for rep in 1 2; do for v in "" sb1 sb2 sb8; do echo "== ${v:-product (batch 4)}"; if [ -z "$v" ]; then python tools/readme_probe.py; else PVAMD_LIB=tools/variants/libpvamd_$v.so python tools/readme_probe.py; fi; done; done; } 2>&1 | grep -v "amdgpu.ids\|A/B build" | tee gpurun_out/r06i/scalar_batch.txt
