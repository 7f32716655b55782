mkdir -p gpurun_out/r06i
export PVAMD_ALLOW_VARIANT=1
{ for rep in 1 2; do for v in tp8 tp2 tp1; do echo "== $v (leaves >= N read a level 1/64 the size: TIMING ONLY)"; PVAMD_LIB=tools/variants/libpvamd_$v.so python tools/readme_probe.py 2>&1 | grep "padding 1.0"; done; done; } 2>&1 | grep -v "amdgpu.ids\|A/B build" | tee gpurun_out/r06i/timing_pool.txt
