mkdir -p gpurun_out/r06b
export PVAMD_ALLOW_VARIANT=1 CQ_P=${CQ_P:-2097152,4194304,8388608,12582912,16777216} CQ_COLD_MB=${CQ_COLD_MB:-640}
{ for rep in 1 2; do for v in "" $VARIANTS; do if [ -z "$v" ]; then python tools/cq_sweep.py; else PVAMD_LIB=tools/variants/libpvamd_$v.so python tools/cq_sweep.py; fi; done; done; } 2>&1 | grep -v "amdgpu.ids\|A/B build" | tee gpurun_out/r06b/stream_from_cold.txt
