mkdir -p gpurun_out/r06h
export PVAMD_ALLOW_VARIANT=1 CQ_P=67108864,100000000,16777216
{ for rep in 1 2; do for v in "" cq_bb8192 cq_bb4096 cq_bb2048; do if [ -z "$v" ]; then python tools/cq_sweep.py; else PVAMD_LIB=tools/variants/libpvamd_$v.so python tools/cq_sweep.py; fi; done; done; } 2>&1 | grep -v "amdgpu.ids\|A/B build" | tee gpurun_out/r06h/bigblocks.txt
