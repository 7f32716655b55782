#!/bin/bash
# SQ counters of the few-points mesh query (tools/run_c1.py) -> gpurun_out/$1/pmc_summary.txt
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $O/pmc1 -o c1 --output-format csv -- python tools/run_c1.py > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM -d $O/pmc2 -o c1 --output-format csv -- python tools/run_c1.py > $O/pmc2.log 2>&1
for k in mesh_parts_all_kernel hand_over_all_kernel order_small_kernel; do echo "== $k"; for d in pmc1 pmc2; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/sq_summary.py $f $k; done; done > $O/pmc_summary.txt 2>&1
find $O -name "*.csv" -size +1M -delete
cat $O/pmc_summary.txt
