#!/bin/bash
# SQ counters of chamfer_mesh_kernel on C5 (tools/run_c5.py) -> gpurun_out/$1/pmc_summary.txt
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $O/pmc1 -o c5 --output-format csv -- python tools/run_c5.py > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM -d $O/pmc2 -o c5 --output-format csv -- python tools/run_c5.py > $O/pmc2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE -d $O/pmc3 -o c5 --output-format csv -- python tools/run_c5.py > $O/pmc3.log 2>&1
for d in pmc1 pmc2 pmc3; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/sq_summary.py $f chamfer_mesh_kernel; echo "-- mesh_parts_kernel"; python tools/sq_summary.py $f mesh_parts_kernel; done > $O/pmc_summary.txt 2>&1
find $O -name "*.csv" -size +1M -delete
cat $O/pmc_summary.txt; tail -2 $O/pmc3.log
