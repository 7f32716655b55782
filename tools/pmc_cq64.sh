#!/bin/bash
# counters of the C2 kernel in the streaming regime (64M points): separate rocprofv3 --pmc passes -> gpurun_out/$1/pmc_cq64_*.md
# usage: tools/pmc_cq64.sh OUTDIR [PVAMD_LIB]
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
[ -n "$2" ] && export PVAMD_LIB=$2
tag=$(basename "${2:-default}" .so)
pass() { # name counters...
  n=$1; shift
  rocprofv3 --pmc "$@" -d $O/pmc_$n -o cq --output-format csv -- python tools/run_cq64.py > $O/pmc_$n.log 2>&1
  f=$(find $O/pmc_$n -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/pmc_table.py $f cached_query_wave 3 "52% out of range" "all in range" "all out of range"; else echo "pass $n failed: $(tail -2 $O/pmc_$n.log)"; fi
  rm -rf $O/pmc_$n
}
{
echo "## $tag"
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pass sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE
pass tcc1 TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_HIT_sum
pass tcc2 TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum
pass tcc3 TCC_BUSY_sum TCC_CYCLE_sum
pass tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pass tcp2 TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum
} > $O/pmc_cq64_$tag.md 2>&1
cat $O/pmc_cq64_$tag.md
