#!/bin/bash
# SQ counters of the 1M-point C2 launch (shipped look-up and the round-4 statements)
export TMPDIR=/tmp
O=gpurun_out/r5cq; mkdir -p $O
for v in "" cq_oldlookup; do
  lib=""; [ -n "$v" ] && lib=tools/variants/libpvamd_$v.so
  T=/tmp/pmc_cq_${v:-ship}; rm -rf $T
  PVAMD_LIB=$lib rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d $T/p1 -o c --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-large --no-legs --detail /tmp/d.json > $T.log 2>&1
  PVAMD_LIB=$lib rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU -d $T/p2 -o c --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-large --no-legs --detail /tmp/d.json >> $T.log 2>&1
  echo "## 1M-point launches of cached_query_wave, build ${v:-shipped}"
  python tools/sq_summary.py $(find $T/p1 $T/p2 -name "*counter_collection.csv") "cached_query_wave<true, false, false, true>"
done > $O/pmc_1m.txt 2>&1
cat $O/pmc_1m.txt
