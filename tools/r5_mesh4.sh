#!/bin/bash
# round 5: counters of chamfer_mesh_kernel on C5 -- the shipped build and the seed-only timing build (what bounds the start of a scan?)
export TMPDIR=/tmp
O=gpurun_out/r5mesh; mkdir -p $O
for v in "" onlyseed; do
  lib=""; [ -n "$v" ] && lib=tools/variants/libpvamd_$v.so
  T=/tmp/pmc_c5_${v:-ship}; rm -rf $T
  PVAMD_LIB=$lib rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $T/p1 -o c --output-format csv -- python tools/run_c5.py > $T.log 2>&1
  PVAMD_LIB=$lib rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM -d $T/p2 -o c --output-format csv -- python tools/run_c5.py >> $T.log 2>&1
  PVAMD_LIB=$lib rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum -d $T/p3 -o c --output-format csv -- python tools/run_c5.py >> $T.log 2>&1
  PVAMD_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d $T/kt -o c -- python tools/run_c5.py >> $T.log 2>&1
  {
    echo "## C5, chamfer_mesh_kernel, build: ${v:-shipped}"
    python tools/sq_summary.py $(find $T/p1 $T/p2 $T/p3 -name "*counter_collection.csv") chamfer_mesh_kernel
    python - <<PY
import csv, glob
d = {}
for f in glob.glob("$T/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"][:60]
        d.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print("kernel-trace us: %-62s n=%d mean %.1f" % (n, len(v), sum(v) / len(v)))
PY
  } > $O/pmc_c5_${v:-shipped}.txt 2>&1
  cat $O/pmc_c5_${v:-shipped}.txt
done
