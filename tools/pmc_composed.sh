#!/bin/bash
# SQ counters of the composed kernel per variant: tools/pmc_composed.sh <out tag> <workload c4|c3> <flags>...
# three rocprofv3 --pmc passes + one kernel-trace pass per variant, all in this session -> gpurun_out/<tag>/pmc_<wl>_<flags>.txt
export TMPDIR=/tmp
O=gpurun_out/$1; WL=$2; shift 2; mkdir -p $O
for FL in "$@"; do
  T=/tmp/pmc_${WL}_$FL; rm -rf $T
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $T/p1 -o c --output-format csv -- python tools/run_composed.py $WL $FL 4 > $T.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD -d $T/p2 -o c --output-format csv -- python tools/run_composed.py $WL $FL 4 >> $T.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_INSTS_SENDMSG SQ_WAVE_CYCLES -d $T/p3 -o c --output-format csv -- python tools/run_composed.py $WL $FL 4 >> $T.log 2>&1
  rocprofv3 --kernel-trace --output-format csv -d $T/kt -o c -- python tools/run_composed.py $WL $FL 12 >> $T.log 2>&1
  {
    echo "## $WL flags $FL"
    python tools/sq_summary.py $(find $T/p1 $T/p2 $T/p3 -name "*counter_collection.csv") composed_query
    python - <<PY
import csv, glob
d = []
for f in glob.glob("$T/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "composed_query" in r["Kernel_Name"]:
            d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
d = d[2:]
print("kernel-trace durations us (same session, calls 3..12): n=%d mean %.1f min %.1f max %.1f" % (len(d), sum(d) / max(len(d), 1), min(d), max(d)))
PY
  } > $O/pmc_${WL}_$FL.txt 2>&1
  cat $O/pmc_${WL}_$FL.txt
done
