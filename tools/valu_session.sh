#!/bin/bash
# ONE session: VALU issue rates + per workload three counter passes and a kernel-trace pass -> gpurun_out/$1/valu_session.json
export TMPDIR=/tmp
O=gpurun_out/$1; S=/tmp/valu_sess; rm -rf $S; mkdir -p $O $S
{ date -u; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4; } > $S/session.txt 2>&1
./tools/valu_rate.bin > $S/valu_rate.txt 2>&1
P1="SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"
P3="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT64 SQ_WAIT_ANY SQ_WAIT_INST_ANY"
for w in "c1 6" "c3 6" "c4 4" "c5 3" "bd1 6" "bd2 4" "bw 3"; do
  set -- $w
  for i in 1 2 3; do
    eval "CNT=\$P$i"
    rocprofv3 --pmc $CNT -d $S/pmc_$1/p$i -o v --output-format csv -- python tools/run_valu.py $1 $2 > $S/pmc_$1_$i.log 2>&1
  done
  rocprofv3 --kernel-trace -d $S/kt_$1 -o v --output-format csv -- python tools/run_valu.py $1 $(( $2 + 2 )) > $S/kt_$1.log 2>&1
done
python tools/valu_session.py $S $O/valu_session.json | tee $O/valu_session.txt
cp $S/valu_rate.txt $O/valu_rate.txt
