#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4stall; mkdir -p $O
rm -rf /tmp/st2
timeout 600 rocprofv3 --hip-trace --hsa-trace --kernel-trace --output-format csv -d /tmp/st2 -o st -- python tools/stall_trace.py > $O/stall_traced.txt 2>&1
grep "steps above" $O/stall_traced.txt
python tools/hip_trace_top.py /tmp/st2 5 > $O/stall_api.txt 2>&1; head -80 $O/stall_api.txt
# the knob candidates: eager code-object loading


