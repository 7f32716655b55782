#!/bin/bash
# round 6: the committed profile set -- driver-shaped bench line + kernel trace + PMC traffic (tools/profile_bench.sh), then the
# VALU session (tools/valu_session.sh)
bash tools/profile_bench.sh r06 > gpurun_out/r06_profile_bench.log 2>&1
tail -20 gpurun_out/r06_profile_bench.log
bash tools/valu_session.sh r06 > gpurun_out/r06_valu_session.log 2>&1
tail -12 gpurun_out/r06_valu_session.log
