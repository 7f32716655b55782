#!/bin/bash
# HBM-side traffic of the un-permute pass (README-size C4, random points): separate rocprofv3 --pmc passes
export TMPDIR=/tmp
O=gpurun_out/r4unp; mkdir -p $O
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  rm -rf /tmp/unp; rocprofv3 --pmc $set -d /tmp/unp -o u --output-format csv -- python tools/unpermute_probe.py > /tmp/unp.log 2>&1
  python tools/sq_summary.py $(find /tmp/unp -name "*counter_collection.csv") composed_unpermute
done | tee $O/pmc_unpermute.txt
