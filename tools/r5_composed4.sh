#!/bin/bash
# round 5: SQ counters of the per-lane (flags 2) and the wave-tile (flags 4) composed kernels on C3 (A = 1, 4 M points), and of C4
bash tools/pmc_composed.sh r5composed c3 2 4 > /dev/null 2>&1
bash tools/pmc_composed.sh r5composed c4 4 > /dev/null 2>&1
cat gpurun_out/r5composed/pmc_c3_2.txt gpurun_out/r5composed/pmc_c3_4.txt gpurun_out/r5composed/pmc_c4_4.txt
