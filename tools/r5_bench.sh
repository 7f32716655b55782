#!/bin/bash
# the driver's command, after the profile set of the final kernels has been committed
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/bench_detail_k20.json > $O/bench_k20.json 2> $O/bench_k20.err ) 2> $O/bench_k20.time
tail -3 $O/bench_k20.time; tail -1 $O/bench_k20.json | wc -c; tail -1 $O/bench_k20.json
python bench.py --detail $O/bench_detail_default.json > $O/bench_default.json 2>/dev/null; tail -1 $O/bench_default.json | head -c 600
