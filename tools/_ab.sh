python -m pytest tests/test_composed_gpu.py tests/test_robot_gpu.py tests/test_golden_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|assert" | tail -5
python tools/c4_sorted_probe.py 2>&1 | tail -2
python tools/bench_c34.py 2>&1 | tail -1
python tools/bench_configs.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
for k in ('C3_composed_8_drills_4M','C4_robot_8links'):
    print(k, 'random %.4f ms' % (d[k]['gpu_s']*1e3), 'grid-ordered %.4f ms' % (d[k]['grid_ordered_points']['gpu_s']*1e3))
"
