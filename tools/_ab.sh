python -m pytest tests/test_composed_gpu.py tests/test_robot_gpu.py tests/test_golden_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
python tools/bench_c34.py 2>&1 | tail -1
for v in noslp noslpmw8 mw8; do echo -n "$v: "; PVAMD_LIB=tools/variants/libpvamd_$v.so python tools/bench_c34.py 2>&1 | tail -1; done
python tools/bench_c34.py 2>&1 | tail -1
