#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4b12; mkdir -p $O
timeout 900 python -m pytest tests/test_robot_gpu.py tests/test_chamfer_gpu.py -q -m gpu > $O/pytest.txt 2>&1; grep -E "^FAILED|passed|failed|Error" $O/pytest.txt | head
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; tail -2 $O/bench_k20.err
python - <<PY
import json
d = json.load(open("$O/bench_k20.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("frac", "frac_best_replay", "frac_rocprof", "frac_of_wall_ms_per_step")})
for k, v in d["legs"].items():
    if "error" in v: print(k, "ERROR", v["error"]); continue
    r = v.get("roofline") or v.get("sharded", {}).get("roofline") or {}
    print(k, v.get("ms_per_call") or v.get("sharded", {}).get("ms_per_step") or v.get("ms_per_step"), {x: r.get(x) for x in ("frac", "frac_same_session")},
          {x: v.get(x) for x in ("set_joint_configuration_ms", "set_joint_configuration_device_q_ms", "configure_plus_query_ms", "configure_plus_query_graph_ms") if x in v})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["spread"], d["cpu_baseline"]["fused_port"]["value"])
print({k: d[k]["frac_of_8TBs"] for k in ("all_in_range_batch", "mid_batch", "large_batch")})
print(d["latency"])
PY
