#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4b13; mkdir -p $O
for i in 1 2 3; do
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_k20_$i.json 2> $O/bench_k20_$i.err
python - <<PY
import json
d = json.load(open("$O/bench_k20_$i.json"))
print("run $i", {k: round(d[k], 6) if isinstance(d[k], float) else d[k] for k in ("value", "ms_per_step")}, round(d["roofline"]["frac"], 4),
      {k: round((v.get("ms_per_call") or v.get("sharded", {}).get("ms_per_step") or v.get("ms_per_step")), 4) for k, v in d["legs"].items()},
      "cpu %.3g / %.3g" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["fused_port"]["value"]), "large %.3f" % d["large_batch"]["frac_of_8TBs"],
      "c4 frac %.3f" % d["legs"]["c4"]["sharded"]["roofline"]["frac"])
PY
done
python -m pytest tests -q -m gpu > $O/pytest_all.txt 2>&1; grep -E "passed|failed" $O/pytest_all.txt | tail -2
