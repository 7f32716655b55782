#!/bin/bash
# SQ_INSTS_VALU (and friends) per CALL of C1 / C4 / C5 -> gpurun_out/$1/valu_counts.json (committed as profiles/r03_valu_counts.json)
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
CNT="SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"
one() { # key calls workload regex
  rocprofv3 --pmc $CNT -d $O/v_$1 -o v --output-format csv -- python tools/run_valu.py $1 $2 > $O/v_$1.log 2>&1
  f=$(find $O/v_$1 -name "*counter_collection.csv" | head -1)
  python tools/valu_counts.py "$5" $2 "$3" "rocprofv3 --pmc $CNT -- python tools/run_valu.py $1 $2" "$4" $f
  rm -rf $O/v_$1
}
{
one c1 6 "C1: MeshSDF(drill), 10,000 grid points, one call = point sort + list / parts / finish launches" "mesh_|hand_over|order_|aabb_|morton" c1_mesh_query
one c4 4 "C4: RobotSDF 8 links (100 KB grids), 200 configurations x 262,144 random points, one launch" "composed_query" c4_composed_query_wave
one c5 3 "C5: chamfer, 2,097,152 points -> 99,500-triangle sphere, one call = point sort + main launch + heavy-group launches" "mesh_|chamfer|hand_over|order_|aabb_|morton" c5_chamfer_mesh
} > $O/valu_counts.jsonl 2>$O/valu_counts.err
python - <<PY
import json
out = {}
for line in open("$O/valu_counts.jsonl"):
    line = line.strip()
    if line.startswith("{"):
        out.update(json.loads(line))
json.dump(out, open("$O/valu_counts.json", "w"), indent=1)
print(json.dumps({k: {c: v for c, v in d.items() if c != "kernels"} for k, d in out.items()}, indent=1))
PY
