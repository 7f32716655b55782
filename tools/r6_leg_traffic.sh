#!/bin/bash
# round 6: HBM-side traffic (FETCH_SIZE x2 on gfx950, WRITE_SIZE; separate --pmc passes) of the composed legs' kernels:
# C4 (group_points_kernel + composed_query_grouped) and C3 (composed_query_fused) -> gpurun_out/r06/leg_traffic.txt
export TMPDIR=/tmp
O=gpurun_out/r06/legpmc; mkdir -p $O
for w in c4 c3; do
  rocprofv3 --pmc FETCH_SIZE -d $O/${w}_fetch -o v --output-format csv -- python tools/run_valu.py $w 4 > $O/${w}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/${w}_write -o v --output-format csv -- python tools/run_valu.py $w 4 > $O/${w}_write.log 2>&1
done
python - <<'PY' | tee gpurun_out/r06/leg_traffic.txt
import csv, glob, collections
def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and ("composed_query" in r["Kernel_Name"] or "group_points" in r["Kernel_Name"]):
                acc[r["Kernel_Name"].split("(")[0][-60:]].append(float(r["Counter_Value"]))
    return acc
for w, algo, what in (("c4", 200 * 262144 * 16 + 262144 * 12, "C4: 16 B x 200 x 262,144 written + 12 B x 262,144 read"),
                      ("c3", 4194304 * 28, "C3: 28 B x 4,194,304")):
    fe, wr = per_kernel(f"gpurun_out/r06/legpmc/{w}_fetch", "FETCH_SIZE"), per_kernel(f"gpurun_out/r06/legpmc/{w}_write", "WRITE_SIZE")
    print(f"== {what} = {algo / 1e6:.1f} MB algorithmic")
    tot = 0.0
    for k in sorted(set(fe) | set(wr)):
        f = 2.0 * 1024.0 * sum(fe.get(k, [0])) / max(len(fe.get(k, [0])), 1)
        x = 1024.0 * sum(wr.get(k, [0])) / max(len(wr.get(k, [0])), 1)
        tot += f + x
        print(f"   {k}: fetch (x2) {f / 1e6:.1f} MB + write {x / 1e6:.1f} MB per launch ({len(fe.get(k, []))} / {len(wr.get(k, []))} launches sampled)")
    print(f"   total {tot / 1e6:.1f} MB = {tot / algo:.3f} x algorithmic")
PY
find $O -name "*.csv" -size +1M -delete
