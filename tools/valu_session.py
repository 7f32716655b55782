"""Aggregate ONE profiling session (tools/valu_session.sh) into profiles/rNN_valu_session.json: per workload the VALU
instruction counters, the dynamic opcode-class counters, the kernels' time (kernel trace of the same commands, same box,
same minutes) and the issue rates tools/valu_rate.bin measured -- everything bench.py's VALU rooflines are built from.
usage: valu_session.py <session dir> <out.json>"""
import csv, glob, json, os, re, sys, collections

# also: valu_session.py --rerate <valu_rate.txt> <existing.json>: recompute the issue rates of a committed session from its table
sess, out_path = sys.argv[1], sys.argv[2]
WORK = {  # key -> (run_valu.py name, calls, kernel regex, description)
    "c1_mesh_query": ("c1", 6, r"mesh_|hand_over|order_|radix_|aabb_|morton|invert_face", "C1: MeshSDF(drill), 10,000 grid points, one call = point sort + list / parts / finish launches"),
    "c3_composed_query": ("c3", 6, r"composed_query|group_points", "C3: ComposedSDF of 8 drills, 4,194,304 random points, one launch"),
    "c4_composed_query_wave": ("c4", 4, r"composed_query|group_points", "C4: RobotSDF 8 links (100 KB grids), 200 configurations x 262,144 random points, one launch"),
    "c5_chamfer_mesh": ("c5", 3, r"mesh_|chamfer|hand_over|order_|radix_|aabb_|morton|invert_face", "C5: chamfer, 2,097,152 points -> 99,500-triangle sphere, one call = point sort + main launch + heavy-group launches"),
    # cache construction: CachedSDF(...) over a MeshSDF = point sort + mesh query over every voxel centre + pack
    "build_drill_0.01": ("bd1", 6, r"mesh_|hand_over|order_|radix_|aabb_|morton|invert_face|pack_grid|grid_points", "cache build: drill, res 0.01 pad 0.1, 37x33x40 = 48,840 voxel centres"),
    "build_drill_0.002": ("bd2", 4, r"mesh_|hand_over|order_|radix_|aabb_|morton|invert_face|pack_grid|grid_points", "cache build: drill, res 0.002 pad 0.01, 92x73x105 = 705,180 voxel centres"),
    "build_wrench_0.001": ("bw", 3, r"mesh_|hand_over|order_|radix_|aabb_|morton|invert_face|pack_grid|grid_points", "cache build: offset_wrench_nogrip, res 0.001 pad 0.05, 218x126x111 = 3,048,948 voxel centres"),
}


def rates(path):
    """tools/valu_rate.bin table -> best ns per wave64 instruction per SIMD over the resident-wave counts, per opcode"""
    table = {}
    for line in open(path):
        if "|" not in line or line.startswith("opcode"):
            continue
        cells = [c.strip() for c in line.split("|")]
        try:
            ns = [float(c.split()[0]) for c in cells[1:] if c]
        except (ValueError, IndexError):
            continue
        table[cells[0]] = min(ns)
    fast = [table[k] for k in ("v_fma_f32", "v_add_f32", "v_mul_f32") if k in table]
    slow = [table[k] for k in ("v_cmp_lt_f32 (vcc)", "v_cmp_lt_f32 (sgpr pair)", "v_max_f32", "v_med3_f32", "v_cndmask_b32 (sgpr-pair mask)",
                               "v_cvt_i32_f32", "v_rndne_f32", "v_mul_lo_u32") if k in table]
    return {"fast": sum(fast) / len(fast), "slow": sum(slow) / len(slow), "trans": table.get("v_sqrt_f32"),
            "best_any": min(table.values()), "best_any_row": min(table, key=table.get),
            "per_opcode_best_ns": table,
            "note": "best over 1..8 resident waves per SIMD of tools/valu_rate.bin, this session; fast = mean of v_fma/add/mul_f32, "
                    "slow = mean of v_cmp (vcc / sgpr pair) / v_max / v_med3 / v_cndmask (sgpr mask) / v_cvt / v_rndne / v_mul_lo_u32; "
                    "best_any = the fastest row of the whole table (a stream that alternates fast-group and slow-group opcodes: the two "
                    "groups issue side by side, so a mix is NOT bounded by the weighted sum of the two rates)"}


def counters(pattern_dir, regex):
    tot = collections.defaultdict(float)
    for path in glob.glob(os.path.join(pattern_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if regex.search(r["Kernel_Name"]) and "mesh_prepare" not in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
    return tot


def kernel_ms(trace_dir, regex, calls):
    per = []
    for path in glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if regex.search(r["Kernel_Name"]) and "mesh_prepare" not in r["Kernel_Name"]:
                per.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    per.sort()
    if not per:
        return None, 0
    groups, last_end = [], None  # calls are separated by a synchronize + 10 ms sleep (tools/run_valu.py)
    for start, d in per:
        if last_end is None or start - last_end > 3e6:
            groups.append([])
        groups[-1].append(d)
        last_end = start + d * 1e6
    use = groups[2:] if len(groups) > 2 else groups  # the first calls carry one-off work (clock ramp, code load, set-up)
    return sum(sum(g) for g in use) / len(use), len(use[-1])


if sess == "--rerate":
    rate_file, existing = sys.argv[2], sys.argv[3]
    doc = json.load(open(existing))
    doc["issue_rates_ns"] = rates(rate_file)
    json.dump(doc, open(existing, "w"), indent=1)
    print("re-rated", existing, {k: doc["issue_rates_ns"][k] for k in ("fast", "slow", "best_any", "best_any_row")})
    sys.exit(0)
out = {"issue_rates_ns": rates(os.path.join(sess, "valu_rate.txt")), "workloads": {},
       "session": open(os.path.join(sess, "session.txt")).read().strip() if os.path.exists(os.path.join(sess, "session.txt")) else None}
for key, (name, calls, pat, text) in WORK.items():
    rx = re.compile(pat)
    c = counters(os.path.join(sess, f"pmc_{name}"), rx)
    if not c.get("SQ_INSTS_VALU"):
        continue
    per = {k: v / calls for k, v in c.items()}
    ms, launches = kernel_ms(os.path.join(sess, f"kt_{name}"), rx, calls + 2)
    valu = per["SQ_INSTS_VALU"]
    arith = sum(per.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32"))
    per.update({"workload": text, "command": f"rocprofv3 --pmc ... -- python tools/run_valu.py {name} {calls} (3 counter passes) + --kernel-trace pass, tools/valu_session.sh",
                "kernel_ms_same_session": ms, "launches_per_call": launches,
                "active_lanes": per.get("SQ_THREAD_CYCLES_VALU", 0.0) / valu if per.get("SQ_THREAD_CYCLES_VALU") else None,
                "mix": {"fast_f32_add_mul_fma": arith / valu, "slow_fraction": 1.0 - arith / valu,
                        "trans_f32": per.get("SQ_INSTS_VALU_TRANS_F32", 0.0) / valu, "int32": per.get("SQ_INSTS_VALU_INT32", 0.0) / valu,
                        "cvt": per.get("SQ_INSTS_VALU_CVT", 0.0) / valu,
                        "f64": sum(per.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64")) / valu,
                        "note": "dynamic: class counters / SQ_INSTS_VALU of the same calls; everything that is not an f32 add / mul / fma "
                                "is priced at the slow rate (v_mov and integer adds issue at the fast rate, so the ceiling is a little "
                                "pessimistic and `frac` a little optimistic -- by at most their share)"}})
    out["workloads"][key] = per
json.dump(out, open(out_path, "w"), indent=1)
r = out["issue_rates_ns"]
for k, w in out["workloads"].items():
    t_mix = max(w["mix"]["slow_fraction"] * r["slow"], r["best_any"])  # ns per instruction: the slow group alone, or the co-issue rate
    peak = 1024 / (t_mix * 1e-9)
    print(f"{k}: VALU {w['SQ_INSTS_VALU']:.4g}/call, {w['kernel_ms_same_session']:.4f} ms, slow fraction {w['mix']['slow_fraction']:.3f}, "
          f"frac of own-mix ceiling {w['SQ_INSTS_VALU'] / (w['kernel_ms_same_session'] * 1e-3) / peak:.3f}")
