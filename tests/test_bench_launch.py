"""bench.py's launcher: `--gpus N` must really start N ranks (the flag used to be parsed and ignored).

The CPU test drives the same spawn path the driver's `python bench.py --gpus N` takes, with `--launch-check` standing in
for the GPU work (ranks rendezvous over gloo on 127.0.0.1 and all-reduce a 1).  The GPU test runs the whole benchmark
with two ranks sharing the one GPU of the test box (collectives through gloo): every leg, rank-0 JSON, n_gpus == 2.
"""
import json
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LINE_LIMIT = 4096  # the driver reads the tail of stdout: round 4's 23.7 KB line came back as `parsed: null`


def long_strings(obj, limit=120, path=""):
    if isinstance(obj, str):
        return [path] if len(obj) > limit else []
    if isinstance(obj, dict):
        return [p for k, v in obj.items() for p in long_strings(v, limit, f"{path}.{k}")]
    if isinstance(obj, list):
        return [p for i, v in enumerate(obj) for p in long_strings(v, limit, f"{path}[{i}]")]
    return []


def run_bench(*flags, timeout=600, want_detail=False):
    """The contract line: the LAST stdout line, which must be JSON and under LINE_LIMIT bytes.  With want_detail also the long
    record bench.py wrote to its --detail file."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "detail.json")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags, "--detail", path], capture_output=True,
                             text=True, timeout=timeout, env=env, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        text = out.stdout.strip().splitlines()[-1]
        assert len(text.encode()) < LINE_LIMIT, f"contract line is {len(text.encode())} bytes"
        line = json.loads(text)
        if not want_detail:
            return line
        detail = json.load(open(path))
        assert not long_strings(detail), long_strings(detail)
        return line, detail


@pytest.mark.parametrize("n", [1, 2, 3])
def test_gpus_flag_spawns_that_many_ranks(n):
    line = run_bench("--gpus", str(n), "--launch-check")
    assert line["n_gpus"] == n and line["ranks_seen_by_all_reduce"] == n and line["gpus_requested"] == n


def test_under_torchrun_the_given_world_is_used():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29741", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks_seen_by_all_reduce"] == 2


def fat_record(world=8):
    """A long record with every optional block present and 17-digit floats everywhere (no GPU needed)."""
    x = 0.123456789012345678
    roof = {"bound": "valu", "achieved": 556.59 + x, "peak": 977.33 + x, "unit": "G wave64 VALU inst/s", "frac": x, "frac_this_run": x,
            "util_valu": x}
    check = {"ranks": world, "ranks_expected": world, "backend": "nccl", "bytes_received_per_rank_expected": 734003200,
             "bytes_match": True, "kernel_only_ms": x, "gathered_ms": 10 * x, "ok": True}
    gathered = {"gather": True, "value": 1e10 + x, "ms_per_step": x, "self_check": check, "bytes_received_per_rank": 734003200,
                "equals_unsharded_call": True, "collective": "packed (val, grad) records, all_gather_into_tensor x1 (nccl), unpack kernel"}
    c4 = {"scaling": "strong", "sharded": {"gather": False, "value": 7.7e10 + x, "ms_per_step": x, "roofline": roof, "frac_8d": x,
                                           "prepared_sorted_ms": x, "prepared_caller_ms": x},
          "gathered": gathered, "gathered_by_configs": {"ms_per_step": x}}
    legs = {"c3": {"value": 4.9e10 + x, "ms_per_step": x, "roofline": roof, "frac_8d": x}, "c4": c4, "c4_readme_grid": c4,
            "c5": {"value": 7e8 + x, "ms_per_step": x, "roofline": roof, "rel_err_vs_analytic": x * 1e-3, "frac_8d": x,
                   "exact_pairs_per_step": 25165824, "collective": "all_reduce of B=1 float64 sums + count (nccl)"},
            "cache_build": {"builds": {k: {"ms": x, "gpu_ms": x, "value": 1e9 + x, "roofline": roof, "frac_8d": x, "exact_pairs": 36587376}
                                       for k in ("drill_0.01", "drill_0.002", "wrench_0.001")}},
            "c1": {"value": 1.2e8 + x, "ms_per_step": x, "roofline": roof, "frac_8d": x, "exact_pairs_per_step": 250000},
            "readme_a20": {"ms_per_call": x, "configure_plus_query_graph_ms": x, "published_ms": 37.688577},
            "readme_a200": {"error": "RuntimeError(" + "y" * 500 + ")", "ranks_without_a_result": 3}}
    batch = {"ms_per_launch": x, "kernel_ms_median": x, "frac_of_8TBs": x}
    return {"metric": "SDF (val+grad) queries/sec", "value": 1.5e11 + x, "unit": "queries/s", "n_gpus": world, "steps": 2000,
            "warmup": 200, "ms_per_step": x, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C2: CachedSDF 0.01 m voxels (37x33x40) on YcbPowerDrill, 1048576 uniform points per GPU per step",
                       "points_per_gpu": 1048576, "oob_fraction": x, "ranks": world, "backend": "nccl", "gather": False,
                       "launch": "one hipGraph of the K steps"},
            "roofline": {"bound": "hbm", "achieved": 4690.0 + x, "peak": 8000.0, "unit": "GB/s", "frac": x, "traffic": 35736551.39 + x,
                         "algorithmic_bytes_per_launch": 29360128, "kernel": "pvamd::cached_query_direct<2 points per lane, 16 waves>", "launch_us": 6 + x,
                         "launch_source": "hip events (this run)", "frac_rocprof": x, "frac_events": x, "launch_us_events": x,
                         "frac_wall": x, "events": "z" * 110},
            "cpu_baseline": {"value": 8.5e6 + x, "unit": "queries/s", "cores": 128, "kind": "port", "host_cpus": 256,
                             "sample": "median of 3 samples of whole passes over the same 1048576 points, 1.9 s wall",
                             "baseline_opforop": 8.5e6 + x, "baseline_fused": 3.1e8 + x, "fused_port": {"value": 3.1e8 + x, "cores": 128}},
            "parity": {"checked_points": 50000, "max_abs_val_err_vs_oracle": 0.0, "grad_mismatches_vs_oracle": 0, "parity_unpinned": True},
            "legs": legs, "all_in_range_batch": batch, "mid_batch": batch, "large_batch": batch, "p1e8_batch": batch,
            "latency": {"cached(points)": {"p50_us": 9 + x, "p99_us": 12 + x}}, "legs_aborted": "w" * 300}


def test_contract_line_stays_under_the_drivers_tail():
    """Round 4's line was 23.7 KB and came back unparsed.  The contract line built from a record with every block present,
    eight ranks and a failing leg with a long message is under 4 KB, carries the contract keys and rounds its floats."""
    sys.path.insert(0, ROOT)
    import bench
    text = bench.compact_text(fat_record(), "bench_detail.json")
    assert len(text.encode()) < LINE_LIMIT, len(text)
    line = json.loads(text)
    assert "dropped_for_size" not in line
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "legs", "detail"):
        assert key in line, key
    assert line["roofline"]["frac"] == 0.12346 and "events" not in line["roofline"]
    assert line["legs"]["c4"]["gathered"] == {"ranks": 8, "backend": "nccl", "bytes_received_per_rank": 734003200,
                                              "kernel_only_ms": 0.12346, "gathered_ms": 1.2346, "equals_unsharded_call": True, "ok": True}
    assert len(line["legs"]["readme_a200"]["error"]) <= 80 and line["legs"]["cache_build"]["wrench_0.001"]["frac_8d"] == 0.12346
    # every leg carries the SURVEY 8(d) fraction of THIS run next to the utilisation figure; nothing in the line names a profiles/ file
    for leg in ("c1", "c3", "c5"):
        assert line["legs"][leg]["frac_8d"] == 0.12346 and line["legs"][leg]["util_valu"] == 0.12346
    assert line["legs"]["c4"]["frac_8d"] == 0.12346 and "profiles/" not in text
    assert line["scaling_table"] == [{"n": 8, "value": 150000000000.0, "frac_hbm": 0.12346, "c4_kernel_only_ms": 0.12346, "c4_gathered_ms": 0.12346}]
    # and a record so large that it cannot fit sheds whole optional blocks instead of growing
    big = fat_record()
    big["legs"] = {f"leg{i}": big["legs"]["c4"] for i in range(40)}
    text = bench.compact_text(big, "bench_detail.json")
    assert len(text.encode()) < LINE_LIMIT and "legs" in json.loads(text)["dropped_for_size"]


SMALL = ("--steps", "5", "--warmup", "2", "--points", "65536", "--small-legs", "--no-large", "--no-cpu-baseline")


@pytest.mark.gpu
def test_two_ranks_share_the_gpu_and_every_leg_reports():
    line, detail = run_bench("--gpus", "2", "--share-gpu", *SMALL, want_detail=True)
    assert line["n_gpus"] == 2 and line["config"]["ranks"] == 2 and line["scaling"] == "weak"
    assert line["parity"]["max_abs_val_err_vs_oracle"] == 0.0 and line["parity"]["grad_mismatches_vs_oracle"] == 0
    assert line["parity"]["parity_unpinned"] is True and line["detail"] == "detail.json"
    legs = detail["legs"]
    for name in ("c3", "c4", "c4_readme_grid", "c5"):
        assert "error" not in legs[name], legs[name]
        assert legs[name]["scaling"] == "strong" and legs[name]["n_gpus"] == 2
        assert line["legs"][name]["ms_per_step"] > 0 and line["legs"][name]["value"] > 0
    assert legs["c4"]["sharded"]["gather"] is False and legs["c4"]["gathered"]["gather"] is True
    assert legs["c4"]["gathered"]["output_shape"] == [[8, 16384], [8, 16384, 3]]
    assert legs["c4"]["gathered"]["path"] == "packed" and legs["c4"]["gathered"]["equals_unsharded_call"] is True
    assert legs["c4"]["gathered"]["bytes_received_per_rank"] == 8 * 8192 * 16 and legs["c4"]["gathered"]["xgmi_lower_bound_ms"] > 0
    assert legs["c4"]["gathered_by_configs"]["equals_unsharded_call"] is True
    assert legs["c5"]["rel_err_vs_analytic"] < 1e-3
    # the self-check the first real multi-GPU line will carry, in the contract line itself: rank count, received bytes =
    # (W - 1) / W of the packed output, kernel-only against gathered time (gloo here, so `ok` only asks for the backend when
    # the ranks own their GPUs)
    chk = legs["c4"]["gathered"]["self_check"]
    assert chk["ranks"] == chk["ranks_expected"] == 2 and chk["bytes_match"] is True and chk["backend"] == "gloo"
    assert chk["bytes_received_per_rank_expected"] == 8 * 8192 * 16 and chk["gathered_ms"] > 0 and chk["kernel_only_ms"] > 0
    g = line["legs"]["c4"]["gathered"]
    assert g["ranks"] == 2 and g["backend"] == "gloo" and g["bytes_received_per_rank"] == 8 * 8192 * 16
    assert g["equals_unsharded_call"] is True and g["kernel_only_ms"] > 0 and g["gathered_ms"] > 0
    assert "gloo" in line["legs"]["c5"]["collective"]
    assert line["legs"]["cache_build"]["drill_0.01"]["ms"] > 0


@pytest.mark.gpu
def test_eight_ranks_share_the_gpu():
    """The driver's 8-GPU run cannot be rehearsed on a 1-GPU box, but its world size can: 8 ranks on the one GPU (gloo) walk
    the W = 8 partition / padding / index arithmetic of every leg; the gathered C4 results must equal the unsharded call.  The
    contract line of the 8-rank run stays under the size limit too (run_bench asserts it)."""
    line, detail = run_bench("--gpus", "8", "--share-gpu", *SMALL, want_detail=True)
    assert line["n_gpus"] == 8 and "legs_aborted" not in line
    legs = detail["legs"]
    for name in ("c4", "c4_readme_grid", "c5"):
        assert "error" not in legs[name] and "skipped" not in legs[name], legs[name]
    g = legs["c4"]["gathered"]
    assert g["path"] == "packed" and g["equals_unsharded_call"] is True and g["output_shape"] == [[8, 16384], [8, 16384, 3]]
    assert g["bytes_received_per_rank"] == 7 * 8 * 2048 * 16  # 7 peers x 8 configurations x 2048 padded points x 16 B
    assert line["legs"]["c4"]["gathered"]["bytes_received_per_rank"] == 7 * 8 * 2048 * 16
    assert legs["c4"]["gathered_by_configs"]["equals_unsharded_call"] is True  # one configuration per rank
    assert legs["c5"]["rel_err_vs_analytic"] < 1e-3


@pytest.mark.gpu
def test_a_rank_that_fails_before_a_legs_collectives_does_not_hang_the_others():
    """Rank 1 raises while preparing the C4 leg (whose packed path all-gathers): every rank skips that leg together -- one
    all-reduced flag before the leg's first collective -- and the remaining legs and the headline line still come out."""
    line, detail = run_bench("--gpus", "2", "--share-gpu", *SMALL, "--fail-rank", "1", "--fail-leg", "c4", timeout=300,
                             want_detail=True)
    legs = detail["legs"]
    assert "skipped" in legs["c4"] and legs["c4"]["ranks_without_a_result"] == 2
    assert "error" in line["legs"]["c4"] and line["legs"]["c4"]["ranks_without_a_result"] == 2
    assert "error" not in legs["c5"] and legs["c5"]["rel_err_vs_analytic"] < 1e-3 and "error" not in legs["c4_readme_grid"]
    assert line["n_gpus"] == 2 and line["value"] > 0


@pytest.mark.gpu
def test_a_rank_lost_inside_a_leg_costs_the_legs_not_the_headline():
    """Rank 1 passes the C4 leg's gate and then never arrives at its first collective (the case the gate cannot see): after
    --legs-deadline rank 0 prints the line -- headline, roofline, the legs finished so far -- and every rank exits 0."""
    line = run_bench("--gpus", "2", "--share-gpu", *SMALL, "--hang-rank", "1", "--fail-leg", "c4", "--legs-deadline", "20",
                     timeout=300)
    assert "legs_aborted" in line and line["n_gpus"] == 2 and line["value"] > 0 and "roofline" in line
    assert "c4" not in line["legs"] or "error" in line["legs"]["c4"]


@pytest.mark.gpu
def test_single_rank_line_has_the_contract_fields():
    line, detail = run_bench("--steps", "20", "--warmup", "5", "--small-legs", "--no-large", "--cpu-seconds", "0.5",
                             want_detail=True)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "legs", "detail"):
        assert key in line, key
    for key in ("workload", "points_per_gpu", "oob_fraction", "ranks", "backend", "gather"):
        assert key in line["config"], key
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "kernel", "launch_us",
                "launch_source", "frac_wall"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4
    assert roof["frac"] <= 1.0 and roof["frac_wall"] <= roof["frac"] * 1.02 and roof["launch_source"] == "hip events (this run)"
    assert detail["roofline"]["timed_regions_ms"]["n"] >= 15
    assert line["scaling_table"][0]["n"] == 1 and line["scaling_table"][0]["value"] == line["value"]
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / roof["launch_us"] / 1e3) < 1e-3 * roof["achieved"]
    long_roof = detail["roofline"]
    assert long_roof["dropin_call"]["ms_per_call"] > 0 and long_roof["dropin_call"]["queries_per_s"] > 1e9
    assert long_roof["launch_us_events_best_replay"] <= long_roof["launch_us_events"] and long_roof["frac_events_best_replay"] <= 1.0
    cpu = line["cpu_baseline"]
    # the op-for-op restatement of what the reference runs on CPU is the headline CPU figure, under a stable key beside the
    # fused C port's
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] == cpu["baseline_opforop"]
    assert cpu["fused_port"]["value"] == cpu["baseline_fused"] > cpu["value"]
    long_cpu = detail["cpu_baseline"]
    assert long_cpu["spread"]["samples"] >= 3 and long_cpu["thread_pinning"]["OMP_PROC_BIND"] == "close"
    assert long_cpu["spread"]["min"] <= long_cpu["value"] <= long_cpu["spread"]["max"]
    assert line["steps"] == 20 and line["warmup"] == 5 and line["n_gpus"] == 1
    assert detail["parity"]["oracle_pinned"] == {"view": False, "transform": False, "embree": False}


@pytest.mark.gpu
def test_rccl_calls_of_the_legs_run_on_one_rank():
    """The 8-GPU run belongs to the driver; what can be checked on a 1-GPU box is that every torch.distributed call of the
    multi-rank path is valid against the nccl (= RCCL) backend: process group with device_id, barrier, MAX all-reduce of the
    timing, ShardedSDF's all_gather_into_tensor of packed records, sharded_chamfer's all-reduces -- with a single rank (--force-pg)."""
    line, detail = run_bench(*SMALL, "--force-pg", want_detail=True)
    assert line["config"]["backend"] == "nccl"
    legs = detail["legs"]
    for name in ("c4", "c4_readme_grid", "c5"):
        assert "error" not in legs[name], legs[name]
    assert legs["c4"]["gathered"]["gather"] is True and "nccl" in legs["c4"]["gathered"]["collective"]
    assert legs["c4"]["gathered"]["output_shape"] == [[8, 16384], [8, 16384, 3]]
    assert legs["c4"]["gathered"]["path"] == "packed" and legs["c4"]["gathered"]["equals_unsharded_call"] is True
    assert "nccl" in legs["c5"]["collective"] and legs["c5"]["rel_err_vs_analytic"] < 1e-3
    assert line["legs"]["c4"]["gathered"]["backend"] == "nccl"
