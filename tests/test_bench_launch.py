"""bench.py's launcher: `--gpus N` must really start N ranks (the flag used to be parsed and ignored).

The CPU test drives the same spawn path the driver's `python bench.py --gpus N` takes, with `--launch-check` standing in
for the GPU work (ranks rendezvous over gloo on 127.0.0.1 and all-reduce a 1).  The GPU test runs the whole benchmark
with two ranks sharing the one GPU of the test box (collectives through gloo): every leg, rank-0 JSON, n_gpus == 2.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True,
                         timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


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


@pytest.mark.gpu
def test_two_ranks_share_the_gpu_and_every_leg_reports():
    line = run_bench("--gpus", "2", "--share-gpu", "--steps", "5", "--warmup", "2", "--points", "65536", "--small-legs",
                     "--no-large", "--no-cpu-baseline")
    assert line["n_gpus"] == 2 and line["config"]["ranks"] == 2 and line["scaling"] == "weak"
    assert line["parity"]["max_abs_val_err_vs_oracle"] == 0.0 and line["parity"]["grad_mismatches_vs_oracle"] == 0
    legs = line["legs"]
    for name in ("c3", "c4", "c4_readme_grid", "c5"):
        assert "error" not in legs[name], legs[name]
        assert legs[name]["scaling"] == "strong" and legs[name]["n_gpus"] == 2
    assert legs["c4"]["sharded"]["gather"] is False and legs["c4"]["gathered"]["gather"] is True
    assert legs["c4"]["gathered"]["output_shape"] == [[8, 16384], [8, 16384, 3]]
    assert legs["c4"]["gathered"]["path"] == "packed" and legs["c4"]["gathered"]["equals_unsharded_call"] is True
    assert legs["c4"]["gathered"]["bytes_received_per_rank"] == 8 * 8192 * 16 and legs["c4"]["gathered"]["xgmi_lower_bound_ms"] > 0
    assert legs["c4"]["gathered_by_configs"]["equals_unsharded_call"] is True
    assert legs["c5"]["rel_err_vs_analytic"] < 1e-3
    # the self-check the first real multi-GPU line will carry: rank count, received bytes = (W - 1) / W of the packed output,
    # kernel-only against gathered time (gloo here, so `ok` only asks for the backend when the ranks own their GPUs)
    chk = legs["c4"]["gathered"]["self_check"]
    assert chk["ranks"] == chk["ranks_expected"] == 2 and chk["bytes_match"] is True and chk["backend"] == "gloo"
    assert chk["bytes_received_per_rank_expected"] == 8 * 8192 * 16 and chk["gathered_ms"] > 0 and chk["kernel_only_ms"] > 0


@pytest.mark.gpu
def test_eight_ranks_share_the_gpu():
    """The driver's 8-GPU run cannot be rehearsed on a 1-GPU box, but its world size can: 8 ranks on the one GPU (gloo) walk
    the W = 8 partition / padding / index arithmetic of every leg; the gathered C4 results must equal the unsharded call."""
    line = run_bench("--gpus", "8", "--share-gpu", "--steps", "5", "--warmup", "2", "--points", "65536", "--small-legs",
                     "--no-large", "--no-cpu-baseline")
    assert line["n_gpus"] == 8 and "legs_aborted" not in line
    legs = line["legs"]
    for name in ("c4", "c4_readme_grid", "c5"):
        assert "error" not in legs[name] and "skipped" not in legs[name], legs[name]
    g = legs["c4"]["gathered"]
    assert g["path"] == "packed" and g["equals_unsharded_call"] is True and g["output_shape"] == [[8, 16384], [8, 16384, 3]]
    assert g["bytes_received_per_rank"] == 7 * 8 * 2048 * 16  # 7 peers x 8 configurations x 2048 padded points x 16 B
    assert legs["c4"]["gathered_by_configs"]["equals_unsharded_call"] is True  # one configuration per rank
    assert legs["c5"]["rel_err_vs_analytic"] < 1e-3


@pytest.mark.gpu
def test_a_rank_that_fails_before_a_legs_collectives_does_not_hang_the_others():
    """Rank 1 raises while preparing the C4 leg (whose packed path all-gathers): every rank skips that leg together -- one
    all-reduced flag before the leg's first collective -- and the remaining legs and the headline line still come out."""
    line = run_bench("--gpus", "2", "--share-gpu", "--steps", "5", "--warmup", "2", "--points", "65536", "--small-legs",
                     "--no-large", "--no-cpu-baseline", "--fail-rank", "1", "--fail-leg", "c4", timeout=300)
    legs = line["legs"]
    assert "skipped" in legs["c4"] and legs["c4"]["ranks_without_a_result"] == 2
    assert "error" not in legs["c5"] and legs["c5"]["rel_err_vs_analytic"] < 1e-3 and "error" not in legs["c4_readme_grid"]
    assert line["n_gpus"] == 2 and line["value"] > 0


@pytest.mark.gpu
def test_a_rank_lost_inside_a_leg_costs_the_legs_not_the_headline():
    """Rank 1 passes the C4 leg's gate and then never arrives at its first collective (the case the gate cannot see): after
    --legs-deadline rank 0 prints the line -- headline, roofline, the legs finished so far -- and every rank exits 0."""
    line = run_bench("--gpus", "2", "--share-gpu", "--steps", "5", "--warmup", "2", "--points", "65536", "--small-legs",
                     "--no-large", "--no-cpu-baseline", "--hang-rank", "1", "--fail-leg", "c4", "--legs-deadline", "20",
                     timeout=300)
    assert "legs_aborted" in line and line["n_gpus"] == 2 and line["value"] > 0 and "roofline" in line
    assert "c4" not in line["legs"] or "error" in line["legs"]["c4"]


@pytest.mark.gpu
def test_single_rank_line_has_the_contract_fields():
    line = run_bench("--steps", "20", "--warmup", "5", "--small-legs", "--no-large", "--cpu-seconds", "0.5")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert "SEPARATE" in roof["timing"] and "traffic_source" in roof
    assert roof["dropin_call"]["ms_per_call"] > 0 and roof["dropin_call"]["queries_per_s"] > 1e9
    assert "launch_ms_best_replay" in roof and roof["launch_ms_best_replay"] <= roof["launch_ms_mean"] and "frac_rocprof" in roof
    assert roof["frac"] <= 1.0 and roof["frac_best_replay"] <= 1.0
    cpu = line["cpu_baseline"]
    # the op-for-op restatement of what the reference runs on CPU is the headline CPU figure; the fused C port is nested
    assert cpu["kind"] == "op-for-op restatement" and cpu["cores"] >= 1 and cpu["spread"]["samples"] >= 3
    assert cpu["fused_port"]["kind"] == "port" and cpu["fused_port"]["value"] > cpu["value"]
    assert cpu["spread"]["min"] <= cpu["value"] <= cpu["spread"]["max"] and cpu["thread_pinning"]["OMP_PROC_BIND"] == "close"
    assert line["steps"] == 20 and line["warmup"] == 5 and line["n_gpus"] == 1
    assert "UNPINNED" in line["parity"]["oracle"]


@pytest.mark.gpu
def test_rccl_calls_of_the_legs_run_on_one_rank():
    """The 8-GPU run belongs to the driver; what can be checked on a 1-GPU box is that every torch.distributed call of the
    multi-rank path is valid against the nccl (= RCCL) backend: process group with device_id, barrier, MAX all-reduce of the
    timing, ShardedSDF's all_gather_into_tensor of packed records, sharded_chamfer's all-reduces -- with a single rank (--force-pg)."""
    line = run_bench("--steps", "5", "--warmup", "2", "--points", "65536", "--small-legs", "--no-large", "--no-cpu-baseline",
                     "--force-pg")
    assert line["config"]["backend"] == "nccl"
    legs = line["legs"]
    for name in ("c4", "c4_readme_grid", "c5"):
        assert "error" not in legs[name], legs[name]
    assert legs["c4"]["gathered"]["gather"] is True and "nccl" in legs["c4"]["gathered"]["collective"]
    assert legs["c4"]["gathered"]["output_shape"] == [[8, 16384], [8, 16384, 3]]
    assert legs["c4"]["gathered"]["path"] == "packed" and legs["c4"]["gathered"]["equals_unsharded_call"] is True
    assert "nccl" in legs["c5"]["collective"] and legs["c5"]["rel_err_vs_analytic"] < 1e-3
