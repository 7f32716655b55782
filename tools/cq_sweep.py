"""C2 kernel at 1M / 8M / 64M points for A/B builds: PVAMD_LIB=tools/variants/libpvamd_X.so python tools/cq_sweep.py
CQ_MARGINS="0.05,-0.001,9" adds, at 64M points, the all-in-range and all-out-of-range cases (what the gathers cost)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pytorch_volumetric_amd as pv
import workloads as Wk

cached = Wk.build_c2_cache()
out = []
margins = [float(m) for m in os.environ.get("CQ_MARGINS", "0.05").split(",")]
sizes = [int(x) for x in os.environ.get("CQ_LOGP", "20,23,26").split(",")]
if os.environ.get("CQ_P"):  # explicit point counts (any integer), e.g. CQ_P=1310720,1572864
    sizes = [float(np.log2(int(x))) for x in os.environ["CQ_P"].split(",")]
for logp in sizes:
    for margin in (margins if logp == 26 else margins[:1]):
        P = int(round(2.0 ** logp))
        pts = Wk.c2_points(cached, P, seed=99, margin=margin)
        # CQ_COLD_MB=N: rotate over enough distinct point buffers (N MB of xyz in total) that a launch never finds its points in the
        # 256 MB Infinity Cache from the replay before
        n_rot = max(1, -(-int(os.environ.get("CQ_COLD_MB", "0")) * (1 << 20) // (12 * P)))
        rot = [pts] + [Wk.c2_points(cached, P, seed=100 + j, margin=margin) for j in range(n_rot - 1)]
        val = torch.empty((P,), dtype=torch.float32, device="cuda")
        grad = torch.empty((P, 3), dtype=torch.float32, device="cuda")
        reps = 2000 if logp <= 20 else (400 if logp <= 22 else (200 if logp == 23 else 40))
        for _ in range(max(60, reps // 4)):
            cached.query_into(pts, val, grad)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for j in range(reps):
                    cached.query_into(rot[j % n_rot], val, grad)
        torch.cuda.current_stream().wait_stream(side)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        tag = ("cold " if n_rot > 1 else "") + (f"2^{logp}" if float(logp).is_integer() else f"{P}") + ("" if margin == margins[0] else f"[margin {margin}]")
        out.append(f"{tag}: {best*1e3:.2f} us {28*P/best/1e6:.0f} GB/s")
        del g, pts, val, grad, rot
print(os.environ.get("PVAMD_LIB", "default"), " | ".join(out), flush=True)
