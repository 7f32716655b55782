"""Per-launch floor of this box's hipGraph harness: a tiny torch kernel, an empty-ish launch, copies of growing size."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from c2_floor_probe import graph_us
out = []
t = torch.zeros(64, device="cuda")
out.append(f"torch fill_ of 64 floats: {graph_us(lambda: t.fill_(1.0)):.2f} us")
for mb in (0.25, 1, 4, 14):
    n = int(mb * (1 << 20) / 4)
    a, b = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    out.append(f"torch copy {mb} MB -> {mb} MB: {graph_us(lambda: b.copy_(a)):.2f} us")
print("\n".join(out), flush=True)
