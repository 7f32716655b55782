"""Which pvamd_grid_t.rule does a given value-range view implement?

The reference's CachedSDF indexes its cache through multidim_indexing's TorchMultidimView (sdf.py:521,537-540), a package that
is neither vendored nor pinned nor installable here, so this library restates the view's behaviour under a DEFAULT rule and
keeps the alternatives behind a field (include/pvamd.h PVAMD_RULE_*, pv.voxel.INDEX_RULE).  Given the real thing -- a
`torch_view.py` dropped next to the repository, or an installed multidim_indexing -- this tool probes it on inputs where the
rules differ and prints the assignment that makes this library agree with it:

    python tools/detect_index_rule.py path/to/multidim_indexing/torch_view.py      # or no argument: import multidim_indexing

`detect(view_cls)` works on any class with the constructor / ensure_index_key / get_valid_values the reference calls; the CPU
test (tests/test_index_rules.py) runs it against stand-in views of every rule.  No GPU needed."""
import ast
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_volumetric_amd import _lib  # noqa: E402  (constants only)


def load_view_class(path=None):
    if path is None:
        from multidim_indexing import torch_view
        return torch_view.TorchMultidimView
    tree = ast.parse(open(path).read())
    ns = {"torch": torch, "__name__": "torch_view_lifted"}
    exec(compile(tree, path, "exec"), ns)  # a one-file module whose only import is torch (and typing)
    return ns["TorchMultidimView"]


def detect(view_cls):
    """-> (rule bits, {observation: what was seen})"""
    seen = {}
    # a grid whose planes float32 hits exactly: origin 0, resolution 2^-3, 9 cells per axis; float64 ranges (numpy-like)
    src = torch.zeros(9, 9, 9)
    v = view_cls(src, value_ranges=[(0.0, 1.0)] * 3, invalid_value=0)

    def key(x):
        p = torch.tensor([[x, 0.5, 0.5]], dtype=torch.float32)
        return int(v.ensure_index_key(p)[0, 0])

    def valid(x):
        p = torch.tensor([[x, 0.5, 0.5]], dtype=torch.float32)
        return bool(v.get_valid_values(p)[0])

    # rounding of exact halves: q = 2.5 and q = 3.5 (x = 0.3125, 0.4375)
    k25, k35 = key(0.3125), key(0.4375)
    seen["index of quotient 2.5 / 3.5"] = (k25, k35)
    if (k25, k35) == (2, 4):
        rounding = 0
    elif (k25, k35) == (3, 4):
        # half away from zero and floor(q + 0.5) agree for positive q: a negative half tells them apart (q = -0.5 at x = -1/16)
        kneg = key(-0.0625)
        seen["index of quotient -0.5"] = kneg
        rounding = _lib.RULE_ROUND_HALF_AWAY if kneg == -1 else _lib.RULE_ROUND_FLOOR_HALF
    else:
        raise SystemExit(f"unknown rounding: quotients 2.5 / 3.5 map to {k25} / {k35} (truncation? then voxel centres do not "
                         "map to themselves and sdf.py:508-512 would fail)")
    # validity: a point a quarter voxel outside the range (x = -1/32, quotient -0.25 -> index 0)
    inside_q, outside_q = valid(-0.03125), valid(-0.09375)  # quotients -0.25 (rounds to 0) and -0.75 (rounds to -1)
    seen["valid at quotient -0.25 / -0.75"] = (inside_q, outside_q)
    if outside_q:
        raise SystemExit("points 3/4 of a voxel outside the range are valid: no rule of this library describes that")
    on_index = _lib.RULE_VALID_ON_INDEX if inside_q else 0
    # resolution dtype for a float32 range: (max - min) / (shape - 1) with numbers where float32 and float64 disagree
    res_bits = 0
    lo, hi, n = 0.1, 0.7, 8
    v32 = view_cls(torch.zeros(n, n, n), value_ranges=[(lo, hi)] * 3, invalid_value=0)
    res = getattr(v32, "_resolution", None)
    if res is not None and res.dtype == torch.float32:
        f32 = (torch.tensor(hi) - torch.tensor(lo)) / torch.tensor(float(n - 1))
        f64 = ((torch.tensor(hi).double() - torch.tensor(lo).double()) / (n - 1)).float()
        seen["float32-range resolution (view, float32 arithmetic, float64 arithmetic)"] = (float(res[0]), float(f32), float(f64))
        if float(f32) != float(f64):
            res_bits = _lib.RULE_RES_F64 if float(res[0]) == float(f64) else 0
    elif res is not None:
        seen["float32-range resolution dtype"] = str(res.dtype)
    return rounding | on_index | res_bits, seen


def describe(rule):
    names = [n for n, b in (("RULE_VALID_ON_INDEX", 1), ("RULE_ROUND_HALF_AWAY", 2), ("RULE_ROUND_FLOOR_HALF", 4), ("RULE_RES_F64", 8))
             if rule & b]
    return " | ".join("pv." + n for n in names) if names else "0  (the default: this library already agrees)"


if __name__ == "__main__":
    cls = load_view_class(sys.argv[1] if len(sys.argv) > 1 else None)
    rule, seen = detect(cls)
    for k, val in seen.items():
        print(f"  {k}: {val}")
    print(f"pv.voxel.INDEX_RULE = {describe(rule)}")
    print("then: python tests/golden/make_golden.py (with the real view instead of the shim) and python -m pytest tests -m gpu")
