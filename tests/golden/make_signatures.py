"""Writes tests/golden/reference_signatures.json: for every public function / method of /root/reference/src/pytorch_volumetric, its
positional parameter NAMES and which of them have defaults (names only -- no source text, no default values).  Run in the build
container, where the reference tree exists; tests/test_host_logic.py::test_reference_signatures_are_kept checks this package against it."""
import ast
import json
import os

REF = "/root/reference/src/pytorch_volumetric"
out = {}


def sig(node):
    names = [a.arg for a in node.args.args]
    nd = len(node.args.defaults)
    return [names, names[len(names) - nd:] if nd else []]


for f in sorted(os.listdir(REF)):
    if not f.endswith(".py") or f == "__init__.py":
        continue
    mod = f[:-3]
    for node in ast.parse(open(os.path.join(REF, f)).read()).body:
        if isinstance(node, ast.FunctionDef) and not node.name.startswith("_") and node.name != "fmt":
            out[f"{mod}.{node.name}"] = sig(node)
        elif isinstance(node, ast.ClassDef):
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and (not sub.name.startswith("_") or sub.name in ("__init__", "__call__")):
                    out[f"{mod}.{node.name}.{sub.name}"] = sig(sub)
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_signatures.json"), "w") as fh:
    json.dump(out, fh, indent=0, sort_keys=True)
print(len(out), "signatures")
