"""Loops of one kernel in a gfx950 disassembly (tools/disasm.sh output): every backward branch = one loop [target, branch]; per
loop the instruction mix (VALU / SALU / SMEM / VMEM / LDS / scratch / branches).  No GPU needed.
    python tools/isa_loops.py /tmp/pvamd_composed.s 'composed_query_wave<2, 0, false, 1>'"""
import re
import subprocess
import sys

path, want = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = end = None
for i, l in enumerate(lines):
    m = re.match(r"^([0-9a-f]+) <(.*)>:", l)
    if m:
        name = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
        if start is not None and end is None:
            end = i
        if want in name and start is None:
            start = i
if end is None:
    end = len(lines)
ins = []
for l in lines[start + 1:end]:
    m = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-F]+):", l)
    if m:
        ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
addr_index = {a: i for i, (a, _, _) in enumerate(ins)}


def kind(op):
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("v_cmp", "v_")):
        return "valu"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


loops = []
for i, (a, op, args) in enumerate(ins):
    if op.startswith(("s_cbranch", "s_branch")):
        m = re.search(r"(-?\d+)\s*$", args)
        # objdump prints the target as a label or an offset; compute from the simm16 when numeric
        t = re.search(r"<[^>]*\+0x([0-9a-f]+)>", args)
        if t is None:
            continue
        tgt = ins[0][0] - 0 + 0  # placeholder
print(f"kernel {want}: {len(ins)} instructions")
# objdump -d without symbolization prints raw simm16: decode
for i, (a, op, args) in enumerate(ins):
    if op.startswith(("s_cbranch", "s_branch")):
        m = re.match(r"^(\d+)$", args.strip().split()[-1]) if args.strip() else None
        if m:
            simm = int(m.group(1))
            if simm >= 32768:
                simm -= 65536
            tgt = a + 4 + 4 * simm
            if tgt <= a and tgt in addr_index:
                loops.append((addr_index[tgt], i))
for lo, hi in sorted(loops, key=lambda x: (x[1] - x[0])):
    mix = {}
    for _, op, _ in ins[lo:hi + 1]:
        mix[kind(op)] = mix.get(kind(op), 0) + 1
    print(f"loop insn[{lo}..{hi}] ({hi - lo + 1} instr): " + " ".join(f"{k}={v}" for k, v in sorted(mix.items())))
