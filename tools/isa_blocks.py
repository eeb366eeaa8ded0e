#!/usr/bin/env python3
"""Basic-block census of a kernel in a hipcc -save-temps .s file: instructions per block by class (VALU, SALU, scalar
memory, LDS, vector memory, waits) and the loop back-edges -- where a pass of a solve kernel spends its instructions.

    python tools/isa_blocks.py <file.s> [first line] [last line] [min block size]
"""
import re
import sys

S = open(sys.argv[1]).read().split("\n")
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = min(int(sys.argv[3]), len(S)) if len(sys.argv) > 3 else len(S)
minsz = int(sys.argv[4]) if len(sys.argv) > 4 else 25
blocks = []
for i in range(lo, hi):
    line = S[i]
    m = re.match(r"^(\.LBB\d+_\d+):", line)
    if m:
        blocks.append([m.group(1), i + 1, {}])
        continue
    t = line.strip()
    if not t or t[0] in ";." or t.endswith(":"):
        continue
    op = t.split()[0]
    if not blocks:
        blocks.append(["entry", i + 1, {}])
    d = blocks[-1][2]
    if op.startswith("v_"):
        key = "valu"
        if "f64" in op:
            d["f64"] = d.get("f64", 0) + 1
        if "dpp" in t:
            d["dpp"] = d.get("dpp", 0) + 1
    elif op.startswith("s_load") or op.startswith("s_buffer_load"):
        key = "smem"
    elif op.startswith("s_waitcnt"):
        key = "wait"
    elif op.startswith("s_cbranch") or op.startswith("s_branch"):
        key = "br"
        d.setdefault("targets", []).append(t.split()[-1])
    elif op.startswith("s_"):
        key = "salu"
    elif op.startswith("ds_"):
        key = "lds"
        if "bpermute" in op:
            d["bperm"] = d.get("bperm", 0) + 1
    else:
        key = "vmem"
    d[key] = d.get(key, 0) + 1
idx = {b[0]: k for k, b in enumerate(blocks)}
print("block        line     n  valu(f64,dpp) salu smem  lds(bperm) vmem wait  back-edges")
for k, b in enumerate(blocks):
    d = b[2]
    tot = sum(v for kk, v in d.items() if kk in ("valu", "salu", "smem", "lds", "vmem", "wait", "br"))
    back = [t for t in d.get("targets", []) if t in idx and idx[t] <= k]
    if tot >= minsz or back:
        print(f"{b[0]:12s} {b[1]:5d} {tot:5d} {d.get('valu', 0):5d}({d.get('f64', 0):3d},{d.get('dpp', 0):3d}) {d.get('salu', 0):4d} "
              f"{d.get('smem', 0):4d} {d.get('lds', 0):4d}({d.get('bperm', 0):3d}) {d.get('vmem', 0):4d} {d.get('wait', 0):4d}  {back}")
