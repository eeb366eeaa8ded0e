#!/usr/bin/env python3
"""CPU: register / spill / scratch / LDS table of every gfx950 kernel the library dispatches, read from the code-object notes
of the objects a build leaves in build/ (llvm-objcopy the .hip_fatbin section, clang-offload-bundler --unbundle, llvm-readelf
--notes) -- so that register and spill claims in DESIGN.md are checkable without a GPU.

    python tools/codeobj_resources.py [build_dir] > profiles/rNN_codeobj_resources.txt
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "f.bin"), os.path.join(td, "k.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return []
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
        if r.returncode != 0:
            return []
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out, cur = [], {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("name"):
            out.append(cur)
            cur = {}
        if k in ("agpr_count", "group_segment_fixed_size", "name", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count",
                 "vgpr_count", "vgpr_spill_count", "max_flat_workgroup_size"):
            cur[k] = v
    if cur.get("name"):
        out.append(cur)
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines()


def main():
    build = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "build")
    rows = []
    for f in sorted(os.listdir(build)):
        if f.endswith(".o"):
            for k in kernels_of(os.path.join(build, f)):
                rows.append((f, k))
    names = demangle([k["name"] for _, k in rows])
    print("# gfx950 code-object resources of every kernel in libdexr.so (tools/codeobj_resources.py: llvm-readelf --notes of the")
    print("# objects in build/).  vgpr = architectural VGPRs (+ agpr), spills = registers spilled (sgpr -> VGPR lanes, vgpr -> scratch),")
    print("# scratch = private_segment_fixed_size (bytes per lane), lds = static group segment (dynamic LDS is set at launch).")
    print(f"{'object':28s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'sgpr_spill':>10s} {'vgpr_spill':>10s} {'scratch':>8s} {'lds':>6s}  kernel")
    for (f, k), n in zip(rows, names):
        n = re.sub(r"\(dexr::KernelParams.*$", "", n).replace("void ", "")
        print(f"{f:28s} {k.get('vgpr_count', '?'):>5s} {k.get('agpr_count', '?'):>5s} {k.get('sgpr_count', '?'):>5s} "
              f"{k.get('sgpr_spill_count', '?'):>10s} {k.get('vgpr_spill_count', '?'):>10s} {k.get('private_segment_fixed_size', '?'):>8s} "
              f"{k.get('group_segment_fixed_size', '?'):>6s}  {n}")


if __name__ == "__main__":
    main()
