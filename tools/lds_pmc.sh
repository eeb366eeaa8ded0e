#!/bin/bash
# Run ON THE GPU BOX: LDS / VALU activity counters of one workload's solve kernel (separate PMC passes, kernel-trace off).
# Usage: bash tools/lds_pmc.sh [workload]  -> gpurun_out/lds_pmc_<workload>.txt
set -u
W=${1:-shadow_dexpilot}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O"/ldspmc_*
for set in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_VALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set -d "$O/ldspmc_$tag" -o out --output-format csv -- python "$R/bench.py" --workload $W --no-cpu-baseline --headline-only --steps 3 --warmup 1 > /dev/null 2> "$O/ldspmc_$tag.err"
done
python - "$O" "$W" <<'PY'
import csv, glob, os, sys, collections
O, W = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(O, "ldspmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "dexr" not in k: continue
        acc[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(os.path.join(O, f"lds_pmc_{W}.txt"), "w") as out:
    for k, d in acc.items():
        out.write(k + "\n")
        for c, v in sorted(d.items()):
            v = v[len(v)//2:]  # the later dispatches (timed steps)
            out.write(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}\n")
print(open(os.path.join(O, f"lds_pmc_{W}.txt")).read())
PY
rm -rf "$O"/ldspmc_*
