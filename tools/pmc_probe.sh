#!/bin/bash
# Run ON THE GPU BOX: one rocprofv3 --pmc pass per counter group around tools/prof_config.py (groups that the device
# rejects are reported and skipped).  Usage: bash tools/pmc_probe.sh <config.yml> <kernel> <batch> "<group 1>" "<group 2>" ...
set -u
CFG=$1; KER=$2; BATCH=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for P in "$@"; do
  D=/tmp/pmcp_$$_$(echo $P | tr ' ' '_' | cut -c1-40)
  if timeout 300 rocprofv3 --pmc $P -d "$D" -- python "$R/tools/prof_config.py" "$CFG" "$KER" "$BATCH" 3 > /tmp/pmcp.log 2>&1; then
    python "$R/tools/pmc_summary.py" $(find "$D" -name "*.db") 2>/dev/null | grep -v "^#" | sed 's/void dexr:://' | cut -c1-30,70-140
  else
    echo "group [$P] rejected: $(grep -i -m1 'error\|invalid\|not' /tmp/pmcp.log | cut -c1-160)"
  fi
  rm -rf "$D"
done
