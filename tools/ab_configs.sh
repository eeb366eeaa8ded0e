#!/bin/bash
# ON THE GPU BOX: tools/all_configs.py on a subset of configs for each library given (DEXR_LIB), interleaved and repeated, so
# that kernel variants built into tools/_prof/ are compared inside ONE box (boxes differ by +-3 %).
#   bash tools/ab_configs.sh "shadow_hand_right_dexpilot,offline/leap_hand_right" libA.so libB.so
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
sel=$1; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    DEXR_LIB=$lib python tools/all_configs.py 65536 "$sel" 2>/dev/null | grep "yml" | awk -v l="$lib" -v r=$rep '{printf "%-28s rep %d %-44s %s ms  it %s max %s\n", l, r, $1, $7, $9, $10}'
  done
done
