#!/bin/bash
# ON THE GPU BOX: the headline bench line (Allegro vector, 65 536 frames) for each library given (DEXR_LIB), to compare kernel
# variants built into tools/_prof/ in one call.   bash tools/ab_headline.sh [lib.so ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
for lib in "$@"; do
  export DEXR_LIB=$lib
  for rep in 1 2; do
  python bench.py --headline-only --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$lib', '%.4f ms' % d['ms_per_step'], 'kernel %.4f' % (d['roofline'].get('kernel_ms') or 0), d['roofline'].get('kernel'))
"
  done
done
