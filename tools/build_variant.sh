#!/bin/bash
# A copy of the library with extra compiler flags for SOME translation units (kernel experiments compared inside one GPU box
# with DEXR_LIB, tools/ab_*.sh): objects matching the egrep pattern are rebuilt with the flags, the rest is reused.
#   bash tools/build_variant.sh <name> "<flags>" "<egrep pattern over build/*.o names>"   -> tools/_prof/libdexr_<name>.so
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; flags=$2; pat=$3
export DEXR_BUILD_DIR=$R/build_v_$name DEXR_LIB_OUT=$R/tools/_prof/libdexr_$name.so DEXR_EXTRA_FLAGS="$flags"
mkdir -p "$DEXR_BUILD_DIR" "$R/tools/_prof"
for f in "$R"/build/*.o; do
  b=$(basename "$f")
  if echo "$b" | egrep -q "$pat"; then rm -f "$DEXR_BUILD_DIR/$b"; else cp -p "$f" "$DEXR_BUILD_DIR/"; fi
done
cd "$R" && python -m dex_retargeting_amd._build > /dev/null
ls -la "$DEXR_LIB_OUT"
