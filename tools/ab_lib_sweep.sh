#!/bin/bash
# Development tool: the same tools/sweep.py run once per library variant (APPFL_AMD_LIB).
# usage: bash tools/ab_lib_sweep.sh "<cfg>" "<shapes>" variant [variant ...]   ("main" = the product build)
CFG=$1; SHAPES=$2; shift 2
for v in "$@"; do
  echo "== lib $v"
  L=$PWD/appfl_amd/lib/variants/lib_$v.so; [ "$v" = main ] && L=$PWD/appfl_amd/lib/libappfl_amd.so
  APPFL_AMD_LIB=$L timeout -k 10 400 python tools/sweep.py --blocks --cfg "$CFG" --shapes "$SHAPES" --reps 10 2>&1 | grep -E "==|shape" || exit 1
done
