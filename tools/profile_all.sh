#!/bin/bash
# Development tool: tools/profile.sh over the benchmark workloads of one round (on the GPU box).
# usage: bash tools/profile_all.sh <outdir> <workload[:tag[:flags]]> ...
set -u
OUT=$1; shift
for spec in "$@"; do
    IFS=':' read -r wl tag flags <<< "$spec"
    tag=${tag:-$wl}
    echo "== $wl -> $tag ($flags)"
    bash tools/profile.sh "$wl" "$OUT/$tag" "$flags" > "$OUT/$tag.log" 2>&1 || { echo "profile of $tag failed"; tail -5 "$OUT/$tag.log"; }
done
