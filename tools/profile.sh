#!/bin/bash
# Development tool: rocprofv3 kernel-trace stats and HBM PMC passes for one bench workload.
# usage (on the GPU box, from the repo root): bash tools/profile.sh <workload> <outdir>
set -u
WL=${1:-resnet18_fedavg_k100}
OUT=${2:-gpurun_out/prof_$WL}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- \
    python bench.py --workload "$WL" --steps 20 --no-cpu-baseline > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- \
    python bench.py --workload "$WL" --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/fetch.log" 2>&1 || { echo "fetch pmc failed"; tail -5 "$OUT/fetch.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- \
    python bench.py --workload "$WL" --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/write.log" 2>&1 || { echo "write pmc failed"; tail -5 "$OUT/write.log"; exit 1; }
find "$OUT" -name "*.csv" | head -20
