#!/bin/bash
# Development tool: rocprofv3 kernel-trace stats and PMC passes (HBM bytes, L2 <-> fabric stalls,
# L2 hit rate) for one bench workload.
# usage (on the GPU box, from the repo root): bash tools/profile.sh <workload> <outdir> ["<bench flags>"]
set -u
WL=${1:-resnet18_fedavg_k100}
OUT=${2:-gpurun_out/prof_$WL}
EXTRA=${3:-}      # further bench.py flags, e.g. "--cold" or "--tune-stash 0"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- \
    python bench.py --workload "$WL" --only --steps 20 --no-cpu-baseline $EXTRA > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
pass() {  # name, output prefix, counters...
    local name=$1 pre=$2; shift 2
    timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$pre" -- \
        python bench.py --workload "$WL" --only --steps 5 --warmup 1 --no-cpu-baseline $EXTRA > "$OUT/$name.log" 2>&1 || { echo "$name pmc failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
pass fetch f FETCH_SIZE
pass write w WRITE_SIZE
pass stall s TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_CYCLE_sum GRBM_GUI_ACTIVE
pass l2 h TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum
find "$OUT" -name "*.csv" | head -20
