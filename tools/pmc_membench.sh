#!/bin/bash
# Development tool: L2 <-> fabric counters of the membench3 kernels (one rocprofv3 --pmc pass per
# counter group).  usage (GPU box, repo root): bash tools/pmc_membench.sh <case> <layout> <outdir>
set -u
CASE=${1:-4}; LAY=${2:-0}; OUT=${3:-gpurun_out/pmc_mb}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
i=0
for grp in "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_CYCLE_sum" \
           "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" \
           "TCC_BUSY_sum TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum TCC_IB_STALL_sum" \
           "TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_NORMAL_WRITEBACK_sum"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $grp --output-format csv -d "$OUT/p$i" -o c -- ./tools/build/membench3 2 "$CASE" "$LAY" > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/p$i.log"; exit 1; }
done
find "$OUT" -name "*counter_collection.csv"
