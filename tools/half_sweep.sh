#!/bin/bash
# Development tool: the bf16 bench workloads under every variant library of the 16-bit reduce kernel
# (tools/build_variant.py ... fedagg_half.hip -DFA_HALF_VPT=.. -DFA_HALF_AHEAD=..).
# usage (GPU box, repo root): bash tools/half_sweep.sh "<workload> ..." > log
cd "$GRAFT_REPO_ROOT" || exit 1
WLS=${1:-resnet50_fedadam_k64_bf16}
for lib in appfl_amd/lib/libappfl_amd.so appfl_amd/lib/variants/lib_h_*.so; do
    for wl in $WLS; do
        out=$(APPFL_AMD_LIB=$PWD/$lib timeout -k 10 120 python bench.py --workload $wl --only --steps 20 --no-cpu-baseline 2>/dev/null) || { echo "$lib $wl failed"; exit 1; }
        echo "$out" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-48s %-28s %.4f ms  frac %.4f' % ('$lib'.split('/')[-1], '$wl', r['kernel_avg_ms'], r['frac']))"
    done
done
