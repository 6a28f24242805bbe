# Development tool: the late-epilogue kernel's epilogue batch (library variants) on the north-star round.
for pass in 1 2; do for l in ${VARIANTS:-main}; do for c in "" "--cold"; do for wl in resnet50_fedadam_k64 resnet50_fedadam_k32; do
  if [ $l = main ]; then unset APPFL_AMD_LIB; else export APPFL_AMD_LIB=appfl_amd/lib/variants/lib_$l.so; fi
  timeout -k 10 100 python bench.py --workload $wl --only --steps 50 --no-cpu-baseline $c 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$l', '$wl', '$c', r['kernel'], r['kernel_avg_ms'], r['frac'])" || exit 1
done; done; done; done
