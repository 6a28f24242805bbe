#!/bin/bash
# One-off longer sweep of the seeded GPU fuzz tests on fresh seeds: one pytest process per
# offset, stops at the first failure.  Usage: tools/long_fuzz.sh 1000 2000 3000 ...
set -u
mkdir -p gpurun_out
for off in "$@"; do
    echo "== offset $off"
    APPFL_AMD_FUZZ_OFFSET=$off timeout -k 10 300 python -u -m pytest tests/test_fuzz_gpu.py tests/test_half_gpu.py \
        -x -q -m gpu --timeout 120 --timeout-method thread -k "random" > gpurun_out/long_fuzz_$off.log 2>&1
    rc=$?
    tail -2 gpurun_out/long_fuzz_$off.log
    if [ $rc -ne 0 ]; then echo "offset $off failed (rc $rc)"; exit $rc; fi
done
