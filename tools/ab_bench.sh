#!/bin/bash
# Alternating A/B runs of bench.py with two builds of the library on the same box: $1 = lib B path, $2.. = bench args
B=$1; shift
for i in 1 2 3; do
    echo "A: $(python bench.py --no-alt --no-cpu-baseline "$@" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"))')"
    echo "B: $(MG_HIP_LIB=$B python bench.py --no-alt --no-cpu-baseline "$@" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"))')"
done
