#!/bin/bash
# Same-box A/B of the current tree against an older checkout under .ab_old/ (git worktree, built in place):
#   denoise headline (old tree; new tree with everything inside every step; new tree hoisted), optionally the training step.
one() { python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("roofline",{}).get("kernel_ms"))'; }
for i in 1 2 3; do
    echo "old denoise: $(cd .ab_old && python bench.py --no-alt --no-cpu-baseline 2>/dev/null | one)"
    echo "new denoise (per-step): $(python bench.py --no-alt --no-cpu-baseline --project-per-step 2>/dev/null | one)"
    echo "new denoise (hoisted): $(python bench.py --no-alt --no-cpu-baseline 2>/dev/null | one)"
done
if [ "$1" = train ]; then
    for i in 1 2 3; do
        echo "old train: $(cd .ab_old && python bench.py --workload train 2>/dev/null | one)"
        echo "new train: $(python bench.py --workload train 2>/dev/null | one)"
    done
fi
