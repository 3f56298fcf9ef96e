#!/bin/bash
# rocprofv3 kernel trace of the GAN training step (bench.py --workload train), summarised per kernel.
out=$1; shift
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 bench.py --workload train --steps 10 --warmup 3 $* > "$out/kt_bench.json" 2> "$out/kt.err" || echo "kernel-trace pass failed"
python3 tools/summarise_rocprof.py "$out" > "$out/summary.txt" 2>&1
head -45 "$out/kernel_stats.csv"
cat "$out/kt_bench.json"
