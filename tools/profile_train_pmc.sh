#!/bin/bash
# PMC passes (one counter group each) over the GAN training step; per-kernel means by tools/summarise_rocprof.py
out=$1; shift
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p "$out"
A="--workload train --steps 4 --warmup 2"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_mfma" -- python3 bench.py $A > "$out/a.json" 2> "$out/a.err" || echo fail
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -- python3 bench.py $A > "$out/b.json" 2> "$out/b.err" || echo fail
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -- python3 bench.py $A > "$out/c.json" 2> "$out/c.err" || echo fail
python3 tools/summarise_rocprof.py "$out" > "$out/summary.txt" 2>&1
python3 - <<PY
import json
for e in json.load(open("$out/pmc_summary.json"))[:8]:
    print(e["kernel"][:60], e["launches"], e.get("mfma_busy_frac"), e.get("avg_us_under_counters"), e.get("hbm_bytes_raw"), e.get("hbm_bytes_fetch_doubled"))
PY
