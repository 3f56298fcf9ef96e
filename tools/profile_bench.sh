#!/bin/bash
# rocprofv3 passes over `bench.py` (kernel trace + stats, then one PMC group per pass, never combined with other trace
# domains), summarised per kernel by tools/summarise_rocprof.py.  Usage: tools/profile_bench.sh <outdir> [bench args...]
out=$1; shift
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p "$out"
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-alt $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 bench.py $ARGS > "$out/kt_bench.json" 2> "$out/kt.err" || echo "kernel-trace pass failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_mfma" -- python3 bench.py $ARGS > "$out/pmc_mfma_bench.json" 2> "$out/pmc_mfma.err" || echo "pmc mfma pass failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -- python3 bench.py $ARGS > "$out/pmc_fetch_bench.json" 2> "$out/pmc_fetch.err" || echo "pmc fetch pass failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -- python3 bench.py $ARGS > "$out/pmc_write_bench.json" 2> "$out/pmc_write.err" || echo "pmc write pass failed"
python3 tools/summarise_rocprof.py "$out"
