#!/bin/bash
# Everything under profiles/r03_* that comes from a GPU box, in one call (run from the repo root on the box, e.g. through
# `gpurun --timeout 1200 -- tools/profile_round.sh gpurun_out/round`).  Each leg is independent; copy what you want judged
# from <out>/ into profiles/ afterwards (see profiles/README.md for which file came from which leg).
out=${1:-gpurun_out/round}
mkdir -p "$out"
set -x
python bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"
tools/profile_bench.sh "$out/bench_prof" > "$out/bench_prof.txt" 2>&1
python bench.py --workload train --steps 20 --warmup 5 > "$out/bench_train.json" 2>/dev/null
MG_BENCH_EXCHANGE=1 python bench.py --workload train --steps 20 --warmup 5 > "$out/bench_train_rccl_world1.json" 2>/dev/null
tools/profile_train.sh "$out/train_prof" > "$out/train_prof.txt" 2>&1
tools/profile_train_pmc.sh "$out/train_pmc" > "$out/train_pmc.txt" 2>&1
python tools/dbg/one_utt_time.py > "$out/one_utterance_teams.txt" 2>/dev/null
python tests/perf_configs.py > "$out/configs.jsonl" 2>/dev/null
python tests/perf_configs.py e2e > "$out/e2e.jsonl" 2>/dev/null
for u in mfma_f32_rate mfma_f32_power mfma_f32_mix; do
    hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o "$out/$u" tools/ubench/$u.hip 2>/dev/null && "$out/$u" > "$out/ubench_$u.txt" 2>&1
done
set +x
ls -la "$out"
