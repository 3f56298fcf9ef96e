#!/bin/bash
# End-of-round profile set (run from the repo root on the GPU box: gpurun --timeout 1200 -- tools/profile_round_k.sh <tag>)
t=${1:-r03_k}
python bench.py > gpurun_out/${t}_bench.json 2> gpurun_out/${t}_bench.err
tools/profile_bench.sh gpurun_out/$t --steps 40 --warmup 8 > gpurun_out/$t.txt 2>&1
python tests/perf_configs.py > gpurun_out/${t}_configs.jsonl 2>/dev/null
python tests/perf_configs.py e2e > gpurun_out/${t}_e2e.jsonl 2>/dev/null
python bench.py --workload train > gpurun_out/${t}_bench_train.json 2>/dev/null
head -4 gpurun_out/$t/kernel_stats.csv
python - "$t" <<'PY'
import json, sys
t = sys.argv[1]
d = json.loads(open("gpurun_out/%s_bench.json" % t).read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["kernel_ms"], r.get("executed"), r["whole_step_frac"])
p = d["projection_in_every_step"]
print(p["value"], p["roofline"]["frac"], p["roofline"]["kernel_ms"])
c = d["cpu_baseline"]
print(c["value"], d["speedup_vs_cpu"], c["parity"]["x0_pred"]["max_abs_over_max_ref"], c["parity"]["final_mel"]["max_abs_over_max_ref"], d["alt"]["value"])
for e in json.load(open("gpurun_out/%s/pmc_summary.json" % t))[:2]:
    print(e["kernel"], e["launches"], e.get("mfma_busy_frac"), e.get("avg_us_under_counters"), e.get("hbm_bytes_fetch_doubled"), e.get("effective_clock_GHz"))
PY
cat gpurun_out/${t}_configs.jsonl gpurun_out/${t}_e2e.jsonl
cut -c1-200 gpurun_out/${t}_bench_train.json
