d=$1; mkdir -p gpurun_out/$d
for f in $FL; do MG_PERSIST_FLAGS=$f timeout -k 10 200 python tools/persist_timeline.py > gpurun_out/$d/timeline_f$f.json 2>gpurun_out/$d/err_f$f.txt; MG_PERSIST_FLAGS=$f timeout -k 10 200 python bench.py --no-cpu-baseline --no-alt --steps 30 > gpurun_out/$d/bench_f$f.json 2>>gpurun_out/$d/err_f$f.txt; done
[ -n "$NOTEST" ] || timeout -k 10 300 python -m pytest tests/test_gpu_persist.py -x -q 2>&1 | tail -2
python - <<PY
import json
for f in [int(x) for x in "$FL".split()]:
    t=json.load(open("gpurun_out/$d/timeline_f%d.json"%f)); b=json.load(open("gpurun_out/$d/bench_f%d.json"%f))
    print(f, b["value"], b["ms_per_step"], "layer", t["layer_cycles_mean"], t["layer_cycles_by_role"], t["tile_total_cycles"], "pro", t["prologue_cycles_mean"], "tail", t["tail_cycles_mean"])
    for k in t["phases_mean_cycles"]:
        print("  %-28s mean %8.0f  p5 %8.0f p50 %8.0f p95 %8.0f"%(k,t["phases_mean_cycles"][k],t["phases_p5_cycles"][k],t["phases_p50_cycles"][k],t["phases_p95_cycles"][k]))
PY
