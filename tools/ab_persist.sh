# A/B of the single-launch forward against the launch-per-layer kernels on ONE box (devices differ by several %)
d=$1; mkdir -p gpurun_out/$d
for shape in "16 1000" "8 1000" "4 256" "1 1000" "1 256" "32 1000"; do
  set -- $shape
  for p in 0 1:16 1:32 1:64; do
    MG_DENOISER_PERSIST=${p%%:*} MG_PERSIST_NT=${p##*:} timeout -k 10 200 python bench.py --no-cpu-baseline --no-alt --steps 30 --batch $1 --frames $2 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read())
print('B=%s L=%s persist=%s: %.1f steps/s  %.3f ms/step  kernel %.4f ms  frac %.3f' % ('$1', '$2', '$p', b['value'], b['ms_per_step'], b['roofline']['kernel_ms'], b['roofline']['whole_step_frac']))
" | tee -a gpurun_out/$d/ab.txt
  done
done
