#!/bin/bash
# same-box A/B of the discriminator and the training step with / without the split-reduction convolutions
for i in 1 2 3; do
  echo "split   : $(python tools/dbg/jcu_time.py 2>&1 | tail -1)"
  echo "no split: $(MG_CONV_SPLIT=0 python tools/dbg/jcu_time.py 2>&1 | tail -1)"
done
for i in 1 2; do
  echo "train split   : $(python bench.py --workload train --steps 30 --warmup 8 2>&1 | tail -1 | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')"
  echo "train no split: $(MG_CONV_SPLIT=0 python bench.py --workload train --steps 30 --warmup 8 2>&1 | tail -1 | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')"
done
