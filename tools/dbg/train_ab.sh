#!/bin/bash
# same-box A/B of the training step under an environment switch: tools/dbg/train_ab.sh VAR=VALUE [rounds]
sw=$1; n=${2:-3}
for i in $(seq $n); do
  echo "default : $(python bench.py --workload train --steps 30 --warmup 8 2>/dev/null | tail -1 | grep -o 'ms_per_step[^,]*')"
  echo "$sw : $(env $sw python bench.py --workload train --steps 30 --warmup 8 2>/dev/null | tail -1 | grep -o 'ms_per_step[^,]*')"
done
