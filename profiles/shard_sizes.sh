#!/bin/bash
# Kernel time of one GPU's shard of a million-set sweep at 8 / 4 / 2 / 1 GPUs
# (strong scaling), 30 timed sweeps each.
cd "$(dirname "$0")/.." || exit 1
for spec in "cemaneigegr4j metric" "gr4j metric" "hbvedu qsim"; do
  set -- $spec
  for n in 125000 250000 500000 1000000; do
    python bench.py --no-cpu-baseline --no-parity-spot --no-extra-configs --no-power-soak --live-counters none --steps 30 --warmup 3 --model $1 --mode $2 --sets $n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('model=$1 mode=$2 sets=$n kernel_ms=%.3f' % d['roofline']['kernel_ms'])"
  done
done
