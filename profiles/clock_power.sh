#!/bin/bash
# Shader clock and socket power while the headline sweep runs back to back,
# qsim mode against score-only mode: which roof the chip is at (rocm-smi
# samples every 0.7 s; the first seconds of each run are setup).
cd "$(dirname "$0")/.." || exit 1
for mode in qsim metric; do
  timeout 120 python bench.py --no-cpu-baseline --no-parity-spot --no-extra-configs --no-power-soak --live-counters none \
      --steps 600 --warmup 3 --mode $mode > /tmp/cp_$mode.json 2>/dev/null &
  BP=$!
  sleep 6
  for i in 1 2 3 4 5 6; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" |
      sed -e 's/GPU\[0\]//' -e 's/^[ \t:]*//' | tr '\n' ';'
    echo
    sleep 0.7
  done
  wait $BP
  python - <<PY
import json
d = json.loads(open('/tmp/cp_$mode.json').read().strip().splitlines()[-1])
print('mode $mode: kernel_ms %.3f' % d['roofline']['kernel_ms'])
PY
done
