#!/bin/bash
# Fuzz soak on the GPU box: tests/test_gpu_fuzz.py once per seed, one line per
# seed in the format of profiles/r04_fuzz_soak.txt.
# Usage (repo root, via gpurun):  bash profiles/fuzz_soak.sh FIRST LAST "note" >> gpurun_out/fuzz_soak.txt
cd "$(dirname "$0")/.." || exit 1
FIRST=${1:?first seed}; LAST=${2:?last seed}; NOTE=${3:-}
for s in $(seq "$FIRST" "$LAST"); do
  line=$(RR_FUZZ_SEED=$s timeout 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -1)
  echo "seed $s: $line${NOTE:+ ($NOTE)}"
done
