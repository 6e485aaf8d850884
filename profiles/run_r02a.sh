#!/bin/bash
# Round 2, first GPU call: instruction-cost microbenchmark, the full GPU test
# suite, the default bench line and the small-sweep table per kernel variant.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r02a
mkdir -p $O
./profiles/ubench/valu_cost 2000 > $O/valu_cost.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err
for n in 20000 65536 100000 125000 250000 500000 1000000; do
  for v in 0 2; do
    python bench.py --no-cpu-baseline --no-parity-spot --steps 20 --warmup 3 \
        --sets $n --hbv-variant $v 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sets=$n variant=$v kernel_ms=%.3f ms_per_step=%.3f' % (d['roofline']['kernel_ms'], d['ms_per_step']))"
  done
done > $O/small_sweeps.txt 2>&1
python bench.py --gpus 2 --backend gloo --share-gpu --steps 10 --warmup 2 > $O/bench_2rank_gloo.json 2> $O/bench_2rank.err
tail -3 $O/pytest.log; cat $O/bench.json; cat $O/small_sweeps.txt; head -60 $O/valu_cost.txt
