#!/bin/bash
# A/B: HBV kernel with VOP3 selects (main) vs hipcc's own selects (nosel);
# small-sweep variant with the mid-day prefetch.  Then the new GPU tests.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r02c
mkdir -p $O
cp rrmpg_amd/librrhip.so /tmp/librrhip_main.so
one() { # lib sets variant
  python bench.py --no-cpu-baseline --no-parity-spot --steps 20 --warmup 3 \
      --sets $2 --hbv-variant $3 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$1 sets=$2 variant=$3 kernel_ms=%.3f' % d['roofline']['kernel_ms'])"
}
for rep in 1 2; do
for lib in main nosel; do
  if [ $lib = main ]; then cp /tmp/librrhip_main.so rrmpg_amd/librrhip.so; else cp scratch_dbg/ab/librrhip_nosel.so rrmpg_amd/librrhip.so; fi
  for n in 1000000 125000 65536 20000; do
    for v in 0 2; do one $lib $n $v; done
  done
done
done > $O/ab.txt 2>&1
cp /tmp/librrhip_main.so rrmpg_amd/librrhip.so
python bench.py --mode metric --no-cpu-baseline --steps 10 > $O/bench_metric.json 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
cat $O/ab.txt; tail -15 $O/pytest.log
