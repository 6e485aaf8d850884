#!/bin/bash
# A/B on one box: HBV kernel of r02c without VOP3 selects (= round-1 kernel,
# "base") vs the current build ("new": box mask as the fastpow vote, no v_mov
# on days without the power, 32-bit day counters, mid-day prefetch variant).
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r02d
mkdir -p $O
cp rrmpg_amd/librrhip.so /tmp/librrhip_main.so
one() { # lib sets variant mode
  python bench.py --no-cpu-baseline --no-parity-spot --steps 20 --warmup 3 \
      --sets $2 --hbv-variant $3 --mode $4 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$1 sets=$2 variant=$3 mode=$4 kernel_ms=%.3f' % d['roofline']['kernel_ms'])"
}
for rep in 1 2; do
for lib in new base; do
  if [ $lib = new ]; then cp /tmp/librrhip_main.so rrmpg_amd/librrhip.so; else cp scratch_dbg/ab/librrhip_nosel.so rrmpg_amd/librrhip.so; fi
  one $lib 1000000 0 qsim
  one $lib 1000000 0 metric
  for n in 500000 250000 125000 65536 20000; do
    for v in 0 2; do one $lib $n $v qsim; done
  done
done
done > $O/ab.txt 2>&1
cp /tmp/librrhip_main.so rrmpg_amd/librrhip.so
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
cat $O/ab.txt; tail -5 $O/pytest.log
