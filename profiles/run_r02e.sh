#!/bin/bash
# A/B on one box: GR4J-family kernels, new (fast division, third-order root,
# integer range votes, no array moves) vs base (round-1 step function).
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r02e
mkdir -p $O
cp rrmpg_amd/librrhip.so /tmp/librrhip_main.so
one() { # lib model mode sets
  python bench.py --no-cpu-baseline --no-parity-spot --steps 5 --warmup 2 \
      --model $2 --mode $3 --sets $4 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$1 model=$2 mode=$3 sets=$4 kernel_ms=%.3f' % d['roofline']['kernel_ms'])"
}
for lib in new base new base; do
  if [ $lib = new ]; then cp /tmp/librrhip_main.so rrmpg_amd/librrhip.so; else cp scratch_dbg/ab/librrhip_nosel.so rrmpg_amd/librrhip.so; fi
  one $lib gr4j qsim 1000000
  one $lib gr4j metric 1000000
  one $lib cemaneigegr4j metric 1000000
  one $lib cemaneigegr4j qsim 125000
  one $lib cemaneigehystgr4j metric 1000000
  one $lib cemaneigegr4jice metric 1000000
  one $lib cemaneige qsim 1000000
done > $O/ab.txt 2>&1
cp /tmp/librrhip_main.so rrmpg_amd/librrhip.so
python bench.py --model gr4j --no-cpu-baseline --steps 3 > $O/bench_gr4j.json 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
cat $O/ab.txt; tail -15 $O/pytest.log; cat $O/bench_gr4j.json
