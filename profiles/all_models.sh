#!/bin/bash
# Every model / mode through bench.py (kernel ms from HIP events, whole-step
# ms, rate, HBM fraction, parity spot) -> one line each.
cd "$(dirname "$0")/.." || exit 1
run() {
  tag=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --no-power-soak --live-counters none --steps ${STEPS:-10} --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s kernel_ms=%8.3f ms_per_step=%8.3f value=%.3e frac=%.3f parity=%s' % ('$tag', d['roofline']['kernel_ms'], d['ms_per_step'], d['value'], d['roofline']['frac'] or 0, d.get('parity_spot')))"
}
run hbv --model hbvedu
run hbv_metric --model hbvedu --mode metric
run hbv_all --model hbvedu --mode storages --sets 400000
run hbv_500k --model hbvedu --sets 500000
run hbv_250k --model hbvedu --sets 250000
run hbv_125k --model hbvedu --sets 125000
run catch --model hbvedu --catchments 125 --sets 10000 --mode metric
run abc --model abc
run gr4j --model gr4j
run gr4j_metric --model gr4j --mode metric
run gr4j_125k_metric --model gr4j --mode metric --sets 125000
run cema --model cemaneige
run cema_metric --model cemaneige --mode metric
run fused_metric --model cemaneigegr4j --mode metric
run fused_125k_metric --model cemaneigegr4j --mode metric --sets 125000
run hyst_metric --model cemaneigehystgr4j --mode metric
run ice_metric --model cemaneigegr4jice --mode metric
run hystice_metric --model cemaneigehystgr4jice --mode metric
