#!/bin/bash
# Round 4 evidence: rocprofv3 kernel trace + stats, FETCH/WRITE traffic and SQ
# counters (separate passes, profiles/collect.sh) for the headline workload
# and for every extra configuration of the bench line, the shard-size table,
# clock / power, every model and mode, and the bench line of the same build.
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
run() { tag=$1; shift; bash profiles/collect.sh $tag "$@" > /dev/null 2>&1;
        find gpurun_out/prof_$tag -name "*.db" -delete 2>/dev/null; }
run r04
run r04_hbv125k --sets 125000
run r04_hbv100k --sets 100000
run r04_hbv400ks --sets 400000 --mode storages
run r04_hbvcat --sets 10000 --catchments 125 --mode metric
run r04_gr4j --model gr4j
run r04_gr4j125k --model gr4j --mode metric --sets 125000
run r04_fused125k --model cemaneigegr4j --mode metric --sets 125000 --score nse
run r04_cema --model cemaneige
run r04_abc --model abc
bash profiles/shard_sizes.sh > gpurun_out/r04_shard_sizes.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
bash profiles/clock_power.sh > gpurun_out/r04_clock_power.txt 2>&1
bash profiles/all_models.sh > gpurun_out/r04_all_models.txt 2>&1
for p in 0 3 4 6 8; do
  python bench.py --no-cpu-baseline --no-parity-spot --no-extra-configs --no-power-soak --steps 30 --warmup 3 --time-tiles $p 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('hbvedu 1M qsim time-tiles=$p kernel_ms=%.3f' % d['roofline']['kernel_ms'])"
done > gpurun_out/r04_tile_pieces.txt 2>&1
