#!/bin/bash
# Round 2 evidence: rocprofv3 kernel trace + stats, FETCH/WRITE traffic and SQ
# counters (separate passes, profiles/collect.sh) for the headline workload
# and the other kernels, plus the bench lines of the same build.
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
bash profiles/collect.sh r02 > /dev/null 2>&1
bash profiles/collect.sh r02_hbv125k --sets 125000 > /dev/null 2>&1
bash profiles/collect.sh r02_gr4j --model gr4j > /dev/null 2>&1
bash profiles/collect.sh r02_fused --model cemaneigegr4j --mode metric > /dev/null 2>&1
bash profiles/collect.sh r02_cema --model cemaneige > /dev/null 2>&1
bash profiles/collect.sh r02_abc --model abc > /dev/null 2>&1
cd $ROOT
mkdir -p gpurun_out/r02_bench
python bench.py > gpurun_out/r02_bench/hbv.json 2> gpurun_out/r02_bench/hbv.err
for spec in "hbv_metric:--model hbvedu --mode metric" "hbv_125k:--sets 125000" "hbv_250k:--sets 250000" "hbv_500k:--sets 500000" \
            "gr4j:--model gr4j" "gr4j_metric:--model gr4j --mode metric" "fused_metric:--model cemaneigegr4j --mode metric" \
            "fused_125k:--model cemaneigegr4j --sets 125000" "fused_125k_metric:--model cemaneigegr4j --sets 125000 --mode metric" \
            "fused_125k_metric_manywaves:--model cemaneigegr4j --sets 125000 --mode metric --fused-variant 1" \
            "gr4j_125k:--model gr4j --sets 125000" "gr4j_125k_metric:--model gr4j --sets 125000 --mode metric" \
            "cema:--model cemaneige" "abc:--model abc" \
            "hyst_metric:--model cemaneigehystgr4j --mode metric" "ice_metric:--model cemaneigegr4jice --mode metric" \
            "hystice_metric:--model cemaneigehystgr4jice --mode metric" "hbv_all:--model hbvedu --mode storages --sets 400000" \
            "catch:--catchments 125 --sets 10000 --mode metric"; do
  tag=${spec%%:*}; args=${spec#*:}
  python bench.py --no-cpu-baseline --steps 20 --warmup 3 $args > gpurun_out/r02_bench/$tag.json 2>/dev/null
done
python bench.py --gpus 2 --backend gloo --share-gpu --steps 10 --warmup 2 --no-parity-spot > gpurun_out/r02_bench/two_ranks_gloo.json 2>/dev/null
for f in gpurun_out/r02_bench/*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-22s kernel_ms=%8.3f ms_per_step=%8.3f value=%.3e frac=%.3f parity=%s" % (sys.argv[1].split('/')[-1][:-5], d['roofline']['kernel_ms'], d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('parity_spot')))
PY
done
