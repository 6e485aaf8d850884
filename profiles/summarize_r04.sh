#!/bin/bash
# Condenses the raw output of profiles/run_r04_profiles.sh (gpurun_out/prof_r04*)
# into the tracked profiles/r04*_summary.json / _kernel_stats.md, then rebuilds
# profiles/traffic.json.  Kernel name patterns and algorithmic bytes per launch
# are the ones of the round-4 summaries (8 B x sets x 10,957 days for the
# qsim modes); the last argument is the launch's grid
# size: the bench's parity spot and end-to-end record launch the same kernels
# on smaller sweeps, which must not enter the averages.
cd "$(dirname "$0")/.." || exit 1
S="python profiles/summarize.py"
$S r04           'true, 1, false>('                                87656000000 262144 > /dev/null
$S r04_hbv125k   'true, 0, false>('                                10957000000 125056 > /dev/null
$S r04_hbv100k   'true, 0, false>('                                 8765600000 100032 > /dev/null
$S r04_hbv400ks  'true, 1, false>('                               175312000000 262144 > /dev/null
$S r04_hbvcat    'true, 2, false>('                                          0 262144 > /dev/null
$S r04_gr4j      'gr4j_opt_kernel<UhRegs<3>, true, false, true'    87656000000 4000000 > /dev/null
$S r04_gr4j125k  'gr4j_opt_kernel<UhRegs<3>, false, false, true'             0 125056 > /dev/null
$S r04_fused125k 'cemaneigegr4j_opt_kernel<5, UhRegs<3>, true>'              0 125056 > /dev/null
$S r04_cema      'cemaneige_kernel<5, true'                        87656000000 4000000 > /dev/null
$S r04_abc       'abc_kernel'                                      87656000000 500032 > /dev/null
python profiles/make_traffic.py r04
for f in r04_shard_sizes.txt r04_clock_power.txt r04_all_models.txt r04_tile_pieces.txt r04_bench.json; do
  [ -s gpurun_out/$f ] && cp gpurun_out/$f profiles/$f
done
