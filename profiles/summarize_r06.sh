#!/bin/bash
# Condenses the raw output of profiles/run_r06_profiles.sh (gpurun_out/prof_r06*)
# into the tracked profiles/r06*_summary.json / _kernel_stats.md, then rebuilds
# profiles/traffic.json (stamped with the kernel sources' ids).  Kernel name
# patterns, algorithmic bytes per launch (8 B x sets x 10,957 days for the
# qsim modes) and the launch's grid size: the bench's
# parity spot launches the same kernels on smaller sweeps, which must not
# enter the averages.
cd "$(dirname "$0")/.." || exit 1
S="python profiles/summarize.py"
$S r06           'true, 1, false>('                                87656000000 262144 > /dev/null
$S r06_hbv125k   'true, 0, false>('                                10957000000 125056 > /dev/null
$S r06_hbv100k   'true, 0, false>('                                 8765600000 100032 > /dev/null
$S r06_hbv400ks  'true, 1, false>('                               175312000000 190000 > /dev/null
$S r06_hbvcat    'true, 2, false>('                                          0 262144 > /dev/null
$S r06_gr4j      'gr4j_opt_kernel<UhRegs<3>, true, false, true'    87656000000 4000000 > /dev/null
$S r06_gr4j125k  'gr4j_opt_kernel<UhRegs<3>, false, false, true'             0 125056 > /dev/null
$S r06_fused125k 'cemaneigegr4j_opt_kernel<5, UhRegs<3>'                     0 125056 > /dev/null
$S r06_cema      'cemaneige_kernel<5, true'                        87656000000 4000000 > /dev/null
$S r06_abc       'abc_kernel'                                      87656000000 500032 > /dev/null
$S r06_hyst      'snow_gr4j_kernel<5, UhRegs<10>, true, false>'              0 500000 > /dev/null
$S r06_ice       'snow_gr4j_kernel<5, UhRegs<3>, false, true>'               0 500000 > /dev/null
$S r06_hystice   'snow_gr4j_kernel<5, UhRegs<10>, true, true>'               0 500000 > /dev/null
# (the couplings' sorted sweeps run the 3-, 5- and 10-register tiers side by
# side: a summary per tier kernel, the instruction count summed over them)
$S r06_hyst_t3   'snow_gr4j_kernel<5, UhRegs<3>, true, false>'               0 500000 r06_hyst > /dev/null
$S r06_hyst_t5   'snow_gr4j_kernel<5, UhRegs<5>, true, false>'               0 500000 r06_hyst > /dev/null
$S r06_hystice_t3 'snow_gr4j_kernel<5, UhRegs<3>, true, true>'               0 500000 r06_hystice > /dev/null
$S r06_hystice_t5 'snow_gr4j_kernel<5, UhRegs<5>, true, true>'               0 500000 r06_hystice > /dev/null
python profiles/make_traffic.py r06
for f in r06_shard_sizes.txt r06_clock_power.txt r06_all_models.txt r06_bench.json r06_bench_detail.json; do
  [ -s gpurun_out/$f ] && cp gpurun_out/$f profiles/$f
done
