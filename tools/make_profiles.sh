#!/bin/bash
# tools/make_profiles.sh <dir written by tools/profile_round.sh> <tag>: the summaries committed under profiles/
set -eu
[ -s "${1:-gpurun_out/r06_prof}/bench_default.json" ] || { echo "no bench_default.json in ${1:-gpurun_out/r06_prof}: profiles/ left untouched"; exit 1; }
P=${1:-gpurun_out/r06_prof}; T=${2:-r06}
f() { find $P/$1 -name "*counter_collection.csv" | head -1; }
python tools/pmc_report.py calib $(f cfetch) $(f cwrite) > profiles/${T}_pmc_calibration.json
python tools/pmc_report.py traffic $(f fetch) $(f write) profiles/${T}_pmc_calibration.json > profiles/${T}_pmc_traffic.json
python tools/pmc_report.py mfma $(f mfma) > profiles/${T}_pmc_mfma.json
python tools/pmc_report.py knn $(f sq) > profiles/${T}_pmc_knn.json
python tools/pmc_report.py traffic $(f s5fetch) $(f s5write) profiles/${T}_pmc_calibration.json > profiles/${T}_pmc_traffic_stress5.json
python tools/stats_md.py $P/trace/t_kernel_stats.csv "python bench.py --steps 2 --warmup 1 --no-extras --cpu-rays 0" > profiles/${T}_bench_kernel_stats.md
cp $P/trace/t_kernel_stats.csv profiles/${T}_bench_kernel_stats.csv
python tools/stats_md.py $P/s5trace/t_kernel_stats.csv "python bench.py --workload stress5 --steps 1 --warmup 1" > profiles/${T}_stress5_kernel_stats.md
cp $P/bench_default.json profiles/${T}_bench.json
cp $P/stress5.json profiles/${T}_stress5.json
