#!/bin/bash
# tools/profile_round.sh <out_dir>: every rocprofv3 pass behind profiles/rNN_* (run on the GPU box through gpurun).
# Counters are collected in their own runs (no trace domains next to --pmc).
set -u
OUT=${1:-gpurun_out/r06_prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python bench.py --steps 2 --warmup 1 --no-extras --cpu-rays 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $BENCH > /dev/null 2> $OUT/fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $BENCH > /dev/null 2> $OUT/write.err
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o p -- $BENCH > /dev/null 2> $OUT/mfma.err
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/sq -o p -- $BENCH > /dev/null 2> $OUT/sq.err
S5="python bench.py --workload stress5 --steps 1 --warmup 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s5trace -o t -- $S5 > $OUT/stress5.json 2> $OUT/s5trace.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/s5fetch -o p -- $S5 > /dev/null 2> $OUT/s5fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/s5write -o p -- $S5 > /dev/null 2> $OUT/s5write.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cfetch -o p -- python tools/pmc_calib.py > /dev/null 2> $OUT/cfetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cwrite -o p -- python tools/pmc_calib.py > /dev/null 2> $OUT/cwrite.err
find $OUT -name "*counter_collection.csv" -size +20M -exec sh -c 'echo "large: $1"; ls -la $1' _ {} \;
for d in fetch write mfma sq s5fetch s5write cfetch cwrite; do f=$(find $OUT/$d -name "*counter_collection.csv" | head -1); echo "$d: $f $(wc -l < $f 2>/dev/null)"; done
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err   # (every step under its own timeout: a stalled run once burnt 40 GPU-minutes here)
echo done
