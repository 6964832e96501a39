#!/bin/bash
# tools/knn_pmc_study.sh <out_dir>: GPU box -- what the K-NN kernels of the bench frame wait for: three SQ counter passes (issue by unit, in-flight
# levels and instruction fetch, per-unit cycles / LDS conflicts); summarised per kernel by tools/knn_pmc_study.py
set -u
OUT=${1:-gpurun_out/r04_knn_study}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python bench.py --steps 2 --warmup 1 --no-extras --cpu-rays 0"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH --output-format csv -d $OUT/a -o p -- $BENCH > /dev/null 2> $OUT/a.err
timeout 600 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/b -o p -- $BENCH > /dev/null 2> $OUT/b.err
timeout 600 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU --output-format csv -d $OUT/c -o p -- $BENCH > /dev/null 2> $OUT/c.err
for d in a b c; do f=$(find $OUT/$d -name "*counter_collection.csv" | head -1); echo "$d: $f $(wc -l < $f 2>/dev/null)"; done
python tools/knn_pmc_study.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
