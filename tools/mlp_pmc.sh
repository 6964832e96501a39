#!/bin/bash
# tools/mlp_pmc.sh <out_dir> [lib.so]: rocprofv3 --pmc passes over the MLP kernels alone (tools/mlp_ab_quick.py child: 2^20 points per
# launch); prints per-kernel counter sums.  Counters in their own runs (no trace domains next to --pmc).
set -u
OUT=${1:-gpurun_out/mlp_pmc}
LIB=${2:-}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export NM_QUICK_CHILD=1 NM_QUICK_NAME=pmc
[ -n "$LIB" ] && export NEUMESH_HIP_LIB=$PWD/$LIB
CMD="python tools/mlp_ab_quick.py"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed: $(tail -3 $OUT/p$i.log)"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mlp" not in k: continue
        k = k.split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()): print("   %-28s %.4g  (%d launches)" % (c, x, n[(k, c)]))
PY
