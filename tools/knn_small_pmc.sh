#!/bin/bash
# tools/knn_small_pmc.sh: GPU box -- counters of the small K-NN launches of tools/knn_small.py (why does a 512-query launch take 0.4 ms?)
OUT=gpurun_out/ks_pmc; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SMEM SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p -- python tools/knn_small.py > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
res = collections.OrderedDict()
for d in sorted(glob.glob("$OUT/p*/")):
    fs = glob.glob(d + "**/p_counter_collection.csv", recursive=True)
    if not fs: continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "nm_distance" in r["Kernel_Name"]]
    byc = collections.defaultdict(list)
    for r in rows:
        byc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, v in byc.items():
        v.sort()
        res[c] = [x for _, x in v]
names = list(res)
n = min(len(v) for v in res.values())
print("launch " + " ".join("%14s" % c[-14:] for c in names))
for i in range(72, n, 6):      # the second block of tools/knn_small.py starts after 12 groups of 6 launches
    print("%6d " % i + " ".join("%14.0f" % res[c][i + 1] for c in names))
PY
