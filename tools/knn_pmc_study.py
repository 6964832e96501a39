"""tools/knn_pmc_study.py <dir of tools/knn_pmc_study.sh>: per-kernel sums of the SQ counters of the three passes (K-NN kernels of the bench frame)."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
for d in "abc":
    fs = glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        key = "probe" if "probe_bounds" in k else "chain" if "distance_kernel<true>" in k else "plain" if "distance_kernel<false>" in k else \
              "mlp_fwd" if "geo_mlp_h2_kernel<false" in k else None
        if key:
            tot[key][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in tot.items():
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1.0
    print(f"== {k}")
    for n in sorted(c):
        print(f"   {n:28s} {c[n]:.4e}   per wave-cycle {c[n] / wc:.4f}")
