#!/bin/bash
# tools/train_trace.sh <out_dir>: GPU box -- kernel trace of a few training steps (tools/train_profile.py without the torch profiler part)
OUT=${1:-gpurun_out/train_trace}; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
NM_TRAIN_STEPS_ONLY=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python tools/train_profile.py > $OUT/run.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: find the last occurrence of nm_rays_setup_kernel
idx = max(i for i, r in enumerate(rows) if "nm_rays_setup" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
out = open("$OUT/last_step.txt", "w")
prev_end = t0
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.write("%9.1f us  +gap %7.1f  dur %8.1f  grid %-9s %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "")), r["Kernel_Name"][:90]))
    prev_end = e
last = rows[idx:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / 1e6
knn = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last if "nm_distance" in r["Kernel_Name"] or "nm_knn" in r["Kernel_Name"] or "nm_probe" in r["Kernel_Name"]) / 1e6
gemm = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last if "nm_gemm" in r["Kernel_Name"] or "Cijk" in r["Kernel_Name"]) / 1e6
span = (max(int(r["End_Timestamp"]) for r in last) - t0) / 1e6
print("last step: %d kernels, span %.2f ms, busy %.2f ms, K-NN %.2f ms, GEMM %.2f ms" % (len(last), span, busy, knn, gemm))
PY
