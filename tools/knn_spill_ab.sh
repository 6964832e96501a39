#!/bin/bash
# tools/knn_spill_ab.sh <out_dir> lib1.so lib2.so ... -- GPU box (round 6): per library build, the bench frame's K-NN ms
# AND the HBM bytes the K-NN kernels write per frame (rocprofv3 --pmc WRITE_SIZE, its own pass): register-spill scratch shows up as write
# traffic above the records' 20.7 GB per frame.  "default" = the in-tree library.  Builds: -DNM_KNN_WAVES / _CHAIN / _PROBE (waves per SIMD the
# fine / chained / probe kernels are compiled for = their register budgets 512 / waves).
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" = "default" ]; then unset NEUMESH_HIP_LIB; else export NEUMESH_HIP_LIB=$PWD/$lib; fi
  timeout 300 python bench.py --no-extras --cpu-rays 0 --steps 3 --warmup 1 > $out/$name.json 2> $out/$name.err
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${name}_write -o p -- python bench.py --no-extras --cpu-rays 0 --steps 1 --warmup 0 > /dev/null 2> $out/${name}_write.err
  python - <<PY
import csv, glob, json
try:
    d = json.loads(open("$out/$name.json").read().strip().splitlines()[-1])
    sh = d["roofline"]["share_of_step_time"]; ms = d["ms_per_frame"]
    wr, n_probe = {}, 0
    for f in glob.glob("$out/${name}_write/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            key = "probe" if "nm_probe_bounds" in k else "chain" if "nm_distance_kernel<true" in k else "fine" if "nm_distance_kernel<false" in k else None
            if key and r["Counter_Name"] == "WRITE_SIZE":
                wr[key] = wr.get(key, 0.0) + float(r["Counter_Value"])
                n_probe += key == "probe"
    frames = float(max(n_probe, 1))   # one probe launch per frame (the pass renders set-up + parity + timed frames)
    print("%-10s frame %.1f ms  knn %.1f ms | K-NN kernels' WRITE_SIZE per frame (WRITE_SIZE x 1024 B): probe %.2f  chain %.2f  fine+mid %.2f  total %.2f GB (records: 20.7)" % (
        "$name", ms, d["knn_kernel"]["ms_per_frame"], wr.get("probe", 0) * 1024 / frames / 1e9, wr.get("chain", 0) * 1024 / frames / 1e9, wr.get("fine", 0) * 1024 / frames / 1e9, sum(wr.values()) * 1024 / frames / 1e9))
except Exception as e:
    print("$name FAILED", e, open("$out/$name.err").read()[-400:])
PY
done
