#!/bin/bash
# tools/knn_variants.sh -- GPU box: the bench frame (K-NN ms per frame, frame ms) for several builds of the library.
# usage: tools/knn_variants.sh out_dir lib1.so lib2.so ...   ("default" = the in-tree library)
out=$1; shift
mkdir -p $out
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" = "default" ]; then unset NEUMESH_HIP_LIB; else export NEUMESH_HIP_LIB=$PWD/$lib; fi
  python bench.py --no-extras --cpu-rays 0 --steps 3 --warmup 1 ${BENCH_ARGS} > $out/$name.json 2> $out/$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$name.json").read().strip().splitlines()[-1])
    sh=d["roofline"]["share_of_step_time"]; ms=d["ms_per_frame"]
    print("%-14s frame %.1f ms  knn %.1f  geo %.1f  tangent %.1f  colour %.1f  rest %.1f" % ("$name", ms, d["knn_kernel"]["ms_per_frame"], sh["geo_mlp"]*ms, sh["geo_mlp_tangent"]*ms, sh["color_mlp"]*ms, ms*(1-sum(sh.values()))))
except Exception as e:
    print("$name FAILED", e, open("$out/$name.err").read()[-400:])
PY
done
