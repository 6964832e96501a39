set -x
python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "headline_scale_surface or other_configs or trainer_step" 2>&1 | tail -60 > gpurun_out/r4_t2.log
python tools/mlp_ab.py f16x2 f16x2s "f16x2+f16col" > gpurun_out/r4_mlp_ab.log 2>&1
for m in f16x2 f16x2s "f16x2+f16col" "f16x2s+f16col"; do
  python bench.py --mlp-precision "$m" --no-extras --steps 5 --warmup 2 --cpu-rays 0 > "gpurun_out/r4_bench_$m.json" 2> "gpurun_out/r4_bench_$m.err"
done
python bench.py --mlp-precision f16x2 --no-extras --steps 5 --warmup 2 --cpu-rays 0 --rayschunk 320000 > gpurun_out/r4_bench_f16x2_2s.json 2>&1
python bench.py --mlp-precision f16x2s --no-extras --steps 5 --warmup 2 --cpu-rays 0 --rayschunk 320000 > gpurun_out/r4_bench_f16x2s_2s.json 2>&1
tail -3 gpurun_out/r4_t2.log; cat gpurun_out/r4_mlp_ab.log | tail -20
