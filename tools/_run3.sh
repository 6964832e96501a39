python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "headline_scale_surface_scene_matches_reference or outside_the_fused or other_configs" 2>&1 | grep -v Warn | tail -40 > gpurun_out/r4_t3.log
for m in f16x2 f16x2s "f16x2+f16col" "f16x2s+f16col" f16x2; do
  python bench.py --mlp-precision "$m" --no-extras --steps 6 --warmup 2 --cpu-rays 0 > "gpurun_out/r4b_bench_$m.json" 2> "gpurun_out/r4b_bench_$m.err"
done
python bench.py --mlp-precision f16x2 --no-extras --steps 6 --warmup 2 --cpu-rays 0 --rayschunk 320000 > gpurun_out/r4b_bench_f16x2_2s.json 2>&1
python bench.py --mlp-precision f16x2s --no-extras --steps 6 --warmup 2 --cpu-rays 0 --rayschunk 320000 > gpurun_out/r4b_bench_f16x2s_2s.json 2>&1
tail -3 gpurun_out/r4_t3.log
