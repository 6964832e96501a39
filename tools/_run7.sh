python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -m gpu -q -x -k "knn or surface_scene_matches_reference_fixture or fused_sampler or render_frame_properties or config5" 2>&1 | grep -v "Warn\|warn" | grep -v "^  ray " | tail -30 > gpurun_out/r4_t7.log
for i in 1 2; do
python bench.py --no-extras --steps 6 --warmup 2 --cpu-rays 0 > gpurun_out/r4d_bench_pf_$i.json 2> gpurun_out/r4d_pf.err
NEUMESH_HIP_LIB=$PWD/tools/_build/lib_nopf.so python bench.py --no-extras --steps 6 --warmup 2 --cpu-rays 0 > gpurun_out/r4d_bench_nopf_$i.json 2> gpurun_out/r4d_nopf.err
done
python tools/train_profile.py 2>&1 | grep "ms per step" > gpurun_out/r4d_train_pf.log
NEUMESH_HIP_LIB=$PWD/tools/_build/lib_nopf.so python tools/train_profile.py 2>&1 | grep "ms per step" > gpurun_out/r4d_train_nopf.log
tail -3 gpurun_out/r4_t7.log
