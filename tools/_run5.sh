python -m pytest tests -m gpu -q -x 2>&1 | grep -v "Warn\|warn" | tail -25 > gpurun_out/r4_full2.log
NEUMESH_HIP_LIB=$PWD/tools/_build/lib_io64.so python bench.py --no-extras --steps 6 --warmup 2 --cpu-rays 0 > gpurun_out/r4c_bench_io64.json 2> gpurun_out/r4c_bench_io64.err
python bench.py --no-extras --steps 6 --warmup 2 --cpu-rays 0 > gpurun_out/r4c_bench_io256.json 2> gpurun_out/r4c_bench_io256.err
NEUMESH_HIP_LIB=$PWD/tools/_build/lib_io64.so python bench.py --no-extras --steps 6 --warmup 2 --cpu-rays 0 > gpurun_out/r4c_bench_io64b.json 2> gpurun_out/r4c_bench_io64b.err
python bench.py --no-extras --steps 6 --warmup 2 --cpu-rays 0 --rayschunk 65536 > gpurun_out/r4c_bench_chunk64k.json 2>&1
python bench.py --no-extras --steps 6 --warmup 2 --cpu-rays 0 --rayschunk 160000 > gpurun_out/r4c_bench_chunk160k.json 2>&1
tail -4 gpurun_out/r4_full2.log
