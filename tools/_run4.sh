python -m pytest tests -m gpu -q -x 2>&1 | grep -v "Warn\|warn" | tail -25 > gpurun_out/r4_full1.log
python bench.py > gpurun_out/r4_bench_full1.json 2> gpurun_out/r4_bench_full1.err
tail -4 gpurun_out/r4_full1.log
