python -m pytest tests -m gpu -q 2>&1 | grep -v "Warn\|warn" | grep -v "^  ray " | tail -30 > gpurun_out/r4_full3.log
python bench.py --no-extras --steps 6 --warmup 2 --cpu-rays 0 > gpurun_out/r4e_bench.json 2> gpurun_out/r4e_bench.err
tail -3 gpurun_out/r4_full3.log
