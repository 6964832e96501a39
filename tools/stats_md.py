"""tools/stats_md.py -- render a `rocprofv3 --kernel-trace --stats --output-format csv` kernel_stats.csv
as the markdown table kept under profiles/.

    python tools/stats_md.py gpurun_out/r01_trace/t_kernel_stats.csv "bench.py --steps 2 --warmup 1" > profiles/r01_bench_kernel_stats.md
"""
import csv
import sys


def main(path, cmd):
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats of `{cmd}` (one MI355X)\n")
    print("| kernel | calls | total ms | avg ms | min ms | max ms | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for r in rows:
        n = r["Name"]
        n = n if len(n) < 100 else n[:97] + "..."
        print(f"| `{n}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e6:.4f} | "
              f"{float(r['MinNs']) / 1e6:.4f} | {float(r['MaxNs']) / 1e6:.4f} | {float(r['Percentage']):.2f} |")
    print(f"\nkernel time total: {total / 1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "bench.py")
