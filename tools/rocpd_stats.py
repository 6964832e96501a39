"""tools/rocpd_stats.py -- summarise a rocprofv3 (rocpd sqlite) kernel trace into the same table
`rocprofv3 --kernel-trace --stats` prints: per kernel calls / total / average / min / max / %.

    python tools/rocpd_stats.py gpurun_out/prof/r1_results.db > profiles/r01_kernel_stats.md
"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace summary of `{path}`\n")
    print("| kernel | calls | total ms | avg ms | min ms | max ms | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, c, s, a, mn, mx in rows:
        short = n if len(n) < 110 else n[:107] + "..."
        print(f"| `{short}` | {c} | {s / 1e6:.3f} | {a / 1e6:.4f} | {mn / 1e6:.4f} | {mx / 1e6:.4f} | {100 * s / total:.2f} |")
    print(f"\nkernel time total: {total / 1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
