#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats output, *.db) per kernel AND launch geometry, so that
launches of one kernel over different problem sizes are not averaged together.  Usage:
    python tools/rocpd_summary.py gpurun_out/prof/r01_results.db > profiles/r01_kernel_stats.md
"""
import sqlite3
import sys


def short(name, n=110):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = cur.execute(
        "select name, grid_x, grid_y, workgroup_x, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name, grid_x, grid_y, workgroup_x "
        "order by sum(duration) desc").fetchall()
    total = sum(r[5] for r in rows) or 1
    print("| kernel | grid (threads) | block | calls | total us | avg us | min us | max us | % | vgpr | sgpr | lds |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, gx, gy, wx, n, tot, avg, mn, mx, vg, sg, lds in rows[:25]:
        print(f"| `{short(name)}` | {gx}x{gy} | {wx} | {n} | {tot / 1e3:.1f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | "
              f"{mx / 1e3:.2f} | {100 * tot / total:.1f} | {vg} | {sg} | {lds} |")


if __name__ == "__main__":
    main(sys.argv[1])
