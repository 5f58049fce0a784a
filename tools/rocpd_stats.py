#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel count / avg / min / max / total, like `--stats`.
Usage: tools/rocpd_stats.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = cur.execute(
        "select name, grid_x, workgroup_x, vgpr_count, lds_size, count(*), avg(end-start), min(end-start), "
        "max(end-start), sum(end-start) from kernels group by name, grid_x order by sum(end-start) desc"
    ).fetchall()
    total = sum(r[-1] for r in rows) or 1
    lines = ["| kernel | grid | wg | vgpr | lds | calls | avg_us | min_us | max_us | total_ms | % |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    for name, gx, wx, vg, lds, n, avg, mn, mx, tot in rows:
        lines.append(f"| `{name[:110]}` | {gx} | {wx} | {vg} | {lds} | {n} | {avg/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {tot/1e6:.3f} | {100*tot/total:.1f} |")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
