#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) from a rocprofv3 rocpd SQLite result file.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [top_n] [--csv out.csv]
"""
import re
import sqlite3
import sys


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 30
    out_csv = sys.argv[sys.argv.index("--csv") + 1] if "--csv" in sys.argv else None
    db = sqlite3.connect(path)
    names = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in names if t.startswith("rocpd_kernel_dispatch_"))
    ks = next(t for t in names if t.startswith("rocpd_info_kernel_symbol_"))
    rows = db.execute(f"select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
                      f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for name, calls, tot, mn, mx in rows:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        lines.append(f"\"{short}\",{calls},{tot},{tot / calls:.1f},{mn},{mx},{100.0 * tot / total:.2f}")
    if out_csv:
        open(out_csv, "w").write("\n".join(lines) + "\n")
    for ln in lines[:top + 1]:
        print(ln[:200])
    print(f"total kernel time {total / 1e6:.2f} ms")


if __name__ == "__main__":
    main()
