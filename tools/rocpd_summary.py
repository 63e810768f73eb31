#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 rocpd SQLite) kernel trace into the per-kernel stats table committed under
profiles/ (name, calls, total/avg/min/max ms, % of GPU kernel time).  Usage: rocpd_summary.py results.db [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = [f"{'kernel':<72} {'calls':>6} {'total_ms':>11} {'avg_ms':>10} {'min_ms':>10} {'max_ms':>10} {'pct':>6}"]
    for n, c, s, a, mn, mx in rows:
        n = n if len(n) <= 72 else n[:69] + "..."
        lines.append(f"{n:<72} {c:>6} {s / 1e6:>11.3f} {a / 1e6:>10.4f} {mn / 1e6:>10.4f} {mx / 1e6:>10.4f} {100.0 * s / total:>6.2f}")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
