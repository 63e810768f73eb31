#!/usr/bin/env python3
"""The dispatches of a rocprofv3 (rocpd SQLite) kernel trace in time order, runs of the same kernel folded into one line:
start (ms from the first dispatch of the window), wall span of the run, summed kernel time, launches, stream/queue, kernel.
Usage: rocpd_timeline.py results.db [out.txt] [--from KERNEL_SUBSTRING --nth N] [--span MS]
--from/--nth: the window starts at the N-th (0-based) dispatch whose name contains the substring; --span: its length."""
import sqlite3
import sys


def main():
    args = sys.argv[1:]
    opt = {"--from": None, "--nth": "0", "--span": "1e9"}
    pos = []
    i = 0
    while i < len(args):
        if args[i] in opt: opt[args[i]] = args[i + 1]; i += 2
        else: pos.append(args[i]); i += 1
    db = sqlite3.connect(pos[0])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = cur.execute(f"select name, start, end, {qcol if qcol else '0'} from kernels order by start").fetchall()
    if opt["--from"]:
        hits = [k for k, r in enumerate(rows) if opt["--from"] in r[0]]
        rows = rows[hits[int(opt["--nth"])]:]
    t0 = rows[0][1]
    span = float(opt["--span"]) * 1e6
    rows = [r for r in rows if r[1] - t0 <= span]
    lines = [f"{'start_ms':>9} {'span_ms':>9} {'kernel_ms':>9} {'n':>5} {'queue':>6}  kernel"]
    k = 0
    while k < len(rows):
        j = k
        while j + 1 < len(rows) and rows[j + 1][0] == rows[k][0] and rows[j + 1][3] == rows[k][3]: j += 1
        name = rows[k][0]
        name = name if len(name) <= 90 else name[:87] + "..."
        lines.append(f"{(rows[k][1] - t0) / 1e6:>9.3f} {(rows[j][2] - rows[k][1]) / 1e6:>9.3f} {sum(r[2] - r[1] for r in rows[k:j + 1]) / 1e6:>9.3f} {j - k + 1:>5} {rows[k][3]:>6}  {name}")
        k = j + 1
    out = "\n".join(lines) + "\n"
    if len(pos) > 1: open(pos[1], "w").write(out)
    else: print(out)


if __name__ == "__main__":
    main()
