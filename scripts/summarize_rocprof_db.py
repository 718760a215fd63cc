#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace output directory (ROCm 7: a sqlite .db with a `kernels` view) into a markdown
table.  usage: summarize_rocprof_db.py <dir> <steps profiled> [title] > profiles/x.md"""
import glob
import os
import sqlite3
import sys


def main():
    d, steps = sys.argv[1], int(sys.argv[2])
    title = sys.argv[3] if len(sys.argv) > 3 else d
    dbs = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
    if not dbs:
        raise SystemExit("no .db under %s" % d)
    con = sqlite3.connect(dbs[0])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = con.execute("select %s, count(*), sum(end - start) from kernels group by %s order by 3 desc" % (name, name)
                       ).fetchall()
    tot = sum(r[2] for r in rows)
    print("# %s\n" % title)
    print("source: `%s`, %d training steps profiled; total kernel time %.2f ms/step\n" % (
        os.path.basename(dbs[0]), steps, tot / 1e6 / steps))
    print("| kernel | calls/step | avg us | ms/step | % |")
    print("|---|---|---|---|---|")
    for n, c, t in rows[:int(os.environ.get("ROWS", "28"))]:
        n = n.replace("dpa::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:100]
        print("| `%s` | %.1f | %.1f | %.3f | %.1f |" % (n, c / steps, t / c / 1e3, t / 1e6 / steps, 100.0 * t / tot))


if __name__ == "__main__":
    main()
