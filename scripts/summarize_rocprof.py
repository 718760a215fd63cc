#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory into a short markdown table.
usage: summarize_rocprof.py <dir with *_kernel_stats.csv> <steps profiled> [title] [rows, default 30] > profiles/x.md"""
import csv
import glob
import os
import sys


def main():
    d, steps = sys.argv[1], int(sys.argv[2])
    title = sys.argv[3] if len(sys.argv) > 3 else d
    f = sorted(glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("# %s\n" % title)
    print("source: `%s`, %d training steps profiled; total kernel time %.2f ms = %.2f ms/step\n" % (
        os.path.basename(f), steps, tot / 1e6, tot / 1e6 / steps))
    print("| kernel | calls/step | avg us | ms/step | % |")
    print("|---|---|---|---|---|")
    top = int(sys.argv[4]) if len(sys.argv) > 4 else 30
    for r in rows[:top]:
        name = r["Name"].replace("dpa::(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0][:90]
        t = float(r["TotalDurationNs"])
        print("| `%s` | %.1f | %.1f | %.3f | %.1f |" % (name, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                                    t / 1e6 / steps, 100 * t / tot))
    rest = sum(float(r["TotalDurationNs"]) for r in rows[top:])
    if rest:
        print("| (%d more kernels) | | | %.3f | %.1f |" % (len(rows) - top, rest / 1e6 / steps, 100 * rest / tot))


if __name__ == "__main__":
    main()
