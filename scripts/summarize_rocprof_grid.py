#!/usr/bin/env python
"""Per (kernel, grid size) table from a rocprofv3 --kernel-trace CSV directory: the same kernel on different layers.
usage: summarize_rocprof_grid.py <dir with *_kernel_trace.csv> <steps profiled> <name filter> [title]"""
import collections
import csv
import glob
import os
import sys


def main():
    d, steps, flt = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    title = sys.argv[4] if len(sys.argv) > 4 else flt
    f = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("dpa::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if flt not in name:
            continue
        if "Grid_Size" in r:
            wg = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
        else:
            wg = 1
            for ax in "XYZ":
                wg *= int(r["Grid_Size_" + ax]) // max(int(r["Workgroup_Size_" + ax]), 1)
        acc[(name[:100], wg)].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("# %s\n" % title)
    print("| kernel | workgroups | calls/step | median us | ms/step |")
    print("|---|---|---|---|---|")
    for (name, wg), ts in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        ts.sort()
        print("| `%s` | %d | %.1f | %.1f | %.3f |" % (name, wg, len(ts) / steps, ts[len(ts) // 2], sum(ts) / 1e3 / steps))


if __name__ == "__main__":
    main()
