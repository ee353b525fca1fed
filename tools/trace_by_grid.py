#!/usr/bin/env python
"""Average kernel durations of a rocprofv3 --kernel-trace CSV grouped by (kernel name, grid, workgroup): tells apart the launches of
one kernel at different problem sizes, which --stats merges.  usage: python tools/trace_by_grid.py <kernel_trace.csv> [name filter]"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if flt and flt not in name:
                continue
            grid = (r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))
            wg = r.get("Workgroup_Size_X")
            acc[(name[:90], grid, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for (name, grid, wg), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        print(f"{len(v):5d} x  median {v[len(v) // 2]:8.1f} us  min {v[0]:8.1f}  grid {'x'.join(g for g in grid if g)} wg {wg}  {name}")


if __name__ == "__main__":
    main()
