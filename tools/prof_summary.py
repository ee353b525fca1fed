"""Summarise a rocprofv3 kernel_stats.csv by category (development aid).  usage: prof_summary.py stats.csv steps [top]"""
import collections
import csv
import re
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40


from prof_summary_cat import cat  # noqa: E402

d = collections.defaultdict(lambda: [0.0, 0])
tot = cnt = 0
for r in rows:
    ns = float(r['TotalDurationNs']); k = cat(r['Name'])
    d[k][0] += ns; d[k][1] += int(r['Calls']); tot += ns; cnt += int(r['Calls'])
print("# note: bench.py's live roofline measurements (35 attention + 5 x 23 GEMM-class launches per run) are inside this trace")
print(f"{'category':26s} {'ms/step':>8s} {'launches/step':>14s}")
for k, v in sorted(d.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:26s} {v[0] / steps / 1e6:8.3f} {v[1] / steps:14.1f}")
print(f"{'TOTAL':26s} {tot / steps / 1e6:8.3f} {cnt / steps:14.1f}")
print()
for r in rows[:top]:
    n = re.sub(r'at::native::|\(anonymous namespace\)::|rocprim::ROCPRIM_400001_NS::detail::', '', r['Name'])
    print(f"{float(r['TotalDurationNs']) / steps / 1e6:7.3f} ms {int(r['Calls']) / steps:6.1f}x {float(r['AverageNs']) / 1e3:8.1f} us  {n[:150]}")
