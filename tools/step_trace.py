"""One training step out of a rocprofv3 kernel TRACE (development aid): one period of the step -- the launches between two
consecutive launches of a kernel that runs once per step (nms_segsort_kernel, in the forward pass; with AdamW applied per gradient
bucket there are several adamw_kernel launches per step, so that kernel no longer delimits it; traces without an RPN fall back to
it) -- by category -- exact per-step counts and times (the kernel_stats.csv summary also holds bench.py's warm-up and its
live roofline measurements).  usage: step_trace.py kernel_trace.csv [which_step_from_the_end=2] [top=40]"""
import collections
import csv
import re
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary_cat import cat  # noqa: E402

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
idx = [i for i, r in enumerate(rows) if 'nms_segsort_kernel' in r['Kernel_Name']]
if len(idx) > back:
    a, b = idx[-back - 1], idx[-back]
    step = rows[a:b]                     # one period, starting in the forward pass
    t0, t1 = int(step[0]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
    what = "one period of the step (nms_segsort_kernel to nms_segsort_kernel)"
else:
    idx = [i for i, r in enumerate(rows) if 'adamw_kernel' in r['Kernel_Name']]
    a, b = idx[-back - 1], idx[-back]
    step = rows[a + 1:b + 1]
    t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])
    what = "from the first launch to the end of adamw_kernel"
streams = collections.Counter(r['Stream_Id'] if 'Stream_Id' in r else r['Queue_Id'] for r in step)
main = streams.most_common(1)[0][0]
d = collections.defaultdict(lambda: [0.0, 0, 0.0, 0])
busy = collections.defaultdict(float)
for r in step:
    ns = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    k = cat(r['Kernel_Name'])
    sid = r['Stream_Id'] if 'Stream_Id' in r else r['Queue_Id']
    d[k][0] += ns; d[k][1] += 1
    if sid != main:
        d[k][2] += ns; d[k][3] += 1
    busy[sid] += ns
# time covered by at least one kernel (union of intervals)
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
cov, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        cov += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
cov += cur_e - cur_s
print(f"one step: {len(step)} launches, {(t1 - t0) / 1e6:.3f} ms {what} "
      f"(under the profiler); some kernel running for {cov / 1e6:.3f} ms of it")
print("kernel time per stream (ms): " + ", ".join(f"{k}: {v / 1e6:.3f} ({streams[k]} launches)" for k, v in sorted(busy.items(), key=lambda kv: -kv[1])))
print(f"{'category':26s} {'ms':>8s} {'launches':>9s} {'of which on the 2nd stream (ms, launches)':>44s}")
tot = [0.0, 0]
for k, v in sorted(d.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:26s} {v[0] / 1e6:8.3f} {v[1]:9d} {v[2] / 1e6:30.3f} {v[3]:9d}")
    tot[0] += v[0]; tot[1] += v[1]
print(f"{'TOTAL':26s} {tot[0] / 1e6:8.3f} {tot[1]:9d}")
print()
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    n = re.sub(r'at::native::|\(anonymous namespace\)::|rocprim::ROCPRIM_400001_NS::detail::', '', r['Kernel_Name'])[:150]
    agg[n][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); agg[n][1] += 1
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{v[0] / 1e6:7.3f} ms {v[1]:5d}x {v[0] / v[1] / 1e3:8.1f} us  {n}")
