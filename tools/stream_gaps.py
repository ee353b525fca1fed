"""Idle gaps of the main stream in one eager (two-stream) step of a rocprofv3 kernel trace, with what the other stream ran meanwhile
(development aid).  usage: python tools/stream_gaps.py <kernel_trace.csv> [min_gap_us]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 15000
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'nms_segsort_kernel' in r['Kernel_Name']]     # once per step (forward); AdamW may run per bucket
if len(idx) >= 3:
    step = rows[idx[-3]: idx[-2]]
else:
    idx = [i for i, r in enumerate(rows) if 'adamw_kernel' in r['Kernel_Name']]
    step = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
streams = collections.Counter(r['Stream_Id'] for r in step)
main = max(streams, key=streams.get)
ms = [r for r in step if r['Stream_Id'] == main]
ss = [r for r in step if r['Stream_Id'] != main]
prev_end, prev_name, gaps, allg = None, '', [], 0
for r in ms:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if prev_end is not None and s > prev_end:
        allg += s - prev_end
        if s - prev_end > thr:
            ov = [x['Kernel_Name'][:34] for x in ss if int(x['Start_Timestamp']) < s and int(x['End_Timestamp']) > prev_end]
            gaps.append(((prev_end - t0) / 1e3, (s - prev_end) / 1e3, prev_name[:44], r['Kernel_Name'][:44], ov[:4]))
    prev_end, prev_name = max(prev_end or 0, e), r['Kernel_Name']
print(f"streams {dict(streams)}; step span {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.0f} us; main-stream idle {allg / 1e3:.0f} us, "
      f"of which in {len(gaps)} gaps > {thr / 1e3:.0f} us: {sum(g[1] for g in gaps):.0f} us")
for g in sorted(gaps, key=lambda g: -g[1])[:20]:
    print('  at %6.0f us: %6.1f us  after %-44s before %-44s | other stream: %s' % g)
