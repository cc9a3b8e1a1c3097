#!/usr/bin/env python3
"""One steady-state step from a rocprofv3 kernel trace: per-kernel totals, queue busy time, idle gaps, long kernels.
usage: timeline.py <r_kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'lstm2_coop_fwd_kernel<unsigned short, 2, 5' in r['Kernel_Name']]
step = rows[idx[-2]:idx[-1]]
t0 = int(step[0]['Start_Timestamp'])
print("kernels in step", len(step), "step wall ms", (int(rows[idx[-1]]['Start_Timestamp']) - t0) / 1e6)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    n = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:46]
    agg[n][0] += 1
    agg[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{n:48s} {c:4d} {t:7.3f} ms")
qs = collections.defaultdict(list)
for r in step:
    qs[r['Queue_Id']].append(r)
for q, lst in qs.items():
    gap = sum(max(0, int(b['Start_Timestamp']) - int(a['End_Timestamp'])) for a, b in zip(lst, lst[1:]))
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in lst)
    print(f"queue {q}: {len(lst)} kernels, busy {busy/1e6:.3f} ms, idle between kernels {gap/1e6:.3f} ms")
main = max(qs.values(), key=len)
big = []
for a, b in zip(main, main[1:]):
    g = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
    if g > 15000:
        big.append((g / 1e3, (int(a['End_Timestamp']) - t0) / 1e6, a['Kernel_Name'][:40], b['Kernel_Name'][:40]))
for b in sorted(big, reverse=True)[:10]:
    print("gap us %.1f at %.3f ms: %s -> %s" % b)
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if e - s > 250e3:
        print(f"{(s-t0)/1e6:8.3f} -> {(e-t0)/1e6:8.3f}  q{r['Queue_Id']}  {r['Kernel_Name'][:70]}")
