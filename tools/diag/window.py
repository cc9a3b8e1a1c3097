#!/usr/bin/env python3
"""Kernel-by-kernel listing of the stretches of one steady-state step that lie BETWEEN the three LSTM launches (the fronts of
the two nets, the loss, the start of the tail).  usage: window.py <r_kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'lstm2_coop_fwd_kernel<unsigned short, 2, 5' in r['Kernel_Name']]
step = rows[idx[-2]:idx[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
name = lambda r: r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:58]
big = [i for i, r in enumerate(step) if 'lstm2_coop' in r['Kernel_Name']]
def dump(a, b, title):
    print(f"--- {title}: {(int(step[b]['Start_Timestamp']) - int(step[a]['End_Timestamp'])) / 1e3:.1f} us wall")
    prev = int(step[a]['End_Timestamp'])
    for r in step[a + 1:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print(f"  q{r['Queue_Id']} +{(s - prev) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  {name(r)}")
        prev = max(prev, e)
dump(big[0], big[1], "restorer LSTM end -> training forward LSTM start (restorer tail + direction-net front)")
dump(big[1], big[2], "training forward LSTM end -> LSTM backward start (head, Gram-Schmidt, loss, their backward)")
last = big[3] if len(big) > 3 else len(step) - 1
tail = step[big[2] + 1:last]
print(f"--- LSTM backward end -> next restorer LSTM start: {(int(step[last]['Start_Timestamp']) - int(step[big[2]]['End_Timestamp'])) / 1e3:.1f} us wall, {len(tail)} kernels")
main_q = step[big[2]]['Queue_Id']
mq = [r for r in tail if r['Queue_Id'] == main_q]
print("  (first 45 and last 40 launches of the main queue)")
prev = int(step[big[2]]['End_Timestamp'])
for r in mq[:45]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"  q{r['Queue_Id']} +{(s - prev) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  {name(r)}")
    prev = max(prev, e)
print("  ...")
front = mq[-40:]
prev = int(front[0]['Start_Timestamp'])
for r in front:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"  q{r['Queue_Id']} +{(s - prev) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  {name(r)}")
    prev = max(prev, e)
