#!/usr/bin/env python3
"""prints avg us of selected kernels from a rocprofv3 kernel_stats csv: kstat.py <csv> name1 name2 ..."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for pat in sys.argv[2:]:
    for r in rows:
        if pat in r["Name"]:
            print(f"{pat:24s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms   {r['Name'][:60]}")
