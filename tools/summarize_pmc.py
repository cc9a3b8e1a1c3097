#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 counter passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, each in its own run
with --output-format csv): sums the counters per kernel name, applies the gfx950 FETCH_SIZE x2 correction of
MI355X_MICROARCH.md (wide coalesced reads are tallied at 64 B per 128-B request) and prints a csv.
usage: summarize_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> [top N]"""
import csv
import sys
from collections import defaultdict


def load(path, counter):
    tot, launches = defaultdict(float), defaultdict(set)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            launches[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in launches.items()}


def main():
    fetch, nf = load(sys.argv[1], "FETCH_SIZE")
    write, _ = load(sys.argv[2], "WRITE_SIZE")
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "launches", "FETCH_SIZE_sum_KB", "WRITE_SIZE_sum_KB", "fetch_MB_per_launch_x2_corrected",
                "write_MB_per_launch"])
    for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0.0)))[:top]:
        n = max(nf[k], 1)
        w.writerow([k[:110], n, int(fetch[k]), int(write.get(k, 0.0)), round(2 * fetch[k] / 1024 / n, 1),
                    round(write.get(k, 0.0) / 1024 / n, 1)])


if __name__ == "__main__":
    main()
