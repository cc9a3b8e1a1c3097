#!/usr/bin/env python3
"""Copy the round-4 evidence that tools/diag/r04_evidence.sh left under gpurun_out/ into profiles/ (tracked) and write the
stamps that tie each PMC summary to the kernel sources it profiled.  Runs on the GPU box (so that the bench lines that
follow report roofline.traffic from the passes just made) and again in the build container (to commit the copies).
usage: r04_collect.py <c2|c3> <git head>"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

TU = {"c2": ["lstm_coop.hip", "common.h"], "c3": ["unet.hip", "common.h"], "c5": ["lstm_coop.hip", "lstm_ws.hip", "common.h"]}
ARGS = {"c2": "--no-fp32 --no-families", "c3": "--config c3", "c5": "--config c5 --no-fp32 --no-families"}


def main():
    cfg, head = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "unknown"
    G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
    for suffix in ("kernel_stats.csv", "pmc_traffic.csv", "timeline.txt", "windows.txt"):
        src = os.path.join(G, f"r04_{cfg}_{suffix}")
        if os.path.exists(src) and os.path.getsize(src):
            shutil.copy(src, os.path.join(P, f"r04_bench_{cfg}_bf16_{suffix}"))
    stamp_path = os.path.join(G, f"r04_{cfg}_pmc_stamp.json")
    if not os.path.exists(stamp_path):      # first call, on the box that made the passes
        json.dump({"kernel_source_hash": bench.kernel_source_hash(TU[cfg]), "files": TU[cfg], "git_head": head,
                   "whole_csrc_hash": bench.kernel_source_hash(),
                   "command": f"rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py {ARGS[cfg]} "
                              f"--steps 2 --warmup 1 --no-cpu-baseline",
                   "note": "bytes per launch = 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE; tools/summarize_pmc.py; the "
                           "hash covers the translation unit of the dominant kernel (files) + include/nppc_hip.h"},
                  open(stamp_path, "w"), indent=1)
    shutil.copy(stamp_path, os.path.join(P, f"r04_bench_{cfg}_bf16_pmc_stamp.json"))
    for name in (f"r04_bench_{cfg}_bf16.json", "r04_gpu_tests.log", "r04_bench_c2_xpacked0.json", "r04_parity_errors.json",
                 "r04_c2_fp32_kernel_stats.csv"):
        src = os.path.join(G, name)
        if os.path.exists(src) and os.path.getsize(src):
            shutil.copy(src, os.path.join(P, name))


if __name__ == "__main__":
    main()
