#!/usr/bin/env python3
"""Builds libnppc_hip.so (gfx950 kernels + C ABI) in-tree with hipcc.  No GPU needed."""
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
OUT = os.path.join(HERE, "nppc_audio", "libnppc_hip.so")
OBJ = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INC, "-I" + CSRC, "-Wno-unused-value"]


def _stamp(src):
    h = hashlib.sha1()
    for p in [src, os.path.join(CSRC, "common.h"), os.path.join(INC, "nppc_hip.h")]:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src):
    name = os.path.basename(src)[:-4]
    obj = os.path.join(OBJ, name + ".o")
    st = obj + ".stamp"
    want = _stamp(src)
    if os.path.exists(obj) and os.path.exists(st) and open(st).read() == want:
        return obj, False
    subprocess.run([HIPCC] + FLAGS + ["-c", src, "-o", obj], check=True)
    with open(st, "w") as f:
        f.write(want)
    return obj, True


def build(verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(_compile, srcs))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or not os.path.exists(OUT):
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs, check=True)
    if verbose:
        print(f"built {OUT} from {len(srcs)} sources ({sum(ch for _, ch in res)} recompiled)")
    return OUT


if __name__ == "__main__":
    build()
    sys.exit(0)
