"""Which ATen operators (= torch kernels on the hot path) one steady-state C2 train step issues, with their Python call sites.
usage (GPU box): python tools/diag/aten_ops.py"""
import os, sys, collections, traceback
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, root)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
torch.cuda.set_device(0)
tr, batch = bench.build_trainer("bf16", 0, 1, 32, 64000)
for _ in range(3):
    tr.train_step(batch)
torch.cuda.synchronize()
seen = collections.Counter()
SKIP = ("aten.view", "aten.detach", "aten.alias", "aten._unsafe_view", "aten.reshape", "aten.as_strided", "aten.select", "aten.slice",
        "aten.t.", "aten.permute", "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.empty", "aten.transpose", "aten.is_", "aten._local_scalar",
        "aten.lift_fresh", "aten.unbind", "aten.split", "aten.stride", "aten.sym_", "aten.new_empty", "aten.record_stream")
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            st = [f for f in traceback.extract_stack() if "/generative-audio_amd/" in f.filename or f.filename.endswith("bench.py")]
            site = f"{os.path.basename(st[-1].filename)}:{st[-1].lineno}" if st else "(autograd / other)"
            seen[(name, site)] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    tr.train_step(batch)
torch.cuda.synchronize()
for (name, site), n in sorted(seen.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"{n:3d}  {name:40s} {site}")
print("total", sum(seen.values()))
