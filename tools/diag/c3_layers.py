"""Per-launch times of the convolutions of one C3 (inpainting) train step: kind, shape, ms, TFLOP/s (HIP events).
usage (GPU box): python tools/diag/c3_layers.py"""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(root, "tools")); sys.path.insert(0, root)
import torch
import bench_inpainting as bi
from nppc_audio import unet_engine
L = 64000
F, T = bi.NFFT // 2 + 1, 1 + (L + 2 * (bi.NFFT // 2) - bi.NFFT) // bi.HOP
import os
tr = bi.build("bf16", 32, F, T)
batch = bi.synth(32, F, T, "cuda")
for _ in range(3):
    tr.train_step(batch)
torch.cuda.synchronize()
orig = unet_engine._timed.__init__
shapes = []
def init(self, kind, flops):
    orig(self, kind, flops)
    import inspect
    fr = inspect.currentframe().f_back
    loc = fr.f_locals
    shapes.append({k: loc.get(k) for k in ("h", "w", "cin", "cout", "ks", "name")})
unet_engine._timed.__init__ = init
unet_engine.PROFILE = []
tr.train_step(batch)
torch.cuda.synchronize()
rows = []
for (kind, flops, e0, e1), sh in zip(unet_engine.PROFILE, shapes):
    ms = e0.elapsed_time(e1)
    rows.append((ms, kind, sh, flops / (ms * 1e-3) / 1e12))
tot = {}
for ms, kind, sh, tf in rows:
    tot[kind] = tot.get(kind, 0) + ms
print({k: round(v, 3) for k, v in tot.items()})
for ms, kind, sh, tf in rows:
    print(f"{kind:14s} {str(sh.get('name')):34s} h={sh.get('h')} w={sh.get('w')} cin={sh.get('cin')} cout={sh.get('cout')} ks={sh.get('ks')}  {ms:7.3f} ms  {tf:7.1f} TFLOP/s")
