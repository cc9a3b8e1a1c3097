"""Which LSTM kernels the g2_k5 bf16 step runs under NPPC_LSTM_WS=<mode>, with checksums of the restorer output and of a few gradients
(GPU box):  NPPC_LSTM_WS=0 python tools/diag/ws_auto_check.py;  NPPC_LSTM_WS=auto python tools/diag/ws_auto_check.py"""
import os, sys, tempfile
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [root, os.path.join(root, "generative-audio_amd"), os.path.join(root, "tests")]
import torch
from golden_util import load, waves
from test_train_step_gpu import build_model
from nppc_audio import ops_lstm
from nppc_audio.trainer import nppc_base_step
z, meta = load("g2_k5")
c = meta["config"]
model, wts = build_model(c, "bf16", tempfile.mkdtemp())
wn, wc = waves(z, meta)
noisy, clean = torch.from_numpy(wn).cuda(), torch.from_numpy(wc).cuda()
ops_lstm.PROFILE = []
rec, obj, log = nppc_base_step(model, (noisy, clean), 500, 500, 1.0)
model.zero_grad()
obj.backward()
torch.cuda.synchronize()
print("mode", ops_lstm.WS_MODE, "launches", [l[0] for l in ops_lstm.PROFILE])
print("objective %.9f  pred_crm sum %.9e  w_mat sum %.9e" % (float(obj), float(log["pred_crm"].double().abs().sum()), float(log["w_mat"].double().abs().sum())))
P = dict(model.named_parameters())
for n in ("audio_pc_wrapper.net.fb_model.fc_output_layer.weight", "audio_pc_wrapper.net.channel_attention.fc2.weight",
          "audio_pc_wrapper.net.fb_model.sequence_model.3.conv1x1.weight", "audio_pc_wrapper.net.sb_model.sequence_model.weight_hh_l1"):
    print("  grad |sum| %-70s %.9e" % (n, float(P[n].grad.double().abs().sum())))
