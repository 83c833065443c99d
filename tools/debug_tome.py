import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamer_yolo_amd import synth
from hamer_yolo_amd.engine import HamerEngine
from oracle import tome_ref as T
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'tools'))
from runlog import banner
banner()
cfg = synth.tome_tiny_config()
sd = synth.hamer_state_dict(cfg, seed=7)
mp = synth.mano_params(seed=0)
img = synth.normalize_crops(synth.crops_u8(3, seed0=70))
eng = HamerEngine(sd, mp, cfg, dtype=torch.float16, token_merge=True)
out = eng.forward(img.cuda(), want_tokens=True)
torch.cuda.synchronize()
tok = out["tokens"].float().cpu()[:3 * 146].reshape(3, 146, -1)
trace = {}
with torch.no_grad():
    feats = T.vit_forward_tome(sd, img[:, :, :, 32:-32], cfg.vit, (8, -1), emu="fp16", trace=trace)
d = (tok - feats).abs().amax(-1)
print("rows differing > 0.05 per crop:", [(d[b] > 0.05).nonzero().flatten().tolist() for b in range(3)])
for b in range(3):
    dist = torch.cdist(tok[b], feats[b])
    nn = dist.argmin(1)
    print("crop", b, "is permutation:", sorted(nn.tolist()) == list(range(146)), "max dist after matching", float(dist.min(1).values.max()),
          "moved:", [(i, int(j)) for i, j in enumerate(nn.tolist()) if i != j][:20])
