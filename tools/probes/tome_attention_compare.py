"""hm_tome_attention: MFMA kernel vs the scalar fp32 kernel (HM_OPT_TOME_SCALAR_ATTENTION) vs an fp64 reference."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hamer_yolo_amd import lib as L
from hamer_yolo_amd import ops, synth
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))), 'tools'))
from runlog import banner
banner()

B, H, d = 3, 4, 80
for tokens, with_size in ((192, False), (192, True), (176, True), (164, True), (155, True), (149, True), (23, True)):
    qkv = synth.uniform("tq", (B * tokens, 3 * H * d), 1.2, seed=tokens).half()
    size = (1.0 + (synth._hash_u32(torch.arange(B * tokens, dtype=torch.int64), 7) % 4).float()) if with_size else None
    q, k, v = qkv.double().reshape(B, tokens, 3, H, d).permute(2, 0, 3, 1, 4)
    a = (q @ k.transpose(-1, -2)) * d ** -0.5
    if with_size:
        a = a + size.double().reshape(B, 1, 1, tokens).log()
    ref = (a.softmax(-1) @ v).transpose(1, 2).reshape(B * tokens, H * d)
    g1 = ops.tome_attention(qkv.cuda(), size.cuda() if with_size else None, B, tokens, H, d, d ** -0.5).double().cpu()
    with L.option(L.HM_OPT_TOME_SCALAR_ATTENTION, 1):
        g2 = ops.tome_attention(qkv.cuda(), size.cuda() if with_size else None, B, tokens, H, d, d ** -0.5).double().cpu()
    mism = (g1 != g2).double().mean().item()
    r16 = ref.half().double()
    print(f"   bitwise mismatch mfma vs scalar {mism:.4f}; vs fp64->fp16: mfma {(g1 != r16).double().mean():.4f} scalar {(g2 != r16).double().mean():.4f}")
    e1, e2 = (g1 - ref).abs(), (g2 - ref).abs()
    i = e1.argmax()
    print(f"T={tokens} size={with_size}: mfma max err {e1.max():.2e} (row {i // (H*d) % tokens}, col {i % (H*d)}), mean {e1.mean():.2e}; scalar max {e2.max():.2e} mean {e2.mean():.2e}; "
          f"rows worst by mfma err: {e1.amax(1).topk(3).indices.tolist()}")
