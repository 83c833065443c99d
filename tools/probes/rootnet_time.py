#!/usr/bin/env python3
"""RootNet (ResNet-34 + depth head) forward time for B hand patches, per-layer table from the library's profiler, with the
round-3 convolution switches on and off.  Usage: python tools/probes/rootnet_time.py [B=66]"""
import os, sys, time, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hamer_yolo_amd import lib as L, synth
from hamer_yolo_amd.rootnet.Model_RGB import get_model
from runlog import banner
banner()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 66
m = get_model()
eng = m.engine if hasattr(m, "engine") else m
img = torch.randn(B, 3, 256, 256, device="cuda")
kv = torch.ones(B, device="cuda")
lib = L.load()
for name, opts in (("default", {}), ("no K groups", {L.HM_OPT_CONV_KGROUPS: 1})):
    for k, v in opts.items(): L.check(lib.hm_set_option(k, v))
    for _ in range(3): eng.forward(img, kv)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): eng.forward(img, kv)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    with L.profile(capacity=1024) as prof:
        eng.forward(img, kv); torch.cuda.synchronize()
    conv = [(r[2], r[3], r[4], r[5]) for r in prof.records if r[0] == "conv"]
    print(f"{name}: {ms:.3f} ms per forward of {B} patches; {len(conv)} conv launches, {sum(c[3] for c in conv):.3f} ms by events")
    agg = collections.OrderedDict()
    for M, N, K, t in conv: agg.setdefault((M, N, K), []).append(t)
    for (M, N, K), ts in agg.items(): print(f"   M={M:7d} N={N:4d} K={K:5d} x{len(ts):2d}  {sum(ts)/len(ts)*1e3:8.1f} us each  {2.0*M*N*K/(sum(ts)/len(ts))/1e9:7.1f} TF/s")
    for k in opts: L.check(lib.hm_set_option(k, 0))
