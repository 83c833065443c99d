import torch
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))), 'tools'))
from runlog import banner
banner()
DT=torch.float16
for (M,K,N) in [(12288,1280,3840),(12288,1280,5120),(12288,5120,1280),(12288,1280,1280),(49152,1280,5120)]:
    x=torch.randn(M,K,device="cuda").to(DT); w=torch.randn(N,K,device="cuda").to(DT); b=torch.randn(N,device="cuda").to(DT)
    for _ in range(3): torch.nn.functional.linear(x,w,b)
    torch.cuda.synchronize()
