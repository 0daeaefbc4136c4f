"""One fused coupling launch of the cfg4 shape (RealNVP d=256, hidden 512 x 3), for the -DZK_CP_TIMING probe build:
    bash scripts/build_tu_variant.sh fused_coupling cp_t -DZK_CP_TIMING=1 && ZUKO_AMD_LIB=scripts/probes/ab/lib_cp_t.so python scripts/cp_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zuko_amd
from zuko_amd.flows import RealNVP

dev = torch.device("cuda:0")
torch.manual_seed(0)
flow = RealNVP(features=256, context=0, transforms=16, hidden_features=[512] * 3).to(dev)
N = 1 << int(os.environ.get("LOG2N", "19"))
x = torch.randn(N, 256, device=dev)
with torch.no_grad():
    t = flow().transform.transforms[0]
    for _ in range(2):
        y, l = t.call_and_ladj(x); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); y, l = t.call_and_ladj(x); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{os.environ.get('ZUKO_AMD_LIB', 'product').split('/')[-1]}: one transform call: min {min(ts):.3f} ms, median {sorted(ts)[2]:.3f} ms")
