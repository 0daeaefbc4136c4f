"""cProfile of the host side of RealNVP cfg4 training steps (run on the GPU box)."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd.flows import RealNVP
dev = torch.device("cuda:0")
torch.manual_seed(0)
flow = RealNVP(256, 0, transforms=16, hidden_features=[512] * 3).to(dev)
opt = torch.optim.Adam(flow.parameters(), lr=1e-3)
x = torch.randn(1 << 14, 256, device=dev)
def step():
    loss = -flow().log_prob(x).mean(); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
