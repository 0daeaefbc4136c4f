"""Does a whole training step (forward, backward, Adam) replay as a HIP graph?  (run on the GPU box under `timeout`)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd.flows import NSF, RealNVP, MAF
dev = torch.device("cuda:0")
which = sys.argv[1]
torch.manual_seed(0)
if which == "cfg1":
    flow = NSF(3, 5, transforms=3, bins=8, hidden_features=[128] * 3).to(dev); D, C, N = 3, 5, 1 << 12
elif which == "cfg4":
    flow = RealNVP(256, 0, transforms=16, hidden_features=[512] * 3).to(dev); D, C, N = 256, 0, 1 << 14
else:
    flow = MAF(64, 0, transforms=8, hidden_features=[256] * 3).to(dev); D, C, N = 64, 0, 1 << 14
opt = torch.optim.Adam(flow.parameters(), lr=1e-3, capturable=True)
x = torch.randn(N, D, device=dev); c = torch.randn(N, C, device=dev) if C else None
loss_out = torch.zeros((), device=dev)
def step():
    loss = -flow(c).log_prob(x).mean()
    opt.zero_grad(set_to_none=False)
    loss.backward()
    opt.step()
    loss_out.copy_(loss.detach())
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(6): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 20
print(f"{which}: eager {eager * 1e3:.3f} ms per step, loss {loss_out.item():.5f}", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
l0 = loss_out.item()
t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize(); rep = (time.perf_counter() - t0) / 20
print(f"{which}: graph replay {rep * 1e3:.3f} ms per step, loss {l0:.5f} -> {loss_out.item():.5f}", flush=True)
