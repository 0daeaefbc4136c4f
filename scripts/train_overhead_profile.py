#!/usr/bin/env python
"""Host-side cost of one training step at a SMALL batch (NSF cfg2, batch 256: the GPU work is a fraction of a millisecond, the step time is Python + launches):
cProfile of 30 steps, top entries by cumulative time."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd.flows import NSF

dev = torch.device("cuda:0")
torch.manual_seed(0)
flow = NSF(64, 0, transforms=8, bins=8, hidden_features=[256] * 3).to(dev)
opt = torch.optim.Adam(flow.parameters(), lr=1e-3)
x = torch.randn(int(os.environ.get("B", "256")), 64, device=dev)


def step():
    loss = -flow().log_prob(x).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    step()
torch.cuda.synchronize()
print(f"step at batch {x.shape[0]}: {(time.perf_counter() - t0) / 30 * 1e3:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
