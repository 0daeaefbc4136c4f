"""Host time to ENQUEUE one RealNVP cfg4 training step vs the time until the GPU has finished it (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd.flows import RealNVP
dev = torch.device("cuda:0")
torch.manual_seed(0)
flow = RealNVP(256, 0, transforms=16, hidden_features=[512] * 3).to(dev)
opt = torch.optim.Adam(flow.parameters(), lr=1e-3)
x = torch.randn(1 << int(os.environ.get("LOG2N", "14")), 256, device=dev)
def step():
    loss = -flow().log_prob(x).mean(); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(5): step()
enq, tot = [], []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print(f"one step from an idle GPU: host enqueue {min(enq):.2f} ms (median {sorted(enq)[5]:.2f}), until the GPU is done {min(tot):.2f} ms (median {sorted(tot)[5]:.2f})")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print(f"20 steps back to back: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per step")
# forward only / backward only
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    loss = -flow().log_prob(x).mean()
torch.cuda.synchronize(); print(f"forward (with graph recording) only: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms")
