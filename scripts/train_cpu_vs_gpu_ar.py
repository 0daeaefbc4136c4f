"""Host time to ENQUEUE one NSF cfg2 / MAF cfg3 training step vs the time until the GPU has finished it (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd.flows import NSF, MAF
dev = torch.device("cuda:0")
for name, mk in (("NSF cfg2", lambda: NSF(64, 0, transforms=8, bins=8, hidden_features=[256] * 3)), ("MAF cfg3", lambda: MAF(64, 0, transforms=8, hidden_features=[256] * 3))):
    torch.manual_seed(0)
    flow = mk().to(dev)
    opt = torch.optim.Adam(flow.parameters(), lr=1e-3)
    x = torch.randn(1 << int(os.environ.get("LOG2N", "16")), 64, device=dev)
    def step():
        loss = -flow().log_prob(x).mean(); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    for _ in range(5): step()
    enq, tot = [], []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    print(f"{name}: one step from an idle GPU: host enqueue {min(enq):.2f} ms, until the GPU is done {min(tot):.2f} ms; 20 steps back to back {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per step")
