import sys, os, time
sys.path.insert(0, '/root/repo')
import torch
from zuko_amd.flows import NSF
dev = torch.device("cuda:0")
for logB in (14, 16, 18):
    torch.manual_seed(0)
    flow = NSF(64, 0, transforms=8, bins=8, hidden_features=[256] * 3).to(dev)
    opt = torch.optim.Adam(flow.parameters(), lr=1e-3)
    B = 1 << logB
    x = torch.randn(B, 64, device=dev)
    def step():
        loss = -flow().log_prob(x).mean()
        opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"NSF cfg2 Adam step at 2^{logB}: {dt*1e3:.2f} ms ({B/dt/1e6:.2f} M samples/s)", flush=True)
