"""Training-step timing of the README loop (zuko README.md:43-49): loss = -flow().log_prob(x).mean(); backward; Adam.
Forward through the layer-wise HIP kernels with autograd Functions, adjoints in csrc/backward.hip, dgrad / wgrad
through torch.mm (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd.flows import MAF, NSF

dev = torch.device("cuda:0")
for name, make, logB in (("NSF cfg2", lambda: NSF(64, 0, transforms=8, bins=8, hidden_features=[256] * 3), 16),
                         ("MAF cfg3", lambda: MAF(64, 0, transforms=8, hidden_features=[256] * 3), 16)):
    torch.manual_seed(0)
    flow = make().to(dev)
    opt = torch.optim.Adam(flow.parameters(), lr=1e-3)
    B = 1 << logB
    x = torch.randn(B, 64, device=dev)
    def step():
        loss = -flow().log_prob(x).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss
    for _ in range(3): l0 = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): l = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    with torch.no_grad():
        flow().log_prob(x); torch.cuda.synchronize()  # (builds the fused plan)
        t1 = time.perf_counter()
        for _ in range(n): flow().log_prob(x)
        torch.cuda.synchronize(); df = (time.perf_counter() - t1) / n
    print(f"{name}: batch 2^{logB}: training step {dt*1e3:.2f} ms ({B/dt/1e6:.2f} M samples/s), loss {float(l0):.3f} -> {float(l):.3f}; inference log_prob {df*1e3:.2f} ms ({B/df/1e6:.2f} M samples/s)")
