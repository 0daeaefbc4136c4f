"""SOSPF / BPF Adam step (64 features, 3 transforms, hidden [256] * 3) at 2^14 / 2^16 rows, with the autograd node types of one step (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd.flows import BPF, SOSPF, MAF
dev = torch.device("cuda:0")
for name, ctor in (("MAF", MAF), ("SOSPF", SOSPF), ("BPF", BPF)):
    torch.manual_seed(0)
    flow = ctor(64, 0, transforms=3, hidden_features=[256] * 3).to(dev)
    opt = torch.optim.Adam(flow.parameters(), lr=1e-3)
    for lg in (14, 16):
        x = 0.8 * torch.randn(1 << lg, 64, device=dev)
        def step():
            loss = -flow().log_prob(x).mean(); opt.zero_grad(set_to_none=True); loss.backward(); opt.step(); return loss
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        loss = -flow().log_prob(x).mean()
        seen, stack = set(), [loss.grad_fn]
        while stack:
            f = stack.pop()
            if f is None or f in seen: continue
            seen.add(f); stack += [n for n, _ in f.next_functions]
        names = sorted({type(f).__name__ for f in seen if "Fn" in type(f).__name__})
        print(f"{name} 2^{lg}: {dt * 1e3:.2f} ms per step ({(1 << lg) / dt / 1e6:.2f} M samples/s); custom nodes: {names}", flush=True)
