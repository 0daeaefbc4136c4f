"""Training step captured in a HIP graph (torch.cuda.CUDAGraph over the C-ABI launches + autograd + capturable Adam):
the ~600 launches of one NSF cfg2 step replay without per-launch host work.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd.flows import MAF, NSF

dev = torch.device("cuda:0")
for name, make in (("NSF cfg2", lambda: NSF(64, 0, transforms=8, bins=8, hidden_features=[256] * 3)), ("MAF cfg3", lambda: MAF(64, 0, transforms=8, hidden_features=[256] * 3))):
    torch.manual_seed(0)
    flow = make().to(dev)
    opt = torch.optim.Adam(flow.parameters(), lr=1e-3, capturable=True)
    B = 1 << int(os.environ.get("LOG2B", "16"))
    x = torch.randn(B, 64, device=dev)
    loss_out = torch.zeros((), device=dev)

    def step():
        loss = -flow().log_prob(x).mean()
        opt.zero_grad(set_to_none=False)
        loss.backward()
        opt.step()
        loss_out.copy_(loss.detach())

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    l0 = float(loss_out)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: batch 2^{B.bit_length()-1}: graph-replayed training step {dt*1e3:.2f} ms ({B/dt/1e6:.2f} M samples/s), loss {l0:.3f} -> {float(loss_out):.3f}")
