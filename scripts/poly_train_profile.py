"""Per-C-ABI-call time of one SOSPF / BPF training step (64 features, 3 transforms, hidden [256] * 3, 2^16 rows; events on the launch stream; run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd import _C
from zuko_amd.flows import BPF, SOSPF
dev = torch.device("cuda:0")
for name, ctor in (("SOSPF", SOSPF), ("BPF", BPF)):
    torch.manual_seed(0)
    flow = ctor(64, 0, transforms=3, hidden_features=[256] * 3).to(dev)
    opt = torch.optim.Adam(flow.parameters(), lr=1e-3)
    x = 0.8 * torch.randn(1 << 16, 64, device=dev)
    def step():
        loss = -flow().log_prob(x).mean(); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    for _ in range(4): step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): step()
    b.record(); torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / 5:.2f} ms per step")
    _C.PROFILE = {}
    step(); torch.cuda.synchronize()
    prof, _C.PROFILE = _C.PROFILE, None
    for nm, recs in sorted(prof.items(), key=lambda kv: -sum(r[0].elapsed_time(r[1]) for r in kv[1]))[:8]:
        print(f"    {nm:28s} calls {len(recs):4d}  total {sum(r[0].elapsed_time(r[1]) for r in recs):8.3f} ms")
