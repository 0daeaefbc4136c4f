"""Host time per segment of CouplingFn.forward / backward (monkey-patched timers; run on the GPU box)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd import coupling_train as ct, train, autograd as AG, _C
from zuko_amd.flows import RealNVP
T = collections.defaultdict(float)
def wrap(mod, name, label=None):
    f = getattr(mod, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[label or name] += time.perf_counter() - t0; return r
    setattr(mod, name, g)
wrap(ct, "amax"); wrap(ct, "wsplit"); wrap(ct, "gemm"); wrap(ct, "_maps"); wrap(AG, "_fwd_any"); wrap(AG, "_adj_any")
wrap(train.SortedPlan, "wgrad_multi")
fw, bw = ct.CouplingFn.forward, ct.CouplingFn.backward
def fwd(ctx, *a):
    t0 = time.perf_counter(); r = fw(ctx, *a); T["CouplingFn.forward (all)"] += time.perf_counter() - t0; return r
def bwd(ctx, *a):
    t0 = time.perf_counter(); r = bw(ctx, *a); T["CouplingFn.backward (all)"] += time.perf_counter() - t0; return r
ct.CouplingFn.forward = staticmethod(fwd); ct.CouplingFn.backward = staticmethod(bwd)
wrap(ct, "coupling", "coupling() incl. apply")
dev = torch.device("cuda:0")
torch.manual_seed(0)
flow = RealNVP(256, 0, transforms=16, hidden_features=[512] * 3).to(dev)
opt = torch.optim.Adam(flow.parameters(), lr=1e-3)
x = torch.randn(1 << 14, 256, device=dev)
tt = collections.defaultdict(float)
def step():
    t0 = time.perf_counter(); loss = -flow().log_prob(x).mean(); t1 = time.perf_counter(); opt.zero_grad(set_to_none=True); loss.backward(); t2 = time.perf_counter(); opt.step(); t3 = time.perf_counter()
    tt["forward"] += t1 - t0; tt["backward"] += t2 - t1; tt["optimizer"] += t3 - t2
for _ in range(5): step()
T.clear(); tt.clear(); torch.cuda.synchronize()
n = 20
for _ in range(n): step()
torch.cuda.synchronize()
for k, v in tt.items(): print(f"{k:34s} {v / n * 1e3:7.3f} ms per step (host)")
for k, v in sorted(T.items(), key=lambda kv: -kv[1]): print(f"  {k:32s} {v / n * 1e3:7.3f} ms per step")
