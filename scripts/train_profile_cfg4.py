"""Per-C-ABI-call time of one RealNVP cfg4 training step at batch 2^14 (events on the launch stream; run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd import _C
from zuko_amd.flows import RealNVP
dev = torch.device("cuda:0")
torch.manual_seed(0)
flow = RealNVP(256, 0, transforms=16, hidden_features=[512] * 3).to(dev)
opt = torch.optim.Adam(flow.parameters(), lr=1e-3)
x = torch.randn(1 << int(os.environ.get("LOG2N", "14")), 256, device=dev)
def step():
    loss = -flow().log_prob(x).mean(); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): step()
b.record(); torch.cuda.synchronize()
print(f"whole step (10 in a row) {a.elapsed_time(b)/10:.3f} ms")
_C.PROFILE = {}
a.record(); step(); b.record(); torch.cuda.synchronize()
prof, _C.PROFILE = _C.PROFILE, None
tot = 0.0
shapes = {}
for name in ("zk_gemm_f32_skip", "zk_wgrad_f32", "zk_wgrad_bias_f32"):
    for r in prof.get(name, []):
        key = (name,) + tuple(v for v in r[2][:3] if isinstance(v, int))
        shapes.setdefault(key, []).append(r[0].elapsed_time(r[1]))
for key, ts in sorted(shapes.items()):
    n, a_, b_ = key[1], key[2], key[3]
    print(f"{key[0]:18s} N={n} {a_}x{b_}: calls {len(ts):3d} avg {sum(ts)/len(ts):7.3f} ms  dense-equiv {2.0*n*a_*b_/(sum(ts)/len(ts))/1e9:6.1f} TF/s")
for name, recs in sorted(prof.items(), key=lambda kv: -sum(r[0].elapsed_time(r[1]) for r in kv[1])):
    t = sum(r[0].elapsed_time(r[1]) for r in recs); tot += t
    print(f"{name:28s} calls {len(recs):4d}  total {t:8.3f} ms")
print(f"sum of C-ABI calls {tot:.3f} ms; profiled step {a.elapsed_time(b):.3f} ms")
