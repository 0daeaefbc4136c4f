"""RealNVP cfg4 parameter gradients vs float64 autograd through the oracle, one-node path vs layer-wise path, several row counts / data seeds:
is the 1-norm distance a property of the path or of which ReLU units sit within float32 rounding of zero?  (run on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(16)
from oracle import zuko_oracle as O
from zuko_amd.flows import RealNVP
dev = torch.device("cuda:0")
torch.manual_seed(0)
flow = RealNVP(256, 0, transforms=16, hidden_features=[512] * 3)
sd = {k: (v.detach().double() if v.is_floating_point() else v.detach()) for k, v in flow.state_dict().items() if v is not None}
pn = [k for k, _ in flow.named_parameters()]
flow = flow.to(dev)
def run(x, off):
    os.environ["ZUKO_AMD_NO_COUPLING_FN"] = "1" if off else "0"
    flow.zero_grad(set_to_none=True)
    (-flow().log_prob(x.to(dev)).mean()).backward()
    return [p.grad.detach().cpu().double() for p in flow.parameters()]
l1 = lambda a, b: max(((u - v).abs().sum() / v.abs().sum().clamp_min(1e-300)).item() for u, v in zip(a, b))
mx = lambda a, b: max(((u - v).abs().max() / v.abs().max().clamp_min(1e-300)).item() for u, v in zip(a, b))
for rows in (1024, 4096):
    for seed in (0, 1, 2):
        x = torch.randn(rows, 256, generator=torch.Generator().manual_seed(seed))
        leaves = {k: sd[k].clone().requires_grad_() for k in pn}
        s2 = dict(sd); s2.update(leaves)
        spec = O.spec_from_state_dict(s2, "coupling", O.UNI_AFFINE, 256)
        (-O.flow_log_prob(spec, x.double(), None).mean()).backward()
        ref = [leaves[k].grad for k in pn]
        a, b = run(x, False), run(x, True)
        print(f"rows {rows} data seed {seed}: 1-norm (max-norm) vs float64: one node {l1(a, ref):.2e} ({mx(a, ref):.2e}), layer-wise f32 {l1(b, ref):.2e} ({mx(b, ref):.2e}); one node vs layer-wise {l1(a, b):.2e}", flush=True)
