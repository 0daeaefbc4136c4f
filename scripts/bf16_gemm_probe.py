"""Times zk_linear_bf16 on the cfg5 layer shapes (run on the GPU box; optionally under rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd import ops
from zuko_amd.nn import _Bf16Plan, masked_mlp_masks

dev = torch.device("cuda:0")
N = 1 << int(os.environ.get("LOG2N", "17"))
reps = int(os.environ.get("REPS", "5"))
D, total = 1024, 47
order = torch.arange(D)
adjacency = (order[:, None] > order).repeat_interleave(total, dim=0)
masks = masked_mlp_masks(adjacency, [1024] * 3)
class L:
    def __init__(s, m): s.mask = m.to(dev)
plan = _Bf16Plan([L(m) for m in masks])
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, 1024, generator=g, device=dev).to(torch.bfloat16)
for li in (3, 1):
    mp = plan.masks_p[li]
    out_f, in_f = mp.shape
    w = (torch.randn(out_f, in_f, generator=g, device=dev) / 32).to(torch.bfloat16) * mp
    b = torch.randn(out_f, generator=g, device=dev).to(torch.bfloat16)
    for name, live in (("live-skip", plan.live[li]), ("dense", None)):
        with torch.no_grad():
            y = ops.linear_bf16(x, w, b, live, 1); torch.cuda.synchronize()
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record()
            for _ in range(reps): y = ops.linear_bf16(x, w, b, live, 1)
            b_.record(); torch.cuda.synchronize()
        ms = a_.elapsed_time(b_) / reps
        frac = 1.0 if live is None else plan.live_fraction()[li]
        print(f"N=2^{N.bit_length()-1} {in_f}->{out_f} {name:9s}: {ms:8.3f} ms  dense-equiv {2*N*in_f*out_f/ms/1e9:7.1f} TF/s  executed {2*N*in_f*out_f*frac/ms/1e9:7.1f} TF/s ({2*N*in_f*out_f*frac/ms/1e9/2500*100:.1f}% of 2.5 PF)")
        del y
