"""The screened bisection of the incremental kernel's polynomial epilogues against the plain loop (library built with -DINC_NO_SCREEN=1, ZUKO_AMD_LIB):
run once per library with OUT=<file>; with two files given as arguments, compares them bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if len(sys.argv) == 3:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    for k in a:
        same = torch.equal(a[k], b[k])
        print(f"{k}: screened == plain loop bit for bit: {same}" + ("" if same else f" (max |d| {(a[k] - b[k]).abs().max().item():.2e}, {int((a[k] != b[k]).sum())} of {a[k].numel()} elements)"))
    sys.exit(0)
from zuko_amd.flows import BPF, SOSPF
dev = torch.device("cuda:0")
out = {}
for name, ctor in (("SOSPF", SOSPF), ("BPF", BPF)):
    for seed, scale in ((0, 1.0), (1, 4.0)):  # (the second: weights x 4 — larger coefficients, flatter and steeper stretches)
        torch.manual_seed(seed)
        flow = ctor(64, 0, transforms=3, hidden_features=[256] * 3).to(dev)
        with torch.no_grad():
            for p in flow.parameters():
                p.mul_(scale)
            tr = flow().transform
            z = 1.5 * torch.randn(1 << 14, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
            out[f"{name} seed {seed} weights x {scale:g}"] = tr.inv(z).cpu()
torch.save(out, os.environ["OUT"])
