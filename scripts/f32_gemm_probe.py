"""Times zk_linear (fp32 MFMA) on the cfg4 / cfg2 layer shapes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd import ops

dev = torch.device("cuda:0")
N = 1 << int(os.environ.get("LOG2N", "19"))
g = torch.Generator(device=dev).manual_seed(0)
for in_f, out_f, act in ((128, 512, 1), (512, 512, 1), (512, 256, 0), (256, 1472, 0), (512, 512, 2)):
    x = torch.randn(N, in_f, generator=g, device=dev)
    w = torch.randn(out_f, in_f, generator=g, device=dev) / in_f**0.5
    b = torch.randn(out_f, generator=g, device=dev)
    with torch.no_grad():
        y = ops.linear(x, w, b, None, act); torch.cuda.synchronize()
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record()
        for _ in range(5): y = ops.linear(x, w, b, None, act)
        b_.record(); torch.cuda.synchronize()
    ms = a_.elapsed_time(b_) / 5
    print(f"N=2^{N.bit_length()-1} {in_f}->{out_f} act {act}: {ms:7.3f} ms  {2*N*in_f*out_f/ms/1e9:6.1f} TF/s ({2*N*in_f*out_f/ms/1e9/157.3*100:.1f}% of 157.3)")
    del x, w, b, y
