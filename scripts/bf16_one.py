"""One dense zk_linear_bf16 launch shape for PMC passes: N = 2^LOG2N, IN -> OUT (defaults 2^17, 1024 -> 8192)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd import ops
dev = torch.device("cuda:0")
N, IN, OUT = 1 << int(os.environ.get("LOG2N", "17")), int(os.environ.get("IN", "1024")), int(os.environ.get("OUT", "8192"))
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, IN, generator=g, device=dev).to(torch.bfloat16)
w = (torch.randn(OUT, IN, generator=g, device=dev) / 32).to(torch.bfloat16)
with torch.no_grad():
    for _ in range(int(os.environ.get("REPS", "3"))):
        y = ops.linear_bf16(x, w, None, None, 1)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); y = ops.linear_bf16(x, w, None, None, 1); b.record(); torch.cuda.synchronize()
print(f"N={N} {IN}->{OUT}: {a.elapsed_time(b):.3f} ms = {2*N*IN*OUT/a.elapsed_time(b)/1e9:.1f} TF/s")
