#!/bin/bash
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from zuko_amd import ops, _C
dev = torch.device('cuda:0')
B, D, K = 1 << 20, 64, 8
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randn(B, D, generator=g, device=dev)
phi = torch.randn(B, D, 3 * K - 1, generator=g, device=dev)
w, h, d = phi[..., :K], phi[..., K:2 * K], phi[..., 2 * K:]
for name, fn in (("rqs_forward reduced", lambda: ops.rqs_forward(x, w, h, d, reduce=True)), ("rqs_forward full ladj", lambda: ops.rqs_forward(x, w, h, d)), ("rqs_inverse", lambda: ops.rqs_inverse(x, w, h, d))):
    with torch.no_grad():
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"{name:24s} {ms:.3f} ms  {B * 6404 / ms / 1e6:.0f} GB/s algorithmic ({B*6404/ms/1e6/8000*100:.1f}% of 8 TB/s)")
PY
