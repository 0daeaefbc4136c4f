#!/usr/bin/env python
"""The one-launch backward of a transform alone (zk_ar_backward_full, csrc/fused_ar_split_impl.h: arxb_kernel): NSF / MAF cfg2 / cfg3 conditioner
at batch 2^16, ms per launch, next to the stand-alone adjoint kernel + dgrad chain it replaces (ZUKO_AMD_NO_FUSED_AR_BACKWARD=1 semantics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd import train
from zuko_amd.autograd import _adj_any
from zuko_amd.flows import MAF, NSF

dev = torch.device("cuda:0")
N = 1 << int(os.environ.get("LOGN", "16"))
for name, make, uni in (("NSF", lambda: NSF(64, 0, transforms=1, bins=8, hidden_features=[256] * 3), (1, 5.0, 1e-3, (8, 8, 7))),
                        ("MAF", lambda: MAF(64, 0, transforms=1, hidden_features=[256] * 3), (0, 5.0, 1e-3, (1, 1)))):
    torch.manual_seed(0)
    flow = make().to(dev)
    hyper = flow.transform.transforms[0].hyper
    plan, lins = train.plan_for(hyper, dev)
    st = train._fused_forward_state(plan, lins, dev)
    x = torch.randn(N, 64, device=dev)
    st.refresh(lins, fine_only=True)
    bk = train._backward_kernel(plan, st, N)
    ch = train._dgrad_chain(plan, lins, N)
    acts, phi, y, ladj = train._fused_forward(st, x, plan.shapes[-1][0], uni=(uni[1], uni[2]))
    phi_packed = train._fused_forward(st, x, plan.shapes[-1][0], uni=(uni[1], uni[2]), packed_width=bk.packed.width)[1]
    hs = [x, *acts]
    gy, gl = torch.randn(N, 64, device=dev), torch.randn(N, device=dev)

    def fused():
        return bk.run_backward(plan, bk.gather(plan, lins), st, uni, x, phi_packed, gy, gl, hs)

    def two():
        gx, gphi = _adj_any((uni[0], uni[1], uni[2], uni[3], ()), x, phi.view(N, 64, -1), gy, gl, True)
        return ch.run(plan, ch.gather(plan, lins), gphi.view(N, -1), hs, gx_add=gx)

    for tag, fn in (("one launch", fused), ("adjoint + chain", two)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{name} backward to g_phi, g_h, g_x at 2^{N.bit_length() - 1}: {tag:16s} {e0.elapsed_time(e1) / 20:.3f} ms", flush=True)
