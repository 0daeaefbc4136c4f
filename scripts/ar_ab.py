"""A/B timing of zk_ar_forward across several builds of the library on ONE box (run on the GPU box).

    python scripts/ar_ab.py name=path/to/lib.so [name=path ...]      (the current build is always included as `cur`)

One transform of the headline flow (NSF cfg2, or MAF cfg3 with CONFIG=cfg3) at batch 2^LOG2N; launches are
interleaved across the libraries round-robin so clock drift hits all of them alike; outputs must agree bit for bit."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import zuko_amd
from zuko_amd import _C
from zuko_amd.flows import MAF, NSF
from zuko_amd.nn import MaskedLinear

dev = torch.device("cuda:0")
N = 1 << int(os.environ.get("LOG2N", "20"))
rounds = int(os.environ.get("ROUNDS", "10"))
torch.manual_seed(0)
flow = (MAF(64, 0, transforms=8, hidden_features=[256] * 3) if os.environ.get("CONFIG") == "cfg3" else NSF(64, 0, transforms=8, bins=8, hidden_features=[256] * 3)).to(dev)
lazy = flow.transform.transforms[0]
st = lazy.fused_state(dev)
st.refresh([m for m in lazy.hyper if isinstance(m, MaskedLinear)])
x = torch.randn(N, 64, generator=torch.Generator().manual_seed(1)).to(dev)
p = st.plan
libs = {"cur": _C.LIB_PATH}
for a in sys.argv[1:]:
    k, v = a.split("=", 1)
    libs[k] = os.path.abspath(v)
fns = {}
for k, path in libs.items():
    fn = ctypes.CDLL(path).zk_ar_forward
    fn.argtypes = _C.SIGNATURES["zk_ar_forward"]
    fn.restype = ctypes.c_int
    fns[k] = fn
outs = {k: (torch.empty(N, 64, device=dev), torch.empty(N, device=dev)) for k in libs}
P = lambda t: ctypes.c_void_p(t.data_ptr())


def launch(k):
    y, l = outs[k]
    err = fns[k](p.layout.kind, N, 64, 64, P(x), 64, P(y), 64, P(l), 0, P(st.stream), P(st.bias), st.bias_floats, P(st.skip), P(st.featmap), p.n_layers, p.n_groups,
                 p.n_chunks, st.act, st.bound, st.slope, 0, torch.cuda.current_stream().cuda_stream)
    assert err == 0, (k, err)


for k in libs:
    launch(k); launch(k)
torch.cuda.synchronize()
times = {k: [] for k in libs}
for r in range(rounds):
    for k in libs:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(4):
            launch(k)
        b.record()
        torch.cuda.synchronize()
        times[k].append(a.elapsed_time(b) / 4)
ref = outs["cur"]
for k in libs:
    ts = sorted(times[k])
    same = torch.equal(outs[k][0], ref[0]) and torch.equal(outs[k][1], ref[1])
    print(f"{k:12s} median {ts[len(ts)//2]:.4f} ms  min {ts[0]:.4f}  max {ts[-1]:.4f}   bit-identical to cur: {same}")
