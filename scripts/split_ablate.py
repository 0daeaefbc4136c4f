#!/usr/bin/env python
"""Timing ablations of the operand-split cfg2 kernel (csrc/fused_ar_split_impl.h, ARX_ABL).

    python scripts/split_ablate.py build      (here: hipcc cross-compiles the variants into zuko_amd/lib/ars/abl/)
    python scripts/split_ablate.py run [log2] (GPU box: ms per launch of every variant; results are garbage by construction)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zuko_amd import static_ar as sa  # noqa: E402

HALF = os.environ.get("ABL_HALF", "0") == "1"  # the two-part (f16 x 2) kernel instead of the three-part one
OUT = os.path.join(sa.ARS_DIR, ("ablh_" if HALF else "abl_") + "x".join(map(str, sa.split_geometry())) + os.environ.get("ABL_TAG", ""))
NAMES = {0: "full kernel", 1: "no DMA into the ring", 2: "no MFMA", 3: "no spline arithmetic", 4: "no barrier at chunk boundaries", 5: "no LDS reads of the weights", 6: "no operand conversion"}


def plan():
    (pa, lay, _), _ = sa._plans_for(*sa.PREBUILT[0][:5])
    return pa, lay


def build():
    pa, lay = plan()
    t, _ = (sa.half_tables if HALF else sa.split_tables)(pa, lay.kind, 1)
    t = dict(t)
    for kv in filter(None, os.environ.get("ABL_SHAPE", "").split(",")):  # probe builds: ABL_SHAPE="NR=2,LAG=2,XLDS=0" overrides entries of the generated shape
        k, v = kv.split("=")
        t[k] = int(v)
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(OUT, "arx_cfg2.hip")
    with open(src, "w") as f:
        f.write((sa.emit_half if HALF else sa.emit_split)(t))
    procs = []
    for k in (NAMES if os.environ.get("ABL_ONLY0", "0") != "1" else [0]):
        cmd = [sa._hipcc(), "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-Wno-uninitialized", "-ffp-contract=off", f"-I{sa.CSRC}", f"-DARX_ABL={k}", "-shared",
               "-no-hip-rt", src, f"-L{sa._torch_lib_dir()}", "-l:libamdhip64.so", "-o", os.path.join(OUT, f"arx_abl{k}.so")] + sys.argv[2:]
        procs.append(subprocess.Popen(cmd))
    assert all(p.wait() == 0 for p in procs)


def run():
    import torch

    from zuko_amd import _C
    from zuko_amd.flows import NSF
    from zuko_amd.nn import MaskedLinear
    from zuko_amd.ops import _ptr, _stream

    lb = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    flow = NSF(64, 0, transforms=2, bins=8, hidden_features=[256] * 3).to(dev)
    lazy = flow.transform.transforms[0]
    st = lazy.fused_state(dev)
    st.ready(1 << 20)  # (a geometry that was not built ahead of time is compiled here)
    assert st.static is not None and st.static[0].meta.get("split") and (st.static[0].meta["WAVES"], st.static[0].meta.get("CH")) == sa.split_geometry()
    st.refresh([m for m in lazy.hyper if isinstance(m, MaskedLinear)])
    N = 1 << lb
    x = torch.randn(N, 64, device=dev)
    y, ladj = torch.empty(N, 64, device=dev), torch.empty(N, device=dev)
    p = st.plan
    only = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
    import zuko_amd

    zuko_amd.set_matmul_precision("f16x2" if HALF else "bf16x3")
    with torch.no_grad():  # the product kernel's result on the same rows: what variant 0 of a probe build has to reproduce bit for bit
        y_ref, l_ref = flow.transform.transforms[0]().call_and_ladj(x)
    assert st._half_serves(y) == HALF
    for k, name in NAMES.items():
        so = os.path.join(OUT, f"arx_abl{k}.so")
        if not os.path.exists(so) or (only is not None and k not in only):
            continue
        lib = ctypes.CDLL(so)
        launcher = ctypes.cast(lib.zk_ars_launch, ctypes.c_void_p)
        if HALF:
            a = st._half_args(N=N, DIN=64, x=_ptr(x), ldx=64, y=_ptr(y), ldy=64, ladj=_ptr(ladj), accumulate=0)
            a.launcher = launcher.value
        else:
            a = _C.args("zk_ar_args_v1", launcher=launcher, rev=0, uni_kind=p.layout.kind, N=N, D=64, DIN=64, x=_ptr(x), ldx=64, y=_ptr(y), ldy=64, ladj=_ptr(ladj), accumulate=0,
                        wstream=_ptr(st.fine_stream), bias=_ptr(st.bias), bias_floats=st.bias_floats, featmap=_ptr(st.featmap), n_layers=p.n_layers, n_groups=p.n_groups,
                        n_chunks=st.fine_n_chunks, act=1, bound=st.bound, slope=st.slope)
        fn = lambda: _C.check(_C.lib().zk_ar_forward_static(a, _stream()), "zk_ar_forward_static")
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        same = "" if k else f"  bitwise == product kernel: y {torch.equal(y, y_ref)} ladj {torch.equal(ladj, l_ref)}  max|dy| {(y - y_ref).abs().max().item():.2e}"
        print(f"ARX_ABL={k} {name:34s} {e0.elapsed_time(e1) / 10:7.3f} ms per launch (2^{lb} rows){same}", flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
