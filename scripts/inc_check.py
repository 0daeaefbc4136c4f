"""Incremental inverse vs partial sweeps vs oracle (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd.flows import MAF, NSF
from oracle import zuko_oracle as O
dev = torch.device("cuda:0")
for name, make, uni in (("NSF cfg2", lambda: NSF(64, 0, transforms=8, bins=8, hidden_features=[256] * 3), O.uni_rqs(8)),
                        ("MAF cfg3", lambda: MAF(64, 0, transforms=8, hidden_features=[256] * 3), O.UNI_AFFINE)):
    torch.manual_seed(0)
    flow = make()
    sd = {k: v.clone() for k, v in flow.state_dict().items() if v is not None}
    spec = O.spec_from_state_dict(sd, "ar", uni, 64)
    flow = flow.to(dev)
    z = torch.randn(1000, 64, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        t0 = flow.transform.transforms[0]
        st = t0.incremental_state(dev)
        print(name, "incremental plan:", None if st is None else (st.plan.n_groups, st.plan.n_blocks))
        x_inc = flow().transform.inv(z.to(dev)).cpu()
        os.environ["ZUKO_AMD_NO_INCREMENTAL"] = "1"
        x_par = flow().transform.inv(z.to(dev)).cpu()
        os.environ.pop("ZUKO_AMD_NO_INCREMENTAL")
        x_or = O.flow_inverse(spec, z)
    print(f"  max |inc - partial| {float((x_inc - x_par).abs().max()):.3e}   max |inc - oracle| {float((x_inc - x_or).abs().max()):.3e}   max |partial - oracle| {float((x_par - x_or).abs().max()):.3e}")
    for logB in (18, 20):
        B = 1 << logB
        zz = torch.randn(B, 64, device=dev)
        for mode in ("incremental", "partial"):
            if mode == "partial":
                os.environ["ZUKO_AMD_NO_INCREMENTAL"] = "1"
            with torch.no_grad():
                flow().transform.inv(zz); torch.cuda.synchronize()
                t0 = time.perf_counter()
                reps = 3 if mode == "incremental" else 1
                for _ in range(reps):
                    flow().transform.inv(zz)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / reps
            os.environ.pop("ZUKO_AMD_NO_INCREMENTAL", None)
            print(f"  batch 2^{logB} {mode:12s}: {dt*1e3:9.2f} ms  {B/dt/1e6:7.2f} M samples/s")
