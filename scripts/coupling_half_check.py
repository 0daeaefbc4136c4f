"""GPU: RealNVP cfg4 (256 features, 16 transforms, hidden [512] * 3) log_prob on the three-part (bf16 x 3) and the two-part (f16 x 2) coupling kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zuko_amd
import zuko_amd.flows as ZF
from oracle import zuko_oracle as O

dev = torch.device("cuda:0")
torch.set_num_threads(16)
torch.manual_seed(0)
flow = ZF.RealNVP(features=256, context=0, transforms=16, hidden_features=[512] * 3)
sd = {k: v for k, v in flow.state_dict().items() if v is not None}
spec = O.spec_from_state_dict(sd, "coupling", O.UNI_AFFINE, 256)
spec64 = O.spec_from_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, "coupling", O.UNI_AFFINE, 256)
flow = flow.to(dev)
xs = torch.randn(2048, 256, generator=torch.Generator().manual_seed(1))
with torch.no_grad():
    lp32 = O.flow_log_prob(spec, xs)
    lp64 = O.flow_log_prob(spec64, xs.double())
print(f"float32 reference vs float64: log_prob rel {((lp32 - lp64).abs() / lp64.abs()).max():.2e}")
x = torch.randn(1 << 19, 256, device=dev)
for mode in ("bf16x3", "f16x2", "bf16x3", "f16x2"):
    zuko_amd.set_matmul_precision(mode)
    with torch.no_grad():
        lp = flow().log_prob(xs.to(dev)).cpu()
        used = flow.transform.transforms[0].fused_state(dev).half_ok and mode == "f16x2"
        for _ in range(2):
            flow().log_prob(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            flow().log_prob(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 4 * 1e3
    print(f"{mode:7s} (two-part kernel: {used}) log_prob rel vs f64 {((lp - lp64).abs() / lp64.abs()).max():.2e}, vs f32 reference {((lp - lp32).abs() / lp32.abs()).max():.2e} | 2^19 rows: {ms:.2f} ms = {(1 << 19) / ms / 1e3:.2f} M samples/s", flush=True)
