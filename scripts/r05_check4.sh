#!/bin/bash
mkdir -p gpurun_out/r05
timeout 300 python -m pytest tests/test_gpu_flows.py tests/test_gpu_bins.py -x -q -m gpu -k "wavefront_form or golden or polynomial or generic_split or without_a_generated or bin_index" 2>&1 | tail -4 | tee gpurun_out/r05/check4_pytest.txt
timeout 200 python - <<'PY' 2>&1 | tail -8 | tee gpurun_out/r05/check4_side.txt
import os, time, torch
import zuko_amd.flows as F
dev = torch.device("cuda:0")
for ctor in ("SOSPF", "BPF"):
    torch.manual_seed(0)
    flow = getattr(F, ctor)(64, 0, transforms=3, hidden_features=[256] * 3).to(dev)
    for lb in (14, 18):
        with torch.no_grad():
            tr = flow().transform
            x0 = 0.8 * torch.randn(1 << lb, 64, device=dev)
            z = tr(x0)
            res = {}
            for mode in ("wavefront", "reference_loop"):
                if mode == "reference_loop":
                    if lb > 14: continue
                    os.environ["ZUKO_AMD_FULL_SWEEPS"] = "1"
                xs = tr.inv(z); torch.cuda.synchronize()
                t0 = time.perf_counter(); xs = tr.inv(z); torch.cuda.synchronize()
                res[mode] = (time.perf_counter() - t0, xs)
                os.environ.pop("ZUKO_AMD_FULL_SWEEPS", None)
            print(ctor, f"2^{lb}", {k: round(v[0] * 1e3, 2) for k, v in res.items()}, "M samples/s", round((1 << lb) / res["wavefront"][0] / 1e6, 3),
                  "equal", torch.equal(res["wavefront"][1], res["reference_loop"][1]) if "reference_loop" in res else None, "round trip", (res["wavefront"][1] - x0).abs().max().item())
PY
