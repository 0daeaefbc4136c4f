"""Test configuration: `gpu` marker, golden-fixture loader, flow registry shared by CPU and GPU tests."""

from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a MI355X (run with -m gpu on the GPU box)")
    # the oracle runs on the host: PyTorch-CPU at the GPU box's default of 128-256 threads is pathologically slow on these sizes (bench.py measured
    # 197 s -> 3 s for one float64 autograd pass of the oracle when capped at 16 threads, round 5); a no-op on smaller machines
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))


def pytest_sessionstart(session):
    """The shared library is a build product (git-ignored): bring it up to date before anything imports zuko_amd.
    A no-op when the hash stamps match (the prebuilt library that ships to the GPU box is used as is)."""
    import importlib.util
    import shutil

    lib = os.path.join(ROOT, "zuko_amd", "lib", "libzuko_amd.so")
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        return  # nothing to build with: importing zuko_amd will say so if the library is missing
    spec = importlib.util.spec_from_file_location("_zuko_amd_build", os.path.join(ROOT, "zuko_amd", "_build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        mod.build(verbose=not os.path.exists(lib))
    except Exception:
        if not os.path.exists(lib):
            raise


def pytest_sessionfinish(session, exitstatus):
    """Measured error ratios of every parity comparison -> gpurun_out/parity_report.json (tests/parity.py)."""
    try:
        import parity

        parity.dump_report()
    except Exception:
        pass


def golden(name: str) -> dict:
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def T(a, device="cpu"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def sd_hash(sd: dict) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        v = sd[k]
        if v is None:
            continue
        h.update(k.encode())
        h.update(str(tuple(v.shape)).encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def flow_registry():
    """name -> (zuko_amd constructor, kwargs, seed, oracle kind, oracle univariate, extra spec kwargs);
    identical configurations to tests/golden/make_golden.py:FLOWS."""
    import zuko_amd.flows as F
    from oracle import zuko_oracle as O

    return {
        "nsf_cfg1": (F.NSF, dict(features=3, context=5, transforms=3, bins=8, hidden_features=[128] * 3), 0, "ar", O.uni_rqs(8), {}),
        "nsf_cfg2": (F.NSF, dict(features=64, context=0, transforms=8, bins=8, hidden_features=[256] * 3), 0, "ar", O.uni_rqs(8), {}),
        "maf_cfg3": (F.MAF, dict(features=64, context=0, transforms=8, hidden_features=[256] * 3), 0, "ar", O.UNI_AFFINE, {}),
        "realnvp_cfg4": (F.RealNVP, dict(features=256, context=0, transforms=16, hidden_features=[512] * 3), 0, "coupling", O.UNI_AFFINE, {}),
        "maf_doc": (F.MAF, dict(features=3, context=4, transforms=3), 0, "ar", O.UNI_AFFINE, {}),
        "nsf_p2": (F.NSF, dict(features=6, context=2, transforms=2, bins=4, passes=2, hidden_features=[32, 32]), 3, "ar", O.uni_rqs(4), dict(passes=2)),
        "nice_small": (F.NICE, dict(features=5, context=3, transforms=3, hidden_features=[32, 32]), 4, "coupling", O.UNI_AFFINE, {}),
        "sospf_small": (F.SOSPF, dict(features=4, context=2, transforms=2, hidden_features=[32, 32]), 5, "ar", O.uni_sos(), dict(softclip=11.0)),
        "maf_res": (F.MAF, dict(features=5, context=2, transforms=2, hidden_features=[24, 32, 32], residual=True), 7, "ar", O.UNI_AFFINE, {}),
        "ncsf_small": (F.NCSF, dict(features=3, context=2, transforms=2, hidden_features=[16, 16]), 8, "ar", O.uni_crqs(8), {}),
        "bpf_small": (F.BPF, dict(features=4, context=2, transforms=2, hidden_features=[32, 32]), 6, "ar", O.uni_bpf(), {}),
    }


def build_flow(name: str):
    """Construct the zuko_amd flow on CPU with the golden seed; returns (flow, entry)."""
    entry = flow_registry()[name]
    ctor, kw, seed = entry[0], entry[1], entry[2]
    torch.manual_seed(seed)
    return ctor(**kw), entry


def oracle_spec(flow, entry):
    from oracle import zuko_oracle as O

    sd = {k: v.detach().cpu() for k, v in flow.state_dict().items() if v is not None}
    return O.spec_from_state_dict(sd, entry[3], entry[4], entry[1]["features"], **entry[5])


@pytest.fixture(scope="session")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture
def matmul():
    """matmul("bf16x3" | "f16x2"): zuko_amd.set_matmul_precision for the duration of a test (the previous mode is restored)."""
    import zuko_amd

    keep = zuko_amd.matmul_precision()
    yield zuko_amd.set_matmul_precision
    zuko_amd.set_matmul_precision(keep)
