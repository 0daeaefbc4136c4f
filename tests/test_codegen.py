"""Static guard on the code the compiler emits for the headline kernel (no GPU needed: hipcc cross-compiles).

The fused autoregressive kernel keeps 6 accumulator tiles live across 16 wave-uniform skip branches; whether the
register allocator reconciles the two sides of those branches with copies is decided by heuristics that flip on
unrelated source edits (measured: +420 v_mov in the last-layer loop = +6 % kernel time, bit-identical results).  The
test compiles the translation unit to ISA (cached per source hash under zuko_amd/lib/) and bounds the instruction mix
of ar_kernel<UniRqs<8>, forward, Ring24x3, LDS-staged>, the instantiation bench.py measures."""

import collections
import hashlib
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zuko_amd", "csrc")
HEADLINE = "_ZN2zk9ar_kernelINS_6UniRqsILi8ELb0EEELb0ENS_5RingTILi24ELi3EEELb1ELb0EEEvNS_6ArArgsE"


def _isa() -> str:
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    h = hashlib.sha256()
    for name in ("fused_ar.hip", "zk_univariate.h", "zk_common.h"):
        h.update(open(os.path.join(CSRC, name), "rb").read())
    out = os.path.join(ROOT, "zuko_amd", "lib", f"fused_ar.{h.hexdigest()[:16]}.s")
    if not os.path.exists(out):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-result", "-ffp-contract=off", "--cuda-device-only", "-S",
                        os.path.join(CSRC, "fused_ar.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def test_headline_kernel_instruction_mix():
    s = _isa()
    i = s.index(HEADLINE + ":")
    j = s.index(".Lfunc_end", i)
    ops = collections.Counter(l.split()[0] for l in (x.strip() for x in s[i:j].split("\n")) if l and not l.startswith((";", "//", ".")))
    mfma = sum(n for k, n in ops.items() if "mfma" in k)
    vmov = sum(n for k, n in ops.items() if k.startswith("v_mov"))
    total = sum(ops.values())
    print(f"headline kernel: {total} instructions, {mfma} MFMA, {vmov} v_mov, {ops['v_readlane_b32']} v_readlane, {ops['v_writelane_b32']} v_writelane")
    assert mfma == 1408  # 64 hidden blocks x 16 + 16 last-layer blocks x 24
    assert vmov <= 600, f"{vmov} v_mov: the accumulators are being copied around the skip branches again"
    assert total <= 24300
    k = s.index(".amdhsa_kernel " + HEADLINE)
    desc = s[k : s.index(".end_amdhsa_kernel", k)]
    assert ".amdhsa_private_segment_fixed_size 0" in desc, "scratch (VGPR spill) in the headline kernel"
