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
    for name in ("fused_ar.hip", "zk_ar_common.h", "zk_univariate.h", "zk_common.h"):
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


def _isa_of(tu: str, extra=()) -> str:
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    h = hashlib.sha256(" ".join(extra).encode())
    for name in (tu, "zk_ar_common.h", "zk_univariate.h", "zk_common.h"):
        h.update(open(os.path.join(CSRC, name), "rb").read())
    out = os.path.join(ROOT, "zuko_amd", "lib", f"{tu.split('.')[0]}.{h.hexdigest()[:16]}.s")
    if not os.path.exists(out):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-result", "-ffp-contract=off", *extra, "--cuda-device-only", "-S",
                        os.path.join(CSRC, tu), "-o", out], check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def _check_raw_reads(s: str, prefix: str):
    """For every kernel whose mangled name starts with `prefix`: between an inline-assembly `ds_read_b128` and an
    `s_waitcnt lgkmcnt(n)` that covers it (LDS operations of a wave complete in order: a wait lgkmcnt(n) covers every read
    except the youngest n) no instruction may mention the destination registers.  Returns [(asm reads, MFMAs)] per kernel."""
    import re

    def regs(t):
        out = set()
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", t):
            if m.group(1):
                out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
            else:
                out.add(int(m.group(3)))
        return out

    stats = []
    for k in [m.start() for m in re.finditer(r"^" + prefix + r"[^\n]*:", s, flags=re.M)]:
        body = s[k : s.index("s_endpgm", k)].split("\n")
        pending, in_asm, n_reads, mfma = [], False, 0, 0
        for line in body:
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t[0] in ";.":
                continue
            mfma += "v_mfma" in t
            m = re.match(r"ds_read_b128 (v\[\d+:\d+\]), ", t)
            if m and in_asm:
                pending.append(regs(m.group(1)))
                n_reads += 1
                continue
            w = re.match(r"s_waitcnt .*lgkmcnt\((\d+)\)", t)
            if w:
                n = int(w.group(1))
                pending = pending[len(pending) - n :] if 0 < n < len(pending) else ([] if n == 0 else pending)
                continue
            used = regs(t)
            assert not any(used & r for r in pending), f"'{t}' touches a weight tile whose LDS read has not been waited for"
        stats.append((n_reads, mfma))
    return stats


def test_static_kernel_raw_lds_reads_are_never_touched_before_their_wait():
    """fused_ar_static.hip issues its weight-tile reads from inline assembly and makes them usable through an
    `s_waitcnt lgkmcnt(n)` it places itself: a register copy inserted between the two by the allocator would read data
    that has not arrived."""
    import re

    s = _isa_of("fused_ar_static.hip")
    stats = _check_raw_reads(s, "_ZN2zk16ar_static_kernel")
    # density + conditioner-only (training) instantiation for each of the two univariate maps; every streamed tile is read once and
    # multiplied by four k-steps (1176 - 46 / 432 - 46 all-zero tiles)
    assert len(stats) == 4 and sorted(set(stats)) == [(386, 1544), (1130, 4520)]
    for name in re.findall(r"\.amdhsa_kernel (_ZN2zk16ar_static_kernel\S+)", s):
        k = s.index(".amdhsa_kernel " + name)
        assert ".amdhsa_private_segment_fixed_size 0" in s[k : s.index(".end_amdhsa_kernel", k)]


@pytest.mark.parametrize("tu,prefix,extra,n_kernels", [
    ("inc_inverse.hip", "_ZN2zk18inc_inverse_kernel", ("-DZK_INC_FAST_BUILD",), 2),
    ("fused_coupling.hip", "_ZN2zk22coupling_kernel_static", ("-mllvm", "-pragma-unroll-threshold=1000000"), 1),
])
def test_raw_lds_reads_of_the_other_ring_kernels(tu, prefix, extra, n_kernels):
    """Same guard for the incremental inverse (the two benchmark instantiations) and the static coupling kernel."""
    stats = _check_raw_reads(_isa_of(tu, extra), prefix)
    assert len(stats) == n_kernels and all(r > 0 and m >= 4 * r for r, m in stats)
