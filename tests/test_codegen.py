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


def _isa_of_generated(cfgs):
    """ISA of the static-shape kernels zuko_amd/static_ar.py generates for `cfgs` (cached per source + header hash)."""
    from concurrent.futures import ThreadPoolExecutor

    from zuko_amd import static_ar

    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    jobs = []
    for cfg in cfgs:
        (pa, lay, _), (pd, _, _) = static_ar._plans_for(*cfg)
        ta, td = static_ar.tables(pa, lay.kind), static_ar.tables(pd, lay.kind)
        (ca, la), (cd, ld) = static_ar._split(ta), static_ar._split(td)
        src = static_ar.emit(ta, ld if (ca == cd and la != ld) else None)
        h = hashlib.sha256((src + static_ar._header_digest()).encode()).hexdigest()[:16]
        out = os.path.join(ROOT, "zuko_amd", "lib", f"ars_isa.{h}.s")
        jobs.append((src, out, ta))

    def build(job):
        src, out, _ = job
        if not os.path.exists(out):
            os.makedirs(os.path.dirname(out), exist_ok=True)
            hip = out[:-2] + ".hip"
            with open(hip, "w") as f:
                f.write(src)
            subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-result", "-Wno-uninitialized", "-ffp-contract=off", f"-I{CSRC}", "--cuda-device-only", "-S",
                            hip, "-o", out], check=True, stderr=subprocess.DEVNULL)
        return open(out).read()

    with ThreadPoolExecutor(max_workers=4) as ex:
        return list(zip(ex.map(build, jobs), (j[2] for j in jobs)))


def test_static_kernel_raw_lds_reads_are_never_touched_before_their_wait():
    """The generated static-shape kernels (csrc/fused_ar_static_impl.h instantiated on the tables of zuko_amd/static_ar.py) issue
    their weight-tile reads from inline assembly and make them usable through an `s_waitcnt lgkmcnt(n)` they place themselves: a
    register copy inserted between the two by the allocator would read data that has not arrived.  Checked on the ISA of the
    headline conditioner (cfg2, spline), cfg3 (affine map) and the 512-wide one-wavefront-per-SIMD instantiation; also: every
    streamed tile is read exactly once and multiplied by four k-steps, and no kernel spills."""
    import re

    cfgs = [("rqs", 64, 0, (256, 256, 256), 8), ("affine", 64, 0, (256, 256, 256), 0), ("rqs", 32, 0, (512, 512), 8)]
    want = {0: [(1130, 4520)] * 2, 1: [(386, 1544)] * 2}  # density + conditioner-only (training) instantiation (1176 - 46 / 432 - 46 all-zero tiles dropped)
    for i, (s, t) in enumerate(_isa_of_generated(cfgs)):
        stats = _check_raw_reads(s, "_ZN2zk10ars_kernel")
        tiles = sum(bin(m).count("1") for m in t["S_MASK"]) + t["GOFF"][-1] * {0: 1, 1: 6}[t["uni"]]
        assert stats and all(st == (tiles, 4 * tiles) for st in stats), (cfgs[i], stats, tiles)
        if i in want:
            assert stats == want[i]
        else:
            assert len(stats) == 1  # (no training instantiation for the wide kernel)
        for name in re.findall(r"\.amdhsa_kernel (_ZN2zk10ars_kernel\S+)", s):
            k = s.index(".amdhsa_kernel " + name)
            assert ".amdhsa_private_segment_fixed_size 0" in s[k : s.index(".end_amdhsa_kernel", k)]


@pytest.mark.parametrize("tu,prefix,extra,n_kernels", [
    ("inc_inverse.hip", "_ZN2zk18inc_inverse_kernel", ("-DZK_INC_FAST_BUILD",), 4),  # (f32 pulls and, round 6, the HALF instantiations: blocks of two images, three matrix instructions)
    ("fused_coupling.hip", "_ZN2zk22coupling_kernel_static", ("-mllvm", "-pragma-unroll-threshold=1000000"), 1),
])
def test_raw_lds_reads_of_the_other_ring_kernels(tu, prefix, extra, n_kernels):
    """Same guard for the incremental inverse (the two benchmark instantiations) and the static coupling kernel."""
    stats = _check_raw_reads(_isa_of(tu, extra), prefix)
    assert len(stats) == n_kernels and all(r > 0 and m >= (1.2 if n_kernels == 4 else 4) * r for r, m in stats)
    if n_kernels == 4:
        assert sum(m >= 4 * r for r, m in stats) >= 2  # (the f32-pull instantiations: four matrix instructions per image, as before)
