r"""Static-shape instantiations of the fused autoregressive kernel, generated per conditioner.

csrc/fused_ar_static_impl.h holds the kernel as a template over a `Shape` struct of constexpr tables: which 16 x 16 weight
tiles of every layer hold non-zero weights and where each of them sits in the weight stream (zuko_amd/fused.py:build_plan —
with the hidden units sorted by dependency count the masks of zuko/nn.py:270-295 are block triangular, so all of this is
fixed per (features, context, hidden widths, order, univariate map)).  This module turns a plan into that struct, compiles the
one-kernel translation unit with hipcc (gfx950, ~10-20 s) into zuko_amd/lib/ars/ars_<signature>.so and hands the launcher's
address to the library (zk_ar_forward_static).

* Ahead of time (`prebuild()`, run by zuko_amd/_build.py and __graft_entry__.build()): the conditioners of BASELINE.json's
  configurations and a few common shapes (PREBUILT below).  Their .so files travel with the tree.
* On first use (`lookup(..., rows)`): any other conditioner whose batch is large enough for the compile to pay
  (ZUKO_AMD_JIT_MIN_ROWS, default 2^15 rows; ZUKO_AMD_JIT=0 disables it).  Without hipcc, or below the threshold, the generic
  tile-skipping kernel (widths <= 256) or the layer-wise kernels (wider) run instead — same results.

The kernels are bit-identical to the generic kernel on the same plan (tests/test_gpu_flows.py), so none of this changes a
number; it removes the run-time tile tests, makes every LDS wait partial and lifts the generic kernel's width limit of 256
(one wavefront per SIMD, 32 + 32 activation tiles, for widths up to 512).
"""

from __future__ import annotations

import ctypes
import fcntl
import hashlib
import json
import os
import shutil
import subprocess
import sys
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ARS_DIR = os.path.join(HERE, "lib", "ars")
ARS_ABI = 3  # == ARS_ABI of csrc/zk_ar_common.h
UNI_TYPES = {0: "zk::UniAffine", 1: "zk::UniRqs8", 2: "zk::UniRqs4", 4: "zk::UniCircRqs8"}  # (16 bins: 12 accumulator tiles per group do not fit the double-buffered last layer)
_HEADERS = ("fused_ar_static_impl.h", "zk_ar_common.h", "zk_univariate.h", "zk_common.h")


def _hipcc() -> str | None:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    return None


def _header_digest() -> str:
    h = hashlib.sha256(str(ARS_ABI).encode())
    for name in _HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:12]


# --------------------------------------------------------------------------------------------------------------------
# plan -> tables
# --------------------------------------------------------------------------------------------------------------------


def tables(plan, uni_kind: int, act: int = 1) -> dict | None:
    """The Shape tables of one plan (fused.ArPlan with its per-tile stream), or None when no static kernel can be built."""
    if plan.fine_gather is None or uni_kind not in UNI_TYPES or act not in range(8):
        return None
    L = plan.n_layers
    NH = L - 1
    if NH < 1 or NH > 6:
        return None
    tm = plan.fine_tilemask
    n_otg = tm.shape[1]
    widths = list(plan.widths)
    HT = [-(-w // 16) for w in widths]
    din4 = -(-plan.din // 4) * 4
    NIT = -(-plan.din // 16)
    TMAX = 4 * -(-max([NIT] + HT) // 4)
    if TMAX > 32:
        return None
    S_OTG, S_IT, S_MASK, NS = [], [], [], []
    for l in range(NH):
        n = 0
        for otg in range(n_otg):
            for it in range(tm.shape[2]):
                m = int(tm[l, otg, it])
                if m:
                    S_OTG.append(otg)
                    S_IT.append(it)
                    S_MASK.append(m)
                    n += 1
        NS.append(n)
    G_IT, GOFF = [], [0]
    for g in range(plan.n_groups):
        bits = int(plan.skip[NH * n_otg + g])
        G_IT += [it for it in range(32) if bits >> it & 1]
        GOFF.append(len(G_IT))
    waves = 8 if TMAX <= 16 else 4
    D = plan.features
    xs = ((D + 3) // 4) * 4 + 4
    bias_floats = plan.bias_off[-1] + len(plan.bias_gather[-1])
    base = (3 * 24 * 256 + bias_floats + 1024 + 256) * 4
    xlds = int(D % 4 == 0 and base + waves * 16 * xs * 4 <= 160 * 1024)
    if base + (waves * 16 * xs * 4 if xlds else 0) > 160 * 1024:
        return None
    if plan.n_groups * plan.layout.fpl * 4 > 1024:
        return None
    return {
        "uni": int(uni_kind), "ACT": int(act), "D": int(D), "DIN": int(din4), "NIT": int(NIT), "NH": int(NH), "HT": HT, "TMAX": int(TMAX), "NG": int(plan.n_groups),
        "NCHUNK": int(plan.fine_n_chunks), "BIAS_STRIDE": int(plan.max_width), "NS": NS, "S_OTG": S_OTG, "S_IT": S_IT, "S_MASK": S_MASK,
        "BASE": [int(b) for b in plan.fine_layer_block0[:NH]], "LAST_BASE": int(plan.fine_layer_block0[NH]), "GOFF": GOFF, "G_IT": G_IT,
        "WAVES": waves, "XLDS": xlds, "TRAIN_OK": int(act == 1 and NH <= 3 and waves == 8 and all(w % 16 == 0 for w in widths)),
    }


def _split(t: dict):
    """(core, l0): everything but the first layer's input-tile list, and that list (a descending feature order changes only it)."""
    n0 = t["NS"][0]
    core = {k: v for k, v in t.items() if k != "S_IT"}
    core["S_IT_rest"] = t["S_IT"][n0:]
    return core, t["S_IT"][:n0]


def _digest(obj) -> str:
    return hashlib.sha256(json.dumps(obj, sort_keys=True).encode()).hexdigest()[:16]


def _arr(name: str, ctype: str, vals) -> str:
    vals = list(vals) or [0]
    return f"  static constexpr {ctype} {name}[{len(vals)}] = {{{', '.join(str(int(v)) for v in vals)}}};"


def emit(t: dict, alt: list | None) -> str:
    """C++ source of the translation unit for tables `t`; `alt`: the first layer's input tiles under the alternative order."""
    n0 = t["NS"][0]
    s_alt = (alt if alt is not None else t["S_IT"][:n0]) + t["S_IT"][n0:]
    soff = [0]
    for n in t["NS"]:
        soff.append(soff[-1] + n)
    lines = [
        "// generated by zuko_amd/static_ar.py — do not edit",
        '#include "fused_ar_static_impl.h"',
        "namespace {",
        "struct Shape {",
        f"  static constexpr int D = {t['D']}, DIN = {t['DIN']}, NIT = {t['NIT']}, NH = {t['NH']}, TMAX = {t['TMAX']}, NG = {t['NG']}, NCHUNK = {t['NCHUNK']};",
        f"  static constexpr int BIAS_STRIDE = {t['BIAS_STRIDE']}, LAST_BASE = {t['LAST_BASE']}, WAVES = {t['WAVES']}, ACT = {t['ACT']};",
        f"  static constexpr bool XLDS = {'true' if t['XLDS'] else 'false'}, HAS_ALT = {'true' if alt is not None else 'false'}, TRAIN_OK = {'true' if t['TRAIN_OK'] else 'false'};",
        _arr("HT", "int", t["HT"]), _arr("NS", "int", t["NS"]), _arr("SOFF", "int", soff), _arr("BASE", "int", t["BASE"]),
        _arr("S_OTG", "unsigned char", t["S_OTG"]), _arr("S_IT", "unsigned char", t["S_IT"]), _arr("S_ALT", "unsigned char", s_alt),
        _arr("S_MASK", "unsigned char", t["S_MASK"]), _arr("GOFF", "int", t["GOFF"]), _arr("G_IT", "unsigned char", t["G_IT"]),
        "};",
        "}  // namespace",
        f'extern "C" int zk_ars_launch(const zk::ArArgs* a, int abi, int args_bytes, int train, void* stream) {{ return zk::ars_launch<Shape, {UNI_TYPES[t["uni"]]}>(a, abi, args_bytes, train, stream); }}',
        "",
    ]
    return "\n".join(lines)


# --------------------------------------------------------------------------------------------------------------------
# compile / load
# --------------------------------------------------------------------------------------------------------------------


class StaticKernel:
    def __init__(self, so: str, meta: dict) -> None:
        self.so, self.meta = so, meta
        self.cdll = ctypes.CDLL(so)
        fn = self.cdll.zk_ars_launch
        self.launcher = ctypes.cast(fn, ctypes.c_void_p)


_LOCK = threading.Lock()
_LOADED: dict[str, StaticKernel] = {}
_INDEX: dict | None = None  # core digest -> list of meta dicts found on disk


def _scan() -> dict:
    global _INDEX
    idx: dict = {}
    stamp = _header_digest()
    if os.path.isdir(ARS_DIR):
        for name in sorted(os.listdir(ARS_DIR)):
            if name.endswith(".json"):
                try:
                    with open(os.path.join(ARS_DIR, name)) as f:
                        meta = json.load(f)
                except (OSError, ValueError):
                    continue
                if meta.get("headers") == stamp and os.path.exists(os.path.join(ARS_DIR, meta["so"])):
                    idx.setdefault(meta["core"], []).append(meta)
    _INDEX = idx
    return idx


def _torch_lib_dir() -> str:
    import importlib.util

    spec = importlib.util.find_spec("torch")
    return os.path.join(os.path.dirname(spec.origin), "lib")


def compile_kernel(t: dict, alt: list | None, verbose: bool = False) -> dict | None:
    """Build lib/ars/ars_<sig>.so for tables `t` (no-op when it is there and current); returns its meta or None (no hipcc / failure)."""
    hipcc = _hipcc()
    if hipcc is None:
        return None
    core, l0 = _split(t)
    stamp = _header_digest()
    sig = _digest({"core": core, "l0": l0, "alt": alt, "headers": stamp})
    os.makedirs(ARS_DIR, exist_ok=True)
    so, meta_path = f"ars_{sig}.so", os.path.join(ARS_DIR, f"ars_{sig}.json")
    meta = {"so": so, "core": _digest(core), "l0": l0, "alt": alt, "headers": stamp, "uni": t["uni"], "ACT": t["ACT"], "D": t["D"], "DIN": t["DIN"], "HT": t["HT"], "WAVES": t["WAVES"],
            "TRAIN_OK": t["TRAIN_OK"], "XLDS": t["XLDS"]}
    with open(os.path.join(ARS_DIR, f".lock_{sig}"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)  # (several ranks / test workers may want the same kernel at once)
        if os.path.exists(os.path.join(ARS_DIR, so)) and os.path.exists(meta_path):
            return meta
        src = os.path.join(ARS_DIR, f"ars_{sig}.hip")
        with open(src, "w") as f:
            f.write(emit(t, alt))
        tmp = os.path.join(ARS_DIR, f".{so}.{os.getpid()}")
        cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-Wno-uninitialized", "-ffp-contract=off", f"-I{CSRC}", "-shared", "-no-hip-rt",
               src, f"-L{_torch_lib_dir()}", "-l:libamdhip64.so", "-o", tmp]
        if verbose:
            print("[zuko_amd static_ar]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            sys.stderr.write(f"[zuko_amd static_ar] hipcc failed for {src}:\n{r.stdout[-2000:]}\n")
            try:
                os.remove(tmp)
            except OSError:
                pass
            return None
        os.replace(tmp, os.path.join(ARS_DIR, so))
        with open(meta_path, "w") as f:
            json.dump(meta, f)
    global _INDEX
    _INDEX = None
    return meta


def _load(meta: dict) -> StaticKernel:
    k = _LOADED.get(meta["so"])
    if k is None:
        k = StaticKernel(os.path.join(ARS_DIR, meta["so"]), meta)
        _LOADED[meta["so"]] = k
    return k


def jit_enabled() -> bool:
    return os.environ.get("ZUKO_AMD_JIT", "1") != "0" and _hipcc() is not None


def jit_min_rows() -> int:
    try:
        return int(os.environ.get("ZUKO_AMD_JIT_MIN_ROWS", str(1 << 15)))
    except ValueError:
        return 1 << 15


def lookup(plan, uni_kind: int, act: int, rows: int | None = None):
    """(StaticKernel, rev) for this plan, or None.  Kernels found on disk are used whatever the batch; a missing one is compiled
    when `rows` reaches the JIT threshold."""
    if os.environ.get("ZUKO_AMD_NO_STATIC_AR", "0") == "1":
        return None
    t = tables(plan, uni_kind, act)
    if t is None:
        return None
    core, l0 = _split(t)
    cd = _digest(core)
    with _LOCK:
        idx = _INDEX if _INDEX is not None else _scan()
        for meta in idx.get(cd, []):
            if meta["l0"] == l0:
                return _load(meta), 0
            if meta["alt"] is not None and meta["alt"] == l0:
                return _load(meta), 1
        if rows is not None and rows >= jit_min_rows() and jit_enabled():
            meta = compile_kernel(t, None, verbose=os.environ.get("ZUKO_AMD_JIT_VERBOSE", "0") == "1")
            if meta is not None:
                return _load(meta), 0
    return None


# --------------------------------------------------------------------------------------------------------------------
# ahead-of-time list
# --------------------------------------------------------------------------------------------------------------------

# (univariate, features, context, hidden widths, bins): the conditioners of BASELINE.json's configurations first
PREBUILT = [
    ("rqs", 64, 0, (256, 256, 256), 8),     # cfg2: NSF(64, T=8, K=8, H=[256]^3)  — the headline
    ("affine", 64, 0, (256, 256, 256), 0),  # cfg3: MAF(64, T=8, H=[256]^3)
    ("rqs", 3, 5, (128, 128, 128), 8),      # cfg1: NSF(3, context 5, H=[128]^3)
    ("rqs", 32, 0, (512, 512), 8),          # widths beyond the generic kernel's 256 (one wavefront per SIMD)
    ("rqs", 32, 0, (256, 256), 8),
    ("affine", 16, 0, (128, 128), 0),
    # shapes the GPU tests exercise (widths that are not multiples of 16 / 64, a context that straddles a tile, one hidden layer, three
    # wide layers): built ahead so that `pytest -m gpu` on a fresh box compiles nothing
    ("rqs", 20, 3, (100, 72), 8),
    ("affine", 7, 2, (40,), 0),
    ("rqs", 24, 8, (384, 512, 320), 8),
    ("rqs", 16, 2, (64, 64), 8, "ELU"),
    ("affine", 12, 0, (48, 32), 0, "Tanh"),
]


def _plans_for(kind: str, features: int, context: int, hidden, bins: int, activation: str | None = None):
    """Plans of the ascending- and descending-order transform of such a flow (MAF / NSF alternate the two).  (The activation does not
    enter the plan: it is a field of the tables.)"""
    import torch

    from . import fused
    from .flows.autoregressive import MaskedAutoregressiveTransform
    from .nn import MaskedLinear
    from .transforms import MonotonicAffineTransform, MonotonicRQSTransform

    out = []
    for order in (torch.arange(features), torch.flipud(torch.arange(features))):
        if kind == "affine":
            t = MaskedAutoregressiveTransform(features, context, order=order, hidden_features=list(hidden), univariate=MonotonicAffineTransform, shapes=[(), ()])
            layout = fused.uni_layout("affine", 2)
        else:
            t = MaskedAutoregressiveTransform(features, context, order=order, hidden_features=list(hidden), univariate=MonotonicRQSTransform, shapes=[(bins,), (bins,), (bins - 1,)])
            layout = fused.uni_layout("rqs", 3 * bins - 1, bins)
        masks = [m.mask for m in t.hyper if isinstance(m, MaskedLinear)]
        wide = max(list(hidden) + [features + context]) > fused.MAX_WIDTH
        out.append((fused.build_plan(masks, features, layout, max_width=fused.MAX_WIDTH_WIDE if wide else fused.MAX_WIDTH), layout))
    return out


def prebuild(verbose: bool = True, jobs: int = 4) -> list[str]:
    """Compile every PREBUILT kernel that is missing or stale; returns the .so names."""
    from concurrent.futures import ThreadPoolExecutor

    # kernels built against another version of the template / headers can never be selected again: drop them
    stamp = _header_digest()
    if os.path.isdir(ARS_DIR):
        for name in os.listdir(ARS_DIR):
            if name.endswith(".json"):
                try:
                    with open(os.path.join(ARS_DIR, name)) as f:
                        stale = json.load(f).get("headers") != stamp
                except (OSError, ValueError):
                    stale = True
                if stale:
                    for ext in (".json", ".so", ".hip"):
                        try:
                            os.remove(os.path.join(ARS_DIR, name[: -len(".json")] + ext))
                        except OSError:
                            pass
            elif name.startswith(".lock_"):
                try:
                    os.remove(os.path.join(ARS_DIR, name))
                except OSError:
                    pass
    work = []
    for entry in PREBUILT:
        kind, features, context, hidden, bins = entry[:5]
        import torch

        from .nn import _act_code

        act = _act_code(getattr(torch.nn, entry[5])()) if len(entry) > 5 else 1
        (pa, layout), (pd, _) = _plans_for(kind, features, context, hidden, bins)
        ta, td = tables(pa, layout.kind, act), tables(pd, layout.kind, act)
        if ta is None or td is None:
            raise RuntimeError(f"zuko_amd.static_ar: no static kernel for the prebuilt shape {entry}")
        (ca, la), (cdesc, ld) = _split(ta), _split(td)
        if ca == cdesc:
            work.append((ta, None if la == ld else ld))
        else:
            work += [(ta, None), (td, None)]
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        metas = list(ex.map(lambda w: compile_kernel(w[0], w[1], verbose), work))
    if any(m is None for m in metas):
        raise RuntimeError("zuko_amd.static_ar: a prebuilt static kernel failed to compile")
    return [m["so"] for m in metas]


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(HERE))
    import zuko_amd.static_ar as me  # (as a package module: the relative imports above need it)

    print("\n".join(me.prebuild()))
