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
  (ZUKO_AMD_JIT_MIN_ROWS, default 2^15 rows in one call or 8 x that in total over the calls; ZUKO_AMD_JIT=0 disables it).  Without hipcc, or below the threshold, the generic
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
import math
import os
import shutil
import subprocess
import sys
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ARS_DIR = os.path.join(HERE, "lib", "ars")  # kernels built ahead of time (prebuild()); also the JIT's directory unless ZUKO_AMD_CACHE_DIR is set
ARS_ABI = 9  # == ARS_ABI of csrc/zk_ar_common.h
UNI_TYPES = {0: "zk::UniAffine", 1: "zk::UniRqs8", 2: "zk::UniRqs4", 3: "zk::UniRqs16", 4: "zk::UniCircRqs8", 5: "zk::UniSos3x5", 6: "zk::UniBern17"}
# 16 bins: the twelve accumulator tiles of a feature group do not fit the f32-instruction template's double-buffered last layer (it would
# spill), but the operand-split template holds them (255 VGPRs, no scratch): that kind exists as a split kernel only, forward only
SPLIT_ONLY_KINDS = {3, 5, 6}
_HEADERS = ("fused_ar_static_impl.h", "fused_ar_split_impl.h", "zk_ar_common.h", "zk_univariate.h", "zk_univariate_bwd.h", "zk_common.h", "zk_half.h")


def _hipcc() -> str | None:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    return None


def _arch() -> str:
    from . import _build

    return _build.ARCH


def _jit_dir() -> str:
    """Where kernels compiled on first use go: ZUKO_AMD_CACHE_DIR when set (a read-only install of the package), else lib/ars."""
    d = os.environ.get("ZUKO_AMD_CACHE_DIR")
    return os.path.join(d, "ars") if d else ARS_DIR


def _dirs() -> list[str]:
    d = _jit_dir()
    return [ARS_DIR] if d == ARS_DIR else [d, ARS_DIR]


def _find(name: str) -> str | None:
    for d in _dirs():
        path = os.path.join(d, name)
        if os.path.exists(path):
            return path
    return None


_WARNED: set = set()


def _warn_once(key: str, msg: str) -> None:
    if key not in _WARNED:
        _WARNED.add(key)
        sys.stderr.write(f"[zuko_amd static_ar] {msg}\n")


def _build_so(stem: str, source, meta: dict | None, verbose: bool, out_dir: str | None = None) -> str | None:
    """Compile the one-kernel translation unit `source()` into <dir>/<stem>.so (+ <stem>.json when `meta` is given); a no-op when it
    is there.  Returns the .so path, or None when there is no hipcc, the compile fails, or the directory cannot be written (read-only
    install, full disk): the caller then stays on the generic / layer-wise kernels — a failed JIT must never fail the user's call."""
    have = _find(stem + ".so")
    if have is not None and (meta is None or os.path.exists(have[: -len(".so")] + ".json")):
        return have
    hipcc = _hipcc()
    if hipcc is None:
        # (this used to be silent — the caller stays on the generic kernels: zk_ar_forward_split, correct but 1.1-1.6x slower)
        _warn_once("nohipcc", f"no hipcc on this machine: the static-shape kernel {stem} cannot be compiled; this conditioner runs on the generic operand-split kernel "
                              "(same results; 1.1-1.6x the generated kernel's time; conditioners wider than 256 run layer by layer).  Build it where a ROCm compiler is installed (zuko_amd.static_ar.prebuild, or one call at the "
                              "target batch size) and ship zuko_amd/lib/ars/ or ZUKO_AMD_CACHE_DIR with the package.")
        return None
    d = out_dir or _jit_dir()
    try:
        os.makedirs(d, exist_ok=True)
        so = os.path.join(d, stem + ".so")
        with open(os.path.join(d, f".lock_{stem}"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)  # (several ranks / test workers may want the same kernel at once)
            if os.path.exists(so) and (meta is None or os.path.exists(os.path.join(d, stem + ".json"))):
                return so
            src = os.path.join(d, stem + ".hip")
            with open(src, "w") as f:
                f.write(source())
            tmp = os.path.join(d, f".{stem}.so.{os.getpid()}")
            cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", f"--offload-arch={_arch()}", "-Wno-unused-result", "-Wno-uninitialized", "-ffp-contract=off", f"-I{CSRC}", "-shared", "-no-hip-rt",
                   *os.environ.get("ZUKO_AMD_STATIC_CXXFLAGS", "").split(),  # (probe / ablation builds into a ZUKO_AMD_CACHE_DIR of their own: NOT part of the kernel's signature)
                   src, f"-L{_torch_lib_dir()}", "-l:libamdhip64.so", "-o", tmp]
            if verbose:
                print("[zuko_amd static_ar]", " ".join(cmd), flush=True)
            elif out_dir is None:
                _warn_once("jit", f"compiling a static-shape kernel for this conditioner on first use ({stem}, 10-25 s; ZUKO_AMD_JIT=0 disables, ZUKO_AMD_CACHE_DIR redirects)")
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                sys.stderr.write(f"[zuko_amd static_ar] hipcc failed for {src}:\n{r.stdout[-2000:]}\n")
                try:
                    os.remove(tmp)
                except OSError:
                    pass
                return None
            os.replace(tmp, so)
            if meta is not None:
                with open(os.path.join(d, stem + ".json"), "w") as f:
                    json.dump(meta, f)
        return so
    except OSError as exc:
        _warn_once("oserror:" + d, f"cannot build kernels in {d} ({exc}); staying on the generic kernels (set ZUKO_AMD_CACHE_DIR to a writable directory)")
        return None


_HEADER_DIGEST: str | None = None


def _header_digest() -> str:
    """Digest of the kernel headers the generated units include (read once per process: it is consulted on every lookup of a kernel)."""
    global _HEADER_DIGEST
    if _HEADER_DIGEST is None:
        h = hashlib.sha256(str(ARS_ABI).encode())
        for name in _HEADERS:
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(f.read())
        _HEADER_DIGEST = h.hexdigest()[:12]
    return _HEADER_DIGEST


_HALF_DIGEST = None


def _half_digest() -> str:
    """_header_digest() extended by the two-part kernels' own header (an edit there rebuilds those kernels only)."""
    global _HALF_DIGEST
    if _HALF_DIGEST is None:
        h = hashlib.sha256(_header_digest().encode())
        with open(os.path.join(CSRC, "fused_ar_half_impl.h"), "rb") as f:
            h.update(f.read())
        _HALF_DIGEST = h.hexdigest()[:12]
    return _HALF_DIGEST


def _stamp_of(meta: dict) -> str:
    return _half_digest() if meta.get("half") else _header_digest()


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
        "WAVES": waves, "XLDS": xlds, "TRAIN_OK": int(act == 1 and NH <= 3 and waves == 8 and all(w % 16 == 0 for w in widths) and uni_kind != 3),  # (kinds 5, 6 — split kernels only: the conditioner-only training forward of SOSPF / BPF)
    }


def split_enabled() -> bool:
    """Whether the bf16x3 operand-split kernels may be used (ZUKO_AMD_EXACT_F32=1 keeps every product on the f32 matrix instruction)."""
    return os.environ.get("ZUKO_AMD_EXACT_F32", "0") != "1"


def split_geometry() -> tuple:
    """(wavefronts per workgroup, images per ring chunk) of the operand-split kernels: 8 x 24 = one workgroup of eight wavefronts per CU
    (two per SIMD) sharing one weight stream; ZUKO_AMD_SPLIT_GEOM=4x16 selects two independent workgroups per CU, one wavefront per
    SIMD each (twice the weight-stream traffic; measured 3-5 % slower on cfg2, profiles/r03/split_kernel.md)."""
    g = os.environ.get("ZUKO_AMD_SPLIT_GEOM", "8x24")
    try:
        w, c = (int(v) for v in g.split("x"))
    except ValueError:
        w, c = 8, 24
    return (w, c) if (w, c) in ((4, 16), (8, 24), (8, 48)) else (8, 24)


def split_tables(plan, uni_kind: int, act: int = 1):
    """Tables + gather indices of the OPERAND-SPLIT twin of a static-shape kernel (csrc/fused_ar_split_impl.h), or None.

    gfx950 has no xf32 / tf32 matrix instruction; its f32 one runs at 1/16 of the bf16 rate.  The split kernels write every f32
    operand as h + m + l with three bf16 numbers (8 + 8 + 8 significant bits: exact to 2^-25) and keep the six partial products down to
    2^-18 (hh, hm, mh, mm, hl, lh) on v_mfma_f32_16x16x32_bf16 with f32 accumulation: f32-grade results at 6/16 of the matrix time.
    A stream BLOCK is the 16 x 32 weight block (one out tile, one PAIR of in tiles) as three 1 KiB bf16 images (h, m, l); a lane's 8
    values of a k-group are [4 units of in tile 2 ip | the same 4 units of in tile 2 ip + 1], which is how the activations sit in the
    accumulator registers.  Returns (tables, gathers): gathers[l] int32 [blocks_l * 512] into W_l.flatten() (-1 = zero), every layer
    padded to whole 24-image chunks (8 blocks)."""
    t = tables(plan, uni_kind, act)
    if t is None:
        return None
    wide = t["WAVES"] != 8  # conditioners 257 - 512 wide: 32 activation tiles + their 16 operand pairs per wavefront = one wavefront per SIMD
    if wide and os.environ.get("ZUKO_AMD_SPLIT_WIDE", "1") == "0":
        return None
    cached = getattr(plan, "_split_cache", None)
    if cached is not None and cached[0] == (uni_kind, act, split_geometry()):
        return cached[1]
    NH, HT, NIT = t["NH"], t["HT"], t["NIT"]
    tm = plan.fine_tilemask
    n_otg, n_itile = tm.shape[1], tm.shape[2]
    waves, ch = (4, 24) if wide else split_geometry()
    B_OT, B_IP, NB, BASE, blocks_of = [], [], [], [], []
    cursor = 0  # in 1 KiB images; the layers follow each other without padding (a chunk boundary may fall anywhere, even inside a block)

    def pair_block(fg, t0, t1):
        idx = -np.ones((64, 8), dtype=np.int64)
        if t0 is not None:
            idx[:, :4] = fg[t0]
        if t1 is not None:
            idx[:, 4:] = fg[t1]
        return idx

    for l in range(NH):
        fg = plan.fine_gather[l].reshape(-1, 64, 4)
        pos, k = {}, 0
        for otg in range(n_otg):
            for it in range(n_itile):
                m = int(tm[l, otg, it])
                for b in range(4):
                    if m >> b & 1:
                        pos[(otg * 4 + b, it)] = k
                        k += 1
        n_in = NIT if l == 0 else HT[l - 1]
        blocks = []
        for ot in range(HT[l]):
            for ip in range(-(-n_in // 2)):
                t0, t1 = pos.get((ot, 2 * ip)), pos.get((ot, 2 * ip + 1))
                if t0 is None and t1 is None:
                    continue
                blocks.append(pair_block(fg, t0, t1))
                B_OT.append(ot), B_IP.append(ip)
        NB.append(len(blocks))
        BASE.append(cursor)
        cursor += 3 * len(blocks)
        blocks_of.append(blocks)
    # last layer: group g, kept in-tile it, tile t of the group  ->  group g, kept in-pair ip, tile t
    nt = plan.layout.nt
    fg = plan.fine_gather[NH].reshape(-1, 64, 4)
    G_IP, GOFFP, blocks, k = [], [0], [], 0
    for g in range(t["NG"]):
        its = t["G_IT"][t["GOFF"][g] : t["GOFF"][g + 1]]
        at = {it: k + i * nt for i, it in enumerate(its)}
        k += len(its) * nt
        for ip in sorted({it // 2 for it in its}):
            for b in range(nt):
                t0, t1 = at.get(2 * ip), at.get(2 * ip + 1)
                blocks.append(pair_block(fg, None if t0 is None else t0 + b, None if t1 is None else t1 + b))
            G_IP.append(ip)
        GOFFP.append(len(G_IP))
    last_base = cursor
    cursor += 3 * len(blocks)
    # The stream is ceil(images / ch) chunks long: every chunk's first image belongs to a real block (the kernel moves the ring on when
    # it reads that image, so a chunk of nothing but padding would never be consumed).  The gather kernel writes whole blocks: the
    # padding blocks may reach up to two images past the last chunk (STREAM_IMAGES is what the buffer must hold).
    n_chunks = -(-cursor // ch)
    pad_blocks = -(-(n_chunks * ch - cursor) // 3)
    blocks_of.append(blocks + [-np.ones((64, 8), dtype=np.int64)] * pad_blocks)
    stream_images = max(n_chunks * ch, cursor + 3 * pad_blocks)
    gathers = [np.stack(b).astype(np.int32).reshape(-1) if b else np.zeros(0, np.int32) for b in blocks_of]
    out = dict(t)
    for key in ("S_OTG", "S_IT", "S_MASK", "NS", "GOFF", "G_IT"):
        out.pop(key)
    nr = 2 if ch == 48 else 3  # (48-image chunks: two ring slots, half as many barriers)
    per_cu = 2 if (waves == 4 and not wide) else 1  # workgroups per CU
    xlds = int(t["D"] % 4 == 0 and (nr * ch * 256 + (t["BIAS_STRIDE"] * NH + t["NG"] * nt * 16) + 1024 + 256 + waves * 16 * (((t["D"] + 3) // 4) * 4 + 4)) * 4 * per_cu <= 160 * 1024)
    out.update({"split": 1, "TMAX": int(2 * -(-t["TMAX"] // 2)), "NB": NB, "B_OT": B_OT, "B_IP": B_IP, "BASE": BASE, "LAST_BASE": last_base, "GOFF": GOFFP, "G_IP": G_IP, "NCHUNK": n_chunks, "STREAM_IMAGES": stream_images,
                "WAVES": waves, "CH": ch, "NR": nr, "XLDS": xlds, "OCC": 1 if wide else 2})
    plan._split_cache = ((uni_kind, act, split_geometry()), (out, gathers))
    return out, gathers


def half_enabled() -> bool:
    """Whether the TWO-PART (2 x f16, three partial products) twins of the operand-split kernels may serve inference launches
    (zuko_amd.set_matmul_precision / ZUKO_AMD_MATMUL: "f16x2" default, "bf16x3" keeps every product on the three-part kernels)."""
    from . import fused

    return split_enabled() and fused.matmul_precision() == "f16x2"


def half_tables(plan, uni_kind: int, act: int = 1):
    """Tables + gather indices of the two-part twin of an operand-split kernel (csrc/fused_ar_half_impl.h), or None.

    Same blocks in the same order as split_tables — a block is the 16 x 32 weight block (one out tile, one pair of in tiles) — as TWO 1 KiB f16
    images (h, l) of the layer's weights times a power of two (zuko_amd/fused.py picks it: the layer's largest magnitude lands in [2^14, 2^15)),
    three partial products (lh, hl, hh) on v_mfma_f32_16x16x32_f16.  Eight wavefronts, conditioners up to 256 wide and three hidden layers (the
    kernel takes one descale factor per linear layer, four at most); the polynomial maps keep the three-part kernels."""
    ts = split_tables(plan, uni_kind, act)
    if ts is None:
        return None
    t, gathers = ts
    if t["WAVES"] != 8 or t["NH"] > 3 or uni_kind in SPLIT_ONLY_KINDS or t["CH"] != 24:
        return None
    cached = getattr(plan, "_half_cache", None)
    if cached is not None and cached[0] == (uni_kind, act):
        return cached[1]
    ch, NH, nt = t["CH"], t["NH"], plan.layout.nt
    n_last = t["GOFF"][-1] * nt
    cursor = 2 * (sum(t["NB"]) + n_last)
    n_chunks = -(-cursor // ch)
    pad_blocks = -(-(n_chunks * ch - cursor) // 2)
    last = gathers[NH][: n_last * 512]
    g = [np.asarray(x) for x in gathers[:NH]] + [np.concatenate([last, -np.ones(pad_blocks * 512, dtype=np.int32)]).astype(np.int32)]
    out = {k: v for k, v in t.items() if k not in ("split", "TRAIN_OK")}
    out.update({"half": 1, "BASE": [2 * sum(t["NB"][:l]) for l in range(NH)], "LAST_BASE": 2 * sum(t["NB"]), "NCHUNK": n_chunks, "STREAM_IMAGES": max(n_chunks * ch, cursor + 2 * pad_blocks),
                "NR": 3, "TRAIN_OK": 0})
    plan._half_cache = ((uni_kind, act), (out, g))
    return out, g


def emit_half(t: dict) -> str:
    boff = [0]
    for n in t["NB"]:
        boff.append(boff[-1] + n)
    lines = [
        "// generated by zuko_amd/static_ar.py — do not edit",
        '#include "fused_ar_half_impl.h"',
        "namespace {",
        "struct Shape {",
        f"  static constexpr int D = {t['D']}, DIN = {t['DIN']}, NIT = {t['NIT']}, NH = {t['NH']}, TMAX = {t['TMAX']}, NG = {t['NG']}, NCHUNK = {t['NCHUNK']};",
        f"  static constexpr int BIAS_STRIDE = {t['BIAS_STRIDE']}, LAST_BASE = {t['LAST_BASE']}, WAVES = {t['WAVES']}, CH = {t['CH']}, NR = {t['NR']}, ACT = {t['ACT']}, OCC = {t['OCC']};",
        f"  static constexpr bool XLDS = {'true' if t['XLDS'] else 'false'};",
        _arr("HT", "int", t["HT"]), _arr("NB", "int", t["NB"]), _arr("BOFF", "int", boff), _arr("BASE", "int", t["BASE"]),
        _arr("B_OT", "unsigned char", t["B_OT"]), _arr("B_IP", "unsigned char", t["B_IP"]), _arr("GOFF", "int", t["GOFF"]), _arr("G_IP", "unsigned char", t["G_IP"]),
        "};",
        "}  // namespace",
        f'extern "C" int zk_ars_launch(const zk::ArArgs* a, int abi, int args_bytes, int train, void* stream) {{ return zk::arh_launch<Shape, {UNI_TYPES[t["uni"]]}>(a, abi, args_bytes, train, stream); }}',
        "",
    ]
    return "\n".join(lines)


def compile_half(t: dict, verbose: bool = False, out_dir: str | None = None) -> dict | None:
    """Build arh_<sig>.so, the two-part operand-split kernel of tables `t` (half_tables); returns its meta or None."""
    stamp = _half_digest()
    sig = _digest({"half": t, "headers": stamp})
    meta = {"so": f"arh_{sig}.so", "core": "h" + _digest(t), "half": 1, "split": 2, "l0": [], "alt": None, "headers": stamp, "uni": t["uni"], "ACT": t["ACT"], "D": t["D"], "DIN": t["DIN"], "HT": t["HT"],
            "WAVES": t["WAVES"], "CH": t["CH"], "TRAIN_OK": 0, "XLDS": t["XLDS"], "NCHUNK": t["NCHUNK"]}
    so = _build_so(f"arh_{sig}", lambda: emit_half(t), meta, verbose, out_dir)
    if so is None:
        return None
    global _INDEX
    _INDEX = None
    return dict(meta, dir=os.path.dirname(so))


def lookup_half(plan, uni_kind: int, act: int, rows: int | None = None):
    """The two-part kernel (StaticKernel) for this plan, or None: found on disk, or compiled when `rows` reaches the JIT threshold."""
    if os.environ.get("ZUKO_AMD_NO_STATIC_AR", "0") == "1" or not split_enabled():
        return None
    ts = half_tables(plan, uni_kind, act)
    if ts is None:
        return None
    cdx = "h" + _digest(ts[0])
    with _LOCK:
        idx = _INDEX if _INDEX is not None else _scan()
        for meta in idx.get(cdx, []):
            return _load(meta)
    if rows is not None and rows >= jit_min_rows() and jit_enabled():
        meta = compile_half(ts[0], verbose=os.environ.get("ZUKO_AMD_JIT_VERBOSE", "0") == "1")
        if meta is not None:
            with _LOCK:
                return _load(meta)
    return None


def emit_split(t: dict) -> str:
    boff = [0]
    for n in t["NB"]:
        boff.append(boff[-1] + n)
    lines = [
        "// generated by zuko_amd/static_ar.py — do not edit",
        '#include "fused_ar_split_impl.h"',
        "namespace {",
        "struct Shape {",
        f"  static constexpr int D = {t['D']}, DIN = {t['DIN']}, NIT = {t['NIT']}, NH = {t['NH']}, TMAX = {t['TMAX']}, NG = {t['NG']}, NCHUNK = {t['NCHUNK']};",
        f"  static constexpr int BIAS_STRIDE = {t['BIAS_STRIDE']}, LAST_BASE = {t['LAST_BASE']}, WAVES = {t['WAVES']}, CH = {t['CH']}, NR = {t['NR']}, ACT = {t['ACT']}, OCC = {t['OCC']};",
        f"  static constexpr bool XLDS = {'true' if t['XLDS'] else 'false'}, HAS_ALT = false, TRAIN_OK = {'true' if t['TRAIN_OK'] else 'false'};",
        _arr("HT", "int", t["HT"]), _arr("NB", "int", t["NB"]), _arr("BOFF", "int", boff), _arr("BASE", "int", t["BASE"]),
        _arr("B_OT", "unsigned char", t["B_OT"]), _arr("B_IP", "unsigned char", t["B_IP"]), _arr("GOFF", "int", t["GOFF"]), _arr("G_IP", "unsigned char", t["G_IP"]),
        "};",
        "}  // namespace",
        f'extern "C" int zk_ars_launch(const zk::ArArgs* a, int abi, int args_bytes, int train, void* stream) {{ return zk::arx_launch<Shape, {UNI_TYPES[t["uni"]]}>(a, abi, args_bytes, train, stream); }}',
        "",
    ]
    return "\n".join(lines)


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
    for d in _dirs():
        try:
            names = sorted(os.listdir(d)) if os.path.isdir(d) else []
        except OSError:
            names = []
        for name in names:
            if name.endswith(".json"):
                try:
                    with open(os.path.join(d, name)) as f:
                        meta = json.load(f)
                except (OSError, ValueError):
                    continue
                if meta.get("headers") == _stamp_of(meta) and os.path.exists(os.path.join(d, meta["so"])):
                    meta["dir"] = d
                    idx.setdefault(meta["core"], []).append(meta)
    _INDEX = idx
    return idx


def _torch_lib_dir() -> str:
    import importlib.util

    spec = importlib.util.find_spec("torch")
    return os.path.join(os.path.dirname(spec.origin), "lib")


def compile_kernel(t: dict, alt: list | None, verbose: bool = False, out_dir: str | None = None) -> dict | None:
    """Build ars_<sig>.so for tables `t` (no-op when it is there and current); returns its meta or None (no hipcc / failure / read-only directory)."""
    core, l0 = _split(t)
    stamp = _header_digest()
    sig = _digest({"core": core, "l0": l0, "alt": alt, "headers": stamp})
    meta = {"so": f"ars_{sig}.so", "core": _digest(core), "l0": l0, "alt": alt, "headers": stamp, "uni": t["uni"], "ACT": t["ACT"], "D": t["D"], "DIN": t["DIN"], "HT": t["HT"], "WAVES": t["WAVES"],
            "TRAIN_OK": t["TRAIN_OK"], "XLDS": t["XLDS"]}
    so = _build_so(f"ars_{sig}", lambda: emit(t, alt), meta, verbose, out_dir)
    if so is None:
        return None
    global _INDEX
    _INDEX = None
    return dict(meta, dir=os.path.dirname(so))


def compile_split(t: dict, verbose: bool = False, out_dir: str | None = None) -> dict | None:
    """Build arx_<sig>.so, the operand-split kernel of tables `t` (split_tables); returns its meta or None."""
    stamp = _header_digest()
    sig = _digest({"split": t, "headers": stamp})
    meta = {"so": f"arx_{sig}.so", "core": "x" + _digest(t), "split": 1, "l0": [], "alt": None, "headers": stamp, "uni": t["uni"], "ACT": t["ACT"], "D": t["D"], "DIN": t["DIN"], "HT": t["HT"],
            "WAVES": t["WAVES"], "CH": t["CH"], "TRAIN_OK": t["TRAIN_OK"], "XLDS": t["XLDS"], "NCHUNK": t["NCHUNK"]}
    so = _build_so(f"arx_{sig}", lambda: emit_split(t), meta, verbose, out_dir)
    if so is None:
        return None
    global _INDEX
    _INDEX = None
    return dict(meta, dir=os.path.dirname(so))


def _load(meta: dict) -> StaticKernel:
    k = _LOADED.get(meta["so"])
    if k is None:
        k = StaticKernel(os.path.join(meta.get("dir", ARS_DIR), meta["so"]), meta)
        _LOADED[meta["so"]] = k
    return k


def jit_enabled() -> bool:
    return os.environ.get("ZUKO_AMD_JIT", "1") != "0" and _hipcc() is not None


def jit_min_rows() -> int:
    try:
        return int(os.environ.get("ZUKO_AMD_JIT_MIN_ROWS", str(1 << 15)))
    except ValueError:
        return 1 << 15


JIT_CUMULATIVE_FACTOR = 8


def effective_rows(holder, rows: int, count: bool = True) -> int:
    """The batch size the JIT threshold is compared with: `rows` itself, or the threshold once `holder` (a plan / state object living as long as
    the conditioner) has seen JIT_CUMULATIVE_FACTOR x the threshold in TOTAL — a training loop with small batches gets its kernel after a few
    dozen steps instead of never.  (ZUKO_AMD_JIT=0 still disables every compile.)"""
    thr = jit_min_rows()
    seen = getattr(holder, "_jit_rows_seen", 0) + (max(int(rows), 0) if count else 0)  # (count=False: a second consultation within the same step)
    try:
        holder._jit_rows_seen = seen
    except AttributeError:
        return rows
    return rows if rows >= thr or seen < JIT_CUMULATIVE_FACTOR * thr else thr


def lookup(plan, uni_kind: int, act: int, rows: int | None = None):
    """(StaticKernel, rev) for this plan, or None.  Kernels found on disk are used whatever the batch; a missing one is compiled
    when `rows` reaches the JIT threshold."""
    if os.environ.get("ZUKO_AMD_NO_STATIC_AR", "0") == "1":
        return None
    t = tables(plan, uni_kind, act)
    if t is None:
        return None
    allow = rows is not None and rows >= jit_min_rows() and jit_enabled()
    verbose = os.environ.get("ZUKO_AMD_JIT_VERBOSE", "0") == "1"
    # (a compile runs OUTSIDE the module lock — the per-kernel file lock serialises ranks / threads that want the same kernel — so that
    #  other threads' lookups do not wait 10-25 s behind it)
    if split_enabled():  # the operand-split kernel (6/16 of the f32 matrix time) when there is one or one may be built
        ts = split_tables(plan, uni_kind, act)
        if ts is not None:
            cdx = "x" + _digest(ts[0])
            with _LOCK:
                idx = _INDEX if _INDEX is not None else _scan()
                for meta in idx.get(cdx, []):
                    return _load(meta), 0
            if allow:
                meta = compile_split(ts[0], verbose=verbose)
                if meta is not None:
                    with _LOCK:
                        return _load(meta), 0
    if uni_kind in SPLIT_ONLY_KINDS:
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
    if allow:
        meta = compile_kernel(t, None, verbose=verbose)
        if meta is not None:
            with _LOCK:
                return _load(meta), 0
    return None


# --------------------------------------------------------------------------------------------------------------------
# the backward twin: dgrad through the hidden layers (csrc/fused_ar_static_impl.h: ars_dgrad_kernel)
# --------------------------------------------------------------------------------------------------------------------


def chain_tables(masks_sorted: list, rows: list, cols: list):
    """Tables + gather indices of the dgrad chain of a masked conditioner.

    masks_sorted[l] (bool [out_l, in_l], l = 0 .. n-2: every linear layer but the last) are the masks in the SORTED unit order of
    zuko_amd/train.py:SortedPlan; rows[l] / cols[l] map a sorted row / column back to the module's (so that the weight stream can be
    gathered straight from the module's parameters).  Chain layer c multiplies by the TRANSPOSE of layer l = n-2-c.  Returns
    (tables, gathers) — gathers[c]: int32 index into W_l.flatten() per stream element (-1 = zero), whole 24-tile chunks per layer —
    or None when the widths do not fit (multiples of 16 up to 256 for the hidden layers, a multiple of 4 for the input)."""
    n1 = len(masks_sorted)
    if n1 < 1 or n1 > 4:
        return None
    widths_out = [m.shape[0] for m in masks_sorted]
    if any(w % 16 or w > 256 for w in widths_out) or masks_sorted[0].shape[1] % 4 or masks_sorted[0].shape[1] > 256:
        return None
    lane = np.arange(64)
    li, lq = lane % 16, lane // 16
    S_OTG, S_IT, S_MASK, NS, HT, BASE, gathers = [], [], [], [], [], [], []
    cursor = 0
    for c in range(n1):
        l = n1 - 1 - c
        M = np.asarray(masks_sorted[l], dtype=bool)  # [out_l, in_l]
        out_l, in_l = M.shape
        n_ot, n_it = -(-in_l // 16), -(-out_l // 16)  # chain outputs = the layer's inputs, chain inputs = the layer's outputs
        HT.append(n_ot)
        blocks, n = [], 0
        for otg in range(-(-n_ot // 4)):
            for it in range(n_it):
                m4 = 0
                for t in range(4):
                    ot = otg * 4 + t
                    if ot < n_ot and M[it * 16 : (it + 1) * 16, ot * 16 : (ot + 1) * 16].any():
                        m4 |= 1 << t
                if m4:
                    S_OTG.append(otg), S_IT.append(it), S_MASK.append(m4)
                    n += 1
                    for t in range(4):
                        if m4 >> t & 1:
                            a_ = (otg * 4 + t) * 16 + li[:, None]                        # chain row  = sorted INPUT unit of layer l
                            b_ = it * 16 + (4 * lq)[:, None] + np.arange(4)[None, :]      # chain col  = sorted OUTPUT unit of layer l
                            ok = (a_ < in_l) & (b_ < out_l)
                            idx = np.where(ok, np.asarray(rows[l])[np.minimum(b_, out_l - 1)] * in_l + np.asarray(cols[l])[np.minimum(a_, in_l - 1)], -1)
                            blocks.append(idx.reshape(-1))
        NS.append(n)
        pad = -(-len(blocks) // 24) * 24 - len(blocks)
        blocks += [-np.ones(256, dtype=np.int64)] * pad
        BASE.append(cursor)
        cursor += len(blocks)
        gathers.append(np.concatenate(blocks).astype(np.int32) if blocks else np.zeros(0, np.int32))
    NIT = -(-widths_out[-1] // 16)
    t = {"chain": 1, "DIN": int(widths_out[-1]), "DOUT": int(masks_sorted[0].shape[1]), "NIT": int(NIT), "NH": n1, "HT": HT, "TMAX": int(4 * -(-max([NIT] + HT) // 4)),
         "NCHUNK": cursor // 24, "NS": NS, "S_OTG": S_OTG, "S_IT": S_IT, "S_MASK": S_MASK, "BASE": BASE}
    if t["TMAX"] > 16 or cursor == 0:
        return None
    return t, gathers


def chain_split_tables(masks_sorted: list, rows: list, cols: list, packed: dict | None = None):
    """Tables + gather indices of the operand-split dgrad chain over ALL n linear layers (csrc/fused_ar_split_impl.h: arxd_kernel), from the
    gradient of the packed parameters down to the conditioner's input.  Arguments as chain_tables, but for every layer l = 0 .. n-1.
    Chain layer c multiplies by the transpose of layer l = n-1-c; a block is (out tile = 16 sorted INPUT units of layer l, in pair = 32
    sorted OUTPUT units of layer l) as three bf16 images (split_tables).  Layer 0 is walked in-pair major (its input g_phi is streamed
    from global memory), the others out-tile major.  Returns (tables, gathers) or None.

    packed = {"uni": kind, "featmap": int array [NG * 4 * FPL], "nt", "fpl", "total"} (the forward plan's grouping of the features,
    zuko_amd/fused.py: build_plan): the tables of arxb_kernel instead — the whole backward of the transform in one launch.  Its first layer
    runs over the FORWARD kernel's packed order of phi (unit 16 (g NT + t) + 4 q + r = parameter 4 t + r of the features of lane q in group
    g), in which a lane owns the parameters of its own features and computes their gradient itself; MODROW maps a packed unit to its row of
    the last linear layer (-1: padding slot)."""
    n = len(masks_sorted)
    if n < 2 or n > 4:
        return None
    hidden = [m.shape[0] for m in masks_sorted[:-1]]
    din, dphi = masks_sorted[0].shape[1], masks_sorted[-1].shape[0]
    if any(w % 16 or w > 256 for w in hidden) or din % 4 or din > 256 or (packed is None and dphi % 4):
        return None  # (module-order g_phi rows are read 16 bytes at a time; packed rows are whole tiles)
    masks_sorted, rows = list(masks_sorted), list(rows)
    if packed is not None:
        fm, nt, fpl, total = np.asarray(packed["featmap"]), int(packed["nt"]), int(packed["fpl"]), int(packed["total"])
        ng = len(fm) // (4 * fpl)
        if ng * 4 * fpl != len(fm) or (ng * nt) % 2 or (nt % 2 and ng % 2) or fpl * total > 4 * nt or dphi % total or dphi // total > din or rows[-1] is None:
            return None  # (dphi / total features, the other din - features inputs are context)
        u = np.arange(ng * nt * 16)
        g_, t_, q_, r_ = u // (16 * nt), (u // 16) % nt, (u % 16) // 4, u % 4
        fi_, k_ = np.divmod(4 * t_ + r_, total)
        f_ = np.where(fi_ < fpl, fm[(g_ * 4 + q_) * fpl + np.minimum(fi_, fpl - 1)], -1)
        mod_row = np.where(f_ >= 0, f_ * total + k_, -1)                                       # packed unit -> row of the last linear layer (module order)
        if sorted(mod_row[mod_row >= 0].tolist()) != list(range(dphi)) or not np.array_equal(np.asarray(rows[-1]), np.arange(dphi)):
            return None
        M_last = np.asarray(masks_sorted[-1], dtype=bool)
        masks_sorted[-1] = np.where((mod_row >= 0)[:, None], M_last[np.maximum(mod_row, 0)], False)
        rows[-1] = mod_row
    ch = 24
    lane = np.arange(64)
    li, lq = lane % 16, lane // 16
    HT, NB, BASE, B_OT, B_IP, blocks_of, P0 = [], [], [], [], [], [], []
    cursor = 0
    for c in range(n):
        l = n - 1 - c
        M = np.asarray(masks_sorted[l], dtype=bool)
        out_l, in_l = M.shape
        n_ot, n_ip = -(-in_l // 16), -(-(-(-out_l // 16)) // 2)
        HT.append(n_ot)

        def block(ot, ip):
            a_ = ot * 16 + li[:, None]                                                        # sorted input unit of layer l
            e = np.arange(8)[None, :]
            b_ = (2 * ip + e // 4) * 16 + (4 * lq)[:, None] + e % 4                            # sorted output unit of layer l
            row = np.asarray(rows[l])[np.minimum(b_, out_l - 1)]
            ok = (a_ < in_l) & (b_ < out_l) & (row >= 0)
            return np.where(ok, row * in_l + np.asarray(cols[l])[np.minimum(a_, in_l - 1)], -1)

        def live(ot, ip):
            return bool(M[ip * 32 : ip * 32 + 32, ot * 16 : ot * 16 + 16].any())

        order = [(ot, ip) for ip in range(n_ip) for ot in range(n_ot)] if c == 0 else [(ot, ip) for ot in range(n_ot) for ip in range(n_ip)]
        blocks = []
        for ot, ip in order:
            if live(ot, ip):
                blocks.append(block(ot, ip))
                B_OT.append(ot), B_IP.append(ip)
                if c == 0 and (not P0 or P0[-1] != ip):
                    P0.append(ip)
        if not blocks:
            return None
        NB.append(len(blocks))
        BASE.append(cursor)
        cursor += 3 * len(blocks)
        blocks_of.append(blocks)
    n_chunks = -(-cursor // ch)
    pad = -(-(n_chunks * ch - cursor) // 3)
    blocks_of[-1] = blocks_of[-1] + [-np.ones((64, 8), dtype=np.int64)] * pad
    t = {"chain": 2, "DIN0": int(masks_sorted[-1].shape[0]), "DOUT": int(din), "NH": n, "HT": HT, "TMAX": int(2 * -(-max(HT) // 2)), "NB": NB, "BASE": BASE, "B_OT": B_OT, "B_IP": B_IP,
         "NP0": len(P0), "P0": P0, "NCHUNK": n_chunks, "STREAM_IMAGES": max(n_chunks * ch, cursor + 3 * pad), "WAVES": 8, "CH": ch}
    if t["TMAX"] > 16:
        return None
    if packed is not None:  # first block of every packed pair (layer 0 is in-pair major: the blocks of a pair are consecutive)
        n_pairs = t["DIN0"] // 32
        ips = np.asarray(B_IP[: NB[0]])
        t.update({"chain": 3, "uni": int(packed["uni"]), "NG": ng, "NT": nt, "PB": [int((ips < pp).sum()) for pp in range(n_pairs + 1)], "MODROW": [int(r) for r in mod_row]})
    return t, [np.stack(b).astype(np.int32).reshape(-1) for b in blocks_of]


def emit_chain_split(t: dict) -> str:
    boff = [0]
    for nb in t["NB"]:
        boff.append(boff[-1] + nb)
    lines = [
        "// generated by zuko_amd/static_ar.py — do not edit",
        '#include "fused_ar_split_impl.h"',
        "namespace {",
        "struct Shape {",
        f"  static constexpr int DIN0 = {t['DIN0']}, DOUT = {t['DOUT']}, NH = {t['NH']}, TMAX = {t['TMAX']}, NCHUNK = {t['NCHUNK']}, WAVES = 8, CH = {t['CH']}, NP0 = {t['NP0']}, ACT = 1;",
        _arr("HT", "int", t["HT"]), _arr("NB", "int", t["NB"]), _arr("BOFF", "int", boff), _arr("BASE", "int", t["BASE"]), _arr("P0", "unsigned char", t["P0"]),
        _arr("B_OT", "unsigned char", t["B_OT"]), _arr("B_IP", "unsigned char", t["B_IP"]),
    ]
    if t.get("chain") == 3:  # the whole backward of the transform (arxb_kernel)
        lines += [f"  static constexpr int NG = {t['NG']};", _arr("PB", "int", t["PB"])]
        launch = f"zk::arxb_launch<Shape, {UNI_TYPES[t['uni']]}>"
    else:
        launch = "zk::arxd_launch<Shape>"
    lines += [
        "};",
        "}  // namespace",
        f'extern "C" int zk_ars_dgrad_launch(const zk::ArArgs* a, int abi, int args_bytes, void* stream) {{ return {launch}(a, abi, args_bytes, stream); }}',
        "",
    ]
    return "\n".join(lines)


def emit_chain(t: dict) -> str:
    soff = [0]
    for n in t["NS"]:
        soff.append(soff[-1] + n)
    lines = [
        "// generated by zuko_amd/static_ar.py — do not edit",
        '#include "fused_ar_static_impl.h"',
        "namespace {",
        "struct Shape {",
        f"  static constexpr int DIN = {t['DIN']}, DOUT = {t['DOUT']}, NIT = {t['NIT']}, NH = {t['NH']}, TMAX = {t['TMAX']}, NCHUNK = {t['NCHUNK']}, WAVES = 8, ACT = 1;",
        "  static constexpr bool HAS_ALT = false;",
        _arr("HT", "int", t["HT"]), _arr("NS", "int", t["NS"]), _arr("SOFF", "int", soff), _arr("BASE", "int", t["BASE"]),
        _arr("S_OTG", "unsigned char", t["S_OTG"]), _arr("S_IT", "unsigned char", t["S_IT"]), _arr("S_ALT", "unsigned char", t["S_IT"]),
        _arr("S_MASK", "unsigned char", t["S_MASK"]),
        "};",
        "}  // namespace",
        'extern "C" int zk_ars_dgrad_launch(const zk::ArArgs* a, int abi, int args_bytes, void* stream) { return zk::ars_dgrad_launch<Shape>(a, abi, args_bytes, stream); }',
        "",
    ]
    return "\n".join(lines)


class ChainKernel:
    def __init__(self, so: str) -> None:
        self.so = so
        self.cdll = ctypes.CDLL(so)
        self.launcher = ctypes.cast(self.cdll.zk_ars_dgrad_launch, ctypes.c_void_p)


_CHAINS: dict[str, ChainKernel] = {}


def chain_kernel(t: dict, allow_compile: bool, verbose: bool = False, out_dir: str | None = None):
    """The compiled dgrad-chain kernel for tables `t` (arsd_<sig>.so), built on demand when allowed; None otherwise."""
    stamp = _header_digest()
    sig = _digest({"t": t, "headers": stamp})
    stem = f"arsd_{sig}"
    with _LOCK:
        k = _CHAINS.get(stem)
        if k is not None:
            return k
        so = _find(stem + ".so")
        if so is None:
            if not allow_compile:
                return None
            meta = {"so": stem + ".so", "headers": stamp, "core": "chain", "l0": [], "alt": None, "DIN": t.get("DIN", t.get("DIN0")), "DOUT": t["DOUT"], "HT": t["HT"]}
            so = _build_so(stem, lambda: emit_chain_split(t) if t.get("chain") in (2, 3) else emit_chain(t), meta, verbose, out_dir)
            if so is None:
                return None
        try:
            k = ChainKernel(so)
        except OSError as exc:
            _warn_once("load:" + so, f"cannot load {so}: {exc}")
            return None
        _CHAINS[stem] = k
        return k


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
    ("rqs", 64, 0, (256, 256, 256), 16),   # NSF(bins=16): operand-split kernel only (SPLIT_ONLY_KINDS)
    ("affine", 12, 0, (64, 64), 0),        # training: a last feature group that is not full (padding slots in the packed phi / g_phi rows)
    # three more shapes of the static-shape table (profiles/r04/static_shapes.jsonl, tests/test_gpu_flows.py::test_static_shape_table): 128 features, a context
    # behind 64 features, three 512-wide layers
    ("rqs", 128, 0, (256, 256, 256), 8), ("rqs", 64, 8, (256, 256), 8), ("rqs", 16, 0, (512, 512, 512), 8),
    # the polynomial flows (forward, operand-split kernels only): the shapes of the golden flows and a 64-feature one of each
    ("sos", 4, 2, (32, 32), 0), ("bern", 4, 2, (32, 32), 0), ("sos", 64, 0, (256, 256, 256), 0), ("bern", 64, 0, (256, 256, 256), 0),
]


def _plans_for(kind: str, features: int, context: int, hidden, bins: int, activation: str | None = None):
    """Plans of the ascending- and descending-order transform of such a flow (MAF / NSF alternate the two).  (The activation does not
    enter the plan: it is a field of the tables.)"""
    import torch

    from . import fused
    from .flows.autoregressive import MaskedAutoregressiveTransform
    from .nn import MaskedLinear
    from .transforms import BoundedBernsteinTransform, MonotonicAffineTransform, MonotonicRQSTransform, ShiftedSOSPolynomialTransform

    out = []
    for order in (torch.arange(features), torch.flipud(torch.arange(features))):
        if kind == "affine":
            t = MaskedAutoregressiveTransform(features, context, order=order, hidden_features=list(hidden), univariate=MonotonicAffineTransform, shapes=[(), ()])
            layout = fused.uni_layout("affine", 2)
        elif kind == "sos":  # SOSPF's defaults
            t = MaskedAutoregressiveTransform(features, context, order=order, hidden_features=list(hidden), univariate=ShiftedSOSPolynomialTransform, shapes=[(3, 5), ()])
            layout = fused.uni_layout("sos", 16)
        elif kind == "bern":  # BPF's default
            t = MaskedAutoregressiveTransform(features, context, order=order, hidden_features=list(hidden), univariate=BoundedBernsteinTransform, shapes=[(17,)])
            layout = fused.uni_layout("bern", 17)
        else:
            t = MaskedAutoregressiveTransform(features, context, order=order, hidden_features=list(hidden), univariate=MonotonicRQSTransform, shapes=[(bins,), (bins,), (bins - 1,)])
            layout = fused.uni_layout("rqs", 3 * bins - 1, bins)
        masks = [m.mask for m in t.hyper if isinstance(m, MaskedLinear)]
        wide = max(list(hidden) + [features + context]) > fused.MAX_WIDTH
        out.append((fused.build_plan(masks, features, layout, max_width=fused.MAX_WIDTH_WIDE if wide else fused.MAX_WIDTH), layout, [m for m in t.hyper if isinstance(m, MaskedLinear)]))
    return out


def chain_tables_for(lins, full: bool = False, packed: dict | None = None):
    """chain_tables (full: chain_split_tables, with `packed` the whole-backward tables) of a masked ReLU conditioner given its linear layers
    (through zuko_amd/train.py:SortedPlan), or None."""
    import torch

    from .train import SortedPlan

    sp = SortedPlan(lins, 1, torch.device("cpu"))
    n = len(lins)
    if n < 2 or n > 4 or any(m is None for m in sp.mask_s_cpu):
        return None
    if full:
        return chain_split_tables(sp.mask_s_cpu, sp.rows_cpu, sp.cols_cpu, packed=packed) if (packed is not None or sp.shapes[-1][0] % 4 == 0) else None
    return chain_tables(sp.mask_s_cpu[: n - 1], sp.rows_cpu[: n - 1], sp.cols_cpu[: n - 1])


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
                        m_ = json.load(f)
                        stale = m_.get("headers") != _stamp_of(m_)
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
    work, chains, splits, halves = [], [], [], []
    for entry in PREBUILT:
        kind, features, context, hidden, bins = entry[:5]
        import torch

        from .nn import _act_code

        act = _act_code(getattr(torch.nn, entry[5])()) if len(entry) > 5 else 1
        (pa, layout, lins_a), (pd, _, lins_d) = _plans_for(kind, features, context, hidden, bins)
        if act == 1:  # the training backward of the same conditioners (one kernel per feature order; the polynomial flows — SPLIT_ONLY_KINDS — train on the two-node path: chains without the packed variant)
            for lins, pl in ((lins_a, pa), (lins_d, pd)):
                cands = [chain_tables_for(lins), chain_tables_for(lins, full=True)]
                if (features + context) % 4 == 0 and layout.kind in (0, 1) and layout.total in (2, 23) and pl is not None:  # what zuko_amd/train.py:autoregressive() covers
                    cands.append(chain_tables_for(lins, full=True, packed={"uni": layout.kind, "featmap": pl.featmap, "nt": layout.nt, "fpl": layout.fpl, "total": layout.total}))
                for tg in cands:
                    if tg is not None and not any(c[0] == tg[0] for c in chains):
                        chains.append(tg)
        for pl in (pa, pd):
            ts = split_tables(pl, layout.kind, act)
            if ts is not None and not any(x == ts[0] for x in splits):
                splits.append(ts[0])
            th = half_tables(pl, layout.kind, act)  # the two-part twin (inference launches)
            if th is not None and not any(x == th[0] for x in halves):
                halves.append(th[0])
        if layout.kind in SPLIT_ONLY_KINDS:
            continue  # (no f32-instruction kernel, no training chain for this kind)
        ta, td = tables(pa, layout.kind, act), tables(pd, layout.kind, act)
        if ta is None or td is None:
            raise RuntimeError(f"zuko_amd.static_ar: no static kernel for the prebuilt shape {entry}")
        (ca, la), (cdesc, ld) = _split(ta), _split(td)
        if ca == cdesc:
            work.append((ta, None if la == ld else ld))
        else:
            work += [(ta, None), (td, None)]
    # the alternative workgroup geometry of the split kernels, for the headline conditioner only (tests/test_gpu_flows.py exercises it)
    prev = os.environ.get("ZUKO_AMD_SPLIT_GEOM")
    if prev is None:
        os.environ["ZUKO_AMD_SPLIT_GEOM"] = "4x16"
        try:
            kind, features, context, hidden, bins = PREBUILT[0][:5]
            for pl, layout, _ in _plans_for(kind, features, context, hidden, bins):
                pl._split_cache = None
                ts = split_tables(pl, layout.kind, 1)
                if ts is not None and not any(x == ts[0] for x in splits):
                    splits.append(ts[0])
        finally:
            del os.environ["ZUKO_AMD_SPLIT_GEOM"]
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        metas = list(ex.map(lambda w: compile_kernel(w[0], w[1], verbose, ARS_DIR), work))
        kerns = list(ex.map(lambda tg: chain_kernel(tg[0], True, verbose, ARS_DIR), chains))
        xmetas = list(ex.map(lambda t: compile_split(t, verbose, ARS_DIR), splits))
        hmetas = list(ex.map(lambda t: compile_half(t, verbose, ARS_DIR), halves))
    if any(m is None for m in metas + xmetas + hmetas) or any(k is None for k in kerns):
        raise RuntimeError("zuko_amd.static_ar: a prebuilt static kernel failed to compile")
    return [m["so"] for m in metas + xmetas + hmetas] + [os.path.basename(k.so) for k in kerns]


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(HERE))
    import zuko_amd.static_ar as me  # (as a package module: the relative imports above need it)

    print("\n".join(me.prebuild()))
