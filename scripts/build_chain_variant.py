#!/usr/bin/env python
"""Probe builds of the headline conditioner's one-launch backward kernel (arxb_kernel) under extra compiler flags:
    python scripts/build_chain_variant.py <tag> [-DARXB_ABL=1 ...]      ->  variants/<tag>/ars/arsd_<sig>.so
Run with ZUKO_AMD_CACHE_DIR=variants/<tag> (searched before zuko_amd/lib/ars) to time it: scripts/arxb_bench.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, flags = sys.argv[1], sys.argv[2:]
os.environ["ZUKO_AMD_STATIC_CXXFLAGS"] = " ".join(flags)
from zuko_amd import static_ar as sa

import shutil

out = os.path.join(ROOT, "variants", tag, "ars")
shutil.rmtree(out, ignore_errors=True)
os.makedirs(out, exist_ok=True)
sa._find = lambda name: None  # (the current kernels of lib/ars have the same names: build anyway)
for kind, bins in (("rqs", 8), ("affine", 0)):
    for plan, layout, lins in sa._plans_for(kind, 64, 0, (256, 256, 256), bins):
        t = sa.chain_tables_for(lins, full=True, packed={"uni": layout.kind, "featmap": plan.featmap, "nt": layout.nt, "fpl": layout.fpl, "total": layout.total})[0]
        stamp = sa._header_digest()
        stem = "arsd_" + sa._digest({"t": t, "headers": stamp})
        meta = {"so": stem + ".so", "headers": stamp, "core": "chain", "l0": [], "alt": None, "DIN": t["DIN0"], "DOUT": t["DOUT"], "HT": t["HT"], "flags": flags}
        print(sa._build_so(stem, lambda: sa.emit_chain_split(t), meta, False, out))
