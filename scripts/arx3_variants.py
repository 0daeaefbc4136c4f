#!/usr/bin/env python
"""Build variant instantiations of the 32-sample split kernel for cfg2 / cfg3 into variants/<name>/ars (ZUKO_AMD_CACHE_DIR), one per
(QB, FILL) pair given on the command line as QBxFILL, e.g.  python scripts/arx2_variants.py 12x3 8x0."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import sys; sys.path.insert(0, %r)
import zuko_amd.static_ar as sa
from concurrent.futures import ThreadPoolExecutor
ts = []
for cfg in (("rqs", 64, 0, (256, 256, 256), 8),) + ((("affine", 64, 0, (256, 256, 256), 0),) if %r else ()):
    for pl, layout, _ in sa._plans_for(*cfg):
        t = sa.split3_tables(pl, layout.kind, 1)[0]
        if not any(t == u for u in ts):
            ts.append(t)
with ThreadPoolExecutor(max_workers=4) as ex:
    print([m and m["so"] for m in ex.map(lambda t: sa.compile_split3(t, False), ts)])
""" % (ROOT, os.environ.get("ARX2_VARIANTS_CFG3", "0") == "1")
procs = []
for v in sys.argv[1:]:  # QBxFILL or QBxFILLxABL (ablation build: -DARX_ABL=k, wrong results by construction)
    qb, fill, *abl = v.split("x")
    env = dict(os.environ, ZUKO_AMD_ARX2_QB=qb, ZUKO_AMD_ARX2_FILL=fill, ZUKO_AMD_CACHE_DIR=os.path.join(ROOT, "variants", v))
    flags = ["-DARX3_ONLY"] if os.environ.get("ARX2_VARIANTS_FULL", "0") != "1" else []
    if abl:
        flags.append(f"-DARX_ABL={abl[0]}" if abl[0].isdigit() else f"-D{abl[0]}")
    env["ZUKO_AMD_STATIC_CXXFLAGS"] = " ".join(flags + os.environ.get("ARX2_VARIANTS_CXXFLAGS", "").split())
    procs.append((v, subprocess.Popen([sys.executable, "-c", CODE], env=env)))
for v, p in procs:
    print(v, p.wait())
print("run with the same ZUKO_AMD_STATIC_CXXFLAGS (it is part of a probe build's signature); this build used:", env.get("ZUKO_AMD_STATIC_CXXFLAGS"))
