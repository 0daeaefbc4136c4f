"""Builds the plain-C cross-check oracle (oracle/zuko_oracle_c.c) with gcc into oracle/_c/libzuko_oracle_c.so
(git-ignored build product).  Test infrastructure only."""

from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "zuko_oracle_c.c")
OUT = os.path.join(HERE, "_c", "libzuko_oracle_c.so")


def build(force: bool = False) -> str:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", SRC, "-lm", "-o", OUT], check=True)
    return OUT


if __name__ == "__main__":
    print(build())
