"""builds csrc/libgct2.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libgct2.so")


def build_library(force: bool = False, jobs: int = 6) -> str:
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", CSRC, f"-j{jobs}"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libgct2.so failed:\n" + r.stdout[-4000:] + r.stderr[-8000:])
    if not os.path.exists(LIB):
        raise RuntimeError("make succeeded but libgct2.so is missing")
    return LIB
