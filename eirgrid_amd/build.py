"""In-tree build of libeirgrid_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libeirgrid_hip.so")


def build(force: bool = False, verbose: bool = False) -> str:
    # `negative`: the two fault-injection builds the GPU parity tests load through EIRGRID_LIB (csrc/Makefile)
    cmd = ["make", "-C", _CSRC, "all", "negative"] + (["-B"] if force else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:]); print(r.stderr[-4000:])
    if r.returncode != 0 or not os.path.exists(LIB):
        raise RuntimeError("hipcc build of libeirgrid_hip.so failed")
    return LIB
