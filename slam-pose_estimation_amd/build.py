"""Build recipe for lib/libukf_batch.so (hipcc, gfx950).  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "lib", "libukf_batch.so")


def build_engine(force: bool = False, jobs: int = 8) -> str:
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    if not os.path.exists(LIB):
        raise RuntimeError("engine build produced no " + LIB)
    return LIB
