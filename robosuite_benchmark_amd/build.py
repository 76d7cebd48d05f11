"""Builds libsac_hip.so (HIP, gfx950 only) in-tree next to this file.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the built .so
travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["replay_buffer.hip", "sac_trainer.hip", "peaks.hip"]
LIB = os.path.join(HERE, "libsac_hip.so")


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libsac_hip.so cannot be built (there is no CPU fallback)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "sac_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-value", "-Wno-pass-failed",
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB)
