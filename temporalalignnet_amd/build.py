"""Build libtan_hip.so (hand-written HIP kernels, gfx950 only) in-tree with hipcc.

    python -m temporalalignnet_amd.build [--force]

The .so lands next to this file (temporalalignnet_amd/libtan_hip.so): git-ignored, but shipped to the
GPU box by gpurun with the working tree.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libtan_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = (["-DTAN_PANEL_LAB"] if os.environ.get("TAN_PANEL_LAB") else []) + os.environ.get("TAN_EXTRA_FLAGS", "").split() + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function",
         "-fgpu-rdc" if False else "-fno-gpu-rdc"]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _deps_mtime():
    hs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
            and os.path.getmtime(obj) >= _deps_mtime()):
        return obj
    cmd = [HIPCC, *FLAGS, "-x", "hip", "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.2f} MB) from {len(srcs)} sources")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
