"""Build libloner_hip.so (gfx950) in-tree with hipcc.

    python -m loner_amd.build [--force]

Objects are cached under loner_amd/_build and rebuilt when a source or header changes.
The library lands in loner_amd/_lib/libloner_hip.so (git-ignored, travels with gpurun).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "_build")
LIBDIR = os.path.join(HERE, "_lib")
LIB = os.path.join(LIBDIR, "libloner_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value"]
# files whose float arithmetic must round exactly like the reference's torch CPU ops
EXACT = {"lnr_sampler.hip", "lnr_rays.hip"}
SOURCES = ["lnr_core.hip", "lnr_density.hip", "lnr_sampler.hip", "lnr_render.hip", "lnr_rays.hip", "lnr_optim.hip"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newest_header():
    t = 0.0
    for d in (CSRC, INCLUDE):
        for f in os.listdir(d):
            if f.endswith(".h"):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def _compile(src, force):
    obj = os.path.join(BUILD, src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(path), _newest_header())
    if stale:
        flags = list(COMMON) + (["-ffp-contract=off"] if src in EXACT else [])
        cmd = [_hipcc()] + flags + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj, stale


def build(force=False, verbose=True):
    os.makedirs(BUILD, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 2)) as ex:
        results = list(ex.map(lambda s: _compile(s, force), SOURCES))
    objs = [o for o, _ in results]
    if any(st for _, st in results) or not os.path.exists(LIB):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"built {LIB}")
    elif verbose:
        print(f"{LIB} up to date")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
