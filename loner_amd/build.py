"""Build libloner_hip.so (gfx950) in-tree with hipcc.

    python -m loner_amd.build [--force]

Objects are cached under loner_amd/_build and rebuilt when a source or header changes.
The library lands in loner_amd/_lib/libloner_hip.so (git-ignored, travels with gpurun).
"""
import hashlib
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# LNR_BUILD_TAG=<name>: a development build beside the product's (objects in _build_<name>, library _lib/libloner_hip_<name>.so)
_TAG = os.environ.get("LNR_BUILD_TAG", "")
BUILD = os.path.join(HERE, "_build" + ("_" + _TAG if _TAG else ""))
LIBDIR = os.path.join(HERE, "_lib")
LIB = os.path.join(LIBDIR, "libloner_hip" + ("_" + _TAG if _TAG else "") + ".so")
MANIFEST = os.path.join(BUILD, "manifest.json")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value"] + \
         os.environ.get("LNR_EXTRA_HIPCC_FLAGS", "").split()        # development switches, e.g. -DLNR_PHASE_TIMING (tools/README.md)
# files whose float arithmetic must round exactly like the reference's torch CPU ops
EXACT = {"lnr_sampler.hip", "lnr_rays.hip"}
# per-source flags.  lnr_render.hip: its 32-samples-per-lane instantiations (2048-sample rays, the inference path) exceed the default
# threshold for "#pragma unroll"; with a loop left rolled the per-lane arrays become scratch memory (820 bytes per lane: the compositing of
# a rendered scan ran at 230 GB/s)
EXTRA = {"lnr_render.hip": ["-mllvm", "-pragma-unroll-threshold=1000000", "-mllvm", "-unroll-threshold=100000"]}
# (source, object name, extra flags); the density kernels are compiled once per hidden width (n_neurons/16)
SOURCES = [("lnr_density_ht.hip", f"lnr_density_ht{ht}.o", [f"-DLNR_HT={ht}"]) for ht in (16, 8, 4, 2, 1)] + \
          [("lnr_density_regs.hip", f"lnr_density_regs{ht}_{nh}.o", [f"-DLNR_HT={ht}", f"-DLNR_NH={nh}"])
           for ht in (8, 4, 16) for nh in (3, 2, 1) if not (ht == 16 and nh > 1)] + \
          [("lnr_density_f16_bwd.hip", f"lnr_density_f16_bwd{part}_fq{fq}.o", [f"-DLNR_BWD_PART={part}", f"-DLNR_BWD_FQ={fq}"]) for part in (2, 1, 0) for fq in (0, 1)] + \
          [("lnr_density_f16_fwd.hip", f"lnr_density_f16_fwd{part}_fq{fq}.o", [f"-DLNR_FWD_PART={part}", f"-DLNR_FWD_FQ={fq}"]) for part in (1, 0) for fq in (0, 1)] + \
          [(s, s.replace(".hip", ".o"), EXTRA.get(s, [])) for s in
           ("lnr_core.hip", "lnr_density.hip", "lnr_density_f16.hip", "lnr_density_bf3.hip", "lnr_density_wide.hip", "lnr_encode.hip", "lnr_sampler.hip", "lnr_render.hip", "lnr_rays.hip", "lnr_optim.hip", "lnr_pose.hip", "lnr_comm.hip")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _deps_digest(path, seen=None):
    """Digest of the headers a source includes (transitively, quoted includes only): an edit rebuilds the objects that see it, not all 28."""
    seen = {} if seen is None else seen
    text = open(path, "rb").read()
    for inc in _INC.findall(text.decode("utf-8", "replace")):
        f = os.path.normpath(os.path.join(os.path.dirname(path), inc))
        if f not in seen and os.path.exists(f):
            seen[f] = None
            seen[f] = hashlib.sha256(open(f, "rb").read()).hexdigest()
            _deps_digest(f, seen)
    return hashlib.sha256("".join(f"{os.path.basename(k)}:{v}" for k, v in sorted(seen.items())).encode()).hexdigest()


# the sources of the kernels profiles/traffic.json holds PMC traffic for (encode forward / backward, table-gradient reduce, the default
# network's MLP kernels, Adam) and what they include
TRAFFIC_SOURCES = ("lnr_encode.hip", "lnr_encoding.h", "lnr_density.hip", "lnr_density_api.h", "lnr_density_bf3.hip", "lnr_density_impl.h",
                   "lnr_common.h", "lnr_optim.hip")


def sources_digest():
    """One digest over the kernel sources a measurement that outlives the build - profiles/traffic.json's PMC passes - was taken on, so
    that bench.py can tell when those kernels have changed since."""
    h = hashlib.sha256()
    for f in TRAFFIC_SOURCES:
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def _load_manifest():
    try:
        return json.load(open(MANIFEST))
    except Exception:
        return {}


def _compile(item, force):
    src, objname, extra = item
    obj = os.path.join(BUILD, objname)
    path = os.path.join(CSRC, src)
    flags = list(COMMON) + list(extra) + (["-ffp-contract=off"] if src in EXACT else [])
    # staleness by content hash (file mtimes do not survive the copy to the GPU box)
    digest = hashlib.sha256(open(path, "rb").read() + _deps_digest(path).encode() + " ".join(flags).encode()).hexdigest()
    stale = force or not os.path.exists(obj) or _load_manifest().get(objname) != digest
    if stale:
        cmd = [_hipcc()] + flags + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj, stale, objname, digest


def build(force=False, verbose=True):
    os.makedirs(BUILD, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 2)) as ex:
        results = list(ex.map(lambda s: _compile(s, force), SOURCES))
    objs = [r[0] for r in results]
    if any(r[1] for r in results):
        manifest = _load_manifest()
        manifest.update({r[2]: r[3] for r in results})
        json.dump(manifest, open(MANIFEST, "w"), indent=1)
    if any(r[1] for r in results) or not os.path.exists(LIB):
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
