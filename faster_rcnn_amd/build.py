"""Builds libfrcnn_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m faster_rcnn_amd.build [--force]

Objects are cached per source by mtime.  "Exact" translation units (integer / IEEE-float
box arithmetic that must round like numpy) are compiled with -ffp-contract=off; the MFMA
conv engine uses the default contraction.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libfrcnn_hip.so")
ARCH = "gfx950"

COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
# per-source extra flags
SOURCES = {
    "boxes.hip": ["-ffp-contract=off"],
    "sort_nms.hip": ["-ffp-contract=off"],
    "roi.hip": ["-ffp-contract=off"],
    "detect.hip": ["-ffp-contract=off"],
}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _sources():
    srcs = dict(SOURCES)
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip") and f not in srcs:
            srcs[f] = []
    return srcs


def _newer(a, deps):
    if not os.path.exists(a):
        return False
    t = os.path.getmtime(a)
    return all(os.path.getmtime(d) <= t for d in deps)


def build_library(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "frcnn_hip.h"))
    hipcc = _hipcc()
    objs = []
    procs = []
    for src, extra in _sources().items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if not force and _newer(o, [s] + headers):
            continue
        cmd = [hipcc, "-c", s, "-o", o] + COMMON + extra
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    if force or procs or not _newer(LIB, objs):
        # -z defs: an unresolved kernel stub must fail HERE, not as an "undefined symbol" when the library is loaded on the GPU box
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-Wl,-z,defs", "-o", LIB] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
