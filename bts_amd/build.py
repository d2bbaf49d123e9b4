"""Build libbts_amd.so (hipcc, gfx950) in-tree under bts_amd/lib/.

Explicit `hipcc -shared -fPIC` of the three kernel translation units; no JIT cache, so the
built library travels with the source tree (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libbts_amd.so")
SOURCES = ["conv_igemm.hip", "conv_wgrad_tr.hip", "conv_igemm_pp.hip", "conv_halo_wide.hip", "conv_c1.hip", "lpg.hip", "lpg_chain.hip", "elementwise.hip", "evalops.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_common.h"), os.path.join(CSRC, "lpg_math.h"), os.path.join(os.path.dirname(HERE), "include", "bts_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS):
            cmd = [_hipcc()] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB)
