"""Build libbts_amd.so (hipcc, gfx950) in-tree under bts_amd/lib/.

Explicit `hipcc -c` of every kernel translation unit (in parallel: one hipcc process per stale source) and one
`hipcc -shared`; no JIT cache, so the built library travels with the source tree (it is git-ignored, not
gpurun-ignored).  `python -m bts_amd.build --force` rebuilds everything from the tracked sources.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libbts_amd.so")
SOURCES = ["conv_igemm.hip", "conv_halo.hip", "conv_halo_wide.hip", "conv_wgrad.hip", "conv_wgrad_tr.hip", "conv_c1.hip", "pack.hip", "lpg.hip", "lpg_chain.hip",
           "elementwise.hip", "evalops.hip"]
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


def build_library(force=False, verbose=True, jobs=None):
    os.makedirs(LIBDIR, exist_ok=True)
    objs, todo = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS):
            todo.append([_hipcc()] + FLAGS + ["-c", s, "-o", o])
        objs.append(o)
    for stale in os.listdir(LIBDIR):                    # objects of translation units that no longer exist
        if stale.endswith(".o") and os.path.join(LIBDIR, stale) not in objs:
            os.remove(os.path.join(LIBDIR, stale))
    if todo:
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=jobs or min(len(todo), os.cpu_count() or 1)) as ex:
            list(ex.map(run, todo))
    if force or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB)
