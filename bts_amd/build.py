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


MANIFEST = os.path.join(LIBDIR, "manifest.json")


def _sha(paths, extra=""):
    import hashlib
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    return h.hexdigest()


def _md5(path):
    import hashlib
    with open(path, "rb") as f:
        return hashlib.md5(f.read()).hexdigest()


def build_library(force=False, verbose=True, jobs=None):
    """Objects and library are rebuilt by CONTENT, not by mtime: `lib/manifest.json` (git-ignored, travels with the binaries) records
    the sha256 of (source + headers + flags) every object was compiled from and the md5 of the linked library.  A snapshot that
    carries binaries of OTHER sources (mtimes are meaningless after a copy) is rebuilt; binaries of exactly these sources are reused."""
    import json
    os.makedirs(LIBDIR, exist_ok=True)
    try:
        with open(MANIFEST) as f:
            man = json.load(f)
    except (OSError, ValueError):
        man = {}
    objs, todo, want = [], [], {}
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        want[src] = _sha([s] + HEADERS, " ".join(FLAGS))
        if force or not os.path.exists(o) or man.get("objects", {}).get(src) != want[src]:
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
    if force or todo or not os.path.exists(LIB) or man.get("library_md5") != _md5(LIB) or man.get("objects") != want:
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with open(MANIFEST, "w") as f:
        json.dump({"objects": want, "library_md5": _md5(LIB), "flags": FLAGS}, f, indent=1)
    return LIB


def verify_library():
    """True when bts_amd/lib/libbts_amd.so is the library the manifest says was linked from the CURRENT sources."""
    import json
    try:
        with open(MANIFEST) as f:
            man = json.load(f)
    except (OSError, ValueError):
        return False
    want = {src: _sha([os.path.join(CSRC, src)] + HEADERS, " ".join(FLAGS)) for src in SOURCES}
    return os.path.exists(LIB) and man.get("objects") == want and man.get("library_md5") == _md5(LIB)


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB)
