"""Run a Python script (bench.py, tools/*.py) or `-m pytest ...` against a CANDIDATE build of the library instead of
bts_amd/lib/libbts_amd.so, without touching the shipped binary:

    python tools/run_with_candidate.py tools/r4_prep/lib/libbts_amd_candidate.so bench.py --dump-launches out.json
    python tools/run_with_candidate.py <lib.so> -m pytest tests/test_gpu_1_kernels.py -q

The loader's path constant is replaced before anything loads the library; bench.py's `library_md5` then names the candidate.
Measurement tooling only: nothing in the product reads an alternative library path.
"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    lib = os.path.abspath(sys.argv[1])
    if not os.path.exists(lib):
        sys.exit("candidate library %s not found" % lib)
    from bts_amd import _lib
    _lib.LIB_PATH = lib
    rest = sys.argv[2:]
    if rest[0] == "-m":
        sys.argv = [rest[1]] + rest[2:]
        runpy.run_module(rest[1], run_name="__main__", alter_sys=True)
    else:
        sys.argv = rest
        runpy.run_path(rest[0], run_name="__main__")


if __name__ == "__main__":
    main()
