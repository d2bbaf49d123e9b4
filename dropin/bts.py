"""Drop-in model module for the reference drivers (bts_main.py:122-133, bts_test.py:68-74).

Copy (or symlink) this file next to bts_main.py / bts_test.py in place of the reference's
pytorch/bts.py.  bts_main.py copies the model file into <log_dir>/<model_name>/<model_name>.py and
re-imports it from there (bts_main.py:569-585), so this file must work from ANY directory: it only
locates the bts_amd package and re-exports its names; the HIP library is found by package path,
never relative to this file.  Search order: an importable ``bts_amd`` (installed or already on
sys.path), ``$BTS_AMD_HOME``, then the checkout this file (or the file it is a symlink to) lives in.
"""
import importlib.util
import os
import sys


def _locate():
    if importlib.util.find_spec("bts_amd") is not None:
        return
    here = os.path.dirname(os.path.realpath(__file__))
    tried = []
    for home in (os.environ.get("BTS_AMD_HOME"), os.path.dirname(here), here):
        if not home:
            continue
        tried.append(home)
        if os.path.isdir(os.path.join(home, "bts_amd")):
            sys.path.insert(0, home)
            return
    raise ImportError("dropin/bts.py: the bts_amd package is not importable; put its checkout on PYTHONPATH or set "
                      "BTS_AMD_HOME (looked in: %s)" % ", ".join(tried))


_locate()

from bts_amd.model import *  # noqa: F401,F403,E402
from bts_amd.model import __all__  # noqa: F401,E402
