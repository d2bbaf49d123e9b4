"""Drop-in model module for the reference drivers (bts_main.py:122-133, bts_test.py:68-74).

Copy (or symlink) this file next to bts_main.py / bts_test.py in place of the reference's
pytorch/bts.py.  bts_main.py copies the model file into <log_dir>/<model_name>/<model_name>.py and
re-imports it from there (bts_main.py:569-585), so this file only locates the bts_amd package --
through $BTS_AMD_HOME or the path recorded below -- and re-exports its names; the HIP library is
found by package path, never relative to this file.
"""
import os
import sys

_HOME = os.environ.get("BTS_AMD_HOME", "/root/repo")
if _HOME not in sys.path:
    sys.path.insert(0, _HOME)

from bts_amd.model import *  # noqa: F401,F403,E402
from bts_amd.model import __all__  # noqa: F401,E402
