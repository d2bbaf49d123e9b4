"""bench.py's `roofline_lpg_op` alone (bare LPG operator at the bench shape; GPU box; measurement tooling)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from bts_amd import _lib  # noqa: E402

_lib.load()
torch.cuda.set_device(0)
r = bench.lpg_op_roofline(8, 352, 1216)
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("BTS_LPG")}, "frac": r["frac"], "per_launch": r["per_launch"],
                  "single_frac": r["single_scale_launches"]["frac"]}), flush=True)
