"""Process-start shims for running the UNMODIFIED reference drivers (pytorch/bts_main.py, bts_test.py) on this stack;
put on PYTHONPATH by tools/run_reference.py so that mp.spawn children (bts_main.py:600-602) get them too.
SURVEY.md section 8b "stack hazards":

  1. bts_main.py:425-427 / 470-472 `np.sum([var.sum() for var in model.parameters() ...])` -- with torch >= 2 / numpy 2 the
     conversion of a tensor that requires grad raises; the torch-1.2-era behaviour was a 0-dim tensor holding the sum.
  2. torch.load defaults to weights_only=True since torch 2.6 and rejects the numpy `best_eval_steps` array the driver
     stores (bts_main.py:530-539): TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD=1 (set by the launcher's environment).
  3. BTS_REF_ALLOW_CPU=1 (plumbing tests on a box without a GPU): Tensor.cuda / Module.cuda become identities.
  4. BTS_REF_POSTIMPORT=<module>: imported at start-up (tests use it to install a CPU checker-executor).
"""
import os

if os.environ.get("BTS_REF_SHIMS") == "1":
    import numpy as _np
    import torch as _torch

    _np_sum = _np.sum

    def _sum(a, *args, **kw):
        if isinstance(a, (list, tuple)) and a and all(isinstance(t, _torch.Tensor) for t in a):
            return _torch.stack([t.detach().reshape(()) for t in a]).sum()
        return _np_sum(a, *args, **kw)
    _np.sum = _sum

    if os.environ.get("BTS_REF_ALLOW_CPU") == "1" and not _torch.cuda.is_available():
        _torch.Tensor.cuda = lambda self, *a, **k: self
        _torch.cuda.set_device = lambda *a, **k: None

    _post = os.environ.get("BTS_REF_POSTIMPORT")
    if _post:
        __import__(_post)
