"""TEST INFRASTRUCTURE ONLY -- import the *unmodified* reference ``pytorch/bts.py``.

Works only where ``/root/reference`` exists (the build container); the GPU box
has no reference tree, so nothing that runs there may call this.  Used by
``tools/make_golden.py`` (golden-vector generation) and by
``tests/test_oracle_golden.py::test_oracle_vs_live_reference_c1_plumbing`` (skipped when the tree is absent).

Two shims are installed, neither edits the reference:
  * ``torch.Tensor.cuda -> identity`` when no GPU is visible, because
    local_planar_guidance.forward hard-codes ``.cuda()`` (bts.py:140,143);
  * a ``torchvision.models`` stub (torchvision is not installed) backed by
    ``bts_amd.tv_models`` -- same child/key names, random init instead of
    ``pretrained=True`` (bts.py:272-298).
"""
import importlib.util
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("BTS_REFERENCE_ROOT", "/root/reference")
REF_BTS = os.path.join(REF_ROOT, "pytorch", "bts.py")


def available():
    return os.path.isfile(REF_BTS)


def install_shims():
    if not torch.cuda.is_available() and not getattr(torch.Tensor, "_bts_cuda_shim", False):
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.Tensor._bts_cuda_shim = True
    if "torchvision" not in sys.modules:
        try:
            import torchvision  # noqa: F401
        except ImportError:
            repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            if repo not in sys.path:
                sys.path.insert(0, repo)
            from bts_amd import tv_models
            tv = types.ModuleType("torchvision")
            tv.models = tv_models
            sys.modules["torchvision"] = tv
            sys.modules["torchvision.models"] = tv_models


def load_reference():
    """Return the reference module object (``ref.bts``, ``ref.BtsModel``, ``ref.silog_loss`` ...)."""
    if not available():
        raise FileNotFoundError(REF_BTS)
    install_shims()
    spec = importlib.util.spec_from_file_location("_reference_bts", REF_BTS)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
