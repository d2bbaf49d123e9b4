"""Minimal stand-in for the parts of torchvision the reference drivers import (pytorch/bts.py:272,
bts_dataloader.py:21, bts_main.py:135-138) -- installed by tools/run_reference.py ONLY when the real package is
absent from the image.  `models` = bts_amd.tv_models (same architectures and state-dict names as torchvision's
densenet.py / resnet.py / mobilenetv2.py, random init); `transforms` = Compose + Normalize."""
from . import models, transforms  # noqa: F401

__version__ = "0.0+bts_amd_shim"
