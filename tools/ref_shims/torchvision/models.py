from bts_amd.tv_models import *  # noqa: F401,F403
from bts_amd.tv_models import densenet121, densenet161, mobilenet_v2, resnet50, resnet101, resnext50_32x4d, resnext101_32x8d  # noqa: F401
