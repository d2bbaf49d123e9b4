"""Drop-in surface of bts_amd.model vs the reference's module (CPU; no kernels run)."""
from types import SimpleNamespace as NS

import pytest
import torch
import torch.nn as nn

from oracle import ref_loader


def _params(enc="densenet161_bts"):
    return NS(encoder=enc, max_depth=80.0, dataset="kitti", bts_size=512)


@pytest.mark.parametrize("enc", ["densenet121_bts", "densenet161_bts", "resnet50_bts", "resnext101_bts", "mobilenetv2_bts"])
def test_state_dict_keys_match_reference(enc):
    from bts_amd.model import BtsModel
    m = BtsModel(_params(enc))
    keys = list(m.state_dict().keys())
    assert sum(k.startswith("decoder.") for k in keys) in (110,)
    if ref_loader.available():
        ref = ref_loader.load_reference()
        r = ref.BtsModel(_params(enc))
        rsd = r.state_dict()
        assert keys == list(rsd.keys())
        assert all(m.state_dict()[k].shape == rsd[k].shape for k in keys)
        # optimizer state is index-keyed: parameter registration order must match (bts_main.py:371-373)
        assert [n for n, _ in m.named_parameters()] == [n for n, _ in r.named_parameters()]


def test_driver_hooks_work_on_containers():
    """weights_init_xavier / bn_init_as_tf / set_misc-style freezing operate on our module tree."""
    from bts_amd.model import BtsModel, bn_init_as_tf, weights_init_xavier
    m = BtsModel(_params("densenet121_bts"))
    before = m.decoder.conv1[0].weight.clone()
    m.decoder.apply(weights_init_xavier)
    assert not torch.equal(before, m.decoder.conv1[0].weight)
    m.train()
    m.apply(bn_init_as_tf)
    assert all(not b.training for b in m.modules() if isinstance(b, nn.BatchNorm2d))
    fixing = ["conv0", "norm"]   # bts_main.py:237
    for name, child in m.named_children():
        if "encoder" not in name:
            continue
        for n2, p in child.named_parameters():
            if any(x in n2 for x in fixing):
                p.requires_grad = False
    assert not m.encoder.base_model.conv0.weight.requires_grad
    assert m.decoder.bn5.weight.requires_grad


def test_no_cpu_fallback():
    """The product path refuses to run without a HIP device (no silent eager fallback)."""
    from bts_amd._lib import BtsAmdError
    from bts_amd.model import bts, local_planar_guidance, silog_loss
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(BtsAmdError):
        local_planar_guidance(8)(torch.randn(1, 4, 2, 2), torch.ones(1))
    with pytest.raises(BtsAmdError):
        silog_loss(0.85)(torch.rand(1, 1, 4, 4) + 1, torch.rand(1, 1, 4, 4) + 1, torch.ones(1, 1, 4, 4, dtype=torch.bool))
    dec = bts(_params(), [96, 96, 192, 384, 2208], 512)
    feats = [torch.randn(1, c, 32 >> (i + 1), 64 >> (i + 1)) for i, c in enumerate([96, 96, 192, 384, 2208])]
    with pytest.raises(BtsAmdError):
        dec(feats, torch.ones(1))
