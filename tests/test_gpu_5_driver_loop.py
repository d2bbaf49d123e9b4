"""The reference drivers' control flow ON THE HIP PATH (-m gpu).  /root/reference does not exist on the GPU box, so the
unmodified scripts cannot run there (they do run, on CPU plumbing, in tests/test_reference_drivers.py); tools/ref_loop.py
restates what `bts_main.py:322-554` and `bts_test.py:84-128` do, and this test drives the drop-in module -- loaded from
dropin/bts.py exactly as `from bts import *` would bind it -- through it: DataParallel wrap, `.cuda()` of the batch dicts,
the per-step poly learning rate, `'{:.12f}'.format(loss)`, `loss.cpu().item()`, checkpoint write / resume with `module.`-
prefixed keys and torch.optim.AdamW state, then the test loop's `.cpu().numpy().squeeze()` of all five outputs, which are
checked against the CPU oracle run on the checkpoint the HIP path trained."""
import importlib.util
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import bts_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_dropin():
    spec = importlib.util.spec_from_file_location("bts_dropin_under_test", os.path.join(ROOT, "dropin", "bts.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _batches(n, B, H, W, dataset, seed):
    gen = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        out.append({"image": torch.randn(B, 3, H, W, generator=gen), "focal": O.synth_focal(B, dataset),
                    "depth": O.synth_depth_gt(B, H, W, dataset, gen)})
    return out


def test_reference_driver_loops_on_the_hip_path(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import ref_loop
    finally:
        sys.path.pop(0)
    bts_mod = _load_dropin()
    for name in ("BtsModel", "silog_loss", "weights_init_xavier", "bn_init_as_tf"):
        assert hasattr(bts_mod, name), name
    H, W, B = 64, 96, 2
    args = NS(encoder="densenet121_bts", dataset="kitti", max_depth=80.0, bts_size=512, variance_focus=0.85, learning_rate=1e-4,
              end_learning_rate=-1, weight_decay=1e-2, adam_eps=1e-3, num_epochs=1, save_freq=2, log_directory=str(tmp_path),
              bn_no_track_stats=False, fix_first_conv_blocks=False, fix_first_conv_block=False, checkpoint_path="")
    lines = []
    torch.manual_seed(0)
    batches = _batches(4, B, H, W, "kitti", 11)
    # ---- uninterrupted run: 4 steps, checkpoints at 2 and 4
    model, opt, gs, losses = ref_loop.train_loop(bts_mod, args, batches, log=lines.append)
    assert gs == 4 and len(losses) == 4 and all(np.isfinite(losses))
    assert os.path.isfile(tmp_path / "model-2") and os.path.isfile(tmp_path / "model-4")
    assert any("loss: " in ln and "lr: " in ln for ln in lines)
    ck2 = torch.load(tmp_path / "model-2")
    assert set(ck2) == {"global_step", "model", "optimizer"} and ck2["global_step"] == 2
    assert all(k.startswith("module.") for k in ck2["model"])
    assert any(k.startswith("module.decoder.reduc8x8.reduc.") for k in ck2["model"])
    frozen = [n for n, p in model.module.encoder.named_parameters() if not p.requires_grad]
    assert frozen and all(("conv0" in n) or ("norm" in n) for n in frozen)
    # ---- resume from step 2 and replay steps 3-4: same losses as the uninterrupted run (the decoder is deterministic; MIOpen's
    # encoder kernels use atomics, hence a tolerance)
    torch.manual_seed(123)                                       # construction-time init must not matter after the load
    model_r, opt_r, gs_r, losses_r = ref_loop.train_loop(bts_mod, NS(**{**vars(args), "save_freq": 0}), batches[2:],
                                                        checkpoint_path=str(tmp_path / "model-2"), log=lines.append,
                                                        steps_per_epoch=len(batches))
    assert gs_r == 4
    # (the poly schedule restarts its step index from the checkpoint's global_step, bts_main.py:383, 456-458)
    for a, b in zip(losses[2:], losses_r):
        assert abs(a - b) / abs(a) < 2e-3, (losses, losses_r)
    # ---- test loop on the checkpoint the HIP path wrote
    targs = NS(**{**vars(args), "checkpoint_path": str(tmp_path / "model-4")})
    samples = _batches(2, 1, H, W, "kitti", 29)
    preds = ref_loop.test_loop(bts_mod, targs, samples, log=lines.append)
    assert len(preds) == 5 and all(len(p) == 2 for p in preds)
    for lst in preds:
        for a in lst:
            assert isinstance(a, np.ndarray) and a.shape == (H, W) and a.dtype == np.float32 and np.isfinite(a).all()
    assert all((d > 0).all() and (d < 80.0 * 721.5377 / 715.0873 + 1e-3).all() for d in preds[0])
    # ---- checker: the CPU oracle on the same checkpoint (encoder = the same stock module on CPU, decoder = oracle formulas)
    ck4 = torch.load(tmp_path / "model-4")
    cpu_model = bts_mod.BtsModel(params=targs)
    cpu_model.load_state_dict({k[len("module."):]: v.cpu() for k, v in ck4["model"].items()})
    cpu_model.eval()
    P = {k: v for k, v in cpu_model.decoder.state_dict().items()}
    with torch.no_grad():
        for i, s in enumerate(samples):
            feats = cpu_model.encoder(s["image"])
            ref, _ = O.decoder_forward(P, feats, s["focal"], 80.0, "kitti", False)
            for got, want in zip((preds[1][i], preds[2][i], preds[3][i], preds[4][i], preds[0][i]), ref):
                w = want.squeeze().numpy()
                err = np.abs(got - w).max() / np.abs(w).max()
                assert err < 1e-4, err
