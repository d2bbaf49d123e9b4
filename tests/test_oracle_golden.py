"""Pin the CPU oracle (oracle/bts_oracle.py) against golden vectors produced by the UNMODIFIED
reference (tools/make_golden.py, run in the build container) -- and, where /root/reference is
present, against the live reference.  CPU only."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import bts_oracle as O
from oracle import ref_loader


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("k", [8, 4, 2])
def test_lpg_matches_reference_golden(golden_dir, k):
    g = np.load(golden_dir + "/lpg.npz")
    eq = torch.tensor(g["k%d_eq" % k], requires_grad=True)
    out = O.lpg(eq, k)
    assert torch.equal(out, torch.tensor(g["k%d_out" % k]))          # bit-identical on CPU
    out.backward(torch.tensor(g["k%d_gout" % k]))
    assert rel(eq.grad, g["k%d_geq" % k]) < 1e-5


@pytest.mark.parametrize("tag", ["r8_128", "r2_128", "r1_128", "r8_512", "r1_512"])
def test_reduction_chain_matches_golden(golden_dir, tag):
    g = np.load(golden_dir + "/reduction.npz")
    final = tag.startswith("r1")
    names = [k[len(tag) + 3:] for k in g.files if k.startswith(tag + "_w_")]
    P = {"x." + n: torch.tensor(g[tag + "_w_" + n], requires_grad=True) for n in names}
    x = torch.tensor(g[tag + "_x"], requires_grad=True)
    y = O.reduction_chain(x, P, "x", 80.0, final)
    assert rel(y, g[tag + "_y"]) < 1e-6
    y.backward(torch.tensor(g[tag + "_gy"]))
    assert rel(x.grad, g[tag + "_gx"]) < 1e-5
    for n in names:
        assert rel(P["x." + n].grad, g[tag + "_gw_" + n]) < 1e-5


@pytest.mark.parametrize("tag", ["kitti", "nyu"])
def test_silog_matches_golden(golden_dir, tag):
    g = np.load(golden_dir + "/silog.npz")
    est = torch.tensor(g[tag + "_est"], requires_grad=True)
    gt = torch.tensor(g[tag + "_gt"])
    loss = O.silog(est, gt, gt > (1.0 if tag == "kitti" else 0.1), float(g[tag + "_vf"]))
    assert rel(loss, g[tag + "_loss"]) < 1e-6
    loss.backward()
    assert rel(est.grad, g[tag + "_gest"]) < 1e-5


@pytest.mark.parametrize("tag,train,ds", [("train_kitti", True, "kitti"), ("eval_nyu", False, "nyu")])
def test_decoder_matches_golden(golden_dir, tag, train, ds):
    g = np.load("%s/decoder_small_%s.npz" % (golden_dir, tag))
    g0 = np.load("%s/decoder_small_train_kitti.npz" % golden_dir)
    P = {k[2:]: torch.tensor(g0[k]) for k in g0.files if k.startswith("P/")}
    for k, v in P.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    feats = [torch.tensor(g0["feat%d" % i], requires_grad=True) for i in range(5)]
    md = 80.0 if ds == "kitti" else 10.0
    outs, upd = O.decoder_forward(P, feats, torch.tensor(g["focal"]), md, ds, train)
    for i, o in enumerate(outs):
        assert rel(o, g["out%d" % i]) < 5e-6, i
    gt = torch.tensor(g["gt"])
    loss = O.silog(outs[4], gt, gt > (1.0 if ds == "kitti" else 0.1), 0.85)
    assert rel(loss, g["loss"]) < 1e-5
    (loss + sum((o * o).mean() for o in outs[:4])).backward()
    worst = max(rel(P[k[2:]].grad, g[k]) for k in g.files if k.startswith("G/"))
    assert worst < 2e-4, worst
    for i, f in enumerate(feats):
        assert rel(f.grad, g["gfeat%d" % i]) < 2e-4
    for k, v in upd.items():
        assert rel(v, g["B/" + k]) < 1e-5, k


def test_param_factory_matches_reference_state_dict():
    """Key names / order / shapes of the decoder state dict (110 tensors for DenseNet161)."""
    specs = O.decoder_param_specs([96, 96, 192, 384, 2208], 512)
    assert len(specs) == 110
    if ref_loader.available():
        ref = ref_loader.load_reference()
        dec = ref.bts(NS(max_depth=80.0, dataset="kitti", encoder="densenet161_bts", bts_size=512),
                      [96, 96, 192, 384, 2208], 512)
        sd = dec.state_dict()
        assert list(sd.keys()) == [k for k, _, _ in specs]
        assert all(tuple(sd[k].shape) == tuple(s) for k, s, _ in specs)


@pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not present (GPU box)")
def test_oracle_vs_live_reference_c1_plumbing(golden_dir):
    """configs[0]: BtsModel densenet121 416x544 batch-1 CPU forward of the reference reproduces the golden samples."""
    ref = ref_loader.load_reference()
    g = np.load(golden_dir + "/model_c1_densenet121.npz")
    torch.manual_seed(int(g["model_seed"]))
    model = ref.BtsModel(NS(encoder="densenet121_bts", max_depth=10.0, dataset="nyu", bts_size=512))
    model.decoder.apply(ref.weights_init_xavier)
    model.eval()
    x = torch.randn(1, 3, 416, 544, generator=torch.Generator().manual_seed(int(g["input_seed"])))
    with torch.no_grad():
        outs = model(x, O.synth_focal(1, "nyu"))
        feats = model.encoder(x)
        P = model.decoder.state_dict()
        outs_o, _ = O.decoder_forward(P, feats, O.synth_focal(1, "nyu"), 10.0, "nyu", False)
    for i, (o, oo) in enumerate(zip(outs, outs_o)):
        assert rel(o[:, :, ::8, ::8], g["out%d_s8" % i]) < 1e-4
        assert rel(oo, o) < 1e-4
    assert int(g["n_params"]) == sum(p.numel() for p in model.parameters())


def _c_oracle():
    import ctypes
    import os

    import __graft_entry__
    so = os.path.join(os.path.dirname(os.path.abspath(__graft_entry__.__file__)), "oracle", "_build", "liblpg_oracle.so")
    if not os.path.exists(so):
        __graft_entry__.build()
    return ctypes.CDLL(so)


@pytest.mark.parametrize("k", [8, 4, 2])
def test_c_oracle_lpg_matches_reference_golden(golden_dir, k):
    """oracle/lpg_oracle.c (TF-op layout) reproduces the reference's PyTorch LPG bit-for-bit."""
    import ctypes
    lib = _c_oracle()
    g = np.load(golden_dir + "/lpg.npz")
    eq = np.ascontiguousarray(g["k%d_eq" % k].transpose(0, 2, 3, 1))        # NCHW -> [B,h,w,4]
    B, h, w, _ = eq.shape
    out = np.empty((B, h * k, w * k), dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.lpg_forward_c(eq.ctypes.data_as(fp), out.ctypes.data_as(fp), B, h, w, k, ctypes.c_float(1.0))
    assert np.array_equal(out, g["k%d_out" % k])
    gout = np.ascontiguousarray(g["k%d_gout" % k])
    geq = np.empty_like(eq)
    lib.lpg_backward_c(gout.ctypes.data_as(fp), eq.ctypes.data_as(fp), geq.ctypes.data_as(fp), B, h, w, k, ctypes.c_float(1.0))
    assert rel(geq.transpose(0, 3, 1, 2), g["k%d_geq" % k]) < 1e-5


def test_c_oracle_silog_matches_reference_golden(golden_dir):
    import ctypes
    lib = _c_oracle()
    lib.silog_c.restype = ctypes.c_double
    g = np.load(golden_dir + "/silog.npz")
    est, gt = np.ascontiguousarray(g["kitti_est"]), np.ascontiguousarray(g["kitti_gt"])
    mask = np.ascontiguousarray((gt > 1.0).astype(np.uint8))
    fp = ctypes.POINTER(ctypes.c_float)
    v = lib.silog_c(est.ctypes.data_as(fp), gt.ctypes.data_as(fp), mask.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)),
                    ctypes.c_long(est.size), ctypes.c_double(float(g["kitti_vf"])))
    assert abs(v - float(g["kitti_loss"])) / float(g["kitti_loss"]) < 1e-5


@pytest.mark.parametrize("tag", ["kitti", "nyu"])
def test_eval_oracle_matches_reference_compute_errors(golden_dir, tag):
    """oracle/eval_oracle.py vs the reference's own compute_errors (bts_main.py:143-165; golden written by executing the
    function's source from the unmodified file) on the masked pixel lists online_eval() would build."""
    import numpy as np
    from oracle import eval_oracle as E
    g = np.load("%s/eval.npz" % golden_dir)
    gh, gw, ph, pw, md, kb, garg, eig, ds = E.EVAL_CASES[tag]
    pred, gt = E.synth_eval_case(tag)
    chk = g[tag + "_checksum"]
    assert abs(float(pred[np.isfinite(pred)].astype(np.float64).sum()) - chk[0]) <= 1e-9 * abs(chk[0])      # same inputs as the generator
    assert abs(float(gt.astype(np.float64).sum()) - chk[1]) <= 1e-9 * abs(chk[1])
    pf, valid = E.eval_prepare(pred, gt, 1e-3, md, ds, kb, garg, eig)
    assert int(valid.sum()) == int(g[tag + "_nvalid"])
    assert np.isfinite(pf).all() and pf.min() >= np.float32(1e-3) and pf.max() <= md
    m = np.array(E.compute_errors(gt[valid], pf[valid]), dtype=np.float64)
    assert np.array_equal(m, g[tag + "_measures"])                   # same numpy ops in the same order: bit-identical


def test_eval_oracle_uint16_payload():
    import numpy as np
    from oracle import eval_oracle as E
    d = np.array([[0.0, 0.0039, 1.0, 79.999, 80.0]], dtype=np.float32)
    assert E.depth_to_uint16(d, "kitti").tolist() == [[0, 0, 256, 20479, 20480]]
    assert E.depth_to_uint16(d[:, :3], "nyu").tolist() == [[0, 3, 1000]]


def test_preprocess_oracle_matches_reference_methods(golden_dir):
    """oracle/data_oracle.py vs random_crop / train_preprocess / augment_image executed from the unmodified
    bts_dataloader.py (tests/golden/preprocess.npz): bit-identical on 16 seeded cases covering flip x augment x dataset;
    and bts_amd.dataops.draw_train_params consumes the generators in the same order."""
    import random

    import numpy as np
    from bts_amd import dataops
    from oracle import data_oracle as D
    g = np.load("%s/preprocess.npz" % golden_dir)
    img, dep = g["image_u8"], g["depth_raw"]
    seen = set()
    for ds in ("kitti", "nyu"):
        for seed in range(1, 9):
            p = D.draw_train_params(seed, 48, 80, 32, 64, ds)
            chw, d, hwc = D.preprocess_train(img, dep, p, 32, 64, ds)
            assert np.array_equal(hwc, g["%s_%d_image" % (ds, seed)]), (ds, seed)
            assert np.array_equal(d.transpose(1, 2, 0), g["%s_%d_depth" % (ds, seed)]), (ds, seed)
            assert chw.shape == (3, 32, 64) and chw.dtype == np.float32
            seen.add((p["flip"], p["augment"]))
            random.seed(seed)
            np.random.seed(seed)
            q = dataops.draw_train_params(48, 80, 32, 64, ds)
            assert (q.crop_x, q.crop_y, q.flip, q.augment) == (p["crop_x"], p["crop_y"], p["flip"], p["augment"])
            assert q.gamma == np.float32(p["gamma"]) and q.brightness == np.float32(p["brightness"])
            assert list(q.color) == [float(c) for c in p["colors"]]
    assert seen == {(0, 0), (0, 1), (1, 0), (1, 1)}
