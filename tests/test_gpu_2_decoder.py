"""GPU parity tests, decoder level: the drop-in ``bts`` / ``BtsModel`` / ``silog_loss`` against
(a) golden outputs produced by the unmodified reference (tests/golden, tools/make_golden.py) and
(b) the CPU oracle on seeded inputs, in train and eval mode, forward and backward.

Bar (north_star): outputs within 1e-4 relative of the reference's PyTorch CPU path in f32.
Gradients of deep parameters accumulate f32 round-off through ~40 layers and atomics; they are
held to 2e-3 relative (max-norm), stated here.  bf16 runs are throughput configurations and are
only sanity-bounded (5e-2).
"""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import bts_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def build(feat, nf, dataset, P, dtype=torch.float32, train=True):
    from bts_amd.model import bts
    md = 80.0 if dataset == "kitti" else 10.0
    dec = bts(NS(max_depth=md, dataset=dataset, encoder="densenet161_bts", bts_size=nf, decoder_dtype=dtype), feat, nf)
    dec.load_state_dict(P)
    dec.to(DEV)
    dec.train(train)
    return dec, md


def run(dec, feats, focal, gt, dataset):
    from bts_amd.model import silog_loss
    fs = [f.to(DEV).requires_grad_(True) for f in feats]
    outs = dec(fs, focal.to(DEV))
    mask = gt.to(DEV) > (1.0 if dataset == "kitti" else 0.1)
    loss = silog_loss(0.85)(outs[4], gt.to(DEV), mask)
    aux = sum((o * o).mean() for o in outs[:4])
    (loss + aux).backward()
    return fs, outs, loss, aux


@pytest.mark.parametrize("tag,train,ds", [("train_kitti", True, "kitti"), ("eval_nyu", False, "nyu")])
def test_decoder_small_golden(golden_dir, tag, train, ds):
    g = np.load("%s/decoder_small_%s.npz" % (golden_dir, tag))
    g0 = np.load("%s/decoder_small_train_kitti.npz" % golden_dir)
    feat, nf = [int(c) for c in g["feat"]], int(g["nf"])
    P = {k[2:]: torch.tensor(g0[k]) for k in g0.files if k.startswith("P/")}
    feats = [torch.tensor(g0["feat%d" % i]) for i in range(5)]
    dec, md = build(feat, nf, ds, P, train=train)
    fs, outs, loss, aux = run(dec, feats, torch.tensor(g["focal"]), torch.tensor(g["gt"]), ds)
    worst = {}
    for i, o in enumerate(outs):
        assert tuple(o.shape) == tuple(g["out%d" % i].shape)
        worst["out%d" % i] = rel(o, torch.tensor(g["out%d" % i]))
    worst["loss"] = abs(loss.item() - float(g["loss"])) / float(g["loss"])
    print("outputs:", {k: "%.2e" % v for k, v in worst.items()})
    assert max(worst.values()) < 1e-4
    gw = {}
    for n, p in dec.named_parameters():
        assert p.grad is not None, n
        gw[n] = rel(p.grad, torch.tensor(g["G/" + n]))
    for i, f in enumerate(fs):
        gw["feat%d" % i] = rel(f.grad, torch.tensor(g["gfeat%d" % i]))
    bad = {k: "%.2e" % v for k, v in gw.items() if v > 2e-3}
    print("worst grads:", sorted(gw.items(), key=lambda kv: -kv[1])[:5])
    assert not bad, bad
    for n, b in dec.named_buffers():
        if n.endswith("num_batches_tracked"):
            assert int(b) == int(g["B/" + n]), n
        else:
            assert rel(b, torch.tensor(g["B/" + n])) < 1e-4, n


def test_decoder_dn161_tiny_golden(golden_dir):
    """Real DenseNet161 channel widths (2208/384/192/96/96, bts_size 512), weights regenerated from the seed."""
    g = np.load("%s/decoder_dn161_tiny.npz" % golden_dir)
    if str(g["torch_version"]) != torch.__version__:
        pytest.skip("golden generated with torch %s (weights are regenerated from the seed)" % g["torch_version"])
    feat, nf = [int(c) for c in g["feat"]], int(g["nf"])
    B, H, W = [int(v) for v in g["shape"]]
    gen = torch.Generator().manual_seed(int(g["seed"]))
    P = O.make_decoder_params(feat, nf, gen, randomize_bn=False)
    feats = O.make_features(feat, B, H, W, gen)
    focal = O.synth_focal(B, "kitti")
    gt = O.synth_depth_gt(B, H, W, "kitti", gen)
    dec, md = build(feat, nf, "kitti", P)
    fs, outs, loss, aux = run(dec, feats, focal, gt, "kitti")
    for i, o in enumerate(outs):
        assert rel(o, torch.tensor(g["out%d" % i])) < 1e-4, i
    assert abs(loss.item() - float(g["loss"])) / float(g["loss"]) < 1e-4
    for n, p in dec.named_parameters():
        l2 = p.grad.double().norm().item()
        assert abs(l2 - float(g["Gl2/" + n])) <= 2e-3 * float(g["Gl2/" + n]) + 1e-7, n


def test_decoder_bf16_sane():
    """bf16 activation path: bounded deviation from the f32 oracle (throughput config, not a parity claim)."""
    feat, nf, B, H, W = [8, 8, 16, 24, 40], 128, 2, 64, 96
    gen = torch.Generator().manual_seed(31)
    P = O.make_decoder_params(feat, nf, gen, randomize_bn=True)
    feats = O.make_features(feat, B, H, W, gen)
    focal = O.synth_focal(B, "kitti")
    ref, _ = O.decoder_forward(P, feats, focal, 80.0, "kitti", True)
    dec, _ = build(feat, nf, "kitti", P, dtype=torch.bfloat16)
    gt = O.synth_depth_gt(B, H, W, "kitti", gen)
    fs, outs, loss, aux = run(dec, feats, focal, gt, "kitti")
    for o, r in zip(outs, ref):
        assert torch.isfinite(o).all()
        assert rel(o, r) < 5e-2
    assert all(torch.isfinite(p.grad).all() for p in dec.parameters())


@pytest.mark.parametrize("nf,feat", [(512, [96, 96, 192, 384, 2208]), (128, [8, 8, 16, 24, 40])])
@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
def test_fused_inference_chain_matches_layerwise(nf, feat, dt, tol, monkeypatch):
    """no-grad forward uses the fused reduction-chain + LPG kernel (csrc/lpg_chain.hip); it must agree with the
    layer-wise path (taken when gradients are recorded) and, in f32, with the oracle to 1e-4."""
    B, H, W = 2, 64, 96
    gen = torch.Generator().manual_seed(41)
    P = O.make_decoder_params(feat, nf, gen, randomize_bn=True)
    feats = O.make_features(feat, B, H, W, gen)
    focal = O.synth_focal(B, "kitti")
    dec, _ = build(feat, nf, "kitti", P, dtype=dt, train=False)
    fs = [f.to(DEV) for f in feats]
    from bts_amd import decoder as decoder_mod, profiler
    prof = profiler.enable()
    with torch.no_grad():
        fused = dec(fs, focal.to(DEV))
    names = {r[0] for r in prof.records}
    monkeypatch.setattr(decoder_mod, "FUSED_CHAIN_BWD", False)      # recorded pass: every chain layer by layer
    layerwise = dec([f.clone().requires_grad_(True) for f in fs], focal.to(DEV))
    names2 = {r[0] for r in prof.records} - names
    profiler.disable()
    assert any(n.startswith("lpg_head_chain_fwd") for n in names), names       # the fused kernel really ran
    assert not any(n.startswith("lpg_head_chain_fwd") for n in names2) and any(n.startswith("lpg_head_fwd") for n in names2)
    for a, b in zip(fused, layerwise):
        assert torch.isfinite(a).all()
        assert rel(a, b) < tol
    if dt == torch.float32:
        ref, _ = O.decoder_forward(P, feats, focal, 80.0, "kitti", False)
        for a, r in zip(fused, ref):
            assert rel(a, r) < 1e-4


@pytest.mark.parametrize("nf,feat", [(512, [96, 96, 192, 384, 2208]), (256, [16, 16, 32, 48, 80]), (128, [8, 8, 16, 24, 40])])
def test_fused_train_chain_matches_layerwise(nf, feat, monkeypatch):
    """bf16 training: the narrow LPG chains (reduc2x2 / reduc1x1 at bts_size 512) run as one fused forward and one
    fused recompute-backward kernel; outputs and every gradient must agree with the layer-wise schedule to bf16 noise."""
    from bts_amd import decoder as decoder_mod, profiler
    B, H, W = 2, 64, 96
    gen = torch.Generator().manual_seed(43)
    P = O.make_decoder_params(feat, nf, gen, randomize_bn=True)
    feats = O.make_features(feat, B, H, W, gen)
    focal = O.synth_focal(B, "kitti")
    gt = O.synth_depth_gt(B, H, W, "kitti", gen)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(decoder_mod, "FUSED_CHAIN_BWD", fused)
        dec, _ = build(feat, nf, "kitti", P, dtype=torch.bfloat16)
        prof = profiler.enable()
        fs, outs, loss, aux = run(dec, feats, focal, gt, "kitti")
        names = {r[0] for r in prof.records}
        profiler.disable()
        assert any(n.startswith("lpg_head_chain_bwd") for n in names) == fused, names
        res[fused] = ([o.detach() for o in outs], {n: p.grad for n, p in dec.named_parameters()}, [f.grad for f in fs])
    for a, b in zip(res[True][0], res[False][0]):
        assert rel(a, b) < 3e-2
    worst = {n: rel(g, res[False][1][n]) for n, g in res[True][1].items()}
    print(sorted(worst.items(), key=lambda kv: -kv[1])[:6])
    assert max(worst.values()) < 6e-2, sorted(worst.items(), key=lambda kv: -kv[1])[:6]
    for a, b in zip(res[True][2], res[False][2]):
        assert rel(a, b) < 6e-2


@pytest.mark.parametrize("dt,tol", [(torch.float32, 5e-5), (torch.bfloat16, 6e-2)])
def test_shared_input_bn_backward_matches_per_layer(dt, tol, monkeypatch):
    """The dense-ASPP first_bn layers (bts.py:51-66, 211-218) each normalise a prefix of the same concatenation; the decoder hands
    the backward of every shared tensor to one bts_bn_bwd_multi launch per tensor (decoder.MULTI_BN_BWD).  Outputs and every
    gradient must agree with the rounds 3-5 schedule (one bts_bn_bwd per BatchNorm), and the new launches really ran."""
    from bts_amd import decoder as decoder_mod, profiler
    feat, nf, B, H, W = [16, 16, 32, 48, 80], 256, 2, 64, 96
    gen = torch.Generator().manual_seed(47)
    P = O.make_decoder_params(feat, nf, gen, randomize_bn=True)
    feats = O.make_features(feat, B, H, W, gen)
    focal = O.synth_focal(B, "kitti")
    gt = O.synth_depth_gt(B, H, W, "kitti", gen)
    res = {}
    for multi in (True, False):
        monkeypatch.setattr(decoder_mod, "MULTI_BN_BWD", multi)
        dec, _ = build(feat, nf, "kitti", P, dtype=dt)
        prof = profiler.enable()
        fs, outs, loss, aux = run(dec, feats, focal, gt, "kitti")
        names = [r[0] for r in prof.records]
        profiler.disable()
        assert (names.count("bn_bwd_multi") > 0) == multi, names
        res[multi] = ([o.detach() for o in outs], {n: p.grad for n, p in dec.named_parameters()}, [f.grad for f in fs],
                      names.count("bn_bwd") + names.count("bn_bwd_multi"))
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b)              # the forward pass is the same launches
    worst = {n: rel(g, res[False][1][n]) for n, g in res[True][1].items()}
    assert max(worst.values()) < tol, sorted(worst.items(), key=lambda kv: -kv[1])[:6]
    for a, b in zip(res[True][2], res[False][2]):
        assert rel(a, b) < tol
    print("bn backward launches: multi %d, per layer %d" % (res[True][3], res[False][3]))


@pytest.mark.parametrize("B,H,W", [(1, 96, 160), (3, 32, 64), (1, 416, 544)])
def test_decoder_ragged_shapes_vs_oracle(B, H, W):
    """Odd grid sizes (H/32 x W/32 = 3x5, 1x2, 13x17 = BASELINE configs[0/1] 416x544), batch sizes that are not powers of
    two: tile tails of every kernel (pixel tiles, halo tiles, LPG rows) against the CPU oracle, f32, eval + train."""
    feat, nf = [8, 8, 16, 24, 40], 128
    gen = torch.Generator().manual_seed(B * 1000 + H)
    P = O.make_decoder_params(feat, nf, gen, randomize_bn=True)
    feats = O.make_features(feat, B, H, W, gen)
    focal = O.synth_focal(B, "nyu")
    for train in (False, True):
        ref, _ = O.decoder_forward(P, feats, focal, 10.0, "nyu", train)
        dec, _ = build(feat, nf, "nyu", P, train=train)
        outs = dec([f.to(DEV).requires_grad_(True) for f in feats], focal.to(DEV))
        for o, r in zip(outs, ref):
            assert tuple(o.shape) == (B, 1, H, W)
            assert rel(o, r) < 1e-4
        with torch.no_grad():
            outs2 = dec([f.to(DEV) for f in feats], focal.to(DEV))      # fused no-grad heads
        for o, r in zip(outs2, ref):
            assert rel(o, r) < 1e-4


@pytest.mark.parametrize("block", ["upconv", "atrous_first_bn", "atrous_plain", "reduction", "reduction_final"])
def test_standalone_blocks_vs_oracle(block):
    """The reference exposes its building blocks as working modules (bts.py:51-122); here they run the decoder's kernels on
    their own: output, input gradient, every parameter gradient and the BN running statistics vs the oracle's functions."""
    from bts_amd.model import atrous_conv, reduction_1x1, upconv
    gen = torch.Generator().manual_seed(11)
    N, H, W = 2, 12, 20
    if block == "upconv":
        mod = upconv(16, 24)
        x = torch.randn(N, 16, H, W, generator=gen)
    elif block.startswith("atrous"):
        mod = atrous_conv(32, 16, 6, apply_bn_first=block == "atrous_first_bn")
        x = torch.randn(N, 32, H, W, generator=gen)
    else:
        mod = reduction_1x1(64, 32, 10.0, is_final=block == "reduction_final") if block == "reduction" else \
            reduction_1x1(32, 16, 10.0, is_final=True)
        x = torch.randn(N, mod.reduc[0][0].in_channels, H, W, generator=gen)
    with torch.no_grad():
        for n, p in mod.named_parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * (0.3 if p.dim() > 1 else 1.0) + (1.0 if n.endswith("bn.weight") or ".2.weight" in n else 0.0))
        for n, b in mod.named_buffers():
            if "running_var" in n:
                b.copy_(torch.rand(b.shape, generator=gen) + 0.5)
            elif "running_mean" in n:
                b.copy_(torch.randn(b.shape, generator=gen) * 0.1)
    P = {"m." + k: v.detach().clone() for k, v in mod.state_dict().items()}
    Pr = {k: (v.requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    st = O.BNState(True)
    if block == "upconv":
        ref = O._upconv(xr, Pr["m.conv.weight"])
    elif block.startswith("atrous"):
        ref = O._atrous(xr, Pr, "m", 6, st, block == "atrous_first_bn")
    else:
        ref = O.reduction_chain(xr, Pr, "m", 10.0, block == "reduction_final")
    gy = torch.randn(ref.shape, generator=gen)
    ref.backward(gy)
    mod.to(DEV).train()
    xd = x.to(DEV).requires_grad_(True)
    out = mod(xd)
    assert out.shape == ref.shape
    assert rel(out, ref) < 1e-4
    out.backward(gy.to(DEV))
    assert rel(xd.grad, xr.grad) < 1e-4
    for n, p in mod.named_parameters():
        assert p.grad is not None, n
        assert rel(p.grad, Pr["m." + n].grad) < 1e-4, n
    for k, v in st.updates.items():
        assert rel(mod.state_dict()[k[2:]], v) < 1e-4, k
    with torch.no_grad():                                   # the no-grad path keeps no tape
        out2 = mod.eval()(x.to(DEV))
    assert out2.shape == ref.shape and torch.isfinite(out2).all()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_channels_last_features_are_taken_in_place(dt):
    """A stock encoder run in torch.channels_last hands over features that already ARE the decoder's NHWC layout
    (DecoderRun.feature): skips are read in place, the dense feature takes its ReLU (bts.py:198) in one same-layout pass,
    gradients return as channels-last views.  Same kernels on the same values as the NCHW entry: outputs, loss and every
    gradient must be bit-identical; a feature modified in place before backward() is refused."""
    from bts_amd._lib import BtsAmdError
    from bts_amd.model import silog_loss
    feat, nf, B, H, W = [96, 96, 192, 384, 2208], 512, 2, 96, 160
    gen = torch.Generator().manual_seed(5)
    P = O.make_decoder_params(feat, nf, gen, randomize_bn=True)
    feats = [f.to(dt) for f in O.make_features(feat, B, H, W, gen)]
    focal = O.synth_focal(B, "kitti").to(DEV)
    gt = O.synth_depth_gt(B, H, W, "kitti", gen).to(DEV)

    def go(cl, poke=False):
        dec, _ = build(feat, nf, "kitti", P, dtype=dt)
        fs = [f.to(DEV) for f in feats]
        if cl:
            fs = [f.contiguous(memory_format=torch.channels_last) for f in fs]
            assert all(not f.is_contiguous() for f in fs)
        fs = [f.requires_grad_(True) for f in fs]
        outs = dec(fs, focal)
        loss = silog_loss(0.85)(outs[4], gt, gt > 1.0) + sum((o * o).mean() for o in outs[:4])
        if poke:
            with torch.no_grad():
                fs[1].add_(1.0)
        loss.backward()
        return fs, outs, loss, {k: p.grad.clone() for k, p in dec.named_parameters()}
    fa, oa, la, ga = go(False)
    fb, ob, lb, gb = go(True)
    assert torch.equal(la, lb)
    for u, v in zip(oa, ob):
        assert torch.equal(u, v)
    for u, v in zip(fa, fb):
        assert v.grad.shape == u.grad.shape and torch.equal(u.grad, v.grad.contiguous())
        assert v.grad.permute(0, 2, 3, 1).is_contiguous()                 # came back as a view of the NHWC gradient buffer
    # weight gradients of the split-K kernels accumulate with f32 atomics (order varies run to run): same bound as two NCHW runs
    for k in ga:
        assert rel(gb[k], ga[k]) < 1e-5, k
    with pytest.raises((BtsAmdError, RuntimeError)):
        go(True, poke=True)
