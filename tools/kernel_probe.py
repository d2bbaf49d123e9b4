#!/usr/bin/env python
"""Micro-benchmark of individual HIP kernels at BASELINE.json shapes (GPU box only).

Used three ways:  plain (`python tools/kernel_probe.py`) prints one JSON line per case with the
event-timed average launch time and the achieved TFLOP/s / GB/s against ALGORITHMIC work;
under `rocprofv3 --kernel-trace --stats` for the per-kernel summary committed in profiles/;
under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) for HBM traffic.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bts_amd import ops  # noqa: E402
from bts_amd._lib import ACT_ELU, ACT_RELU  # noqa: E402
from bts_amd.conv import ConvLayer  # noqa: E402

DEV = "cuda"


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def timeit_rot(fns, iters):
    """Like timeit, but cycles through `fns` (same kernel on DIFFERENT buffers, together > 512 MB) and keeps every
    returned tensor alive until the next lap, so neither inputs nor outputs of a launch can still sit in the 256 MiB
    Infinity Cache from the previous one: the rate it reports is an HBM rate (MI355X_MICROARCH.md, Infinity Cache)."""
    keep = [fn() for fn in fns]
    torch.cuda.synchronize()
    laps = max(1, (iters + len(fns) - 1) // len(fns))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(laps):
        for i, fn in enumerate(fns):
            keep[i] = fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (laps * len(fns)) * 1e-3


def lpg_case_rot(B, H, W, k, iters, footprint=768 << 20):
    """LPG op / fused head at one shape with a rotating working set (HBM-resident numbers)."""
    h, w = H // k, W // k
    per = B * H * W * 4 * 2 + B * h * w * 16 * 2
    nrot = max(2, (footprint + per - 1) // per)
    raws = [torch.randn(B, h, w, 4, device=DEV) for _ in range(nrot)]
    gs = [torch.randn(B, H, W, device=DEV) for _ in range(nrot)]
    out = []
    tag = "k=%d B=%d %dx%d rot%d" % (k, B, H, W, nrot)
    t = timeit_rot([(lambda r=r: ops.lpg_head_fwd(r, k, 80.0)) for r in raws], iters)
    byts = B * h * w * (16 + 4 * k * k)
    out.append(dict(case="lpg_head_fwd " + tag, sec=t, alg_gbs=byts / t / 1e9, alg_bytes=byts, hbm_resident=True))
    t = timeit_rot([(lambda r=r, g=g: ops.lpg_head_bwd(r, g, k, 80.0, torch.bfloat16, 8)) for r, g in zip(raws, gs)], iters)
    byts = B * h * w * (16 + 4 * k * k + 16)
    out.append(dict(case="lpg_head_bwd " + tag, sec=t, alg_gbs=byts / t / 1e9, alg_bytes=byts, hbm_resident=True))
    t = timeit_rot([(lambda r=r: ops.lpg_fwd(r, k)) for r in raws], iters)
    byts = B * H * W * 4 * (1 + 4.0 / (k * k))
    out.append(dict(case="lpg_op_fwd " + tag, sec=t, alg_gbs=byts / t / 1e9, alg_bytes=byts, hbm_resident=True))
    t = timeit_rot([(lambda r=r, g=g: ops.lpg_bwd(g, r, k)) for r, g in zip(raws, gs)], iters)
    byts = B * H * W * 4 * (1 + 8.0 / (k * k))
    out.append(dict(case="lpg_op_bwd " + tag, sec=t, alg_gbs=byts / t / 1e9, alg_bytes=byts, hbm_resident=True))
    return out


def conv_case(name, dt, cout, segc, kk, dil, up, N, H, W, iters, which):
    L = ConvLayer(name, cout, segc, kk, dil, up)
    v = 4 if dt == torch.float32 else 8
    segs = [torch.randn(N, H, W, (c + v - 1) // v * v, device=DEV).to(dt) for c in segc]
    k = 3 if kk == 9 else 1
    w = torch.randn(cout, sum(segc), k, k, device=DEV) * 0.05
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    cp = (cout + v - 1) // v * v
    out = torch.empty(N, Ho, Wo, cp, dtype=dt, device=DEV)
    wp = L.pack_fwd(w, dt)
    flops = 2.0 * N * H * W * L.nphase * L.T * sum(segc) * cout
    esz = 4 if dt == torch.float32 else 2
    res = []
    if "fwd" in which:
        t = timeit(lambda: L.forward(segs, wp, out, ACT_ELU), iters)
        byts = (N * H * W * sum(segc) + N * Ho * Wo * cout + cout * sum(segc) * kk) * esz
        res.append(dict(case=name + ".fwd", dtype=str(dt), sec=t, tflops=flops / t / 1e12, alg_gbs=byts / t / 1e9))
    if "wgrad" in which:
        dz = torch.randn(N, Ho, Wo, cp, device=DEV).to(dt)
        t = timeit(lambda: L.wgrad(segs, dz), iters)
        res.append(dict(case=name + ".wgrad(+zero,unpack)", dtype=str(dt), sec=t, tflops=flops / t / 1e12))
    if "wgradk" in which:      # the weight-gradient kernel alone, into a caller-zeroed packed buffer (no memset, no unpack)
        dz = torch.randn(N, Ho, Wo, cp, device=DEV).to(dt)
        tb = L.tables(dt, dz.device)
        dwp = torch.zeros((cout, L.nphase * L.T, tb["ktot"]), dtype=torch.float32, device=DEV)
        t = timeit(lambda: L.wgrad_packed(segs, dz, dwp), iters)
        res.append(dict(case=name + ".wgrad_kernel", dtype=str(dt), sec=t, tflops=flops / t / 1e12))
    if "dgrad" in which:
        dz = torch.randn(N, Ho, Wo, cp, device=DEV).to(dt)
        wd = L.pack_dgrad(w, dt, 0)
        gx = torch.empty_like(segs[0])
        t = timeit(lambda: L.dgrad(dz, wd, 0, gx, False), iters)
        f0 = 2.0 * N * H * W * len(L.taps) * segc[0] * cout
        res.append(dict(case=name + ".dgrad0", dtype=str(dt), sec=t, tflops=f0 / t / 1e12))
    return res


def lpg_case(B, H, W, k, iters):
    h, w = H // k, W // k
    raw = torch.randn(B, h, w, 4, device=DEV)
    g = torch.randn(B, H, W, device=DEV)
    t = timeit(lambda: ops.lpg_head_fwd(raw, k, 80.0), iters)
    byts = B * h * w * (16 + 4 * k * k)
    out = [dict(case="lpg_head_fwd k=%d B=%d %dx%d" % (k, B, H, W), sec=t, alg_gbs=byts / t / 1e9, alg_bytes=byts)]
    t = timeit(lambda: ops.lpg_head_bwd(raw, g, k, 80.0, torch.bfloat16, 8), iters)
    byts = B * h * w * (16 + 4 * k * k + 16)
    out.append(dict(case="lpg_head_bwd k=%d B=%d %dx%d" % (k, B, H, W), sec=t, alg_gbs=byts / t / 1e9, alg_bytes=byts))
    eq = torch.randn(B, h, w, 4, device=DEV)
    t = timeit(lambda: ops.lpg_fwd(eq, k), iters)
    byts = B * H * W * 4 * (1 + 4.0 / (k * k))
    out.append(dict(case="lpg_op_fwd k=%d B=%d %dx%d" % (k, B, H, W), sec=t, alg_gbs=byts / t / 1e9, alg_bytes=byts))
    t = timeit(lambda: ops.lpg_bwd(g, eq, k), iters)
    byts = B * H * W * 4 * (1 + 8.0 / (k * k))
    out.append(dict(case="lpg_op_bwd k=%d B=%d %dx%d" % (k, B, H, W), sec=t, alg_gbs=byts / t / 1e9, alg_bytes=byts))
    return out


def chain_cases(B, H, W, dt, iters):
    """Fused inference LPG heads (reduction chain + plane + LPG in one pass) at bts_size 512."""
    from bts_amd import chain
    out = []
    esz = 4 if dt == torch.float32 else 2
    for k, c0, same, dims in ((8, 128, 1, [128, 128, 64, 32, 16, 8, 3]), (4, 128, 0, [128, 64, 32, 16, 8, 3]),
                               (2, 64, 0, [64, 32, 16, 8, 3]), (1, 32, 0, [32, 16, 8, 1])):
        h, w = H // k, W // k
        ws = [torch.randn(dims[i + 1], dims[i], 1, 1, device=DEV) * (1.0 / dims[i]) ** 0.5 for i in range(len(dims) - 1)]
        frags = chain.pack_chain(ws, dt)
        x = torch.randn(B, h, w, c0, device=DEV).to(dt)
        t = timeit(lambda: chain.chain_fwd(x, frags, c0, same, k, 80.0), iters)
        cells = B * h * w
        byts = cells * (c0 * esz + 4 * k * k)
        macs = cells * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
        out.append(dict(case="lpg_chain_fwd k=%d C0=%d %s B=%d %dx%d" % (k, c0, "bf16" if esz == 2 else "f32", B, H, W), sec=t,
                        alg_gbs=byts / t / 1e9, alg_bytes=byts, tflops=2 * macs / t / 1e12))
    return out


def misc_cases(B, H, W, iters):
    out = []
    est = torch.rand(B, 1, H, W, device=DEV) * 70 + 1
    gt = torch.rand(B, 1, H, W, device=DEV) * 70 + 0.5
    mask = gt > 1.0
    t = timeit(lambda: ops.silog_fwd(est, gt, mask, 0.85), iters)
    out.append(dict(case="silog_fwd B=%d %dx%d" % (B, H, W), sec=t, alg_gbs=B * H * W * 9 / t / 1e9))
    x = torch.randn(B, H // 2, W // 2, 64, device=DEV).to(torch.bfloat16)
    t = timeit(lambda: ops.bn_stats(x), iters)
    out.append(dict(case="bn_stats bf16 C=64 %dx%d" % (H // 2, W // 2), sec=t, alg_gbs=x.numel() * 2 / t / 1e9))
    sc = torch.rand(64, device=DEV)
    t = timeit(lambda: ops.affine_act(x, sc, sc, ACT_RELU), iters)
    out.append(dict(case="affine_relu bf16 C=64", sec=t, alg_gbs=x.numel() * 4 / t / 1e9))
    # evaluation side: online_eval metrics of a kitti batch (kb crop + garg crop), uint16 payload
    from bts_amd import evalops
    gtf = torch.rand(B, 1, 375, 1242, device=DEV) * 80
    acc = torch.zeros(10, device=DEV)
    t = timeit(lambda: evalops.compute_errors(est, gtf, 1e-3, 80.0, "kitti", True, True, False, eval_measures=acc), iters)
    win = evalops.crop_window("kitti", 375, 1242, True, False)
    out.append(dict(case="eval_errors kitti B=%d (garg window %dx%d)" % (B, win[1] - win[0], win[3] - win[2]), sec=t,
                    alg_gbs=B * (win[1] - win[0]) * (win[3] - win[2]) * 8 / t / 1e9))
    big = torch.rand(32, 1, 704, 1216, device=DEV) * 80
    t = timeit(lambda: evalops.depth_to_uint16(big, "kitti"), iters)
    out.append(dict(case="depth_to_uint16 B=32 704x1216", sec=t, alg_gbs=big.numel() * 6 / t / 1e9))
    # training-sample preprocessing: kitti kb-cropped 352x1216 sources, batch 8, all samples augmented
    from bts_amd import dataops
    src = torch.randint(0, 256, (B, 352, 1216, 3), dtype=torch.uint8, device=DEV)
    draw = torch.randint(0, 20000, (B, 352, 1216), dtype=torch.int32, device=DEV)
    ps = []
    for _ in range(B):
        p = dataops.draw_train_params(352, 1216, 352, 1216, "kitti")
        p.augment, p.gamma, p.brightness = 1, 1.05, 0.95
        ps.append(p)
    t = timeit(lambda: dataops.preprocess_train(src, draw, ps, 352, 1216, "kitti"), iters)
    out.append(dict(case="preprocess_train kitti B=%d 352x1216 (augmented)" % B, sec=t, alg_gbs=B * 352 * 1216 * (3 + 4 + 12 + 4) / t / 1e9))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--set", default="all", help="all | conv | fwd | mid | wgrad | narrow | pmc | lpg | lpgrot | chain | misc")
    a = ap.parse_args()
    bf, f32 = torch.bfloat16, torch.float32
    res = []
    if a.set in ("all", "conv"):
        B = 8   # DenseNet161-BTS decoder layers at 352x1216, batch 8 (BASELINE.json configs[2] per GPU)
        res += conv_case("conv5", bf, 512, [512, 384], 9, 1, False, B, 22, 76, a.iters, ("fwd", "dgrad", "wgrad"))
        res += conv_case("upconv5", bf, 512, [2208], 9, 1, True, B, 11, 38, a.iters, ("fwd", "wgrad"))
        res += conv_case("daspp_conv", bf, 128, [256, 128, 128, 128, 128, 128], 9, 1, False, B, 44, 152, a.iters, ("fwd", "wgrad"))
        res += conv_case("daspp12_3x3", bf, 128, [256], 9, 12, False, B, 44, 152, a.iters, ("fwd",))
        res += conv_case("conv3", bf, 128, [128, 96, 1], 9, 1, False, B, 88, 304, a.iters, ("fwd", "wgrad"))
        res += conv_case("conv2", bf, 64, [64, 96, 1], 9, 1, False, B, 176, 608, a.iters, ("fwd", "wgrad"))
        res += conv_case("upconv1", bf, 32, [64], 9, 1, True, B, 176, 608, a.iters, ("fwd", "wgrad"))
        res += conv_case("conv1", bf, 32, [32, 4], 9, 1, False, B, 352, 1216, a.iters, ("fwd", "wgrad"))
        res += conv_case("conv5_f32", f32, 512, [512, 384], 9, 1, False, B, 22, 76, max(2, a.iters // 3), ("fwd",))
    if a.set == "fwd":   # forward-only subset for A/B and ablation runs
        B = 8
        res += conv_case("conv5", bf, 512, [512, 384], 9, 1, False, B, 22, 76, a.iters, ("fwd",))
        res += conv_case("daspp_conv", bf, 128, [256, 128, 128, 128, 128, 128], 9, 1, False, B, 44, 152, a.iters, ("fwd",))
        res += conv_case("conv2", bf, 64, [64, 96, 1], 9, 1, False, B, 176, 608, a.iters, ("fwd",))
        res += conv_case("upconv1", bf, 32, [64], 9, 1, True, B, 176, 608, a.iters, ("fwd",))
        res += conv_case("conv1", bf, 32, [32, 4], 9, 1, False, B, 352, 1216, a.iters, ("fwd",))
    if a.set == "mid":      # layers whose 128x256 tiling gives < 2 workgroups per CU
        B = 8
        res += conv_case("daspp3x3", bf, 128, [256], 9, 6, False, B, 44, 152, a.iters, ("fwd", "dgrad"))
        res += conv_case("daspp1x1_24", bf, 256, [256, 192, 128, 128, 128, 128], 1, 1, False, B, 44, 152, a.iters, ("fwd", "dgrad"))
        res += conv_case("conv4", bf, 256, [256, 192], 9, 1, False, B, 44, 152, a.iters, ("fwd", "dgrad"))
        res += conv_case("upconv4", bf, 256, [512], 9, 1, True, B, 22, 76, a.iters, ("fwd",))
        res += conv_case("conv5", bf, 512, [512, 384], 9, 1, False, B, 22, 76, a.iters, ("fwd",))
        res += conv_case("conv3", bf, 128, [128, 96, 1], 9, 1, False, B, 88, 304, a.iters, ("fwd", "dgrad"))
        res += conv_case("daspp_conv", bf, 128, [256, 128, 128, 128, 128, 128], 9, 1, False, B, 44, 152, a.iters, ("fwd",))
    if a.set == "pmc":      # compact set for the HBM-traffic counter passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)
        B = 8
        res += conv_case("conv5", bf, 512, [512, 384], 9, 1, False, B, 22, 76, a.iters, ("fwd", "wgrad"))
        res += conv_case("daspp_conv", bf, 128, [256, 128, 128, 128, 128, 128], 9, 1, False, B, 44, 152, a.iters, ("fwd",))
        res += conv_case("conv3", bf, 128, [128, 96, 1], 9, 1, False, B, 88, 304, a.iters, ("fwd", "dgrad"))
        res += conv_case("conv2", bf, 64, [64, 96, 1], 9, 1, False, B, 176, 608, a.iters, ("fwd", "wgrad"))
        res += conv_case("upconv1", bf, 32, [64], 9, 1, True, B, 176, 608, a.iters, ("fwd", "wgrad"))
        res += conv_case("conv1", bf, 32, [32, 4], 9, 1, False, B, 352, 1216, a.iters, ("fwd", "wgrad"))
        res += conv_case("get_depth", bf, 1, [32], 9, 1, False, B, 352, 1216, a.iters, ("wgrad",))
        res += chain_cases(8, 352, 1216, bf, a.iters)
    if a.set == "wgrad":    # wide-layer weight gradients, kernel alone
        B = 8
        res += conv_case("conv5", bf, 512, [512, 384], 9, 1, False, B, 22, 76, a.iters, ("wgradk",))
        res += conv_case("upconv5", bf, 512, [2208], 9, 1, True, B, 11, 38, a.iters, ("wgradk",))
        res += conv_case("conv4", bf, 256, [256, 192], 9, 1, False, B, 44, 152, a.iters, ("wgradk",))
        res += conv_case("upconv4", bf, 256, [512], 9, 1, True, B, 22, 76, a.iters, ("wgradk",))
        res += conv_case("daspp_conv", bf, 128, [256, 128, 128, 128, 128, 128], 9, 1, False, B, 44, 152, a.iters, ("wgradk",))
        res += conv_case("daspp12_3x3", bf, 128, [256], 9, 12, False, B, 44, 152, a.iters, ("wgradk",))
        res += conv_case("daspp1x1_24", bf, 256, [256, 192, 128, 128, 128, 128], 1, 1, False, B, 44, 152, a.iters, ("wgradk",))
        res += conv_case("conv3", bf, 128, [128, 96, 1], 9, 1, False, B, 88, 304, a.iters, ("wgradk",))
        res += conv_case("upconv3", bf, 128, [128], 9, 1, True, B, 44, 152, a.iters, ("wgradk",))
    if a.set == "narrow":   # full-resolution narrow layers: weight gradients
        B = 8
        res += conv_case("get_depth", bf, 1, [32], 9, 1, False, B, 352, 1216, a.iters, ("wgrad",))
        res += conv_case("conv1", bf, 32, [32, 4], 9, 1, False, B, 352, 1216, a.iters, ("wgrad",))
        res += conv_case("upconv1", bf, 32, [64], 9, 1, True, B, 176, 608, a.iters, ("wgrad",))
        res += conv_case("conv2", bf, 64, [64, 96, 1], 9, 1, False, B, 176, 608, a.iters, ("wgrad",))
        res += conv_case("upconv2", bf, 64, [128], 9, 1, True, B, 88, 304, a.iters, ("wgrad",))
    if a.set in ("all", "lpg"):
        for k in (8, 4, 2):
            res += lpg_case(8, 352, 1216, k, a.iters)      # train shape (configs[2], per GPU)
        for k in (8, 4, 2):
            res += lpg_case(32, 704, 1216, k, a.iters)     # inference shape (configs[4])
    if a.set in ("all", "lpgrot"):   # the same LPG kernels with a rotating > 512 MB working set: HBM rates, not cache rates
        for k in (8, 4, 2):
            res += lpg_case_rot(8, 352, 1216, k, max(a.iters, 40))
        for k in (8, 4, 2):
            res += lpg_case_rot(32, 704, 1216, k, max(a.iters, 12))
    if a.set in ("all", "chain"):
        res += chain_cases(8, 352, 1216, bf, a.iters)
        res += chain_cases(32, 704, 1216, bf, a.iters)
        res += chain_cases(8, 352, 1216, f32, a.iters)
    if a.set in ("all", "misc"):
        res += misc_cases(8, 352, 1216, a.iters)
    for r in res:
        r["sec"] = round(r["sec"], 7)
        for k in ("tflops", "alg_gbs"):
            if k in r:
                r[k] = round(r[k], 1)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
