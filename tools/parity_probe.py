"""Three-way gradient comparison at a benchmarked configuration (test infrastructure, runs on the GPU box).

Which side of `test_decoder_parity_at_bench_config[*-f32]` carries the error?  Evaluates the oracle's formulas on the
device in f64 (the arbiter) and in f32 (what the test uses as its checker; convolutions go through MIOpen), runs the
product path in f32, and prints the L2-relative error of every gradient of (product f32, oracle f32) against oracle f64.

    python tools/parity_probe.py [--cfg c3] [--tf32 0|1] [--out gpurun_out/parity_probe.json]
"""
import argparse
import json
import os
import sys
from types import SimpleNamespace as NS

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CONFIGS = {"c3": (8, 352, 1216, "kitti", 80.0), "c2": (16, 416, 544, "nyu", 10.0), "tiny": (2, 96, 160, "kitti", 80.0)}


def l2rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="c3")
    ap.add_argument("--tf32", type=int, default=-1, help="-1 leave torch defaults; 0/1 set torch.backends.cudnn.allow_tf32")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from bts_amd.model import bts, silog_loss
    from oracle import bts_oracle as O
    dev = "cuda"
    if args.tf32 >= 0:
        torch.backends.cudnn.allow_tf32 = bool(args.tf32)
        torch.backends.cuda.matmul.allow_tf32 = bool(args.tf32)
    B, H, W, ds, md = CONFIGS[args.cfg]
    feat, nf = [96, 96, 192, 384, 2208], 512
    gen = torch.Generator().manual_seed(2024)
    P = O.make_decoder_params(feat, nf, gen, randomize_bn=True)
    feats = O.make_features(feat, B, H, W, gen)
    focal = O.synth_focal(B, ds)
    gt = O.synth_depth_gt(B, H, W, ds, gen).to(dev)
    mask = gt > (1.0 if ds == "kitti" else 0.1)

    def objective(outs, loss):
        return loss + sum((o * o).mean() for o in outs[:4])

    def run_oracle(dt):
        Pd = {k: ((v.to(dev).to(dt)).requires_grad_(True) if v.dtype.is_floating_point and "running" not in k
                  else (v.to(dev).to(dt) if v.dtype.is_floating_point else v.to(dev))) for k, v in P.items()}
        fr = [f.to(dev).to(dt).requires_grad_(True) for f in feats]
        ref, _ = O.decoder_forward(Pd, fr, focal.to(dev).to(dt), md, ds, True)
        loss = O.silog(ref[4], gt.to(dt), mask, 0.85)
        objective(ref, loss).backward()
        g = {k: v.grad.detach() for k, v in Pd.items() if v.dtype.is_floating_point and v.requires_grad}
        for i, f in enumerate(fr):
            g["feat%d" % i] = f.grad.detach()
        return [r.detach() for r in ref], loss.detach(), g

    rep = {"config": args.cfg, "tf32_flag": args.tf32}
    try:
        o64, l64, g64 = run_oracle(torch.float64)
        rep["arbiter"] = "oracle f64 on device"
    except Exception as e:          # no f64 convolution on this stack: fall back to the f32 oracle with TF32 off
        print("f64 oracle failed:", repr(e)[:300], flush=True)
        torch.backends.cudnn.allow_tf32 = False
        o64, l64, g64 = run_oracle(torch.float32)
        rep["arbiter"] = "oracle f32 on device, allow_tf32=False (f64 failed: %s)" % repr(e)[:120]
    o32, l32, g32 = run_oracle(torch.float32)
    dec = bts(NS(max_depth=md, dataset=ds, encoder="densenet161_bts", bts_size=nf, decoder_dtype=torch.float32), feat, nf)
    dec.load_state_dict(P)
    dec.to(dev).train()
    fs = [f.to(dev).requires_grad_(True) for f in feats]
    outs = dec(fs, focal.to(dev))
    loss = silog_loss(0.85)(outs[4], gt, mask)
    objective(outs, loss).backward()
    gp = {n: p.grad for n, p in dec.named_parameters()}
    for i, f in enumerate(fs):
        gp["feat%d" % i] = f.grad
    rep["outputs"] = {"out%d" % i: {"product": l2rel(outs[i], o64[i]), "oracle_f32": l2rel(o32[i], o64[i])} for i in range(5)}
    rep["loss"] = {"product": abs(loss.item() - l64.item()) / abs(l64.item()), "oracle_f32": abs(l32.item() - l64.item()) / abs(l64.item())}
    rep["grads"] = {k: {"product": l2rel(gp[k], g64[k]), "oracle_f32": l2rel(g32[k], g64[k])} for k in g64}
    print("arbiter:", rep["arbiter"])
    print("outputs:", {k: "%.1e / %.1e" % (v["product"], v["oracle_f32"]) for k, v in rep["outputs"].items()})
    print("loss   : %.1e / %.1e" % (rep["loss"]["product"], rep["loss"]["oracle_f32"]))
    print("%-55s %10s %10s" % ("gradient (L2-rel vs arbiter)", "product", "oracle_f32"))
    for k, v in rep["grads"].items():
        print("%-55s %10.2e %10.2e" % (k, v["product"], v["oracle_f32"]))
    print("max product %.2e  max oracle_f32 %.2e" % (max(v["product"] for v in rep["grads"].values()),
                                                     max(v["oracle_f32"] for v in rep["grads"].values())), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
