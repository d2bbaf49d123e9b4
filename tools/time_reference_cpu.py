"""Anchor for bench.py's `cpu_baseline` (kind "port"): time the UNMODIFIED reference (pytorch/bts.py BtsModel +
silog_loss, imported through oracle/ref_loader.py) and the oracle port on the SAME host, same threads, same seeded
input -- one train step (fwd + loss + bwd), f32.  BASELINE.md section 2 protocol: 2 warm-up + >= 5 timed, median.
Runs only where /root/reference exists (the build container); writes one JSON object.

    python tools/time_reference_cpu.py --height 160 --width 608 --out profiles/r02_cpu_reference_vs_port.json
"""
import argparse
import json
import os
import statistics
import sys
import time
from types import SimpleNamespace as NS

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--encoder", default="densenet161_bts")
    ap.add_argument("--height", type=int, default=160)
    ap.add_argument("--width", type=int, default=608)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from oracle import bts_oracle as O
    from oracle import ref_loader
    ref = ref_loader.load_reference()
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    params = NS(encoder=a.encoder, max_depth=80.0, dataset="kitti", bts_size=512)
    torch.manual_seed(0)
    model = ref.BtsModel(params)
    model.train()
    model.decoder.apply(ref.weights_init_xavier)
    crit = ref.silog_loss(0.85)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(a.batch, 3, a.height, a.width, generator=gen)
    focal = O.synth_focal(a.batch, "kitti")
    gt = O.synth_depth_gt(a.batch, a.height, a.width, "kitti", gen)
    mask = gt > 1.0

    def step_reference():
        model.zero_grad(set_to_none=True)
        outs = model(x, focal)
        loss = crit(outs[4], gt, mask)
        loss.backward()
        return float(loss)

    P = {k: (v.detach().clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
         for k, v in model.decoder.state_dict().items()}

    def step_port():
        for v in P.values():
            if v.requires_grad:
                v.grad = None
        model.encoder.zero_grad(set_to_none=True)
        feats = model.encoder(x)
        outs, _ = O.decoder_forward(P, feats, focal, 80.0, "kitti", True)
        loss = O.silog(outs[4], gt, mask, 0.85)
        loss.backward()
        return float(loss)

    res = {}
    for name, fn in (("reference", step_reference), ("port", step_port)):
        for _ in range(a.warmup):
            fn()
        ts = []
        for _ in range(a.iters):
            t0 = time.time()
            loss = fn()
            ts.append(time.time() - t0)
        res[name] = {"median_s": statistics.median(ts), "min_s": min(ts), "max_s": max(ts), "loss": loss,
                     "images_per_s": a.batch / statistics.median(ts)}
    out = {"what": "one train step (fwd + silog + bwd), f32, CPU", "encoder": a.encoder, "shape": [a.batch, 3, a.height, a.width],
           "threads": threads, "host_cpu_count": os.cpu_count(), "torch": torch.__version__, "warmup": a.warmup, "iters": a.iters,
           "reference": res["reference"], "port": res["port"],
           "port_over_reference_time": res["port"]["median_s"] / res["reference"]["median_s"],
           "note": "reference = unmodified /root/reference/pytorch/bts.py (torchvision stub for the backbone, .cuda() identity); "
                   "port = oracle/bts_oracle.py with the same stock encoder: what bench.py's cpu_baseline times on the GPU host"}
    s = json.dumps(out, indent=1)
    print(s)
    if a.out:
        with open(a.out, "w") as f:
            f.write(s + "\n")


if __name__ == "__main__":
    main()
