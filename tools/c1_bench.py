"""Event-timed get_depth kernels (csrc/conv_c1.hip) at the bench shape: forward, data gradient, weight gradient.

    python tools/c1_bench.py [--n 8 --h 352 --w 1216 --c 32 --dtype bf16]

Prints one JSON line per kernel: average microseconds over `--iters` back-to-back launches and the algorithmic GB/s.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bts_amd import ops  # noqa: E402


def timed(fn, iters):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--h", type=int, default=352)
    ap.add_argument("--w", type=int, default=1216)
    ap.add_argument("--c", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(a.n, a.h, a.w, a.c, device=dev, generator=g).to(dt)
    w = torch.randn(1, a.c, 3, 3, device=dev, generator=g) * 0.05
    sc = torch.full((a.n,), 1.01, device=dev)
    y = ops.conv3x3_c1_fwd(x, w, 80.0, sc)
    gy = torch.randn(a.n, a.h, a.w, device=dev, generator=g)
    gx = torch.zeros_like(x)
    dwp = torch.zeros(1, 9, a.c, device=dev)
    px, es = a.n * a.h * a.w, x.element_size()
    rows = [
        ("conv_c1_fwd", lambda: ops.conv3x3_c1_fwd(x, w, 80.0, sc), px * (a.c * es + 4)),
        ("conv_c1_dgrad", lambda: ops.conv3x3_c1_dgrad(gy, y, w, gx, False, 80.0, sc), px * (8 + a.c * es)),
        ("conv_c1_dgrad+acc+fold", lambda: ops.conv3x3_c1_dgrad(gy, y, w, gx, True, 80.0, sc, x), px * (8 + 3 * a.c * es)),
        ("conv_c1_dgrad+fold", lambda: ops.conv3x3_c1_dgrad(gy, y, w, gx, False, 80.0, sc, x), px * (8 + 2 * a.c * es)),
        ("conv_c1_wgrad", lambda: ops.conv3x3_c1_wgrad(gy, y, x, dwp, 80.0, sc), px * (8 + a.c * es)),
    ]
    for name, fn, nbytes in rows:
        us = timed(fn, a.iters)
        print(json.dumps({"kernel": name, "shape": [a.n, a.h, a.w, a.c], "dtype": a.dtype, "us": round(us, 2),
                          "alg_GBps": round(nbytes / us * 1e-3, 1)}), flush=True)


if __name__ == "__main__":
    main()
