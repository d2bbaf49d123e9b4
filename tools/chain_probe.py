#!/usr/bin/env python
"""Fused LPG chain kernels alone at the bench shape (8 x 352 x 1216): time of `bts_lpg_chain_fwd` / `bts_lpg_chain_bwd` per
(C0, k) against the batch size, timed as a hipGraph replay over a rotation of input buffers (> the 256 MiB Infinity Cache when it
fits).  A linear fit t(B) = t0 + B * t1 splits a launch into its FIXED cost (weights to LDS, scratch zeroing, the cross-wave
reduction and the atomics of the weight gradients: per workgroup, independent of the tile count) and its streaming part.

    python tools/chain_probe.py [out.jsonl]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bts_amd import chain as chain_mod  # noqa: E402
from bts_amd.decoder import reduction_specs  # noqa: E402

DEV = "cuda"
CASES = [("reduc8x8", 128, 8, 44, 152, 128, 128), ("reduc4x4", 128, 4, 88, 304, 128, 64), ("reduc2x2", 64, 2, 176, 608, 64, 32),
         ("reduc1x1", 32, 1, 352, 1216, 32, 16)]


def timed(fn, nrot, replays=5):
    for i in range(nrot):
        fn(i)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for i in range(nrot):
                fn(i)
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(replays):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (replays * nrot)


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    gen = torch.Generator(device=DEV).manual_seed(0)
    for name, c0, k, h, w, cin, cout in CASES:
        specs = reduction_specs(cin, cout, k == 1)
        # the fused backward covers the halving tail: for reduc8x8 everything behind the 128 -> 128 layer
        ws = [torch.randn(b, a, 1, 1, device=DEV, generator=gen) * (1.0 / a ** 0.5) for _, a, b in specs]
        if name == "reduc8x8":
            ws = ws[1:]
        frags, frags_t = chain_mod.pack_chain(ws, torch.bfloat16), chain_mod.pack_chain_t(ws, torch.bfloat16)
        rows = []
        for B in (1, 2, 4, 8, 16):
            per = B * h * w * c0 * 2 * 2 + B * h * w * k * k * 4
            nrot = max(2, min(24, (384 << 20) // per))
            xs = [(torch.randn(B, h, w, c0, device=DEV, generator=gen)).to(torch.bfloat16) for _ in range(nrot)]
            gs = [torch.randn((B, h * k, w * k) if k > 1 else (B, h, w), device=DEV, generator=gen) for _ in range(nrot)]
            dxs = [torch.empty_like(x) for x in xs]
            gws = [torch.zeros(wt.shape[0], wt.shape[1], device=DEV) for wt in ws]
            t_f = timed(lambda i: chain_mod.chain_fwd(xs[i], frags, c0, False, k, 80.0), nrot)
            t_b = timed(lambda i: chain_mod.chain_bwd(xs[i], frags, frags_t, c0, k, 80.0, gs[i], dxs[i], False, gws, True), nrot)
            cells = B * h * w
            rec = {"chain": name, "c0": c0, "k": k, "B": B, "cells": cells, "tiles32": (cells + 31) // 32, "fwd_us": round(t_f, 2), "bwd_us": round(t_b, 2),
                   "bwd_alg_GBps": round(cells * (c0 * 2 * 2 + 4 * k * k) / t_b / 1e3, 1), "fwd_alg_GBps": round(cells * (c0 * 2 + 4 * k * k) / t_f / 1e3, 1)}
            rows.append(rec)
            print(json.dumps(rec), flush=True)
            if out:
                out.write(json.dumps(rec) + "\n")
            del xs, gs, dxs
            torch.cuda.empty_cache()
        # least squares t = t0 + B t1 over the rows
        n = len(rows)
        sx = sum(r["B"] for r in rows); sy = sum(r["bwd_us"] for r in rows)
        sxx = sum(r["B"] ** 2 for r in rows); sxy = sum(r["B"] * r["bwd_us"] for r in rows)
        t1 = (n * sxy - sx * sy) / (n * sxx - sx * sx)
        t0 = (sy - t1 * sx) / n
        fit = {"chain": name, "fit_bwd_us": {"fixed_t0": round(t0, 2), "per_image_t1": round(t1, 2)}}
        print(json.dumps(fit), flush=True)
        if out:
            out.write(json.dumps(fit) + "\n")


if __name__ == "__main__":
    main()
