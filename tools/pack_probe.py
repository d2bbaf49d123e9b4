#!/usr/bin/env python
"""Times the batched weight pack / unpack launches of the DenseNet161-BTS decoder (GPU box only)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bts_oracle as O  # noqa: E402
from bts_amd.decoder import DecoderPlan, PackSet  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    feat, nf = [96, 96, 192, 384, 2208], 512
    plan = DecoderPlan(feat, nf)
    gen = torch.Generator().manual_seed(0)
    P = {k: v.cuda() for k, v in O.make_decoder_params(feat, nf, gen).items()}
    for dt in (torch.bfloat16, torch.float32):
        ps = PackSet(plan, P, dt)
        dwp = torch.randn(ps.dwp_total, device="cuda")
        gw = torch.empty(ps.gw_total, device="cuda")
        print(dt, "weights %.1f M, blocks f/d/u = %d/%d/%d" % (ps.gw_total / 1e6, ps.fblocks, ps.dblocks, ps.ublocks))
        print("  pack_forward %.1f us  pack_dgrad %.1f us  unpack_all %.1f us" % (
            timeit(ps.pack_forward), timeit(ps.pack_dgrad), timeit(lambda: ps.unpack_all(dwp, gw))))


if __name__ == "__main__":
    main()
