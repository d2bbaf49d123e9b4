#!/usr/bin/env python
"""Least-squares split of the launch times of `conv_igemm_dma<bf16,128x128>` in a `bench.py --dump-launches` table (DenseNet161-BTS at the
bench shape) into a FIXED cost per output tile and a cost per 64-deep K chunk:

    t(launch) = rounds x (t0 + chunks x tc),     rounds = max(1, tiles / 512)        (two 128 x 128 workgroups per CU, 256 CUs)

    python tools/fit_fixed_cost.py profiles/r04_launches.json

Prints t0, tc, the asymptotic rate 512 x 2 x 128 x 128 x 64 FLOP / tc, every launch with the model's fixed share, and the family totals:
how much of the dominant family's time is per-tile fixed cost (pipeline fill behind a cold first chunk, address set-up, epilogue) -- what
a persistent walk over several tiles per workgroup could overlap -- and how much is the K loop itself."""
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bts_amd.decoder import DecoderPlan  # noqa: E402

FEAT, NF, N, H, W = [96, 96, 192, 384, 2208], 512, 8, 352, 1216
RES = {"upconv5.conv": 32, "conv5.0": 16, "upconv4.conv": 16, "upconv3.conv": 8, "upconv2.conv": 4, "upconv1.conv": 2}


def grid(name):
    s = 8 if (name.startswith("daspp_") or name.startswith("reduc8x8")) else RES.get(name)
    if s is None:
        raise KeyError(name)
    return H // s, W // s


def main():
    plan = DecoderPlan(FEAT, NF)
    rows = []
    for r in json.load(open(sys.argv[1]))["rows"]:
        if r["family"] != "conv_igemm_dma<bf16,128x128>":
            continue
        name, kind = r["tag"].rsplit(".", 1)
        L = plan.layers[name]
        h, w = grid(name)
        pad = lambda c: (c + 7) // 8 * 8    # noqa: E731
        if kind == "fwd":
            K, T, cout, phases = sum(pad(c) for c in L.seg_channels), L.T, L.cout, L.nphase
        else:
            K, T, cout, phases = pad(L.cout), len(L.taps), pad(L.seg_channels[int(kind[5:])]), 1
        kv = K // 8
        chunks = (kv // 8) * T if (T > 1 and phases == 1 and kv % 8 == 0) else math.ceil(T * kv / 8)
        tiles = math.ceil(cout / 128) * math.ceil(N * h * w / 128) * phases
        rows.append((r["tag"], r["us_per_launch"], chunks, tiles))
    A = np.array([[max(t / 512.0, 1.0), max(t / 512.0, 1.0) * c] for _, _, c, t in rows])
    y = np.array([us for _, us, _, _ in rows])
    (t0, tc), *_ = np.linalg.lstsq(A, y, rcond=None)
    print("fixed cost per tile (per workgroup slot) t0 = %.1f us, per chunk tc = %.3f us -> asymptote %.0f TFLOP/s"
          % (t0, tc, 512 * 2 * 128 * 128 * 64 / tc / 1e6))
    fixed = var = meas = 0.0
    for (tag, us, c, t), a in zip(rows, A):
        print("%-52s %6.1f us  chunks %4d  tiles %5d  model %6.1f  fixed share %3.0f %%" % (tag, us, c, t, a @ np.array([t0, tc]), 100 * t0 * a[0] / (a @ np.array([t0, tc]))))
        fixed += t0 * a[0]
        var += tc * a[1]
        meas += us
    print("family: measured %.0f us, model %.0f us = %.0f us fixed + %.0f us in the K loop" % (meas, fixed + var, fixed, var))


if __name__ == "__main__":
    main()
