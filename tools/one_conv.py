#!/usr/bin/env python
"""One convolution layer of the decoder, launched a few times: the smallest target for `rocprofv3 --pmc` experiments on ONE kernel
family (tools/pmc_wide_probe.sh).  --layer conv4 (conv_halo_wide<4>), conv2 (conv_halo_wide<2>), aspp1x1_dgrad (conv_igemm_res),
conv5 (conv_igemm_dma) at the bench shape."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bts_amd._lib import ACT_ELU, ACT_NONE  # noqa: E402
from bts_amd.conv import ConvLayer  # noqa: E402

LAYERS = {   # cout, segments, (N, H, W), dgrad of segment (or None = forward)
    "conv4": (256, [256, 192], (8, 44, 152), None),
    "conv3": (128, [128, 96, 1], (8, 88, 304), None),
    "conv2": (64, [64, 96, 1], (8, 176, 608), None),
    "conv5": (512, [512, 384], (8, 22, 76), None),
    "aspp1x1_dgrad": (256, [960], (8, 44, 152), 0),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="conv4")
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    cout, segc, (N, H, W), dg = LAYERS[a.layer]
    dev = "cuda"
    L = ConvLayer(a.layer, cout, segc, 1 if a.layer.startswith("aspp1x1") else 9)
    gen = torch.Generator(device=dev).manual_seed(0)
    segs = [torch.randn(N, H, W, (c + 7) // 8 * 8, device=dev, generator=gen).to(torch.bfloat16) for c in segc]
    for sg, c in zip(segs, segc):
        sg[..., c:] = 0
    w = torch.randn(cout, sum(segc), *((1, 1) if L.kk == 1 else (3, 3)), device=dev, generator=gen) * 0.02
    out = torch.empty(N, H, W, cout, dtype=torch.bfloat16, device=dev)
    if dg is None:
        wp = L.pack_fwd(w, torch.bfloat16)
        for _ in range(a.iters):
            L.forward(segs, wp, out, ACT_ELU)
    else:
        wd = L.pack_dgrad(w, torch.bfloat16, dg)
        dz = torch.randn(N, H, W, cout, device=dev, generator=gen).to(torch.bfloat16)
        gx = torch.empty_like(segs[dg])
        tb = L.tables(torch.bfloat16, torch.device(dev))
        lay = L.frag_layout(torch.bfloat16, segs[dg].shape[3], tb["cout_pad"], True)
        if lay:
            wd = L.to_frag(wd, L._launch_taps(True)[0])
        for _ in range(a.iters):
            L.dgrad(dz, wd, dg, gx, False, w_frag=lay)
    torch.cuda.synchronize()
    print("ok", a.layer)


if __name__ == "__main__":
    main()
