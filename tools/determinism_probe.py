"""Root-cause probe for run-to-run / model-to-model bit differences of the no-grad forward.

Per iteration compares, bitwise: (i) the encoder features of two BtsModels carrying the same state dict,
(ii) the decoders of both on the SAME feature tensors, (iii) one decoder twice.  Optional NaN poisoning of every
torch.empty the decoder allocates flushes out reads of uninitialised memory.

    python tools/determinism_probe.py [--iters 200] [--poison 1] [--encoder densenet121_bts] [--h 64 --w 96]
"""
import argparse
import os
import sys
from types import SimpleNamespace as NS

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--poison", type=int, default=0)
    ap.add_argument("--encoder", default="densenet121_bts")
    ap.add_argument("--h", type=int, default=64)
    ap.add_argument("--w", type=int, default=96)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--train", type=int, default=0)
    args = ap.parse_args()
    from bts_amd.model import BtsModel
    from oracle import bts_oracle as O
    dev = "cuda"
    if args.poison:
        real_empty = torch.empty

        def poisoned(*a, **k):
            t = real_empty(*a, **k)
            if t.is_cuda and t.dtype.is_floating_point:
                t.fill_(float("nan"))
            return t
        real_like = torch.empty_like

        def poisoned_like(*a, **k):
            t = real_like(*a, **k)
            if t.is_cuda and t.dtype.is_floating_point:
                t.fill_(float("nan"))
            return t
        torch.empty, torch.empty_like = poisoned, poisoned_like      # looked up at call time by bts_amd.*
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    params = NS(encoder=args.encoder, max_depth=10.0, dataset="nyu", bts_size=512, decoder_dtype=cdt)
    torch.manual_seed(0)
    m = BtsModel(params).to(dev).eval()
    m2 = BtsModel(params).to(dev).eval()
    m2.load_state_dict(m.state_dict())
    if args.train:
        m.train()
        m2.train()
    gen = torch.Generator().manual_seed(5)
    bad = {"enc": 0, "dec_pair": 0, "dec_self": 0, "e2e": 0, "nan": 0}
    first = {}
    for it in range(args.iters):
        x = torch.randn(args.batch, 3, args.h, args.w, generator=gen).to(dev)
        focal = O.synth_focal(args.batch, "nyu").to(dev)
        with torch.no_grad():
            fa = m.encoder(x)
            fb = m2.encoder(x)
            if not all(torch.equal(u, v) for u, v in zip(fa, fb)):
                bad["enc"] += 1
                first.setdefault("enc", (it, [int((u != v).sum()) for u, v in zip(fa, fb)]))
            da = m.decoder(fa, focal)
            db = m2.decoder(fa, focal)
            dc = m.decoder(fa, focal)
            if not all(torch.equal(u, v) for u, v in zip(da, db)):
                bad["dec_pair"] += 1
                first.setdefault("dec_pair", (it, [int((u != v).sum()) for u, v in zip(da, db)],
                                              [float((u - v).abs().max()) for u, v in zip(da, db)]))
            if not all(torch.equal(u, v) for u, v in zip(da, dc)):
                bad["dec_self"] += 1
                first.setdefault("dec_self", (it, [int((u != v).sum()) for u, v in zip(da, dc)],
                                              [float((u - v).abs().max()) for u, v in zip(da, dc)]))
            ea = m(x, focal)
            eb = m2(x, focal)
            if not all(torch.equal(u, v) for u, v in zip(ea, eb)):
                bad["e2e"] += 1
            if any(not torch.isfinite(u).all() for u in da):
                bad["nan"] += 1
                first.setdefault("nan", (it, [int((~torch.isfinite(u)).sum()) for u in da]))
    print("determinism_probe", vars(args), "mismatching iterations:", bad, "first:", first, flush=True)


if __name__ == "__main__":
    main()
