#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the UNMODIFIED reference pytorch/bts.py on CPU.

Run in the build container only (needs /root/reference):
    python tools/make_golden.py
The reference has no golden vectors of its own; these
fixtures are outputs of the reference code itself on seeded synthetic inputs and
are what pins oracle/bts_oracle.py and the HIP path on the GPU box, where the
reference tree does not exist.  Inputs that are cheap to store are stored; the
large-channel case regenerates its weights from the seed with
oracle.bts_oracle.make_decoder_params (torch CPU RNG is deterministic for a
given torch version; the version is recorded in each file).
"""
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bts_oracle as O  # noqa: E402
from oracle import ref_loader  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(max(1, os.cpu_count() or 1))


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    arrs["torch_version"] = np.array(torch.__version__)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def gen_lpg(ref):
    """reference local_planar_guidance fwd + autograd bwd, k = 8, 4, 2 (bts.py:124-146)."""
    g = torch.Generator().manual_seed(11)
    out = {}
    for k, (h, w) in ((8, (5, 7)), (4, (6, 10)), (2, (9, 12))):
        B = 2
        raw = torch.randn(B, 3, h, w, generator=g)
        eq = O.normalize_plane(O.plane_from_raw(raw, 80.0)).requires_grad_(True)
        mod = ref.local_planar_guidance(k)
        y = mod(eq, torch.ones(B))
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        out["k%d_eq" % k] = npy(eq)
        out["k%d_out" % k] = npy(y)
        out["k%d_gout" % k] = npy(gy)
        out["k%d_geq" % k] = npy(eq.grad)
    save("lpg", **out)


def gen_reduction(ref):
    """reference reduction_1x1 fwd + bwd for the four head variants at bts_size=128/512."""
    g = torch.Generator().manual_seed(12)
    torch.manual_seed(12)        # weights_init_xavier (bts.py:26-32) draws from the GLOBAL generator: seed it, or the file is not reproducible
    out = {}
    for tag, cin, cout, final, hw in (("r8_128", 32, 32, False, (4, 6)), ("r2_128", 16, 8, False, (6, 8)),
                                      ("r1_128", 8, 4, True, (8, 8)), ("r8_512", 128, 128, False, (3, 5)),
                                      ("r1_512", 32, 16, True, (4, 4))):
        mod = ref.reduction_1x1(cin, cout, 80.0, is_final=final)
        mod.apply(ref.weights_init_xavier)
        x = torch.randn(2, cin, *hw, generator=g, requires_grad=True)
        y = mod(x)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        out[tag + "_x"] = npy(x)
        out[tag + "_y"] = npy(y)
        out[tag + "_gy"] = npy(gy)
        out[tag + "_gx"] = npy(x.grad)
        for n, p in mod.named_parameters():
            out[tag + "_w_" + n] = npy(p)
            out[tag + "_gw_" + n] = npy(p.grad)
    save("reduction", **out)


def gen_silog(ref):
    """reference silog_loss fwd + bwd (bts.py:41-48), kitti- and nyu-style masks."""
    g = torch.Generator().manual_seed(13)
    out = {}
    for tag, ds, vf in (("kitti", "kitti", 0.85), ("nyu", "nyu", 0.5)):
        B, H, W = 2, 24, 40
        md = 80.0 if ds == "kitti" else 10.0
        est = (torch.rand(B, 1, H, W, generator=g) * (md - 1) + 0.5).requires_grad_(True)
        gt = O.synth_depth_gt(B, H, W, ds, g)
        mask = gt > (1.0 if ds == "kitti" else 0.1)
        loss = ref.silog_loss(vf)(est, gt, mask)
        loss.backward()
        out[tag + "_est"] = npy(est)
        out[tag + "_gt"] = npy(gt)
        out[tag + "_loss"] = npy(loss)
        out[tag + "_gest"] = npy(est.grad)
        out[tag + "_vf"] = np.array(vf)
    save("silog", **out)


def run_decoder(ref, feat, nf, B, H, W, dataset, seed, train, randomize_bn=True):
    g = torch.Generator().manual_seed(seed)
    md = 80.0 if dataset == "kitti" else 10.0
    P = O.make_decoder_params(feat, nf, g, randomize_bn=randomize_bn)
    feats = [f.requires_grad_(True) for f in O.make_features(feat, B, H, W, g)]
    focal = O.synth_focal(B, dataset)
    gt = O.synth_depth_gt(B, H, W, dataset, g)
    dec = ref.bts(NS(max_depth=md, dataset=dataset, encoder="densenet161_bts", bts_size=nf), feat, nf)
    dec.load_state_dict(P)
    dec.train(train)
    outs = dec(feats, focal)
    mask = gt > (1.0 if dataset == "kitti" else 0.1)
    loss = ref.silog_loss(0.85)(outs[4], gt, mask)
    # a loss that touches all five outputs, so every head's backward is exercised
    aux = sum((o * o).mean() for o in outs[:4])
    (loss + aux).backward()
    return P, feats, focal, gt, dec, outs, loss, aux


def gen_decoder_small(ref):
    """Full decoder (bts.py:148-266) fwd+bwd, tiny channel config, train and eval mode."""
    feat, nf, B, H, W = [8, 8, 16, 24, 40], 128, 2, 64, 96
    for tag, train, ds in (("train_kitti", True, "kitti"), ("eval_nyu", False, "nyu")):
        P, feats, focal, gt, dec, outs, loss, aux = run_decoder(ref, feat, nf, B, H, W, ds, 21, train)
        arr = {"feat": np.array(feat), "nf": np.array(nf), "shape": np.array([B, H, W]), "seed": np.array(21),
               "dataset": np.array(ds), "focal": npy(focal), "gt": npy(gt), "loss": npy(loss), "aux": npy(aux)}
        if train:   # params / features depend only on the seed: stored once, in the train file
            for k, v in P.items():
                arr["P/" + k] = npy(v)
        for i, f in enumerate(feats):
            if train:
                arr["feat%d" % i] = npy(f)
            arr["gfeat%d" % i] = npy(f.grad)
        for i, o in enumerate(outs):
            arr["out%d" % i] = npy(o)
        for n, p in dec.named_parameters():
            arr["G/" + n] = npy(p.grad)
        for n, b in dec.named_buffers():
            arr["B/" + n] = npy(b)            # running stats AFTER the forward
        save("decoder_small_" + tag, **arr)


def gen_decoder_dn161(ref):
    """Real DenseNet161 channel config, tiny spatial size; weights regenerated from the seed."""
    feat, nf, B, H, W = [96, 96, 192, 384, 2208], 512, 1, 32, 64
    P, feats, focal, gt, dec, outs, loss, aux = run_decoder(ref, feat, nf, B, H, W, "kitti", 22, True, randomize_bn=False)
    arr = {"feat": np.array(feat), "nf": np.array(nf), "shape": np.array([B, H, W]), "seed": np.array(22),
           "dataset": np.array("kitti"), "loss": npy(loss), "aux": npy(aux)}
    for i, o in enumerate(outs):
        arr["out%d" % i] = npy(o)
    for i, f in enumerate(feats):
        arr["gfeat%d_l2" % i] = npy(f.grad.norm())
    for n, p in dec.named_parameters():
        arr["Gl2/" + n] = npy(p.grad.double().norm())
        arr["Ghead/" + n] = npy(p.grad.flatten()[:32])
    save("decoder_dn161_tiny", **arr)


def gen_model_c1(ref):
    """BASELINE.json configs[0]: BtsModel densenet121, 416x544, batch 1, CPU forward (eval)."""
    torch.manual_seed(23)
    params = NS(encoder="densenet121_bts", max_depth=10.0, dataset="nyu", bts_size=512)
    model = ref.BtsModel(params)
    model.decoder.apply(ref.weights_init_xavier)
    model.eval()
    g = torch.Generator().manual_seed(24)
    x = torch.randn(1, 3, 416, 544, generator=g)
    focal = O.synth_focal(1, "nyu")
    with torch.no_grad():
        outs = model(x, focal)
    arr = {"model_seed": np.array(23), "input_seed": np.array(24)}
    for i, o in enumerate(outs):
        arr["out%d_s8" % i] = npy(o[:, :, ::8, ::8])       # strided sample
        arr["out%d_mean" % i] = npy(o.double().mean())
        arr["out%d_l2" % i] = npy(o.double().norm())
    arr["n_params"] = np.array(sum(p.numel() for p in model.parameters()))
    arr["state_keys"] = np.array(list(model.state_dict().keys()))
    save("model_c1_densenet121", **arr)


def gen_eval():
    """compute_errors (bts_main.py:143-165): bts_main cannot be imported here (tensorboardX / cv2 missing), so the
    function's own source text is executed from the unmodified file.  Inputs: masked pixel lists as online_eval()
    builds them (oracle.eval_oracle.eval_prepare) for a kitti kb-crop + garg-crop case and a nyu eigen-crop case."""
    from oracle import eval_oracle as E
    src = open(os.path.join(ref_loader.REF_ROOT, "pytorch", "bts_main.py")).read().splitlines()
    start = next(i for i, l in enumerate(src) if l.startswith("def compute_errors("))
    end = next(i for i in range(start + 1, len(src)) if src[i].startswith("def "))
    ns = {"np": np}
    exec("\n".join(src[start:end]), ns)
    ref_fn = ns["compute_errors"]
    out = {}
    for tag, (gh, gw, ph, pw, md, kb, garg, eig, ds) in E.EVAL_CASES.items():
        pred, gt = E.synth_eval_case(tag)
        pf, valid = E.eval_prepare(pred, gt, 1e-3, md, ds, kb, garg, eig)
        out[tag + "_measures"] = np.array(ref_fn(gt[valid], pf[valid]), dtype=np.float64)
        out[tag + "_nvalid"] = np.array(int(valid.sum()))
        out[tag + "_checksum"] = np.array([float(pred[np.isfinite(pred)].astype(np.float64).sum()), float(gt.astype(np.float64).sum())])
    save("eval", **out)


def gen_preprocess():
    """random_crop / train_preprocess / augment_image executed from the source of the unmodified bts_dataloader.py
    (methods of DataLoadPreprocess; the module needs torchvision, so the three method bodies are exec'd into a stub
    class).  Small synthetic decoded images; seeds chosen to cover flip / no flip and augment / no augment."""
    import random as pyrandom
    import textwrap
    src = open(os.path.join(ref_loader.REF_ROOT, "pytorch", "bts_dataloader.py")).read().splitlines()

    def method(name):
        start = next(i for i, l in enumerate(src) if l.startswith("    def %s(" % name))
        end = next(i for i in range(start + 1, len(src)) if src[i].startswith("    def ") or src[i].startswith("class "))
        return textwrap.dedent("\n".join(src[start:end]))
    ns = {"np": np, "random": pyrandom}
    body = "\n".join(textwrap.indent(method(m), "    ") for m in ("random_crop", "train_preprocess", "augment_image"))
    exec("class Ref:\n" + body, ns)
    out = {}
    cases = []
    for ds in ("kitti", "nyu"):
        for seed in range(1, 9):
            cases.append((ds, seed))
    rs = np.random.RandomState(11)
    img = rs.randint(0, 256, size=(48, 80, 3)).astype(np.uint8)
    dep = rs.randint(0, 20000, size=(48, 80)).astype(np.int32)
    out["image_u8"], out["depth_raw"] = img, dep
    flips, augs = set(), set()
    for ds, seed in cases:
        r = ns["Ref"]()
        r.args = NS(dataset=ds)
        pyrandom.seed(seed)
        np.random.seed(seed)
        image = np.asarray(img, dtype=np.float32) / 255.0                     # bts_dataloader.py:126-133
        depth = np.expand_dims(np.asarray(dep, dtype=np.float32), axis=2)
        depth = depth / 1000.0 if ds == "nyu" else depth / 256.0
        image, depth = r.random_crop(image, depth, 32, 64)
        image, depth = r.train_preprocess(image, depth)
        out["%s_%d_image" % (ds, seed)] = np.ascontiguousarray(image, dtype=np.float32)
        out["%s_%d_depth" % (ds, seed)] = np.ascontiguousarray(depth, dtype=np.float32)
    save("preprocess", **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = ref_loader.load_reference()
    gen_lpg(ref)
    gen_reduction(ref)
    gen_silog(ref)
    gen_decoder_small(ref)
    gen_decoder_dn161(ref)
    gen_model_c1(ref)
    gen_eval()
    gen_preprocess()


if __name__ == "__main__" and "--eval-only" in sys.argv:
    gen_eval()
    gen_preprocess()
    sys.exit(0)

if __name__ == "__main__":
    main()
