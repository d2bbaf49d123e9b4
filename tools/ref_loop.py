"""The reference's two driver loops as a small harness, for boxes where /root/reference does not exist (the GPU box).

The unmodified `bts_main.py` / `bts_test.py` run on the drop-in through tools/run_reference.py wherever the reference tree is
present (tests/test_reference_drivers.py, build container, CPU plumbing).  The GPU box has no reference tree, so the HIP path
would never see the drivers' control flow; this file restates that control flow -- written from what the drivers DO, not copied
from them -- so a `-m gpu` test can put the same sequence of calls on the real kernels:

  train_loop  <- bts_main.py:322-554 (main_worker): build `BtsModel(args)`, `model.train()`, `model.decoder.apply(
                 weights_init_xavier)` (:336-338), `set_misc` freezing by name substrings (:217-247), `torch.nn.DataParallel`
                 wrap + `.cuda()` (:357-358), AdamW over `model.module.encoder/decoder.parameters()` with per-group weight decay
                 (:371-373), optional resume from {'global_step', 'model', 'optimizer'} (:376-397), then per batch (:439-466):
                 `optimizer.zero_grad()`, batch dict -> `.cuda(non_blocking=True)`, five outputs, dataset-dependent mask,
                 `silog_criterion.forward(...)`, `loss.backward()`, per-step poly learning rate written into every param group,
                 `optimizer.step()`, the '{:.12f}'.format(loss) log line and the `np.isnan(loss.cpu().item())` abort; checkpoints
                 written with `torch.save` every `save_freq` steps (:498-503).
  test_loop   <- bts_test.py:84-128: `BtsModel` -> `DataParallel` -> `load_state_dict(checkpoint['model'])` -> `eval()` ->
                 `.cuda()`, parameter count via numpy, then under `torch.no_grad()` per sample: `.cuda()`, five outputs,
                 `depth_est.cpu().numpy().squeeze()` and `lpgNxN[0].cpu().numpy().squeeze()` appended to five lists.

  test_loop_device / online_eval_device / train_batches_device: the same loops with the work either side of the model on the
                 device kernels (bts_amd/loops.py: uint16 payload, compute_errors, sample preprocessing) -- SURVEY.md 8f rows 2-4.

`model_module` is whatever `from bts import *` would have bound (the drop-in module): it must provide BtsModel, silog_loss,
weights_init_xavier, bn_init_as_tf -- nothing else is assumed.
"""
import os

import numpy as np
import torch


def set_misc(model, args):
    """Freezing rules of bts_main.py:217-247 (name-substring matching on the child whose name contains 'encoder')."""
    if getattr(args, "bn_no_track_stats", False):
        model.apply(args._module.bn_init_as_tf)
    resnet = "resne" in args.encoder
    if getattr(args, "fix_first_conv_blocks", False):
        fixing = ["base_model.conv1", "base_model.layer1.0", "base_model.layer1.1", ".bn"] if resnet else \
                 ["conv0", "denseblock1.denselayer1", "denseblock1.denselayer2", "norm"]
    elif getattr(args, "fix_first_conv_block", False):
        fixing = ["base_model.conv1", "base_model.layer1.0", ".bn"] if resnet else ["conv0", "denseblock1.denselayer1", "norm"]
    else:
        fixing = ["base_model.conv1", ".bn"] if resnet else ["conv0", "norm"]
    for name, child in model.named_children():
        if "encoder" not in name:
            continue
        for pname, p in child.named_parameters():
            if any(x in pname for x in fixing):
                p.requires_grad = False


def train_loop(model_module, args, batches, checkpoint_path="", log=print, steps_per_epoch=None):
    """One `main_worker` pass over `batches` (dicts with CPU tensors 'image' [B,3,H,W] f32, 'focal' [B] f64, 'depth' [B,1,H,W]).
    steps_per_epoch: length of the epoch the poly schedule is computed for (bts_main.py:430-431), default len(batches) -- a
    resumed run that is handed only the remaining batches passes the full epoch length.
    Returns (model, optimizer, global_step, losses)."""
    args._module = model_module
    model = model_module.BtsModel(args)
    model.train()
    model.decoder.apply(model_module.weights_init_xavier)
    set_misc(model, args)
    n_all = sum(np.prod(p.size()) for p in model.parameters())
    n_train = sum(np.prod(p.shape) for p in model.parameters() if p.requires_grad)
    log("Total number of parameters: {}".format(n_all))
    log("Total number of learning parameters: {}".format(n_train))
    model = torch.nn.DataParallel(model)
    model.cuda()
    optimizer = torch.optim.AdamW([{"params": model.module.encoder.parameters(), "weight_decay": args.weight_decay},
                                   {"params": model.module.decoder.parameters(), "weight_decay": 0}],
                                  lr=args.learning_rate, eps=args.adam_eps)
    global_step = 0
    if checkpoint_path:
        if not os.path.isfile(checkpoint_path):
            raise FileNotFoundError(checkpoint_path)
        checkpoint = torch.load(checkpoint_path)
        global_step = checkpoint["global_step"]
        model.load_state_dict(checkpoint["model"])
        optimizer.load_state_dict(checkpoint["optimizer"])
        log("Loaded checkpoint '{}' (global_step {})".format(checkpoint_path, global_step))
    silog_criterion = model_module.silog_loss(variance_focus=args.variance_focus)
    end_lr = args.end_learning_rate if args.end_learning_rate != -1 else 0.1 * args.learning_rate
    steps_per_epoch = steps_per_epoch or len(batches)
    num_total_steps = args.num_epochs * steps_per_epoch
    losses = []
    for step, sample in enumerate(batches):
        optimizer.zero_grad()
        image = sample["image"].cuda(non_blocking=True)
        focal = sample["focal"].cuda(non_blocking=True)
        depth_gt = sample["depth"].cuda(non_blocking=True)
        lpg8x8, lpg4x4, lpg2x2, reduc1x1, depth_est = model(image, focal)
        mask = depth_gt > (0.1 if args.dataset == "nyu" else 1.0)
        loss = silog_criterion.forward(depth_est, depth_gt, mask.to(torch.bool))
        loss.backward()
        current_lr = (args.learning_rate - end_lr) * (1 - global_step / num_total_steps) ** 0.9 + end_lr
        for group in optimizer.param_groups:
            group["lr"] = current_lr
        optimizer.step()
        log("[epoch][s/s_per_e/gs]: [{}][{}/{}/{}], lr: {:.12f}, loss: {:.12f}".format(global_step // steps_per_epoch, step, steps_per_epoch, global_step, current_lr, loss))
        if np.isnan(loss.cpu().item()):
            raise FloatingPointError("NaN in loss occurred. Aborting training.")
        losses.append(loss.cpu().item())
        global_step += 1
        if args.save_freq and global_step % args.save_freq == 0:
            torch.save({"global_step": global_step, "model": model.state_dict(), "optimizer": optimizer.state_dict()},
                       os.path.join(args.log_directory, "model-{}".format(global_step)))
    return model, optimizer, global_step, losses


def test_loop(model_module, args, samples, log=print):
    """bts_test.py's `test`: returns the five lists of squeezed numpy maps."""
    model = model_module.BtsModel(params=args)
    model = torch.nn.DataParallel(model)
    checkpoint = torch.load(args.checkpoint_path)
    model.load_state_dict(checkpoint["model"])
    model.eval()
    model.cuda()
    log("Total number of parameters: {}".format(sum(np.prod(p.size()) for p in model.parameters())))
    pred_depths, pred_8x8s, pred_4x4s, pred_2x2s, pred_1x1s = [], [], [], [], []
    with torch.no_grad():
        for sample in samples:
            image = sample["image"].cuda()
            focal = sample["focal"].cuda()
            lpg8x8, lpg4x4, lpg2x2, reduc1x1, depth_est = model(image, focal)
            pred_depths.append(depth_est.cpu().numpy().squeeze())
            pred_8x8s.append(lpg8x8[0].cpu().numpy().squeeze())
            pred_4x4s.append(lpg4x4[0].cpu().numpy().squeeze())
            pred_2x2s.append(lpg2x2[0].cpu().numpy().squeeze())
            pred_1x1s.append(reduc1x1[0].cpu().numpy().squeeze())
    return pred_depths, pred_8x8s, pred_4x4s, pred_2x2s, pred_1x1s


def _load_for_test(model_module, args):
    model = model_module.BtsModel(params=args)
    model = torch.nn.DataParallel(model)
    model.load_state_dict(torch.load(args.checkpoint_path)["model"])
    model.eval()
    model.cuda()
    return model


def test_loop_device(model_module, args, samples, keep_lpg=False):
    """bts_test.py's `test` with the saved payload formed on the device (bts_test.py:119-124 + 179-185): the uint16 images that
    are written as PNG, one device -> host copy of 2 bytes per pixel per batch instead of five f32 maps per image."""
    from bts_amd import loops
    model = _load_for_test(model_module, args)
    return loops.predict_payloads(model, samples, args.dataset, device="cuda", keep_lpg=keep_lpg)


def online_eval_host(model, eval_samples, args):
    """bts_main.py:250-319 as the reference does it (per image: five-output forward, prediction and ground truth to the host,
    numpy masks + compute_errors) -- the checker of online_eval_device.  Uses the oracle's restatement of the host arithmetic."""
    from oracle import eval_oracle as E
    em = np.zeros(10, dtype=np.float64)
    with torch.no_grad():
        for s in eval_samples:
            if not s.get("has_valid_depth", True):
                continue
            pred = model(s["image"].cuda(), s["focal"].cuda())[4].cpu().numpy().squeeze()
            gt = s["depth"].cpu().numpy().squeeze()
            pf, valid = E.eval_prepare(pred, gt, args.min_depth_eval, args.max_depth_eval, args.dataset, args.do_kb_crop,
                                       args.garg_crop, args.eigen_crop)
            em[:9] += np.array(E.compute_errors(gt[valid], pf[valid]), dtype=np.float64)
            em[9] += 1
    return em / em[9]


def online_eval_device(model, eval_samples, args, log=None):
    """The same evaluation on `bts_eval_errors`: nothing but the 10-float measure vector leaves the device."""
    from bts_amd import loops
    return loops.online_eval(model, eval_samples, args.dataset, args.min_depth_eval, args.max_depth_eval, args.do_kb_crop,
                             args.garg_crop, args.eigen_crop, device="cuda", log=log)
