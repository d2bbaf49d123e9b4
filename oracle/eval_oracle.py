"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the reference's evaluation arithmetic next to the hot path
(SURVEY.md section 8f, rows 3 and 4).  Nothing in the product path may import this file; tests, smoke() and bench.py's
cpu_baseline leg only.

  compute_errors      pytorch/bts_main.py:143-165   the nine depth metrics on the masked pixel lists
  eval_prepare        pytorch/bts_main.py:268-296   kb-crop paste-back, clamping, validity + garg / eigen crop masks
  depth_to_uint16     pytorch/bts_test.py:179-185   the 16-bit PNG payload (depth * 256 for kitti, * 1000 for nyu)

Pinned: compute_errors against the reference's own function body (tests/golden/eval.npz, written by
tools/make_golden.py, which executes the function's source text from the unmodified bts_main.py -- the module itself
cannot be imported here: tensorboardX / cv2 are missing).  eval_prepare is inline code of online_eval() in the
reference, not a function: restated line by line, parity unpinned beyond the compute_errors it feeds.
"""
import numpy as np


def compute_errors(gt, pred):
    """bts_main.py:143-165 -- returns [silog, abs_rel, log10, rms, sq_rel, log_rms, d1, d2, d3]."""
    thresh = np.maximum((gt / pred), (pred / gt))
    d1 = (thresh < 1.25).mean()
    d2 = (thresh < 1.25 ** 2).mean()
    d3 = (thresh < 1.25 ** 3).mean()
    rms = np.sqrt(((gt - pred) ** 2).mean())
    log_rms = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    err = np.log(pred) - np.log(gt)
    silog = np.sqrt(np.mean(err ** 2) - np.mean(err) ** 2) * 100
    log10 = np.mean(np.abs(np.log10(pred) - np.log10(gt)))
    return [silog, abs_rel, log10, rms, sq_rel, log_rms, d1, d2, d3]


def crop_window(dataset, gt_h, gt_w, garg_crop, eigen_crop):
    """Row / column window [y0, y1) x [x0, x1) of the evaluation mask (bts_main.py:282-293); full image if no crop."""
    if garg_crop:
        return int(0.40810811 * gt_h), int(0.99189189 * gt_h), int(0.03594771 * gt_w), int(0.96405229 * gt_w)
    if eigen_crop:
        if dataset == "kitti":
            return int(0.3324324 * gt_h), int(0.91351351 * gt_h), int(0.0359477 * gt_w), int(0.96405229 * gt_w)
        return 45, 471, 41, 601
    return 0, gt_h, 0, gt_w


def eval_prepare(pred, gt, min_depth, max_depth, dataset, do_kb_crop, garg_crop, eigen_crop):
    """bts_main.py:268-296 for ONE image: returns (pred_full, valid_mask)."""
    pred = np.array(pred, dtype=np.float32, copy=True)
    if do_kb_crop:                                                   # :268-274
        height, width = gt.shape
        top_margin = int(height - 352)
        left_margin = int((width - 1216) / 2)
        unc = np.zeros((height, width), dtype=np.float32)
        unc[top_margin:top_margin + 352, left_margin:left_margin + 1216] = pred
        pred = unc
    pred[pred < min_depth] = min_depth                              # :276-279
    pred[pred > max_depth] = max_depth
    pred[np.isinf(pred)] = max_depth
    pred[np.isnan(pred)] = min_depth
    valid = np.logical_and(gt > min_depth, gt < max_depth)          # :281
    if garg_crop or eigen_crop:                                     # :283-295
        y0, y1, x0, x1 = crop_window(dataset, gt.shape[0], gt.shape[1], garg_crop, eigen_crop)
        em = np.zeros(valid.shape)
        em[y0:y1, x0:x1] = 1
        valid = np.logical_and(valid, em)
    return pred, valid


def depth_to_uint16(pred_depth, dataset):
    """bts_test.py:179-185."""
    scaled = pred_depth * 256.0 if dataset in ("kitti", "kitti_benchmark") else pred_depth * 1000.0
    return scaled.astype(np.uint16)


EVAL_CASES = {   # tag: gt_h, gt_w, pred_h, pred_w, max_depth_eval, do_kb_crop, garg_crop, eigen_crop, dataset
    "kitti": (375, 1242, 352, 1216, 80.0, True, True, False, "kitti"),      # arguments_train_eigen.txt: kb crop + garg crop
    "nyu": (480, 640, 480, 640, 10.0, False, False, True, "nyu"),           # arguments_train_nyu.txt: eigen crop
}


def synth_eval_case(tag):
    """Seeded synthetic (pred, gt) pair of an evaluation case (legacy RandomState: the stream is frozen across numpy
    versions, so the golden file stores only the reference's outputs).  Sparse gt, predictions within x0.5..x2.2 of it,
    plus the inf / nan / negative pixels that bts_main.py:276-279 patches."""
    gh, gw, ph, pw, md, kb, garg, eig, ds = EVAL_CASES[tag]
    rng = np.random.RandomState(5 + len(tag))
    gt = rng.uniform(0.3, md * 1.05, size=(gh, gw)).astype(np.float32)
    gt[rng.uniform(size=gt.shape) < 0.4] = 0.0
    base = gt[gh - ph:, (gw - pw) // 2:(gw - pw) // 2 + pw] if kb else gt
    pred = base * rng.uniform(0.5, 2.2, size=(ph, pw)).astype(np.float32)
    pred = np.where(pred == 0, rng.uniform(0.5, md, size=pred.shape), pred).astype(np.float32)
    pred[0, 0], pred[1, 1], pred[2, 2] = np.inf, np.nan, -1.0
    return pred, gt
