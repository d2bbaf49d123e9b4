"""Host side of the evaluation kernels (csrc/evalops.hip): device-resident replacement of the per-image numpy loop of
``online_eval`` (pytorch/bts_main.py:255-303) and of the uint16 depth payload of ``bts_test.py:179-185``.

``online_eval`` moves every prediction and ground-truth map to the host and runs ~20 numpy passes per image; here the
nine metrics of a whole batch are one reduction on the device and only the 10-float ``eval_measures`` tensor (the one
the reference all-reduces, bts_main.py:301-303) is ever read back.
"""
import ctypes as C

import torch

from ._lib import BtsAmdError, call, require_gpu, stream_ptr

EVAL_METRICS = ["silog", "abs_rel", "log10", "rms", "sq_rel", "log_rms", "d1", "d2", "d3"]   # bts_main.py:140


def crop_window(dataset, gt_h, gt_w, garg_crop=False, eigen_crop=False):
    """[y0, y1) x [x0, x1) of the evaluation mask (bts_main.py:283-293)."""
    if garg_crop:
        return int(0.40810811 * gt_h), int(0.99189189 * gt_h), int(0.03594771 * gt_w), int(0.96405229 * gt_w)
    if eigen_crop:
        if dataset == "kitti":
            return int(0.3324324 * gt_h), int(0.91351351 * gt_h), int(0.0359477 * gt_w), int(0.96405229 * gt_w)
        return 45, 471, 41, 601
    return 0, gt_h, 0, gt_w


def compute_errors(pred_depth, gt_depth, min_depth_eval, max_depth_eval, dataset="kitti", do_kb_crop=False, garg_crop=False,
                   eigen_crop=False, has_valid_depth=None, eval_measures=None):
    """pred_depth [B,1,Hp,Wp] or [B,Hp,Wp] (the model's final_depth), gt_depth [B,(1,)Hg,Wg], both f32 on the GPU.

    Returns measures [B, 9] (order = EVAL_METRICS).  If ``eval_measures`` (f32[10] on the GPU) is given it is updated in
    place exactly as bts_main.py:298-299 does, image by image."""
    require_gpu(pred_depth)
    require_gpu(gt_depth)
    pred = pred_depth.reshape(pred_depth.shape[0], pred_depth.shape[-2], pred_depth.shape[-1])
    gt = gt_depth.reshape(gt_depth.shape[0], gt_depth.shape[-2], gt_depth.shape[-1])
    if pred.dtype != torch.float32 or gt.dtype != torch.float32:
        raise BtsAmdError("compute_errors expects f32 tensors")
    pred, gt = pred.contiguous(), gt.contiguous()
    B, Hp, Wp = pred.shape
    _, Hg, Wg = gt.shape
    top, left = 0, 0
    if do_kb_crop:                                              # bts_main.py:268-274
        top, left = int(Hg - 352), int((Wg - 1216) / 2)
        if (Hp, Wp) != (352, 1216):
            raise BtsAmdError("do_kb_crop expects 352x1216 predictions")
    elif (Hp, Wp) != (Hg, Wg):
        raise BtsAmdError("prediction and ground truth sizes differ without do_kb_crop")
    y0, y1, x0, x1 = crop_window(dataset, Hg, Wg, garg_crop, eigen_crop)
    ws = torch.empty(call("bts_eval_workspace_bytes", B), dtype=torch.uint8, device=pred.device)
    measures = torch.empty((B, 9), dtype=torch.float32, device=pred.device)
    hv = None
    if has_valid_depth is not None:
        hv = has_valid_depth.to(device=pred.device, dtype=torch.uint8).contiguous()
    if eval_measures is not None and (eval_measures.dtype != torch.float32 or eval_measures.numel() != 10 or not eval_measures.is_cuda):
        raise BtsAmdError("eval_measures must be f32[10] on the GPU")
    call("bts_eval_errors", C.c_void_p(pred.data_ptr()), C.c_void_p(gt.data_ptr()), C.c_void_p(hv.data_ptr() if hv is not None else None),
         B, Hp, Wp, Hg, Wg, top, left, float(min_depth_eval), float(max_depth_eval), y0, y1, x0, x1, C.c_void_p(ws.data_ptr()),
         C.c_void_p(measures.data_ptr()), C.c_void_p(eval_measures.data_ptr() if eval_measures is not None else None), stream_ptr())
    return measures


def depth_to_uint16(pred_depth, dataset):
    """bts_test.py:179-185: the uint16 image that is written as PNG (depth * 256 for kitti, * 1000 otherwise)."""
    require_gpu(pred_depth)
    if pred_depth.dtype != torch.float32:
        raise BtsAmdError("depth_to_uint16 expects an f32 tensor")
    d = pred_depth.contiguous()
    out = torch.empty(d.shape, dtype=torch.uint16, device=d.device)
    scale = 256.0 if dataset in ("kitti", "kitti_benchmark") else 1000.0
    call("bts_depth_to_u16", C.c_void_p(d.data_ptr()), C.c_void_p(out.data_ptr()), d.numel(), scale, stream_ptr())
    return out
