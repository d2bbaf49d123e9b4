"""Device-side training-sample preprocessing (csrc/evalops.hip::preprocess_train_kernel): the arithmetic of
``DataLoadPreprocess.__getitem__`` after decoding (pytorch/bts_dataloader.py:126-136, 190-235) and of ``ToTensor``
(:240-250) for a whole batch in one kernel.  The random draws are made on the host, in the reference's order, by
``draw_train_params`` -- so the augmentation statistics, and with equal seeds the exact samples, are the reference's.
"""
import ctypes as C
import random

import numpy as np
import torch

from ._lib import AugParams, BtsAmdError, call, require_gpu, stream_ptr


def draw_train_params(src_h, src_w, height, width, dataset):
    """One sample's draws from python `random` / `np.random`, consumed as random_crop (:195-196), train_preprocess
    (:203, :209) and augment_image (:217, :221-224, :228) consume them."""
    p = AugParams()
    p.crop_x = random.randint(0, src_w - width)
    p.crop_y = random.randint(0, src_h - height)
    p.flip = int(random.random() > 0.5)
    p.augment = int(random.random() > 0.5)
    p.gamma, p.brightness = 1.0, 1.0
    cols = (1.0, 1.0, 1.0)
    if p.augment:
        p.gamma = random.uniform(0.9, 1.1)
        p.brightness = random.uniform(0.75, 1.25) if dataset == "nyu" else random.uniform(0.9, 1.1)
        cols = np.random.uniform(0.9, 1.1, size=3)
    for i in range(3):
        p.color[i] = float(cols[i])
    return p


def preprocess_train(images_u8, depth_raw, params, height, width, dataset):
    """images_u8 [B,Hs,Ws,3] uint8, depth_raw [B,Hs,Ws] int32 (both on the GPU), params: list of AugParams.
    Returns (image f32 [B,3,H,W] normalised, depth f32 [B,1,H,W] in metres) = the 'image' / 'depth' entries of the
    reference's collated training batch."""
    require_gpu(images_u8)
    require_gpu(depth_raw)
    if images_u8.dtype != torch.uint8 or depth_raw.dtype != torch.int32 or images_u8.shape[-1] != 3:
        raise BtsAmdError("preprocess_train expects uint8 [B,Hs,Ws,3] images and int32 [B,Hs,Ws] depth payloads")
    B, Hs, Ws, _ = images_u8.shape
    if len(params) != B or tuple(depth_raw.shape) != (B, Hs, Ws):
        raise BtsAmdError("batch / shape mismatch")
    for p in params:
        if not (0 <= p.crop_x <= Ws - width and 0 <= p.crop_y <= Hs - height):
            raise BtsAmdError("crop window outside the source image")
    raw = b"".join(bytes(p) for p in params)
    dev_params = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(images_u8.device)
    img = torch.empty((B, 3, height, width), dtype=torch.float32, device=images_u8.device)
    dep = torch.empty((B, 1, height, width), dtype=torch.float32, device=images_u8.device)
    call("bts_preprocess_train", C.c_void_p(images_u8.contiguous().data_ptr()), C.c_void_p(depth_raw.contiguous().data_ptr()),
         C.c_void_p(dev_params.data_ptr()), B, Hs, Ws, height, width, 1000.0 if dataset == "nyu" else 256.0,
         C.c_void_p(img.data_ptr()), C.c_void_p(dep.data_ptr()), stream_ptr())
    return img, dep
