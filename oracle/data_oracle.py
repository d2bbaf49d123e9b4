"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the reference's training-sample preprocessing that sits
between the decoded image and the model input; nothing in the product path imports this.

  draw_train_params   pytorch/bts_dataloader.py:190-235  the order in which random_crop / train_preprocess /
                                                         augment_image consume `random` and `np.random`
  preprocess_train    bts_dataloader.py:126-136, 190-235, 240-250  uint8 RGB + raw depth -> normalised CHW f32 image
                                                         and f32 depth in metres (crop, flip, gamma / brightness /
                                                         colour augmentation, clip, ToTensor, ImageNet normalise)

Pinned (tests/golden/preprocess.npz, tools/make_golden.py): random_crop / train_preprocess / augment_image are executed
from the source text of the unmodified bts_dataloader.py (the module itself needs torchvision) with seeded generators;
the Normalize step of ToTensor is torchvision's (x - mean) / std in f32, restated here.
"""
import random

import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)      # bts_dataloader.py:243
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def draw_train_params(seed, src_h, src_w, height, width, dataset):
    """Consumes python `random` and `np.random` exactly as random_crop (:190-199), train_preprocess (:201-213) and
    augment_image (:215-235) do, after seeding both with `seed`."""
    random.seed(seed)
    np.random.seed(seed)
    p = {}
    p["crop_x"] = random.randint(0, src_w - width)               # :195
    p["crop_y"] = random.randint(0, src_h - height)              # :196
    p["flip"] = int(random.random() > 0.5)                       # :203-204
    p["augment"] = int(random.random() > 0.5)                    # :209-210
    p["gamma"], p["brightness"], p["colors"] = 1.0, 1.0, np.ones(3)
    if p["augment"]:
        p["gamma"] = random.uniform(0.9, 1.1)                    # :217
        p["brightness"] = random.uniform(0.75, 1.25) if dataset == "nyu" else random.uniform(0.9, 1.1)   # :221-224
        p["colors"] = np.random.uniform(0.9, 1.1, size=3)        # :228
    return p


def preprocess_train(image_u8, depth_raw, p, height, width, dataset):
    """image_u8 [Hs, Ws, 3] uint8, depth_raw [Hs, Ws] (PNG payload: metres * 256 for kitti, * 1000 for nyu)."""
    image = np.asarray(image_u8, dtype=np.float32) / 255.0       # :126
    depth = np.asarray(depth_raw, dtype=np.float32)[:, :, None]  # :127-128
    depth = depth / 1000.0 if dataset == "nyu" else depth / 256.0     # :130-133
    x, y = p["crop_x"], p["crop_y"]
    image = image[y:y + height, x:x + width, :]                  # :197-198
    depth = depth[y:y + height, x:x + width, :]
    if p["flip"]:                                                # :204-206
        image = image[:, ::-1, :].copy()
        depth = depth[:, ::-1, :].copy()
    if p["augment"]:                                             # :215-235
        aug = image ** p["gamma"]
        aug = aug * p["brightness"]
        white = np.ones((image.shape[0], image.shape[1]))
        color_image = np.stack([white * p["colors"][i] for i in range(3)], axis=2)
        aug *= color_image
        image = np.clip(aug, 0, 1)
    chw = np.ascontiguousarray(image.transpose(2, 0, 1))         # ToTensor.to_tensor (:262-264)
    chw = (chw - MEAN[:, None, None]) / STD[:, None, None]       # transforms.Normalize (:243, 248)
    return chw.astype(np.float32), np.ascontiguousarray(depth.transpose(2, 0, 1)), image
