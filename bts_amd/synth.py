"""Synthetic inputs of the benchmark: focal lengths as the reference's file lists carry them, and a
sparse ground-truth depth map (KITTI ground truth is sparse: ~30 % holes exercise the loss mask, bts_main.py:449-454).

Data generation only -- no model arithmetic.  The test-side recipe (oracle/bts_oracle.py) defines the same two functions; a CPU
test holds them bit-identical, so bench.py does not import anything from oracle/ to build its batch.
"""
import torch

KITTI_FOCALS = (721.5377, 718.856, 707.0912, 718.3351, 707.0493)   # train_test_inputs/eigen_*.txt column 3
NYU_FOCAL = 518.8579                                               # train_test_inputs/nyudepthv2_*.txt column 3


def synth_focal(batch, dataset):
    """float64 [batch], as the reference's collate produces it (bts_dataloader.py:140)."""
    vals = KITTI_FOCALS if dataset == "kitti" else (NYU_FOCAL,)
    return torch.tensor([vals[i % len(vals)] for i in range(batch)], dtype=torch.float64)


def synth_depth_gt(batch, H, W, dataset, gen):
    """U(lo, max_depth - 0.5) with 30 % of the pixels set to 0 (= invalid: below the gt > 1.0 / gt > 0.1 mask threshold)."""
    max_depth = 80.0 if dataset == "kitti" else 10.0
    lo = 1.5 if dataset == "kitti" else 0.5
    gt = torch.rand(batch, 1, H, W, generator=gen) * (max_depth - 0.5 - lo) + lo
    hole = torch.rand(batch, 1, H, W, generator=gen) < 0.3
    return gt.masked_fill(hole, 0.0)
