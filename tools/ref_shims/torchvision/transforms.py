"""torchvision.transforms.Compose / Normalize with torchvision's semantics (per-channel (x - mean) / std on a CHW
tensor, out of place)."""
import torch


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class Normalize:
    def __init__(self, mean, std, inplace=False):
        self.mean, self.std = list(mean), list(std)

    def __call__(self, tensor):
        mean = torch.as_tensor(self.mean, dtype=tensor.dtype, device=tensor.device).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=tensor.dtype, device=tensor.device).view(-1, 1, 1)
        return (tensor - mean) / std
