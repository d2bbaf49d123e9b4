"""TEST INFRASTRUCTURE ONLY -- lets the unmodified reference drivers exercise the drop-in's PLUMBING on a box without
a GPU (tests/test_reference_drivers.py; the build container has /root/reference but no GPU, the GPU box has a GPU but
no /root/reference).

Imported at interpreter start-up of the driver process (BTS_REF_POSTIMPORT, tools/ref_shims/site/sitecustomize.py).
It replaces the three forward() bodies of bts_amd.model that launch HIP kernels with the CPU oracle, so everything
around them is the product's real code: module tree and names, parameter registration order, state-dict keys, the
drop-in file being copied and re-imported by name, optimizer parameter groups, checkpoint save / load / resume.
The product path itself has no CPU execution (bts_amd never imports oracle/, tests/test_model_structure.py).
"""
import torch

import bts_amd.model as M
from oracle import bts_oracle as O


def _decoder_forward(self, features, focal):
    P = dict(self.named_parameters())
    P.update(dict(self.named_buffers()))
    training = any(m.training for m in self.modules() if isinstance(m, torch.nn.BatchNorm2d))
    outs, updates = O.decoder_forward(P, [f.float() for f in features[:5]], focal, self.params.max_depth,
                                      self.params.dataset, training)
    with torch.no_grad():
        for k, v in updates.items():
            P[k].copy_(v)
        if training:
            for n, b in self.named_buffers():
                if n.endswith("num_batches_tracked"):
                    b.add_(1)
    return outs


def _silog_forward(self, depth_est, depth_gt, mask):
    return O.silog(depth_est, depth_gt, mask, self.variance_focus)


M.bts.forward = _decoder_forward
M.silog_loss.forward = _silog_forward
