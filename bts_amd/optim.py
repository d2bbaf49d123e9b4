"""Fused multi-tensor AdamW on the HIP kernel `bts_adamw_step`.

Semantics and state layout of ``torch.optim.AdamW`` (bts_main.py:371-373): per-parameter state
``{'step', 'exp_avg', 'exp_avg_sq'}``, index-keyed ``state_dict()``, so optimizer checkpoints written by
the reference's AdamW load here and vice versa (bts_main.py:383-387, 498-503).  One kernel launch per
parameter group replaces the per-tensor update chain; the per-step poly learning rate (bts_main.py:456-458)
and the bias corrections are read from a small device tensor, so a captured hipGraph replays correctly.

Gradients must keep their storage between steps (call ``zero_grad(set_to_none=False)``, the default here):
the device pointer tables are built once.
"""
import ctypes as C

import torch

from ._lib import call, stream_ptr


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._tables = {}
        self._hyper = None
        self._steps = 0

    def zero_grad(self, set_to_none=False):
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()
        if set_to_none:
            self._tables = {}

    def _table(self, gi, plist):
        key = tuple(p.grad.data_ptr() for p in plist)
        tb = self._tables.get(gi)
        if tb is not None and tb[0] == key:
            return tb[1]
        dev = plist[0].device
        st = [self.state[p] for p in plist]

        def arr(ts):
            return torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64, device=dev)
        t = dict(params=arr(plist), grads=arr([p.grad for p in plist]), m1=arr([s["exp_avg"] for s in st]),
                 m2=arr([s["exp_avg_sq"] for s in st]),
                 sizes=torch.tensor([p.numel() for p in plist], dtype=torch.int64, device=dev),
                 n=len(plist), max_size=max(p.numel() for p in plist))
        self._tables[gi] = (key, t)
        return t

    def prepare_step(self, lrs=None):
        """Host side of a step: advance the step count and refresh the device-resident {lr, bc1, bc2} rows.
        Call OUTSIDE a captured graph (step() calls it itself when not capturing)."""
        self._steps += 1
        rows = []
        for gi, g in enumerate(self.param_groups):
            lr = float(g["lr"] if lrs is None else lrs[gi])
            b1, b2 = g["betas"]
            rows.append([lr, 1.0 - b1 ** self._steps, 1.0 - b2 ** self._steps, 0.0])
        dev = self.param_groups[0]["params"][0].device
        host = torch.tensor(rows, dtype=torch.float32)
        if self._hyper is None:
            self._hyper = host.to(dev)
        else:
            self._hyper.copy_(host, non_blocking=False)

    @torch.no_grad()
    def step(self, closure=None, prepared=False):
        if not prepared:
            self.prepare_step()
        for gi, g in enumerate(self.param_groups):
            plist = [p for p in g["params"] if p.grad is not None]
            if not plist:
                continue
            for p in plist:
                st = self.state[p]
                if not st:
                    st["step"] = torch.zeros((), dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
                st["step"] += 1
                dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                if not (dense and p.grad.stride() == p.stride() and p.dtype == torch.float32 and p.grad.dtype == torch.float32):
                    raise RuntimeError("FusedAdamW needs dense f32 parameters with identically laid out gradients")
            t = self._table(gi, plist)
            b1, b2 = g["betas"]
            hyper = self._hyper[gi]
            call("bts_adamw_step", C.c_void_p(t["params"].data_ptr()), C.c_void_p(t["grads"].data_ptr()),
                 C.c_void_p(t["m1"].data_ptr()), C.c_void_p(t["m2"].data_ptr()), C.c_void_p(t["sizes"].data_ptr()),
                 t["n"], t["max_size"], 0.0, float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]), 1.0, 1.0,
                 C.c_void_p(hyper.data_ptr()), stream_ptr())
            # parameters were updated through raw pointers: tell autograd / any version-keyed cache
            torch.autograd.graph.increment_version(plist)
        return None
