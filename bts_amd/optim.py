"""Fused multi-tensor AdamW on the HIP kernels `bts_adamw_advance` + `bts_adamw_step`.

Semantics and state layout of ``torch.optim.AdamW`` (bts_main.py:371-373): per-parameter state
``{'step', 'exp_avg', 'exp_avg_sq'}``, index-keyed ``state_dict()``, so optimizer checkpoints written by
the reference's AdamW load here and vice versa (bts_main.py:383-387, 498-503).  One kernel launch per
parameter group replaces the per-tensor update chain.

The step count lives on the DEVICE, one 8-float row per parameter group
(``{lr, bias_c1, bias_c2, step, beta1, beta2, -, -}``): ``bts_adamw_advance`` increments it and refreshes the bias
corrections in front of the update kernels, so a captured hipGraph advances the optimizer on every replay with no
host involvement.  The host only rewrites ``lr`` (the per-step poly schedule, bts_main.py:456-458: ``prepare_step``)
and reads ``step`` back when a checkpoint is written (``state_dict``); ``load_state_dict`` restores the counter
from the checkpoint's per-parameter ``step``, so resumed training continues with the right bias corrections.

Restriction (documented, checked): the step count is ONE value per parameter group, where ``torch.optim.AdamW`` keeps one per
parameter.  The two coincide whenever every parameter of a group receives a gradient on every step -- the case of
bts_main.py, where frozen parameters are filtered out before the groups are built.  ``load_state_dict`` refuses a checkpoint
whose per-parameter steps differ inside a group instead of silently rewriting them, and ``step(closure)`` is not supported.

Gradients must keep their storage between steps (call ``zero_grad(set_to_none=False)``, the default here):
the device pointer tables are rebuilt only when a pointer changes.
"""
import ctypes as C

import torch

from . import profiler
from ._lib import BtsAmdError, call, stream_ptr

_ROW = 8        # floats per group row (include/bts_amd.h: bts_adamw_advance)


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._tables = {}
        self._hyper = None          # [n_groups, 8] f32 on the device

    # ---- device-resident schedule / step counter ------------------------------------------------
    def _device(self):
        for g in self.param_groups:
            for p in g["params"]:
                return p.device
        raise BtsAmdError("FusedAdamW has no parameters")

    def _group_step_from_state(self, g):
        """Step count recorded in the per-parameter state of group g (0 if the group has no state yet)."""
        seen = set()
        for p in g["params"]:
            st = self.state.get(p)
            if st and "step" in st:
                seen.add(float(st["step"]))
        if len(seen) > 1:
            raise BtsAmdError("FusedAdamW keeps one step count per parameter group; this state has steps %s inside one group "
                              "(load it with torch.optim.AdamW, or regroup the parameters)" % sorted(seen))
        return seen.pop() if seen else 0.0

    def _rows(self):
        """The device rows, created on first use from the groups' hyper-parameters and recorded state."""
        if self._hyper is None:
            rows = []
            for g in self.param_groups:
                b1, b2 = g["betas"]
                rows.append([float(g["lr"]), 1.0, 1.0, self._group_step_from_state(g), float(b1), float(b2), 0.0, 0.0])
            self._hyper = torch.tensor(rows, dtype=torch.float32).to(self._device())
        return self._hyper

    def prepare_step(self, lrs=None):
        """Host side of a step: write the groups' learning rates into the device rows.  Call OUTSIDE a captured
        graph (step() calls it itself unless ``prepared=True``)."""
        rows = self._rows()
        vals = [float(g["lr"] if lrs is None else lrs[gi]) for gi, g in enumerate(self.param_groups)]
        rows[:, 0].copy_(torch.tensor(vals, dtype=torch.float32))

    def device_steps(self):
        """Per-group step counts as python floats (synchronises)."""
        if self._hyper is None:
            return [self._group_step_from_state(g) for g in self.param_groups]
        return [float(v) for v in self._hyper[:, 3].cpu()]

    # ---- torch.optim.Optimizer surface -------------------------------------------------------------
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

    def state_dict(self):
        """torch.optim.AdamW layout; the per-parameter ``step`` entries are refreshed from the device counter first
        (they are not touched by step(), which may be running inside a replayed graph)."""
        steps = self.device_steps()
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                st = self.state.get(p)
                if st:
                    st["step"] = torch.tensor(steps[gi], dtype=torch.float32)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # the moment buffers were replaced and the step count comes from the checkpoint: drop every cached pointer
        # table and rebuild the device rows (step, betas, lr) from what was just loaded
        self._tables = {}
        self._hyper = None
        self._rows()

    def _table(self, gi, plist):
        st = [self.state[p] for p in plist]
        key = tuple((p.data_ptr(), p.grad.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr())
                    for p, s in zip(plist, st))
        tb = self._tables.get(gi)
        if tb is not None and tb[0] == key:
            return tb[1]
        dev = plist[0].device

        def arr(ts):
            return torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64, device=dev)
        t = dict(params=arr(plist), grads=arr([p.grad for p in plist]), m1=arr([s["exp_avg"] for s in st]),
                 m2=arr([s["exp_avg_sq"] for s in st]),
                 sizes=torch.tensor([p.numel() for p in plist], dtype=torch.int64, device=dev),
                 n=len(plist), max_size=max(p.numel() for p in plist))
        self._tables[gi] = (key, t)
        return t

    @torch.no_grad()
    def step(self, closure=None, prepared=False):
        if closure is not None:
            raise BtsAmdError("FusedAdamW.step() does not support a closure (bts_main.py never passes one)")
        if not prepared:
            self.prepare_step()
        rows = self._rows()
        call("bts_adamw_advance", C.c_void_p(rows.data_ptr()), len(self.param_groups), stream_ptr())
        for gi, g in enumerate(self.param_groups):
            plist = [p for p in g["params"] if p.grad is not None]
            if not plist:
                continue
            for p in plist:
                st = self.state[p]
                if not st:
                    st["step"] = torch.zeros((), dtype=torch.float32)       # refreshed by state_dict()
                    st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
                dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                same = all(a == b for a, b, n in zip(p.grad.stride(), p.stride(), p.shape) if n != 1)     # (a size-1 dimension's stride is free)
                if not (dense and same and p.dtype == torch.float32 and p.grad.dtype == torch.float32):
                    raise RuntimeError("FusedAdamW needs dense f32 parameters with identically laid out gradients")
            t = self._table(gi, plist)
            b1, b2 = g["betas"]
            if profiler.ACTIVE is not None:      # p, g, m, v read; p, m, v written
                profiler.note("adamw_step", "hbm", 28.0 * sum(p.numel() for p in plist))
            call("bts_adamw_step", C.c_void_p(t["params"].data_ptr()), C.c_void_p(t["grads"].data_ptr()),
                 C.c_void_p(t["m1"].data_ptr()), C.c_void_p(t["m2"].data_ptr()), C.c_void_p(t["sizes"].data_ptr()),
                 t["n"], t["max_size"], 0.0, float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]), 1.0, 1.0,
                 C.c_void_p(rows.data_ptr() + 4 * _ROW * gi), stream_ptr())
            # parameters were updated through raw pointers: tell autograd / any version-keyed cache
            torch.autograd.graph.increment_version(plist)
        return None
