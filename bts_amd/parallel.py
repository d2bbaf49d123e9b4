"""Data-parallel gradient exchange for BTS training: one process per GPU, RCCL over xGMI.

The reference wraps the model in ``DistributedDataParallel`` (bts_main.py:352) and the only
exchange step of the hot path is the mean of the gradients across ranks, once per step
(BatchNorm statistics and the loss are rank-local, there is no SyncBN).

``GradAllReducer`` is that exchange, written for the way this decoder produces gradients:

* gradients live in a few large flat buckets (``param.grad`` are views into them, so autograd
  accumulates in place and there is no flatten/unflatten copy);
* buckets follow REVERSE parameter order -- the decoder's parameters are registered last and its
  gradients all appear at the end of the fused decoder backward, i.e. first; the encoder's follow
  layer by layer.  Each bucket's all-reduce is launched the moment its last gradient has been
  accumulated (post-accumulate-grad hooks) with ``async_op=True``: RCCL runs it on its own stream
  behind an event on the compute stream, so the exchange of bucket i overlaps the backward
  compute of everything registered before it;
* bucket size defaults to 64 MiB: on MI355X nodes the 8 GPUs are fully connected by point-to-point
  xGMI links (7 x ~153 GB/s per GPU), so per-message latency, not switch bandwidth, is what small
  buckets pay for; ~47 M parameters (DenseNet161-BTS, 188 MB f32) become 3 large messages -- plus one
  small one: the LAST bucket (the earliest encoder layers, whose gradients close the backward pass) is
  capped at ``tail_bytes`` (8 MiB), because its exchange is the only one nothing can overlap with;
* the mean costs no extra pass at the end of the step: ``ReduceOp.AVG`` where the backend has it (RCCL),
  otherwise the bucket is scaled by 1/world when it is launched, i.e. under the rest of the backward pass.

One synchronising backward per step (hooks count gradients down to zero); gradient accumulation uses
``no_sync()`` for the earlier micro-batches, and a stray second backward raises instead of racing the in-flight exchange.

``launch_log`` records, for every bucket exchange launched from a hook, how many gradient hooks had fired by then: the
tests use it to hold "the decoder bucket is on the wire before the encoder's backward has finished".
"""
import contextlib

import torch
import torch.distributed as dist


def _grad_view(flat, off, p):
    """The slice [off, off + p.numel()) of a flat bucket as a gradient laid out LIKE THE PARAMETER: the same strides over the
    bucket's storage.  A contiguous view under a channels-last parameter (the stock encoder in torch.channels_last, round 6) is a
    layout autograd has to convert into on every accumulation and the fused optimizer refuses (it walks parameter and gradient with
    one index)."""
    # dense in some dimension order (what .to(memory_format=...) produces): sorted by stride, every stride is the product of the
    # sizes behind it
    dims = sorted(range(p.dim()), key=lambda d: (p.stride(d), -p.size(d)), reverse=True)
    expect = 1
    for d in reversed(dims):
        if p.size(d) != 1 and p.stride(d) != expect:
            raise ValueError("GradAllReducer: parameter of shape %s with strides %s is not dense" % (tuple(p.shape), p.stride()))
        expect *= p.size(d)
    return torch.as_strided(flat, p.size(), p.stride(), off)


class GradAllReducer:
    def __init__(self, params, bucket_bytes=64 << 20, process_group=None, reduce_single=False, tail_bytes=8 << 20):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        backend = dist.get_backend(process_group) if dist.is_initialized() else ""
        self.use_avg = backend == "nccl"        # RCCL averages in the collective; gloo has no ReduceOp.AVG
        self.launch_log = []                    # (bucket index, gradient hooks fired so far) per hook-driven launch
        self._fired = 0
        # reduce_single: issue the collectives even at world size 1 (exercises the RCCL path on a one-GPU box)
        self.collective = self.world > 1 or (reduce_single and dist.is_initialized())
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []        # (flat tensor, [params])
        self._pending = {}
        self._works = []
        self._hooks = []
        self._defer = False
        # diagnostics of the exchange (bench.py --gpus N): time the compute stream spends waiting for collectives AFTER the last
        # gradient of the backward pass (finish()), i.e. the part of the exchange nothing overlapped
        self.timing = False
        self._exposed = []       # (event before the waits, event after them) per finish(), or wall-clock seconds without a GPU
        self.fraction_log = []   # per step: [(bucket, hooks fired at its launch / hooks of the step)] (launch_log is reset per step)
        self._build(bucket_bytes, tail_bytes)

    def _build(self, bucket_bytes, tail_bytes):
        cur, cur_bytes = [], 0
        groups = []
        remaining = sum(p.numel() * p.element_size() for p in self.params)
        in_tail = False
        for p in reversed(self.params):
            nb = p.numel() * p.element_size()
            start_tail = not in_tail and remaining <= tail_bytes and remaining < bucket_bytes
            if cur and (cur_bytes + nb > bucket_bytes or start_tail or p.dtype != cur[0].dtype or p.device != cur[0].device):
                groups.append(cur)
                cur, cur_bytes = [], 0
            in_tail = in_tail or start_tail
            cur.append(p)
            cur_bytes += nb
            remaining -= nb
        if cur:
            groups.append(cur)
        for bi, ps in enumerate(groups):
            flat = torch.zeros(sum(p.numel() for p in ps), dtype=ps[0].dtype, device=ps[0].device)
            off = 0
            for p in ps:
                p.grad = _grad_view(flat, off, p)                  # autograd accumulates into the bucket
                off += p.numel()
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
            self.buckets.append((flat, ps))
        self._reset()

    def _make_hook(self, bi):
        def hook(param):
            if self._defer:                         # inside no_sync(): gradients only accumulate in the bucket
                return
            self._fired += 1
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self.launch_log.append((bi, self._fired))
                self._launch(bi)
            elif self._pending[bi] < 0:
                raise RuntimeError(
                    "GradAllReducer: a second backward() reached bucket %d after its all-reduce was launched; the "
                    "contract is ONE synchronising backward per zero_grad()/finish() pair -- wrap the earlier "
                    "micro-batch backwards of a gradient-accumulation step in `with reducer.no_sync():`" % bi)
        return hook

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation (the DDP idiom): backward passes inside this context only add into the flat buckets;
        the exchange is launched by the first backward outside it (or by finish())."""
        self._defer = True
        try:
            yield
        finally:
            self._defer = False

    def _reset(self):
        if self.timing and self.launch_log:
            total = float(max(1, len(self.params)))
            self.fraction_log.append([(bi, fired / total) for bi, fired in self.launch_log])
        self._pending = {bi: len(ps) for bi, (_, ps) in enumerate(self.buckets)}
        self._works = []
        self._fired = 0
        self.launch_log = []

    def _launch(self, bi):
        flat = self.buckets[bi][0]
        if self.collective:
            if self.use_avg:
                op = dist.ReduceOp.AVG
            else:
                op = dist.ReduceOp.SUM
                if self.world > 1:
                    flat.mul_(1.0 / self.world)       # pre-scale at launch time: hidden under the remaining backward
            self._works.append((bi, dist.all_reduce(flat, op=op, group=self.group, async_op=True)))

    def zero_grad(self):
        """Zero the buckets in place (keeps param.grad views valid; replaces optimizer.zero_grad())."""
        for flat, _ in self.buckets:
            flat.zero_()
        self._reset()

    def finish(self):
        """Call after loss.backward(): launches buckets whose parameters received no gradient this step and
        waits for every exchange (the buckets then hold the means)."""
        for bi, n in self._pending.items():
            if n > 0:                      # unused parameters (e.g. ResNet fc.*): still exchange, grads are zero
                self._pending[bi] = 0
                self._launch(bi)
        self._wait_all()

    def _wait_all(self):
        """Wait for every launched exchange.  On RCCL `wait()` makes the CURRENT STREAM wait (the host returns at once), so with
        `timing` the exposed part is measured on the device: one event behind the last kernel of the backward pass, one behind
        the waits."""
        gpu = self.timing and self.collective and bool(self.buckets) and self.buckets[0][0].is_cuda
        if gpu:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        elif self.timing:
            import time
            t0 = time.perf_counter()
        for bi, w in self._works:
            w.wait()
        if gpu:
            e1.record()
            self._exposed.append((e0, e1))
        elif self.timing:
            self._exposed.append(time.perf_counter() - t0)
        self._works = []

    def exchange_stats(self):
        """{'exposed_comm_ms': mean per step of the time the step waited for collectives after its last gradient,
        'bucket_launch_fraction': per bucket, the mean fraction of the step's gradient hooks that had fired when its exchange was
        launched (1.0 = launched by the very last gradient: nothing left to overlap with), 'bucket_mbytes': sizes}.
        Synchronises the device.  Collected only while `timing` is True."""
        ms = []
        for e in self._exposed:
            if isinstance(e, tuple):
                e[1].synchronize()
                ms.append(e[0].elapsed_time(e[1]))
            else:
                ms.append(e * 1e3)
        if self.launch_log:                       # the step in flight
            total = float(max(1, len(self.params)))
            steps = self.fraction_log + [[(bi, fired / total) for bi, fired in self.launch_log]]
        else:
            steps = self.fraction_log
        frac = {}
        for st in steps:
            for bi, f in st:
                frac.setdefault(bi, []).append(f)
        return {"exposed_comm_ms": round(sum(ms) / len(ms), 4) if ms else None, "steps": len(ms),
                "bucket_launch_fraction": [round(sum(frac[bi]) / len(frac[bi]), 4) if bi in frac else None for bi in range(len(self.buckets))],
                "bucket_mbytes": [round(f.numel() * f.element_size() / 2 ** 20, 2) for f, _ in self.buckets]}

    def reduce_all(self):
        """Exchange every bucket NOW, on the current stream, and turn the sums into means.  For steps whose backward ran
        with the hooks deferred (``no_sync()``) -- in particular a forward+backward replayed from a captured hipGraph, where
        no Python hook runs at all: graph(fwd+bwd) -> reduce_all() -> graph(optimizer).  The exchange is then not
        overlapped with the backward pass (3 messages, 188 MB for DenseNet161-BTS), which is the price of replaying ~2000
        kernel launches from one graph instead of issuing them from Python."""
        self._reset()
        for bi in range(len(self.buckets)):
            self._pending[bi] = 0
            self._launch(bi)
        self._wait_all()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_parameters(module, src=0, process_group=None):
    """Rank-0 -> all broadcast of parameters and buffers at start-up (what DDP's constructor does)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    # one broadcast per (dtype, device) over a flat copy instead of ~2 000 small ones (DenseNet161-BTS: 1 200 parameters + 1 450 buffers)
    by_key, seen = {}, set()
    for t in list(module.parameters()) + list(module.buffers()):
        if id(t) in seen:
            continue
        seen.add(id(t))
        by_key.setdefault((t.dtype, t.device), []).append(t.data)
    with torch.no_grad():
        for ts in by_key.values():
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, src=src, group=process_group)
            off = 0
            for t in ts:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n


class BufferSync:
    """DDP's ``broadcast_buffers=True`` (the default the reference trains with, bts_main.py:352): at the START of every forward
    pass rank 0's module buffers -- here the BatchNorm running_mean / running_var / num_batches_tracked of encoder and decoder --
    overwrite every other rank's.  BatchNorm batch statistics stay rank-local (no SyncBN), but every rank evaluates
    (`online_eval`, bts_main.py:250-304, runs `model.eval()` on all ranks) and checkpoints with RANK 0's running statistics.
    `GradAllReducer` alone would leave them rank-local, so this is its companion:

        sync = BufferSync(model)        # after model.to(device)
        for batch in loader:
            sync()                      # before the forward pass, train or eval (what DDP.forward does)
            ...

    Layout, as for the gradients: the buffers of one dtype become VIEWS of one flat tensor (DenseNet161-BTS has ~1 450 of them,
    0.5 M elements), so a sync is one broadcast per dtype and nothing else -- no gather before it, no 1 450 copies after it.
    Names, shapes and state-dict entries are unchanged (a view is a tensor; `load_state_dict` copies into it in place); BatchNorm
    kernels update their running statistics in place through the views.  `flatten=False` keeps the buffers where they are and
    packs / unpacks around the broadcast instead."""

    def __init__(self, module, src=0, process_group=None, flatten=True):
        self.src, self.group = src, process_group
        self.active = dist.is_initialized() and dist.get_world_size(process_group) > 1
        self.flat = []          # flatten: [flat tensor]; else: [[buffers]]
        self._views = []        # (module, buffer name, data_ptr of the view) -- checked before every sync
        self._module = module
        self.flattened = bool(flatten)
        by_key = {}
        for m in module.modules():
            for name, b in m._buffers.items():
                if b is not None:
                    by_key.setdefault((b.dtype, b.device), []).append((m, name, b))
        if not self.flattened:
            return
        seen = {}
        with torch.no_grad():
            for (dtype, device), entries in by_key.items():
                uniq = []
                for m, name, b in entries:              # a buffer registered under two modules is one tensor
                    if id(b) not in seen:
                        seen[id(b)] = None
                        uniq.append(b)
                flat = torch.cat([b.reshape(-1) for b in uniq]) if uniq else torch.empty(0, dtype=dtype, device=device)
                off = 0
                for b in uniq:
                    seen[id(b)] = flat[off:off + b.numel()].view(b.shape)
                    off += b.numel()
                for m, name, b in entries:
                    m._buffers[name] = seen[id(b)]
                    self._views.append((m, name, seen[id(b)].data_ptr()))
                self.flat.append(flat)

    def _views_intact(self):
        """The modules must still point at the views: `module.to(dtype)` / `.cuda()` / `load_state_dict(assign=True)` /
        `register_buffer` after construction silently replace them, and a broadcast of the flat tensor would then sync nothing."""
        return all(m._buffers.get(name) is not None and m._buffers[name].data_ptr() == ptr for m, name, ptr in self._views)

    def _live_sets(self):
        """The module's buffers AS THEY ARE NOW, one list per (dtype, device), a shared buffer once: the non-flattened path looks
        them up on every call, so a later `.to()` / `load_state_dict(assign=True)` can never leave the sync pointing at dead tensors."""
        by_key, seen = {}, set()
        for m in self._module.modules():
            for b in m._buffers.values():
                if b is not None and id(b) not in seen:
                    seen.add(id(b))
                    by_key.setdefault((b.dtype, b.device), []).append(b)
        return list(by_key.values())

    def __call__(self):
        if not self.active:
            return
        if self.flattened and not self._views_intact():
            # something re-bound the buffers after construction: fall back to gather / broadcast / scatter over the live ones
            self.flattened = False
        with torch.no_grad():
            if self.flattened:
                for flat in self.flat:
                    if flat.numel():
                        dist.broadcast(flat, src=self.src, group=self.group)
                return
            for bufs in self._live_sets():
                flat = torch.cat([b.reshape(-1) for b in bufs])
                dist.broadcast(flat, src=self.src, group=self.group)
                off = 0
                for b in bufs:
                    n = b.numel()
                    b.copy_(flat[off:off + n].view_as(b))
                    off += n
