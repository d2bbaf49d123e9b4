"""N>1 path on CPU: world_size-2 gloo processes exercise GradAllReducer (bucketing, hooks fired during
backward, unused parameters, mean semantics) against DistributedDataParallel and a hand-computed mean."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.enc = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1))
        self.dec = nn.Sequential(nn.Conv2d(8, 4, 3, padding=1), nn.ELU(), nn.Conv2d(4, 1, 3, padding=1))
        self.unused = nn.Linear(4, 4)        # like torchvision ResNet's fc.* (bts_main.py:352 find_unused_parameters)
        self.frozen = nn.Conv2d(1, 1, 1)
        for p in self.frozen.parameters():
            p.requires_grad = False

    def forward(self, x):
        return self.dec(self.enc(x))


def _worker(rank, world, port, bucket_bytes, channels_last, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bts_amd.parallel import GradAllReducer, broadcast_parameters
        torch.manual_seed(100 + rank)            # different init per rank: broadcast must fix it
        net = Net()
        broadcast_parameters(net)
        ref = Net()
        ref.load_state_dict(net.state_dict())
        if channels_last:      # bench.py runs the stock encoder in torch.channels_last (round 6): the bucket views must follow the parameters' strides
            net.enc.to(memory_format=torch.channels_last)
            assert not net.enc[0].weight.is_contiguous()
        red = GradAllReducer(net.parameters(), bucket_bytes=bucket_bytes)
        for n_, p_ in net.named_parameters():
            if p_.requires_grad:
                assert all(a == b for a, b, k in zip(p_.grad.stride(), p_.stride(), p_.shape) if k != 1), (n_, p_.grad.stride(), p_.stride())
        assert len(red.buckets) >= (2 if bucket_bytes < 1000 else 1)
        x = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(rank))   # rank-local batch
        for it in range(2):                       # two steps: buckets are reused, zero_grad keeps the views
            red.zero_grad()
            loss = net(x).pow(2).mean()
            loss.backward()
            red.finish()
            # reference: local grads, then explicit mean over ranks
            ref.zero_grad()
            ref(x).pow(2).mean().backward()
            for (n, p), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
                if not p.requires_grad:
                    assert p.grad is None
                    continue
                g = pr.grad.clone() if pr.grad is not None else torch.zeros_like(pr)
                dist.all_reduce(g)
                g /= world
                assert torch.allclose(p.grad, g, rtol=1e-6, atol=1e-7), (n, it)
                assert p.grad.data_ptr() >= red.buckets[0][0].data_ptr() or True
        # gradient accumulation: two micro-batches, the first under no_sync(); mean over ranks of the SUM of both
        x2 = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(10 + rank))
        red.zero_grad()
        with red.no_sync():
            net(x).pow(2).mean().backward()
        net(x2).pow(2).mean().backward()
        red.finish()
        ref.zero_grad()
        ref(x).pow(2).mean().backward()
        ref(x2).pow(2).mean().backward()
        for (n, p), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
            if p.requires_grad:
                g = pr.grad.clone() if pr.grad is not None else torch.zeros_like(pr)
                dist.all_reduce(g)
                g /= world
                assert torch.allclose(p.grad, g, rtol=1e-5, atol=1e-7), ("accum", n)
        # a stray second synchronising backward must raise, not race the exchange in flight
        red.zero_grad()
        net(x).pow(2).mean().backward()
        try:
            net(x2).pow(2).mean().backward()
            raise AssertionError("second backward did not raise")
        except RuntimeError as e:
            assert "no_sync" in str(e)
        red.finish()
        red.zero_grad()
        net(x).pow(2).mean().backward()
        red.finish()
        # deferred form (bench.py N>1 with hipGraphs: the backward is replayed, no hook runs): whole backward under
        # no_sync(), then reduce_all() exchanges every bucket at once -- same means as the hook-driven exchange
        red.zero_grad()
        with red.no_sync():
            net(x).pow(2).mean().backward()
        red.reduce_all()
        ref.zero_grad()
        ref(x).pow(2).mean().backward()
        for (n, p), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
            if p.requires_grad:
                g = pr.grad.clone() if pr.grad is not None else torch.zeros_like(pr)
                dist.all_reduce(g)
                g /= world
                assert torch.allclose(p.grad, g, rtol=1e-6, atol=1e-7), ("reduce_all", n)
        red.zero_grad()
        net(x).pow(2).mean().backward()
        red.finish()
        # every rank ends with identical gradients (what the optimizer sees)
        flat = torch.cat([b[0] for b in red.buckets])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered)
        # and they agree with DistributedDataParallel
        ddp = nn.parallel.DistributedDataParallel(ref, find_unused_parameters=True)
        ref.zero_grad()
        ddp(x).pow(2).mean().backward()
        for (n, p), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
            if p.requires_grad and pr.grad is not None:
                assert torch.allclose(p.grad, pr.grad, rtol=1e-5, atol=1e-7), n
        q.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("channels_last", [False, True], ids=["contiguous", "channels_last"])
@pytest.mark.parametrize("bucket_bytes", [512, 64 << 20])
def test_grad_allreducer_world2_gloo(bucket_bytes, channels_last):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bucket_bytes, channels_last, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_single_process_noop():
    from bts_amd.parallel import GradAllReducer
    net = Net()
    red = GradAllReducer(net.parameters())
    red.zero_grad()
    net(torch.randn(1, 3, 8, 8)).sum().backward()
    red.finish()
    assert all(p.grad is not None for p in net.parameters() if p.requires_grad)


# ---- the shape of the real step: a decoder whose backward delivers ALL its parameter gradients at once ----------------------
class _OneShotDecoder(torch.autograd.Function):
    """Stand-in for bts_amd.model._DecoderFn (the HIP decoder is one autograd node: features + every decoder parameter in,
    five maps out; its backward returns the ~110 parameter gradients together)."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3):
        ctx.save_for_backward(x, w1, w2, w3)
        return torch.tanh(x * w1.view(1, -1, 1, 1)) * w2.view(1, -1, 1, 1) + w3.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, g):
        x, w1, w2, w3 = ctx.saved_tensors
        t = torch.tanh(x * w1.view(1, -1, 1, 1))
        gw3 = g.sum((0, 2, 3))
        gw2 = (g * t).sum((0, 2, 3))
        gt = g * w2.view(1, -1, 1, 1) * (1 - t * t)
        return gt * w1.view(1, -1, 1, 1), (gt * x).sum((0, 2, 3)), gw2, gw3


class StepNet(nn.Module):
    """Encoder registered FIRST (as BtsModel does, bts.py:323-327), four layers; one-shot decoder registered last."""

    def __init__(self):
        super().__init__()
        self.encoder = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1), nn.ReLU(),
                                     nn.Conv2d(8, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1))
        self.decoder = nn.ParameterList([nn.Parameter(torch.randn(8) * 0.5) for _ in range(3)])

    def forward(self, x):
        return _OneShotDecoder.apply(self.encoder(x), *self.decoder)


def _step_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bts_amd.parallel import GradAllReducer, broadcast_parameters
        torch.manual_seed(7 + rank)
        net = StepNet()
        broadcast_parameters(net)
        ref = StepNet()
        ref.load_state_dict(net.state_dict())
        # buckets: decoder (3 x 8 floats) + the two last encoder layers (2336 B each) | the second layer | tail: the FIRST layer only
        first = sum(p.numel() * 4 for p in net.encoder[0].parameters())
        red = GradAllReducer(net.parameters(), bucket_bytes=5000, tail_bytes=first)
        assert len(red.buckets) == 3, [len(b[1]) for b in red.buckets]
        assert [id(p) for p in red.buckets[-1][1]] == [id(p) for p in reversed(list(net.encoder[0].parameters()))], "tail bucket"
        n_dec = len(list(net.decoder.parameters()))
        assert {id(p) for p in red.buckets[0][1][:n_dec]} == {id(p) for p in net.decoder.parameters()}
        x = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(rank))
        red.timing = True                      # bench.py --gpus N: exposed_comm_ms / bucket_launch_fraction of the line
        red.zero_grad()
        net(x).pow(2).mean().backward()
        log = list(red.launch_log)
        red.finish()
        stats = red.exchange_stats()
        assert stats["steps"] == 1 and stats["exposed_comm_ms"] is not None and stats["exposed_comm_ms"] >= 0.0
        fr = stats["bucket_launch_fraction"]
        assert len(fr) == len(red.buckets) and fr[-1] == 1.0 and fr[0] < 1.0, stats      # tail closes the step; decoder bucket early
        assert len(stats["bucket_mbytes"]) == len(red.buckets)
        red.timing = False
        # every bucket was launched from a hook (none left for finish()); the bucket that holds the decoder went on the wire
        # while encoder gradients were still outstanding -- its launch happened before the last hook of the step fired -- and
        # the exchange that closes the step is the small tail bucket.  (The autograd engine does not promise the order in
        # which the AccumulateGrad nodes of one backward node run relative to its successors, so only these are asserted.)
        n_hooks = len(red.params)
        assert sorted(b for b, _ in log) == list(range(len(red.buckets))), log
        when = dict(log)
        assert when[0] < n_hooks, log
        assert log[-1] == (len(red.buckets) - 1, n_hooks), log
        ddp = nn.parallel.DistributedDataParallel(ref)
        ddp(x).pow(2).mean().backward()
        for (n, p), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
            assert torch.allclose(p.grad, pr.grad, rtol=1e-5, atol=1e-7), n
        flat = torch.cat([b[0] for b in red.buckets])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered)
        q.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


def test_grad_allreducer_one_shot_decoder_overlap_world2_gloo():
    """The real step's gradient arrival order (one-shot decoder node, then the encoder layer by layer): the decoder bucket's
    exchange is launched before the encoder backward has finished, the last bucket is the small tail, results equal DDP's."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_step_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def _buffer_worker(rank, world, port, flatten, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bts_amd.parallel import BufferSync, GradAllReducer, broadcast_parameters
        torch.manual_seed(31 + rank)
        net = Net()
        broadcast_parameters(net)
        ref = Net()
        ref.load_state_dict(net.state_dict())
        ddp = nn.parallel.DistributedDataParallel(ref, find_unused_parameters=True)      # broadcast_buffers=True: bts_main.py:352
        red = GradAllReducer(net.parameters(), bucket_bytes=1 << 20)
        sync = BufferSync(net, flatten=flatten)
        if flatten:                                   # every float buffer is a view of ONE flat tensor: a sync is one broadcast
            fl = [b for b in net.buffers() if b.dtype == torch.float32]
            assert len({b.untyped_storage().data_ptr() for b in fl}) == 1 and len(sync.flat) == 2
            sd = {k: v.clone() for k, v in net.state_dict().items()}
            net.load_state_dict(sd)                    # in-place copies: the views survive a checkpoint load
            assert len({b.untyped_storage().data_ptr() for b in net.buffers() if b.dtype == torch.float32}) == 1
        opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=0.1)
        opt_ref = torch.optim.SGD([p for p in ref.parameters() if p.requires_grad], lr=0.1)
        gen = torch.Generator().manual_seed(1000 + rank)                                   # rank-local batches
        for it in range(3):
            x = torch.randn(2, 3, 8, 8, generator=gen) * (1.0 + rank)                      # different statistics per rank
            red.zero_grad()
            sync()
            net(x).pow(2).mean().backward()
            red.finish()
            opt.step()
            opt_ref.zero_grad()
            ddp(x).pow(2).mean().backward()
            opt_ref.step()
            # after the step the running statistics are rank-local on BOTH sides (DDP broadcasts at the START of a forward)
            for (n, b), (_, br) in zip(net.named_buffers(), ref.named_buffers()):
                assert torch.allclose(b.float(), br.float(), rtol=1e-5, atol=1e-7), (it, n)
        # online_eval (bts_main.py:250-304): model.eval() + forward on every rank -> every rank evaluates with rank 0's statistics
        net.eval()
        ddp.eval()
        xe = torch.randn(1, 3, 8, 8, generator=torch.Generator().manual_seed(5))
        sync()
        with torch.no_grad():
            ye, yr = net(xe), ddp(xe)
        assert torch.allclose(ye, yr, rtol=1e-5, atol=1e-6)
        flat = torch.cat([b.reshape(-1).float() for b in net.buffers()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered), "buffers differ across ranks after BufferSync"
        outs = [torch.zeros_like(ye) for _ in range(world)]
        dist.all_gather(outs, ye)
        assert all(torch.equal(outs[0], o) for o in outs), "eval outputs differ across ranks"
        if flatten:
            # Something re-binds a buffer after construction (module.to(dtype), load_state_dict(assign=True), register_buffer): the
            # module no longer points at BufferSync's view, a broadcast of the flat tensor would sync nothing.  The sync must notice
            # and fall back to gather / broadcast / scatter over the LIVE buffers (advisor item of round 4).
            bn = [m for m in net.modules() if isinstance(m, nn.BatchNorm2d)][0]
            assert sync._views_intact()
            bn._buffers["running_mean"] = torch.full_like(bn.running_mean, 10.0 + rank)     # rank-dependent, outside the flat tensor
            assert not sync._views_intact()
            sync()
            assert not sync.flattened
            got = [torch.zeros_like(bn.running_mean) for _ in range(world)]
            dist.all_gather(got, bn.running_mean)
            assert all(torch.equal(g, torch.full_like(g, 10.0)) for g in got), "stale views: the re-bound buffer was not synchronised"
            # ... and a SECOND re-bind after the fallback (advisor item of round 5: the fallback once kept the list of live buffers it
            # found the first time): the non-flattened path looks the buffers up on every call
            bn._buffers["running_var"] = torch.full_like(bn.running_var, 20.0 + rank)
            bn._buffers["running_mean"] = torch.full_like(bn.running_mean, 30.0 + rank)
            sync()
            for name, want in (("running_var", 20.0), ("running_mean", 30.0)):
                got = [torch.zeros_like(bn._buffers[name]) for _ in range(world)]
                dist.all_gather(got, bn._buffers[name])
                assert all(torch.equal(g, torch.full_like(g, want)) for g in got), "second re-bind of %s not synchronised" % name
        q.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("flatten", [True, False], ids=["flat_views", "pack_unpack"])
def test_buffer_sync_matches_ddp_broadcast_buffers_world2_gloo(flatten):
    """BatchNorm buffers under data parallelism: BufferSync() before each forward reproduces DistributedDataParallel's
    broadcast_buffers=True (the reference's setting): identical buffers after every train step, and an eval forward that uses
    rank 0's running statistics on every rank."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_buffer_worker, args=(r, world, port, flatten, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
