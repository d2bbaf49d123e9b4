"""Host-side logic of the HIP path that can be checked without a GPU: weight-fragment packing indices, the block
ranges of the batched pack / unpack launches, the decoder plan, and the profiler's kernel labels."""
import os
import sys

import pytest
import torch

from oracle import bts_oracle as O


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("dims", [[64, 32, 16, 8, 3], [32, 16, 8, 1], [128, 128, 64, 32, 16, 8, 3], [16, 8, 3]])
def test_chain_packer_matches_per_layer_packing(dims, dt):
    """ChainPacker (two gathers through cached indices, used every training step) == layer-by-layer fragment packing,
    for the forward fragments and for the W^T fragments of the recompute backward."""
    from bts_amd import chain
    gen = torch.Generator().manual_seed(len(dims))
    ws = [torch.randn(dims[i + 1], dims[i], 1, 1, generator=gen) for i in range(len(dims) - 1)]
    pk = chain.ChainPacker([(w.shape[0], w.shape[1]) for w in ws], dt, "cpu")
    f, t = pk.pack(ws, True)
    assert torch.equal(f, chain.pack_chain(ws, dt))
    assert torch.equal(t, chain.pack_chain_t(ws, dt))
    assert f.numel() % 1024 == 0 and t.numel() % 1024 == 0          # whole 64-lane x 16-byte fragments
    f2, t2 = pk.pack(ws, False)
    assert t2 is None and torch.equal(f, f2)


def test_chain_fragment_orders():
    """Layer 0 uses the natural K order (B operand from memory), later layers the MFMA accumulator order
    k = 16 s + {0,1,2,3,8,9,10,11}[e] + 4 g (include/bts_amd.h, csrc/lpg_chain.hip header)."""
    from bts_amd import chain
    w = torch.arange(32 * 32, dtype=torch.float32).reshape(32, 32)         # w[row][k] = 32 row + k
    first = chain._pack_layer(w, torch.float32, True).view(torch.float32).reshape(1, 4, 64, 4)
    later = chain._pack_layer(w, torch.bfloat16, False).view(torch.bfloat16).float().reshape(1, 2, 64, 8)
    lane = 37                                                               # row 5, g = 1
    assert first[0, 2, lane].tolist() == [32 * 5 + 8 * 2 + 4 + j for j in range(4)]
    assert later[0, 1, lane].tolist() == [32 * 5 + 16 + 4 + p for p in (0, 1, 2, 3, 8, 9, 10, 11)]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_packset_block_ranges(dt):
    """Jobs of the batched pack / unpack launches own contiguous, non-overlapping block ranges whose sizes follow the
    rule in include/bts_amd.h (32x32 tiles; 256 (co, ci) pairs)."""
    from bts_amd import _lib
    from bts_amd.decoder import DecoderPlan, PackSet
    feat, nf = [96, 96, 192, 384, 2208], 512
    plan = DecoderPlan(feat, nf)
    P = {k: torch.empty(s) for k, s in ((n, sh) for n, sh in _param_shapes(feat, nf))}
    ps = PackSet(plan, P, dt)

    def jobs(buf, cls, n):
        raw = bytes(buf.numpy().tobytes())
        sz = len(raw) // n
        assert sz == __import__("ctypes").sizeof(cls)
        return [cls.from_buffer_copy(raw[i * sz:(i + 1) * sz]) for i in range(n)]

    cdiv = lambda a, b: (a + b - 1) // b
    for buf, n, total in ((ps.fjobs, ps.nf, ps.fblocks), (ps.djobs, ps.nd, ps.dblocks)):
        js = jobs(buf, _lib.PackJob, n)
        nxt = 0
        for j in js:
            assert j.first_block == nxt
            nco, ne = (j.R, j.K) if j.mode == 0 else (j.K, j.R)
            nxt += cdiv(nco, 32) * cdiv(ne, 32)
        assert nxt == total
    js = jobs(ps.ujobs, _lib.UnpackJob, ps.nu)
    nxt = 0
    for j in js:
        assert j.first_block == nxt
        nxt += cdiv(j.Cout * j.Cin, 256)
    assert nxt == ps.ublocks
    assert ps.gw_total == sum(v.numel() for k, v in P.items() if k.endswith(".weight") and v.dim() == 4)


def _param_shapes(feat, nf):
    gen = torch.Generator().manual_seed(0)
    return [(k, tuple(v.shape)) for k, v in O.make_decoder_params(feat, nf, gen).items()]


def test_decoder_plan_matches_reference_layer_list():
    """40 convolutions, the reduction chains of bts.py:83-108 with the channel halving down to 8."""
    from bts_amd.decoder import DecoderPlan
    plan = DecoderPlan([96, 96, 192, 384, 2208], 512)
    assert len(plan.layers) == 40
    assert [plan.layers[k].cin for k in plan.reduc["reduc8x8"]] == [128, 128, 64, 32, 16, 8]
    assert [plan.layers[k].cout for k in plan.reduc["reduc8x8"]] == [128, 64, 32, 16, 8, 3]
    assert [plan.layers[k].cout for k in plan.reduc["reduc1x1"]] == [16, 8, 1]
    assert plan.layers["upconv5.conv"].nphase == 4 and plan.layers["upconv5.conv"].T == 4      # sub-pixel phases
    assert plan.layers["daspp_24.atrous_conv.aconv_sequence.4"].dil == 24


def test_profiler_labels_follow_dispatch():
    """Labels used for `roofline` must name the kernel family launch_fwd()/launch_wgrad() pick."""
    from bts_amd import conv
    bf, f32 = torch.bfloat16, torch.float32
    assert conv._fwd_kernel(bf, 512, True) == "conv_igemm_dma<bf16,128x128>"
    # wide radius-1 layers: 2-D halo tiles with per-tap weight streaming where the tiles fill the chip (conv4 / conv3 / daspp_conv
    # at the train shape), implicit GEMM where they do not (conv5: 22x76 map, 288 workgroups)
    assert conv._fwd_kernel(bf, 256, True, (8, 44, 152), 56) == "conv_halo_wide<bf16,128x256>"
    assert conv._fwd_kernel(bf, 128, True, (8, 88, 304), 29) == "conv_halo_wide<bf16,128x256>"
    assert conv._fwd_kernel(bf, 512, True, (8, 22, 76), 112) == "conv_igemm_dma<bf16,128x128>"
    assert conv._fwd_kernel(bf, 96, True, (8, 176, 608), 8) == "conv_halo_wide<bf16,96x256>"     # conv2's data gradient towards the 96-channel skip
    assert conv._fwd_kernel(bf, 192, True, (8, 44, 152), 32) == "conv_halo_wide<bf16,96x256>"    # conv4's, towards the 192-channel skip
    assert conv._fwd_kernel(torch.float32, 256, True, (8, 44, 152), 112) == "conv_igemm_dma<f32,128x128>"
    assert conv._fwd_kernel(bf, 32, True) == "conv_halo<bf16>"
    assert conv._fwd_kernel(bf, 64, True, (8, 176, 608), 21) == "conv_halo_wide<bf16,64x256>"      # conv2: 161 -> 64, three chunks
    assert conv._fwd_kernel(bf, 64, True, (8, 176, 608), 8) == "conv_halo<bf16>"                   # one chunk: persistent conv_halo
    assert conv._fwd_kernel(torch.float32, 64, True, (8, 176, 608), 41) == "conv_halo<f32>"
    assert conv._fwd_kernel(bf, 32, False) == "conv_igemm_dma<bf16,32x256>"
    assert conv._wgrad_kernel(bf, 1, True, False, 8, 352, 1216) == "conv_wgrad_c1<bf16>"
    assert conv._wgrad_kernel(bf, 32, True, False, 8, 352, 1216) == "conv_wgrad_halo_tr<bf16>"
    assert conv._wgrad_kernel(bf, 32, True, True, 8, 176, 608) == "conv_wgrad_halo_tr_up<bf16>"      # upconv1 (r6: transposing reads)
    assert conv._wgrad_kernel(bf, 16, True, True, 8, 176, 608) == "conv_wgrad_halo_up<bf16>"         # other widths keep the scatter kernel
    assert conv._wgrad_kernel(bf, 64, True, False, 8, 176, 608) == "conv_wgrad_halo_tr<bf16>"          # conv2: LDS-halo tile + transposing reads
    assert conv._wgrad_kernel(bf, 64, True, True, 8, 88, 304) == "conv_wgrad_halo_tr_up<bf16>"         # upconv2 (r6; was the 64-co ring form)
    assert conv._wgrad_kernel(bf, 64, True, True, 1, 32, 64) == "conv_wgrad_ring<bf16,64x256>"         # small map: ring form
    assert conv._wgrad_kernel(bf, 128, True, False, 8, 88, 304, 9 * 232) == "conv_wgrad_halo_tr<bf16>"   # conv3: two 64-channel output tiles
    assert conv._wgrad_kernel(bf, 256, True, False, 8, 44, 152, 9 * 448) == "conv_wgrad_ring<bf16,128x256>"   # conv4: 240 tiles, stays
    assert conv._wgrad_kernel(f32, 32, True, False, 8, 352, 1216) == "conv_wgrad<f32,32x128k4>"
    assert conv._wgrad_kernel(bf, 32, True, False, 1, 32, 64) == "conv_wgrad<bf16,32x128k4>"      # < 256 tiles
    # wide bf16: LDS-DMA + transposing reads; 128 x 256 ring unless its tiles alone exceed two rounds of the chip (upconv5)
    assert conv._wgrad_kernel(bf, 512, True, False, 8, 22, 76, 9 * 896) == "conv_wgrad_ring<bf16,128x256>"
    assert conv._wgrad_kernel(bf, 512, True, True, 8, 11, 38, 4 * 2208) == "conv_wgrad_tr<bf16,128x128>"
    assert conv._wgrad_kernel(f32, 512, True, False, 8, 22, 76) == "conv_wgrad<f32,128x128>"


def test_bench_synthetic_inputs_match_the_test_recipe():
    """bench.py builds its batch with bts_amd.synth (no import from oracle/ on the product side); the recipe the tests use
    (oracle/bts_oracle.py) must be the same data, bit for bit."""
    from bts_amd import synth
    from oracle import bts_oracle as O
    for ds in ("kitti", "nyu"):
        assert torch.equal(synth.synth_focal(7, ds), O.synth_focal(7, ds))
        a = synth.synth_depth_gt(2, 32, 64, ds, torch.Generator().manual_seed(5))
        b = O.synth_depth_gt(2, 32, 64, ds, torch.Generator().manual_seed(5))
        assert torch.equal(a, b)


def test_dual_and_group_predicates_follow_the_launchers():
    """bts_amd/conv.py mirrors of the two round-4 launch forms: which layers of DenseNet161-BTS at the bench shape take them."""
    from bts_amd.decoder import DecoderPlan
    bf, f32 = torch.bfloat16, torch.float32
    plan = DecoderPlan([96, 96, 192, 384, 2208], 512)
    # conv1: 32 + 4 -> 32 channels: both data gradients from one pass over dz (bf16 only; f32 keeps two launches)
    c1 = plan.layers["conv1.0"]
    assert c1.dual_dgrad_ok(bf, 0, 1) and not c1.dual_dgrad_ok(f32, 0, 1)
    assert not plan.layers["conv2.0"].dual_dgrad_ok(bf, 0, 2)            # dz has 64 channels: the weights do not fit the registers
    small = DecoderPlan([64, 64, 128, 256, 1024], 128)                    # bts_size 128: conv1 is 8 + 4 -> 8, no full 32-channel block
    assert not small.layers["conv1.0"].dual_dgrad_ok(bf, 0, 1)
    # grouped weight gradients: the ten dense-ASPP layers and reduc8x8's 128 -> 128 layer (<= 16 tiles of the ring kernel), not the
    # big ring layers, not the up-convolutions, not the LDS-halo-tile layers, nothing in f32
    N, H, W = 8, 44, 152
    want = ["daspp_%d.atrous_conv.aconv_sequence.%d" % (d, k) for d in (3, 6, 12, 18, 24) for k in (1, 4)] + ["reduc8x8.reduc.inter_128_128.0"]
    assert all(plan.layers[n].wgrad_groupable(bf, N, H, W) for n in want)
    for n, shape in (("conv4.0", (8, 44, 152)), ("daspp_conv.0", (8, 44, 152)), ("conv5.0", (8, 22, 76)), ("conv3.0", (8, 88, 304)),
                     ("upconv4.conv", (8, 22, 76)), ("upconv3.conv", (8, 44, 152)), ("conv2.0", (8, 176, 608)), ("get_depth.0", (8, 352, 1216))):
        assert not plan.layers[n].wgrad_groupable(bf, *shape), n
    assert not any(L.wgrad_groupable(f32, N, H, W) for L in plan.layers.values())


def test_deferred_weight_gradients_group_in_arrival_order(monkeypatch):
    """DecoderRun._wgrad / _flush_wgrads: groupable layers leave five at a time in arrival order, a single leftover and every
    non-groupable layer go through the per-layer call, and BTS_ERR_UNSUPPORTED from the grouped entry point falls back to it."""
    from bts_amd import conv as conv_mod
    from bts_amd import decoder as dec
    from bts_amd._lib import ERR_UNSUPPORTED, BtsAmdError
    log = []

    class FakeLayer:
        def __init__(self, name, groupable):
            self.name, self.groupable = name, groupable

        def wgrad_groupable(self, dtype, n, h, w):
            return self.groupable

        def wgrad_packed(self, x, dz, dwp):
            log.append(("single", self.name))

    class FakeT:
        dtype = torch.bfloat16
        shape = (8, 44, 152, 128)

    fail = {"on": False}

    def fake_group(items):
        if fail["on"]:
            raise BtsAmdError("bts_conv_wgrad_group failed: BTS_ERR_UNSUPPORTED (-3)", ERR_UNSUPPORTED)
        log.append(("group", tuple(L.name for L, _, _, _ in items)))
    monkeypatch.setattr(conv_mod.ConvLayer, "wgrad_group", staticmethod(fake_group))
    run = object.__new__(dec.DecoderRun)
    run.wgrad_pending = []
    names = ["g%d" % i for i in range(7)]
    for i, n in enumerate(names):
        run._wgrad(FakeLayer(n, True), [FakeT()], FakeT(), None)
        if i == 2:
            run._wgrad(FakeLayer("big", False), [FakeT()], FakeT(), None)       # not groupable: launched at once, order kept
    run._flush_wgrads()
    assert log == [("single", "big"), ("group", tuple(names[:5])), ("group", tuple(names[5:]))], log
    log.clear()
    run._wgrad(FakeLayer("lonely", True), [FakeT()], FakeT(), None)
    run._flush_wgrads()
    assert log == [("single", "lonely")] and run.wgrad_pending == []
    log.clear()
    fail["on"] = True
    for n in names[:5]:
        run._wgrad(FakeLayer(n, True), [FakeT()], FakeT(), None)
    assert log == [("single", n) for n in names[:5]], log                      # the group of five fell back, layer by layer


def test_shared_batchnorm_backward_groups_tensors_by_their_batchnorms(monkeypatch):
    """DecoderRun._flush_shared_bn: input tensors that were normalised by the SAME BatchNorms leave in one bts_bn_bwd_multi launch
    (three at most), a tensor with another set of BatchNorms in its own; a tensor whose gradient buffer exists accumulates, one
    without gets a fresh buffer; entries without contributions launch nothing."""
    from types import SimpleNamespace as NS

    from bts_amd import decoder as dec
    from bts_amd import ops
    calls = []
    monkeypatch.setattr(ops, "bn_bwd_multi", lambda tensors, eps, relu, ubs, tag=None: calls.append((tag, eps, relu, ubs, tensors)))
    run = object.__new__(dec.DecoderRun)
    run.dtype = torch.float32

    def act(c, with_grad):
        return NS(t=torch.zeros(1, 2, 2, c), g=torch.zeros(1, 2, 2, c) if with_grad else None)

    def item(c, name, eps=1e-5, relu=True):
        return (torch.zeros(1, 2, 2, c), torch.zeros(c), torch.zeros(c), torch.zeros(c), torch.zeros(c), eps, relu, (torch.zeros(c), torch.ones(c)), name)
    four = ["daspp_24.first_bn", "daspp_18.first_bn", "daspp_12.first_bn", "daspp_6.first_bn"]
    ents = []
    for c, has_g in ((128, False), (192, False), (64, True), (32, False)):        # four tensors that saw all four BatchNorms
        a = act(c, has_g)
        ents.append({"act": a, "left": 0, "items": [item(c, n) for n in four]})
    lone = {"act": act(64, True), "left": 0, "items": [item(64, n) for n in four[:3]]}        # another BatchNorm set
    empty = {"act": act(8, False), "left": 0, "items": []}
    run._flush_shared_bn(ents + [lone, empty])
    assert [len(c[4]) for c in calls] == [3, 1, 1], [(c[0], len(c[4])) for c in calls]
    assert calls[0][0] == calls[1][0] == "daspp_24+daspp_18+daspp_12+daspp_6" and calls[2][0] == "daspp_24+daspp_18+daspp_12"
    accs = [t[4] for c in calls[:2] for t in c[4]]
    assert accs == [False, False, True, False]
    assert all(e["act"].g is not None and e["items"] == [] for e in ents + [lone]) and empty["act"].g is None
    assert all(len(t[5]) == 4 for c in calls[:2] for t in c[4]) and len(calls[2][4][0][5]) == 3
    assert all(c[1:4] == (1e-5, True, True) for c in calls)


def test_bench_gpus_n_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher around it must start two ranks itself (the reference spawns its own:
    bts_main.py:600-602) and report the world it ran in; under a launcher whose world differs from --gpus it must refuse."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--plumbing-only", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(out.stdout.splitlines()) == 1, out.stdout          # ONE line on stdout, whatever the libraries underneath print
    j = json.loads(out.stdout)
    assert j["n_gpus"] == 2 and j["sum_ranks"] == 3.0 and abs(j["max_elapsed"] - 0.002) < 1e-12
    # a launcher world that disagrees with --gpus: refuse rather than print a line with the wrong n_gpus
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--plumbing-only", "1"],
                         capture_output=True, text=True, timeout=120, env=env2)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)


def test_bench_result_line_is_alone_on_stdout():
    """bench.py's contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio (it lands after the line when
    stdout is a pipe); own_stdout() keeps a private copy of stdout for the line and points fd 1 at stderr for everything else."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r); import bench; bench.own_stdout(); "
            "os.write(1, b'RCCL version : x\\n'); print('python-level noise'); bench.emit({'a': 1}); os.write(1, b'late noise\\n')" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout == json.dumps({"a": 1}) + "\n"
    assert "RCCL version" in out.stderr and "python-level noise" in out.stderr and "late noise" in out.stderr


def test_build_manifest_matches_sources():
    """build() reuses binaries by content, not mtime: after a build the manifest must describe the current sources and library."""
    from bts_amd import build as b
    if not os.path.exists(b.LIB):
        pytest.skip("library not built")
    assert b.verify_library()
