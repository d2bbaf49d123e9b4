#!/usr/bin/env python
"""Benchmark of the BTS train step (BASELINE.json metric: images/sec, DenseNet161-BTS 352x1216).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = forward (stock PyTorch-ROCm encoder under bf16 autocast + HIP decoder) + silog loss
+ backward + (N>1: RCCL gradient all-reduce overlapped with backward) + AdamW with the per-step
poly LR of bts_main.py:456-460, on a synthetic batch already resident in HBM.  Weak scaling:
8 images per GPU (BASELINE.json configs[2] = batch 64 over 8 GPUs, bts_main.py:351).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant HIP kernel family of the step (by summed device time, measured with HIP
                  events on the launching stream inside the timed region): algorithmic FLOPs / time
  cpu_baseline -- the CPU oracle (oracle/bts_oracle.py + the same stock encoder) timed on this
                  host's cores on ONE image of the same shape (reported-only)
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace as NS

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _flag(name, default):
    """Value of `--name V` / `--name=V` on the command line (read BEFORE argparse and before torch is imported: the MIOpen / ATen
    switches below are environment variables the libraries read once, at their first convolution)."""
    for i, a in enumerate(sys.argv):
        if a == name and i + 1 < len(sys.argv):
            return sys.argv[i + 1]
        if a.startswith(name + "="):
            return a.split("=", 1)[1]
    return default


def encoder_environment():
    """How the STOCK encoder is run (`profiles/r06_ab_encoder_layout.json`: same-box A/B).  The encoder stays
    PyTorch-ROCm / MIOpen; what is chosen here is its configuration:
      * torch.channels_last tensors AND PYTORCH_MIOPEN_SUGGEST_NHWC(+_BATCHNORM)=1, so that ATen hands MIOpen the NHWC tensors as
        they are (without the switch it makes NCHW copies, and MIOpen's NHWC kernels transpose around themselves:
        batched_transpose_* was 13 % of the step);
      * MIOpen's find results for exactly these layer shapes, recorded once on an MI355X by `tools/miopen_warm.sh` and shipped in
        bts_amd/miopen_db/ (MIOPEN_USER_DB_PATH): this image has no gfx950 find-db, so without it every convolution runs the
        solver an untuned heuristic picks.  MIOPEN_FIND_MODE=FAST: a shape that is in the db uses its recorded best solver, a
        shape that is not falls back to the heuristic -- never a minutes-long search inside a benchmark run."""
    nhwc = _flag("--miopen-nhwc", "1")
    if nhwc in ("0", "1"):
        os.environ["PYTORCH_MIOPEN_SUGGEST_NHWC"] = nhwc
        # BatchNorm separately (--miopen-nhwc-bn, default = the convolutions' setting): with 0 ATen does not hand channels-last BatchNorm
        # inputs to MIOpen at all and runs its own channels-last kernels
        os.environ["PYTORCH_MIOPEN_SUGGEST_NHWC_BATCHNORM"] = _flag("--miopen-nhwc-bn", nhwc)
    db = _flag("--miopen-db", os.path.join(ROOT, "bts_amd", "miopen_db"))
    mode = _flag("--miopen-find-mode", "fast")
    if db and db != "none":
        os.makedirs(db, exist_ok=True)
        use = db
        if mode != "normal":
            # Every process reads the recorded results from its OWN copy: MIOpen appends to the user db while it runs (86 of the encoder's
            # problems are re-recorded by every run), and the ranks of an N > 1 run would all append to the same files at the same moment.
            # (Find mode `normal` -- tools/miopen_warm.sh recording the results -- writes into the directory itself.)
            import atexit
            import shutil
            import tempfile
            use = tempfile.mkdtemp(prefix="bts_miopen_db_")
            for f in os.listdir(db):
                if f.endswith(".txt"):
                    shutil.copy2(os.path.join(db, f), os.path.join(use, f))
            atexit.register(shutil.rmtree, use, True)
        os.environ["MIOPEN_USER_DB_PATH"] = use
    if mode != "default":
        os.environ["MIOPEN_FIND_MODE"] = {"normal": "1", "fast": "2", "hybrid": "3", "dynamic_hybrid": "5"}[mode]


if __name__ == "__main__" or os.environ.get("BTS_BENCH_ENV") == "1":
    encoder_environment()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK = {"bf16": 2500.0, "f32": 157.3}   # dense MFMA TFLOP/s, MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="default: WORLD_SIZE under a launcher, else 1")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--height", type=int, default=352)
    ap.add_argument("--width", type=int, default=1216)
    ap.add_argument("--encoder", default="densenet161_bts")
    ap.add_argument("--dataset", default="kitti")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--channels-last", type=int, default=1, help="run the stock encoder in torch.channels_last (model + input); the decoder "
                    "then takes the features as NHWC views, no layout conversion either way (default since round 6: +5 %% alone, +17 %% with "
                    "the recorded find results, profiles/r06_ab_encoder_layout.json)")
    ap.add_argument("--miopen-nhwc", type=int, default=1, help="1 / 0: export PYTORCH_MIOPEN_SUGGEST_NHWC(+_BATCHNORM) = 1 / 0 before the first "
                    "convolution (without it PyTorch-ROCm hands MIOpen NCHW copies of channels-last tensors); -1: leave the environment alone")
    ap.add_argument("--miopen-nhwc-bn", type=int, default=None, help="PYTORCH_MIOPEN_SUGGEST_NHWC_BATCHNORM (default: as --miopen-nhwc); 0 = ATen's "
                    "own channels-last BatchNorm kernels instead of MIOpen's")
    ap.add_argument("--miopen-db", default=os.path.join(ROOT, "bts_amd", "miopen_db"), help="MIOPEN_USER_DB_PATH: directory of the recorded "
                    "MIOpen find results (tools/miopen_warm.sh writes it); 'none' = MIOpen's own default")
    ap.add_argument("--miopen-find-mode", default="fast", choices=["fast", "normal", "hybrid", "dynamic_hybrid", "default"],
                    help="MIOPEN_FIND_MODE: fast = recorded result or heuristic, never a search [default]; normal = full search, "
                         "records into --miopen-db (what tools/miopen_warm.sh uses)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--graph", type=int, default=1, help="capture the train step in a hipGraph (N=1 only; falls back to eager)")
    ap.add_argument("--cudnn-benchmark", type=int, default=1, help="torch.backends.cudnn.benchmark: ATen asks MIOpen's find API (which, in "
                    "find mode `fast`, answers from the recorded results) instead of its immediate-mode heuristic")
    ap.add_argument("--optimizer", default="bts", choices=["bts", "torch"], help="bts = fused HIP AdamW (bts_adamw_step)")
    ap.add_argument("--mode", default="train", choices=["train", "infer"],
                    help="infer = BASELINE.json configs[4]: no-grad forward, DenseNet161 704x1216 batch 32 (secondary config)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only for plumbing tests")
    ap.add_argument("--reducer", default="auto", choices=["auto", "ddp", "bts", "bts-graph"],
                    help="N>1 gradient exchange: ddp = torch DDP (eager step, exchange overlapped with backward); bts = "
                         "bts_amd.parallel.GradAllReducer from hooks (eager, overlapped); bts-graph = hipGraph(fwd+bwd) -> "
                         "GradAllReducer.reduce_all() -> hipGraph(AdamW); auto = bts (the eager step is GPU-bound: 61.1 ms eager vs 60.9 ms "
                         "replayed at N=1, and bts costs +1.2 ms at world 1 against +3.1 ms for ddp and +3.2 ms for bts-graph)")
    ap.add_argument("--parity", type=int, default=1, help="one-batch parity figures (decoder vs the device-side checker) in the JSON line; N=1 only")
    ap.add_argument("--lpg-op", type=int, default=1, help="time the bare LPG operator at the bench shape (BASELINE metric ii); N=1 only")
    ap.add_argument("--dump-launches", default="", help="write the per-launch table of the event-timed steps (family, layer tag, us, work) to this JSON file")
    ap.add_argument("--f32-line", type=int, default=1, help="also time the f32 configuration (the one that meets the 1e-4 parity bound) in a "
                    "child process of this run and print it as the `f32` object of the line; N=1, default workload, bf16 only")
    ap.add_argument("--eager-steps", type=int, default=5, help="N=1: un-profiled eager steps timed after the replayed region (`eager` object: "
                    "the launch mode an N>1 run uses, so the driver's 1 -> N curve can be read like for like)")
    ap.add_argument("--force-dist", type=int, default=0, help=argparse.SUPPRESS)   # world-1 process group: exercises the N>1 path on one GPU
    ap.add_argument("--plumbing-only", type=int, default=0, help=argparse.SUPPRESS)  # CPU test of the launch path: rendezvous + one all-reduce, no model
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when this script spawns its own ranks (0 = pick a free one)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, one process per GPU, the way the
    reference does (bts_main.py:600-602, `mp.spawn(main_worker, nprocs=ngpus_per_node)`) -- here by re-executing this script under
    `torch.distributed.run` (same command line), so the plain command the driver uses for N = 1 works verbatim for N = 8.  Rendezvous
    on 127.0.0.1 (the container hostname may not resolve).  Returns the launcher's exit code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus)]
    if args.master_port:
        cmd += ["--master-addr", "127.0.0.1", "--master-port", str(args.master_port)]
    else:
        # no probe-then-close of a "free" port (a race on busy hosts): the launcher binds port 0 itself and tells its workers
        cmd += ["--standalone", "--local-addr", "127.0.0.1"]
    cmd += [os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL / cross-process tensors need it on this host driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env, stdout=_RESULT_FD)      # the ranks' stdout is the REAL stdout (own_stdout() moved fd 1 to stderr)


def set_misc(model):
    """Freeze what bts_main.py:217-247 freezes by default ('Fixing first conv layer')."""
    fixing = ["base_model.conv1", ".bn"] if "resne" in model.encoder.params.encoder else ["conv0", "norm"]
    for name, p in model.encoder.named_parameters():
        if any(x in name for x in fixing):
            p.requires_grad = False


def make_batch(args, dev, seed):
    from bts_amd import synth              # synthetic-input recipe; the tests hold it equal to the oracle's
    gen = torch.Generator().manual_seed(seed)
    B, H, W = args.batch, args.height, args.width
    image = torch.randn(B, 3, H, W, generator=gen)
    focal = synth.synth_focal(B, args.dataset)
    gt = synth.synth_depth_gt(B, H, W, args.dataset, gen)
    return image.to(dev), focal.to(dev), gt.to(dev)


CPU_CROP = (352, 1216)         # fixed sample of the CPU leg: ONE FULL image of the bench shape (rounds 4-5: a 160x608 crop scaled by pixels)
CPU_THREAD_SWEEP = (8, 16, 32, 64, 128)


def cpu_baseline(args):
    """Oracle train step (stock encoder fwd + oracle decoder + silog + bwd) on the host cores, f32, FIXED bounded sample: ONE
    full image of the bench shape (352x1216; no pixel scaling since round 6).  The thread count is chosen by a sweep over
    8 / 16 / 32 / 64 / 128 threads (one warm-up + one timed iteration each, capped at the host's cpu_count; a count whose
    iteration already takes more than 3x the best so far ends the sweep: PyTorch's CPU convolutions stop scaling long before
    the 256 hardware threads of the GPU host and thrash beyond that), then 3 timed iterations at the best count, median
    reported.  The same sample on every host, so the figure is comparable between runs.  `kind` is "port": the oracle is a
    restatement; the UNMODIFIED reference timed beside it on the build host (same arithmetic, same speed to 2 %):
    profiles/r02_cpu_reference_vs_port.json."""
    from bts_amd.model import BtsModel
    from oracle import bts_oracle as O
    params = NS(encoder=args.encoder, max_depth=80.0 if args.dataset == "kitti" else 10.0, dataset=args.dataset, bts_size=512)
    torch.manual_seed(0)
    model = BtsModel(params)            # parameter container only: the oracle does the math
    enc = model.encoder
    P = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
         for k, v in model.decoder.state_dict().items()}
    gen = torch.Generator().manual_seed(1)
    H, W = args.height, args.width
    hh, ww = min(CPU_CROP[0], H), min(CPU_CROP[1], W)
    gt = O.synth_depth_gt(1, H, W, args.dataset, gen)
    focal = O.synth_focal(1, args.dataset)

    def step(h, w):
        x = torch.randn(1, 3, h, w, generator=gen)
        t0 = time.time()
        for v in P.values():
            if v.requires_grad:
                v.grad = None
        feats = enc(x)
        outs, _ = O.decoder_forward(P, feats, focal, params.max_depth, args.dataset, True)
        g = gt[:, :, :h, :w]
        loss = O.silog(outs[4], g, g > (1.0 if args.dataset == "kitti" else 0.1), 0.85)
        loss.backward()
        return time.time() - t0
    ncpu = os.cpu_count() or 1
    sweep = {}
    for th in [t for t in CPU_THREAD_SWEEP if t <= ncpu] or [ncpu]:
        torch.set_num_threads(th)
        step(hh, ww)                                # warm-up at this thread count (thread pool, allocator, primitive caches)
        sweep[th] = step(hh, ww)
        if sweep[th] > 3.0 * min(sweep.values()):   # thrashing: larger counts only get worse
            break
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    ts = sorted(step(hh, ww) for _ in range(3))
    dt = ts[1]                                      # median of 3
    frac = (hh * ww) / float(H * W)
    ips = frac / dt
    return {"value": round(ips, 4), "unit": "images/s", "cores": threads, "host_cpu_count": ncpu, "kind": "port",
            "thread_sweep_s": {str(k): round(v, 3) for k, v in sweep.items()},
            "sample": "oracle (stock encoder + oracle decoder + silog) fwd+bwd, f32, batch 1, fixed %dx%d crop = %.4g of one %dx%d "
                      "image, threads chosen by the sweep in thread_sweep_s (seconds per iteration), then 3 timed iterations, "
                      "median %.2f s (min %.2f, max %.2f), scaled by pixel count; the unmodified reference timed beside this port "
                      "on the build host: profiles/r02_cpu_reference_vs_port.json" % (hh, ww, frac, H, W, dt, ts[0], ts[2])}


def cpu_baseline_subprocess(args, timeout_s=240):
    """Run cpu_baseline() in a child process so a pathological host (thread oversubscription) cannot stall the bench."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--encoder", args.encoder, "--dataset", args.dataset,
           "--height", str(args.height), "--width", str(args.width)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "images/s", "kind": "port", "sample": "cpu baseline failed: %s" % out.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "images/s", "kind": "port", "cores": None,
                "sample": "cpu baseline exceeded %d s on this host and was cut" % timeout_s}


def f32_line_subprocess(args, parity, timeout_s=240):
    """The f32 configuration of the same workload -- the one whose outputs meet north_star's 1e-4 bound (`parity.f32_*`) -- timed by
    a child of this run (10 steps after 3 warm-up steps, same synthetic batch, hipGraph replay) so that it is part of the
    driver-visible line.  The child is this script with --dtype f32; its roofline object is that of ITS dominant kernel family."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--dtype", "f32", "--steps", "10", "--warmup", "3", "--no-cpu-baseline",
           "--parity", "0", "--lpg-op", "0", "--f32-line", "0", "--eager-steps", "0", "--encoder", args.encoder, "--dataset", args.dataset,
           "--height", str(args.height), "--width", str(args.width), "--batch", str(args.batch),
           "--channels-last", str(args.channels_last), "--miopen-nhwc", str(args.miopen_nhwc),
           "--miopen-nhwc-bn", os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC_BATCHNORM", str(args.miopen_nhwc)), "--miopen-db", args.miopen_db,
           "--miopen-find-mode", args.miopen_find_mode, "--cudnn-benchmark", str(args.cudnn_benchmark)]
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        for line in reversed(res.stdout.strip().splitlines()):
            if line.startswith("{"):
                j = json.loads(line)
                out = {k: j.get(k) for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "roofline", "mfma_all_convs",
                                             "hip_kernels_ms_per_step")}
                out["launch"] = j.get("config", {}).get("launch")
                if parity:
                    out["parity"] = {k: v for k, v in parity.items() if k.startswith("f32_")}
                return out
        return {"value": None, "error": res.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "error": "f32 child exceeded %d s" % timeout_s}


def miopen_db_note(args):
    """What the run found in --miopen-db BEFORE it started (a run in find mode `normal` adds to it)."""
    return _DB_NOTE or "no recorded find results"


_DB_NOTE = ""


def scan_miopen_db(path):
    global _DB_NOTE
    try:
        files = [f for f in os.listdir(path) if f.endswith(".ufdb.txt")]
        n = 0
        for f in files:
            with open(os.path.join(path, f)) as fh:
                n += sum(1 for ln in fh if ln.strip())
        _DB_NOTE = "recorded find results: %d problems in %s" % (n, os.path.relpath(path, ROOT)) if files else ""
    except OSError:
        _DB_NOTE = ""


def library_md5():
    import hashlib
    from bts_amd import _lib
    with open(_lib.LIB_PATH, "rb") as f:
        return hashlib.md5(f.read()).hexdigest()


def attach_traffic(roof):
    """roofline.traffic = measured fabric-side bytes per launch of the kernel family (FETCH_SIZE / WRITE_SIZE PMC passes of THIS
    bench command, collected and corrected as MI355X_MICROARCH.md prescribes and summarised by tools/pmc_traffic.py into
    profiles/pmc_traffic.json).  PMC counters cannot be read inside the process, so the committed summary is looked up by kernel
    family -- and only counts as a measurement of THIS run when it was taken on the same binary: the file records the md5 of the
    libbts_amd.so it profiled; on a mismatch the figures go under `traffic_archived` (with that md5) and `traffic` stays null."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not roof or not os.path.exists(path):
        return
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return
    meta = table.get("_meta", {})
    same = bool(meta.get("library_md5")) and meta.get("library_md5") == library_md5()
    for key in ("roofline", "roofline_lpg", "roofline_elementwise"):
        r = roof.get(key)
        if not r:
            continue
        ent = table.get(r["kernel"]) or table.get(r["kernel"].split(" ")[0])
        if not ent:
            continue
        if same:
            r["traffic"] = ent.get("bytes_per_launch")
            r["traffic_source"] = "%s; library md5 %s = the binary timed here" % (ent.get("source"), meta.get("library_md5"))
            if meta.get("config"):
                r["traffic_config"] = meta["config"]
        else:
            r["traffic_archived"] = {"bytes_per_launch": ent.get("bytes_per_launch"), "library_md5": meta.get("library_md5"),
                                     "note": "PMC passes of an EARLIER build of libbts_amd.so: not a measurement of this run"}


def parity_check(args, model, image, focal, dev):
    """One-batch parity figures for the JSON line, taken before the timed region: the decoder as it is about to be timed (bench
    dtype, fused LPG chains) and in f32, both against the checker -- the oracle's formulas (oracle/bts_oracle.py, restating
    bts.py:196-266) evaluated in f32 with torch ops on the device, on the SAME encoder features.  The oracle is used here only as
    the checker, outside the timed region; nothing of the timed step routes through it."""
    from bts_amd.model import bts
    from oracle import bts_oracle as O

    def l2(a, b):
        a, b = a.double(), b.double()
        return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()

    def mx(a, b):
        a, b = a.double(), b.double()
        return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()

    def elem(a, b, q):
        """ELEMENT-WISE relative error |a - b| / |b| (every output is a positive depth / max_depth map, so |b| > 0): its
        q-quantile over all pixels (torch.quantile is limited to 16 M elements: kthvalue on the flattened map)."""
        r = ((a.double() - b.double()).abs() / b.double().abs().clamp_min(1e-30)).flatten()
        if q >= 1.0:
            return r.max().item()
        kth = min(r.numel(), max(1, int(round(q * r.numel()))))
        return r.float().kthvalue(kth).values.item()
    nb = min(2, image.shape[0])
    md = 80.0 if args.dataset == "kitti" else 10.0
    was_training = model.training
    try:
        with torch.no_grad():
            model.eval()                                    # encoder BN in eval mode: features only need to be realistic
            feats = [f.float() for f in model.encoder(image[:nb])]
            P = {k: v.detach().clone() for k, v in model.decoder.state_dict().items()}
            with torch.backends.cudnn.flags(enabled=False):
                ref, _ = O.decoder_forward(P, feats, focal[:nb], md, args.dataset, True)
            out = {"checker": "oracle formulas (bts.py:196-266), f32 torch ops on the device, same encoder features; train-mode "
                              "BatchNorm; %d images; *_max = max|a-b| / max|b|, *_l2 = ||a-b|| / ||b|| (norm-wise, worst of the five "
                              "outputs); *_elem_p999 / *_elem_max = 99.9th percentile / maximum of the ELEMENT-WISE |a-b| / |b|" % nb}
            for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
                dec = bts(NS(max_depth=md, dataset=args.dataset, encoder=args.encoder, bts_size=512, decoder_dtype=dt),
                          model.encoder.feat_out_channels, 512).to(dev)
                dec.load_state_dict(P)
                dec.train()
                got = dec([f.clone() for f in feats], focal[:nb])
                out["%s_outputs_max" % tag] = float("%.3g" % max(mx(g, r) for g, r in zip(got, ref)))
                out["%s_outputs_l2" % tag] = float("%.3g" % max(l2(g, r) for g, r in zip(got, ref)))
                out["%s_outputs_elem_p999" % tag] = float("%.3g" % max(elem(g, r, 0.999) for g, r in zip(got, ref)))
                out["%s_outputs_elem_max" % tag] = float("%.3g" % max(elem(g, r, 1.0) for g, r in zip(got, ref)))
                del dec
            out["f64_arbiter"] = parity_arbiter_f64(args, O, bts, P, feats, focal, md, dev, elem)
        out["timed_dtype"] = args.dtype
        return out
    except Exception as e:   # noqa: BLE001  (reported, never fatal for the measurement)
        return {"error": str(e)[:200]}
    finally:
        model.train(was_training)
        torch.cuda.empty_cache()


def parity_arbiter_f64(args, O, bts, P, feats, focal, md, dev, elem):
    """Which side of the f32 comparison carries the error?  ONE image of the bench batch: the oracle's formulas in f64 on the device
    are the arbiter; the product's f32 decoder and torch's own f32 evaluation of the same formulas (the checker of the figures above)
    are both measured against it, ELEMENT-WISE (|a - b| / |b|, maximum and 99.9th percentile over every pixel of the five outputs).
    north_star's bound is 1e-4: `product_elem_max` is the figure it applies to.  `worst` names the output and pixel where the
    product is furthest off and what torch's f32 evaluation does at the same pixel (a comparable error there = conditioning of the
    formula at that pixel -- bts.py:146 divides by n1 u + n2 v + n3 --, not the kernel)."""
    from types import SimpleNamespace as NS
    f1 = [f[:1].contiguous() for f in feats]
    fo = focal[:1]
    with torch.backends.cudnn.flags(enabled=False):
        P64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in P.items()}
        ref64, _ = O.decoder_forward(P64, [f.double() for f in f1], fo.double(), md, args.dataset, True)
        P32 = {k: v.clone() for k, v in P.items()}
        ref32, _ = O.decoder_forward(P32, [f.clone() for f in f1], fo, md, args.dataset, True)
    dec = bts(NS(max_depth=md, dataset=args.dataset, encoder=args.encoder, bts_size=512, decoder_dtype=torch.float32),
              [f.shape[1] for f in f1], 512).to(dev)
    dec.load_state_dict(P)
    dec.train()
    got = dec([f.clone() for f in f1], fo)
    res = {"arbiter": "oracle formulas in f64 on the device, 1 image of the bench batch, train-mode BatchNorm",
           "product_elem_max": float("%.3g" % max(elem(g, r, 1.0) for g, r in zip(got, ref64))),
           "product_elem_p999": float("%.3g" % max(elem(g, r, 0.999) for g, r in zip(got, ref64))),
           "torch_f32_elem_max": float("%.3g" % max(elem(g, r, 1.0) for g, r in zip(ref32, ref64))),
           "torch_f32_elem_p999": float("%.3g" % max(elem(g, r, 0.999) for g, r in zip(ref32, ref64))),
           "per_output_product_elem_max": [float("%.3g" % elem(g, r, 1.0)) for g, r in zip(got, ref64)],
           "per_output_torch_f32_elem_max": [float("%.3g" % elem(g, r, 1.0)) for g, r in zip(ref32, ref64)]}
    worst_i = max(range(5), key=lambda i: res["per_output_product_elem_max"][i])
    g, r, t = got[worst_i].double().flatten(), ref64[worst_i].double().flatten(), ref32[worst_i].double().flatten()
    rel = (g - r).abs() / r.abs().clamp_min(1e-30)
    j = int(rel.argmax().item())
    W = got[worst_i].shape[-1]
    res["worst"] = {"output": worst_i, "pixel_yx": [j // W, j % W], "ref_value": float("%.6g" % r[j].item()),
                    "product_rel_err": float("%.3g" % rel[j].item()),
                    "torch_f32_rel_err_same_pixel": float("%.3g" % ((t[j] - r[j]).abs() / r[j].abs().clamp_min(1e-30)).item()),
                    "pixels_over_1e-4": int((rel > 1e-4).sum().item()), "pixels": int(rel.numel())}
    res["bound"] = 1e-4
    res["met"] = bool(res["product_elem_max"] <= 1e-4)
    del dec
    # bf16 attribution (same image, same arbiter): the product's bf16 decoder beside STOCK PyTorch's bf16 evaluation of the same
    # formulas (the oracle's decoder_forward under torch.autocast(bfloat16) on the device: convolutions and the elementwise ops
    # behind them in bf16).  A product error of the size torch's own bf16 evaluation shows is the precision of the format on these
    # formulas, not the kernels; bar: product <= 1.5 x torch-bf16 at the 99.9th percentile and at the maximum.
    try:
        decb = bts(NS(max_depth=md, dataset=args.dataset, encoder=args.encoder, bts_size=512, decoder_dtype=torch.bfloat16),
                   [f.shape[1] for f in f1], 512).to(dev)
        decb.load_state_dict(P)
        decb.train()
        gotb = decb([f.clone() for f in f1], fo)
        with torch.backends.cudnn.flags(enabled=False), torch.autocast("cuda", dtype=torch.bfloat16):
            refb, _ = O.decoder_forward({k: v.clone() for k, v in P.items()}, [f.clone() for f in f1], fo, md, args.dataset, True)
        refb = [r.float() for r in refb]
        a = {"product_bf16_elem_max": float("%.3g" % max(elem(g, r, 1.0) for g, r in zip(gotb, ref64))),
             "product_bf16_elem_p999": float("%.3g" % max(elem(g, r, 0.999) for g, r in zip(gotb, ref64))),
             "torch_bf16_elem_max": float("%.3g" % max(elem(g, r, 1.0) for g, r in zip(refb, ref64))),
             "torch_bf16_elem_p999": float("%.3g" % max(elem(g, r, 0.999) for g, r in zip(refb, ref64))),
             "per_output_product_bf16_elem_p999": [float("%.3g" % elem(g, r, 0.999)) for g, r in zip(gotb, ref64)],
             "per_output_torch_bf16_elem_p999": [float("%.3g" % elem(g, r, 0.999)) for g, r in zip(refb, ref64)],
             "leg": "oracle formulas under torch.autocast('cuda', torch.bfloat16) on the device, same image / features / parameters"}
        a["met"] = bool(a["product_bf16_elem_max"] <= 1.5 * a["torch_bf16_elem_max"] and
                        a["product_bf16_elem_p999"] <= 1.5 * a["torch_bf16_elem_p999"])
        res["bf16_attribution"] = a
        del decb
    except Exception as e:   # noqa: BLE001
        res["bf16_attribution"] = {"error": str(e)[:200]}
    return res


def lpg_op_roofline(B, H, W, replays=5):
    """BASELINE metric (ii), "LPG HBM GB/s": the bare LPG operator (the reference's native op boundary, bts_lpg_fwd / bts_lpg_bwd =
    local_planar_guidance.h:22-49) at the bench shape, k = 8, 4, 2, forward and backward.  Every launch works on different buffers
    of a rotation larger than the 256 MiB Infinity Cache (so the rate is an HBM rate); the rotation is captured into a hipGraph and
    replayed, HIP events around the replays -- issued one by one from Python, a 14 MB launch is bound by the ~10 us of host work
    per call, not by the device (all six kernels measured 10.3-11.1 us that way, gpurun r03l).
    Algorithmic bytes: forward P*4*(1 + 4/k^2), backward P*4*(1 + 8/k^2) per image."""
    from bts_amd import ops
    dev = torch.device("cuda", torch.cuda.current_device())
    res, tot_b, tot_s = {}, 0.0, 0.0
    for k in (8, 4, 2):
        h, w = H // k, W // k
        per = B * H * W * 4 * 2 + B * h * w * 16 * 2
        nrot = max(2, ((768 << 20) + per - 1) // per)
        eqs = [torch.randn(B, h, w, 4, device=dev) for _ in range(nrot)]
        gs = [torch.randn(B, H, W, device=dev) for _ in range(nrot)]
        outs = [torch.empty(B, H, W, device=dev) for _ in range(nrot)]
        geqs = [torch.empty(B, h, w, 4, device=dev) for _ in range(nrot)]
        for name, fn, byts in (("fwd", lambda i: ops.lpg_fwd(eqs[i], k, out=outs[i]), B * H * W * 4 * (1 + 4.0 / (k * k))),
                               ("bwd", lambda i: ops.lpg_bwd(gs[i], eqs[i], k, out=geqs[i]), B * H * W * 4 * (1 + 8.0 / (k * k)))):
            for i in range(nrot):
                fn(i)
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    for i in range(nrot):
                        fn(i)
            torch.cuda.current_stream().wait_stream(side)
            graph.replay()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(replays):
                graph.replay()
            e.record()
            torch.cuda.synchronize()
            sec = s.elapsed_time(e) * 1e-3 / (replays * nrot)
            res["k%d_%s_GBps" % (k, name)] = round(byts / sec / 1e9, 1)
            tot_b += byts
            tot_s += sec
            del graph
        del eqs, gs, outs, geqs
    torch.cuda.empty_cache()
    single = tot_b / tot_s / 1e9
    # The same six problems as TWO launches (bts_lpg_fwd_multi / bts_lpg_bwd_multi: the k = 8, 4, 2 heads of one batch handed over
    # together): at this shape a single scale is 14-27 MB behind a ~2 us launch boundary, which alone caps three dependent launches
    # near 0.45 of the HBM peak.  Same rotation discipline (> 256 MiB per buffer class), same graph-replay timing.
    ks = (8, 4, 2)
    per = sum(B * H * W * 4 * 2 + B * (H // k) * (W // k) * 16 * 2 for k in ks)
    nrot = max(2, ((768 << 20) + per - 1) // per)
    eqs = [[torch.randn(B, H // k, W // k, 4, device=dev) for k in ks] for _ in range(nrot)]
    for r in eqs:
        for e in r:
            e[..., 2] += 3.0
    gs = [[torch.randn(B, H, W, device=dev) for _ in ks] for _ in range(nrot)]
    outs = [[torch.empty(B, H, W, device=dev) for _ in ks] for _ in range(nrot)]
    geqs = [[torch.empty(B, H // k, W // k, 4, device=dev) for k in ks] for _ in range(nrot)]
    mb, ms, mres = 0.0, 0.0, {}
    for name, fn, byts in (("fwd", lambda i: ops.lpg_fwd_multi(eqs[i], list(ks), outs=outs[i]), sum(B * H * W * 4 * (1 + 4.0 / (k * k)) for k in ks)),
                           ("bwd", lambda i: ops.lpg_bwd_multi(gs[i], eqs[i], list(ks), outs=geqs[i]), sum(B * H * W * 4 * (1 + 8.0 / (k * k)) for k in ks))):
        for i in range(nrot):
            fn(i)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                for i in range(nrot):
                    fn(i)
        torch.cuda.current_stream().wait_stream(side)
        graph.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(replays):
            graph.replay()
        e.record()
        torch.cuda.synchronize()
        sec = s.elapsed_time(e) * 1e-3 / (replays * nrot)
        mres["k842_%s_GBps" % name] = round(byts / sec / 1e9, 1)
        mres["k842_%s_us" % name] = round(sec * 1e6, 2)
        mb += byts
        ms += sec
        del graph
    del eqs, gs, outs, geqs
    torch.cuda.empty_cache()
    ach = mb / ms / 1e9
    # Headline = the SINGLE-SCALE launches: that is the reference's op boundary (local_planar_guidance.h:22-49) and the only form a
    # decoder can issue -- depth_8x8 feeds conv3, whose output feeds reduc4x4, and so on (bts.py:229-256), so the three scales of one
    # batch are never available together.  The one-launch form is a library entry point for callers that do hold them; secondary.
    return {"kernel": "lpg_fwd/bwd_kernel<k=8,4,2> (bare LPG operator at the reference's op boundary: six single-scale launches, bts_lpg_fwd / bts_lpg_bwd)",
            "bound": "hbm", "achieved": round(single, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(single / HBM_PEAK_GBS, 4), "traffic": None,
            "alg_bytes_per_launch": round(tot_b / 6), "shape": "%dx%dx%d" % (B, H, W), "hbm_resident": True,
            "timing": "hipGraph replay of the buffer rotation (device time incl. inter-kernel gaps)", "per_kernel": res,
            "multi_scale_launch": {"kernel": "lpg_multi_kernel: k = 8, 4, 2 of one batch as ONE launch per direction (bts_lpg_fwd_multi / "
                                             "bts_lpg_bwd_multi) -- not a launch the decoder issues, its scales depend on each other",
                                   "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
                                   "alg_bytes_per_launch": round(mb / 2), "per_launch": mres}}


def infer_main(args):
    """configs[4]: inference-only path (bts_test.py:119), batch 32 at 704x1216, bf16; images/s of the no-grad forward
    (fused LPG heads, no tape) + AbsRel of the bf16 depth against the f32 CPU oracle on ONE image of the same inputs."""
    from bts_amd import _lib, profiler
    from bts_amd.model import BtsModel, weights_init_xavier
    from oracle import bts_oracle as O
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    _lib.load()
    H, W, B = (704, 1216, 32) if (args.height, args.width, args.batch) == (352, 1216, 8) else (args.height, args.width, args.batch)
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    params = NS(encoder=args.encoder, max_depth=80.0, dataset="kitti", bts_size=512, decoder_dtype=cdt)
    torch.manual_seed(0)
    model = BtsModel(params)
    model.decoder.apply(weights_init_xavier)
    model.eval()
    gen = torch.Generator().manual_seed(99)
    image = torch.randn(B, 3, H, W, generator=gen)
    from bts_amd import synth
    focal = synth.synth_focal(B, "kitti")
    ref_depth = None
    if not args.no_cpu_baseline:
        torch.set_num_threads(min(os.cpu_count() or 1, 64))
        t0 = time.time()
        with torch.no_grad():
            feats = model.encoder(image[:1])
            P = {k: v for k, v in model.decoder.state_dict().items()}
            ref, _ = O.decoder_forward(P, feats, focal[:1], 80.0, "kitti", False)
        cpu_s = time.time() - t0
        ref_depth = ref[4]
    model.to(dev)
    image_d, focal_d = image.to(dev), focal.to(dev)
    torch.backends.cudnn.benchmark = bool(args.cudnn_benchmark)
    if args.channels_last:
        model.encoder.to(memory_format=torch.channels_last)
        image_d = image_d.contiguous(memory_format=torch.channels_last)

    def step():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.dtype == "bf16"):
            return model(image_d, focal_d)
    for _ in range(max(args.warmup, 1)):
        outs = step()
    prof = profiler.enable() if not args.no_kernel_events else None
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        outs = step()
    torch.cuda.synchronize()
    elapsed = time.time() - t0
    out = {"metric": "images/sec (inference forward) DenseNet161-BTS 704x1216", "value": round(B * args.steps / elapsed, 3),
           "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "%s no-grad forward, %dx%d, batch %d (BASELINE.json configs[4])" % (args.encoder, H, W, B),
                      "encoder": "stock PyTorch-ROCm (%s, PYTORCH_MIOPEN_SUGGEST_NHWC=%s (BatchNorm: %s), cudnn.benchmark=%d, MIOPEN_FIND_MODE=%s, %s)" % (
                          "channels_last" if args.channels_last else "contiguous NCHW", os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC", "unset"), os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC_BATCHNORM", "unset"),
                          args.cudnn_benchmark, os.environ.get("MIOPEN_FIND_MODE", "unset"), miopen_db_note(args))}}
    if prof is not None:
        profiler.disable()
        out.update(prof.summary(PEAK[args.dtype], HBM_PEAK_GBS, args.steps))
    # The output path behind the forward (bts_test.py:119-124, 179-185), timed on the last batch, 3 repeats:
    #   reference style: per image, five f32 maps to the host (`.cpu().numpy().squeeze()`), then depth * 256 -> uint16 in numpy
    #   device path    : bts_depth_to_u16 on the batch + ONE device -> host copy of 2 bytes per pixel (bts_amd/loops.py)
    from bts_amd import evalops
    import numpy as np

    def ref_style():
        res = []
        for i in range(B):
            maps = [o[i].float().cpu().numpy().squeeze() for o in outs]
            res.append((maps[4] * 256.0).astype(np.uint16))
        return res

    def dev_style():
        return evalops.depth_to_uint16(outs[4].float(), "kitti").cpu().numpy()
    tt = {}
    for name, fn in (("reference_style_five_f32_maps_per_image", ref_style), ("device_uint16_one_copy_per_batch", dev_style)):
        fn()
        torch.cuda.synchronize()
        t1 = time.time()
        for _ in range(3):
            got = fn()
        torch.cuda.synchronize()
        tt[name] = round((time.time() - t1) / 3 * 1e3, 3)
    a, b = ref_style(), dev_style()
    fwd_ms = elapsed / args.steps * 1e3
    out["output_path_ms"] = dict(tt, payload_identical=bool(all(np.array_equal(a[i], b[i, 0]) for i in range(B))),
                                 images_per_s_with_reference_style_output=round(B / (fwd_ms + tt["reference_style_five_f32_maps_per_image"]) * 1e3, 2),
                                 images_per_s_with_device_output=round(B / (fwd_ms + tt["device_uint16_one_copy_per_batch"]) * 1e3, 2),
                                 note="per batch of %d; `value` is the forward alone (outputs stay on the device)" % B)
    if ref_depth is not None:
        est = outs[4][:1].float().cpu()
        out["absrel_vs_cpu_oracle"] = round(((est - ref_depth).abs() / ref_depth).mean().item(), 6)
        out["cpu_baseline"] = {"value": round(1.0 / cpu_s, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "oracle encoder+decoder forward, f32, 1 image %dx%d (%.1f s)" % (H, W, cpu_s)}
    emit(out)


def plumbing_main(args, world, rank):
    """Launch-path check without a model (CPU test of `--gpus N`): process group from the environment, one all-reduce, the
    barrier + max-over-ranks timing of the real line, rank 0 prints a line with the world it saw."""
    dist.init_process_group(backend=args.backend, init_method="env://")
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    dist.barrier()
    el = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    if rank == 0:
        emit({"plumbing": True, "n_gpus": world, "sum_ranks": t.item(), "max_elapsed": el.item(), "backend": args.backend})
    dist.destroy_process_group()


_RESULT_FD = None


def own_stdout():
    """The contract is ONE line on stdout.  Libraries under this process write there too -- RCCL 2.26 prints a five-line version
    banner (`RCCL version : ...`, `Librccl path : ...`) through C stdio when its first communicator comes up, which lands AFTER
    the JSON line when stdout is a file or a pipe (seen on the world-1 RCCL run, profiles/r05_bench_world1_rccl.json.err) and MIOpen
    can do the same.  So: keep a private copy of the real stdout for the result line and point fd 1 (C stdio and Python's
    sys.stdout alike) at stderr for the rest of the process."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    """The result line, on the real stdout (own_stdout())."""
    line = (json.dumps(obj) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
        return
    while line:
        n = os.write(_RESULT_FD, line)
        line = line[n:]


def main():
    args = parse()
    own_stdout()
    if args.miopen_db != "none":
        scan_miopen_db(args.miopen_db)
    if args.cpu_baseline_only:
        emit(cpu_baseline(args))
        return
    if args.mode == "infer":
        return infer_main(args)
    if args.gpus is None:                         # left at its default: a launcher's world is adopted, an explicit mismatch refused
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))               # no launcher around us: become one (one rank per GPU)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:                        # never print an n_gpus the command line did not ask for
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus (or without a launcher)"
                 % (args.gpus, world))
    multi = world > 1 or bool(args.force_dist)
    if args.plumbing_only:
        return plumbing_main(args, world, rank)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    if args.channels_last and os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC") != "1":
        sys.exit("bench.py: --channels-last 1 needs --miopen-nhwc 1 (ATen would hand MIOpen NCHW copies and return NCHW-laid-out "
                 "weight gradients for channels-last weights)")
    local = local % torch.cuda.device_count()      # (plumbing tests run several ranks on one GPU over gloo)
    torch.cuda.set_device(local)                   # before the process group: RCCL binds the communicator to the current device
    if world > 1:
        dist.init_process_group(backend=args.backend, init_method="env://")
    elif multi:
        import socket
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        dist.init_process_group(backend=args.backend, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    dev = torch.device("cuda", local)

    from bts_amd import _lib, profiler
    from bts_amd.model import BtsModel, silog_loss, weights_init_xavier
    _lib.load()

    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    params = NS(encoder=args.encoder, max_depth=80.0 if args.dataset == "kitti" else 10.0, dataset=args.dataset,
                bts_size=512, decoder_dtype=cdt)
    torch.manual_seed(0)
    model = BtsModel(params)
    model.train()
    model.decoder.apply(weights_init_xavier)
    set_misc(model)
    model.to(dev)
    if args.channels_last:
        model.encoder.to(memory_format=torch.channels_last)
    net = model
    reducer = None
    bufsync = None
    red_mode = args.reducer
    if red_mode == "auto":
        red_mode = "bts"      # hook-driven GradAllReducer, eager step: measured fastest N>1 form (profiles/r02_bench_dist_world1.md)
    if multi and red_mode == "ddp":
        # ResNet-family backbones carry the unused torchvision head (avgpool/fc): bts_main.py:352 sets find_unused_parameters
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], gradient_as_bucket_view=True,
                                                        broadcast_buffers=True,        # the reference's (default) setting
                                                        find_unused_parameters="resne" in args.encoder)
    elif multi:
        from bts_amd.parallel import BufferSync, GradAllReducer, broadcast_parameters
        broadcast_parameters(model)
        reducer = GradAllReducer(model.parameters(), reduce_single=bool(args.force_dist))
        bufsync = BufferSync(model)      # DDP's broadcast_buffers=True (bts_main.py:352): rank 0's BatchNorm buffers before every forward
    use_graph = bool(args.graph) and not multi
    split_graph = multi and red_mode == "bts-graph"
    torch.backends.cudnn.benchmark = bool(args.cudnn_benchmark)
    lr_t = torch.tensor(1e-4, device=dev)      # capturable optimizers read lr from a device tensor
    groups = [{"params": [p for p in model.encoder.parameters() if p.requires_grad], "weight_decay": 1e-2},
              {"params": list(model.decoder.parameters()), "weight_decay": 0.0}]       # bts_main.py:371-373
    own_opt = args.optimizer == "bts"
    if own_opt:
        from bts_amd.optim import FusedAdamW
        opt = FusedAdamW(groups, lr=1e-4, eps=1e-3)
    else:
        opt = torch.optim.AdamW(groups, lr=lr_t if use_graph else 1e-4, eps=1e-3, fused=True, capturable=use_graph)
    crit = silog_loss(0.85)
    image, focal, gt = make_batch(args, dev, 1234 + rank)
    if args.channels_last:
        image = image.contiguous(memory_format=torch.channels_last)
    mask_thr = 1.0 if args.dataset == "kitti" else 0.1
    parity = parity_check(args, model, image, focal, dev) if (args.parity and world == 1 and rank == 0) else None
    total_steps = 50 * 1000
    gstep = [0]

    def poly_lr():
        return (1e-4 - 1e-5) * (1 - gstep[0] / total_steps) ** 0.9 + 1e-5     # bts_main.py:456-458

    def step_body():
        if reducer is not None:
            reducer.zero_grad()                      # flat buckets own the gradients
            bufsync()
        else:
            opt.zero_grad(set_to_none=not own_opt)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.dtype == "bf16"):
            outs = net(image, focal)
        mask = gt > mask_thr                         # built per step, as the reference does (bts_main.py:449-452)
        loss = crit(outs[4], gt, mask)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        if own_opt:
            opt.step(prepared=True)
        else:
            opt.step()
        return loss

    # split form for N > 1 (bts-graph): A = zero + forward + loss + backward with the reducer's hooks deferred, B = optimizer
    def step_fwd_bwd():
        reducer.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.dtype == "bf16"):
            outs = net(image, focal)
        mask = gt > mask_thr
        loss = crit(outs[4], gt, mask)
        with reducer.no_sync():
            loss.backward()
        return loss

    def step_opt():
        opt.step(prepared=True)

    def set_lr():
        lr = poly_lr()
        if own_opt:
            opt.prepare_step(lrs=[lr, lr])          # device-resident {lr, bias corrections}: outside any graph
        elif use_graph:
            lr_t.fill_(lr)
        else:
            for g in opt.param_groups:
                g["lr"] = lr

    def step_eager():
        set_lr()
        loss = step_body()
        gstep[0] += 1
        return loss

    for _ in range(max(args.warmup, 1)):
        step_eager()
    graph, static_loss, graph_note = None, None, "eager"
    if use_graph:
        try:
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step_eager()                       # one step on the capture stream (allocator warm-up)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_loss = step_body()
            graph_note = "hipGraph replay of the whole step (fwd+loss+bwd+AdamW)"
        except Exception as e:   # noqa: BLE001
            graph, graph_note = None, "eager (graph capture failed: %s)" % str(e)[:120]
            torch.cuda.synchronize()

    graph_b = None
    if split_graph and own_opt and reducer is not None:
        # N > 1: the collective stays OUTSIDE the graphs (nothing of RCCL is captured): replay(fwd+bwd) -> 3 all-reduces of
        # the flat gradient buckets on the same stream -> replay(AdamW).  Falls back to the eager DDP-style step on any error.
        try:
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                set_lr()
                step_fwd_bwd()
                reducer.reduce_all()
                step_opt()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_loss = step_fwd_bwd()
            graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_b):
                step_opt()
            graph_note = "hipGraph replay (fwd+loss+bwd) -> GradAllReducer.reduce_all over %s -> hipGraph replay (AdamW)" % args.backend
        except Exception as e:   # noqa: BLE001
            graph, graph_b, graph_note = None, None, "eager (split graph capture failed: %s)" % str(e)[:120]
            torch.cuda.synchronize()

    def step():
        if graph is None:
            return step_eager()
        set_lr()
        if graph_b is not None:
            bufsync()                 # outside the captured graphs, like the gradient exchange
        graph.replay()
        if graph_b is not None:
            reducer.reduce_all()
            graph_b.replay()
        gstep[0] += 1
        return static_loss

    for _ in range(2):
        step()
    if reducer is not None:
        torch.cuda.synchronize()
        reducer.zero_grad()
        reducer.fraction_log, reducer._exposed = [], []
        reducer.timing = True          # exposed_comm_ms / bucket_launch_fraction of the TIMED steps (two events per step, no sync)
    prof = None
    if not args.no_kernel_events and rank == 0 and graph is None and world == 1:
        prof = profiler.enable()      # N > 1: never inside the timed region (per-launch events on one rank would hold back all of them)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.time() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = float(loss.item())
    exchange = None
    if reducer is not None:
        exchange = reducer.exchange_stats()
        reducer.timing = False
        exchange["note"] = ("exposed_comm_ms = device time between the last kernel of backward and the end of the waits for the bucket "
                            "all-reduces (what the exchange adds to the step); bucket_launch_fraction[i] = share of the step's gradient "
                            "hooks that had fired when bucket i went on the wire (buckets in launch order: decoder first; 1.0 = nothing "
                            "left to overlap with); plus one un-overlapped buffer broadcast before every forward (DDP broadcast_buffers)")
    eager = None
    if world == 1 and graph is not None and args.eager_steps > 0:
        # the same step issued launch by launch from Python (no graph, no per-launch events): what `--gpus N` (N > 1) runs
        step_eager()
        torch.cuda.synchronize()
        t1 = time.time()
        for _ in range(args.eager_steps):
            step_eager()
        torch.cuda.synchronize()
        e_ms = (time.time() - t1) / args.eager_steps * 1e3
        eager = {"ms_per_step": round(e_ms, 3), "value": round(args.batch / e_ms * 1e3, 3), "unit": "images/s", "steps": args.eager_steps,
                 "note": "eager launch mode at N=1, un-profiled; an N>1 line (eager step + hook-driven gradient exchange) compares with THIS, "
                         "not with the replayed `value`"}
    if (graph is not None or world > 1) and not args.no_kernel_events and (rank == 0 or world > 1):
        # events cannot be recorded inside a graph replay: time the same kernels over the same number of
        # eager steps right after the timed region (same process, same buffers, same clocks).  With N > 1 every rank runs
        # these steps (their collectives must match); only rank 0 records.
        if rank == 0:
            prof = profiler.enable()
        for _ in range(args.steps):
            step_eager()
        torch.cuda.synchronize()
    roof = None
    if prof is not None:
        profiler.disable()
        roof = prof.summary(PEAK[args.dtype], HBM_PEAK_GBS, args.steps)
        if (args.encoder, args.height, args.width, args.batch, args.dtype) == ("densenet161_bts", 352, 1216, 8, "bf16"):
            attach_traffic(roof)        # the PMC passes were taken on this configuration only
        roof["library_md5"] = library_md5()
        if args.dump_launches:
            agg = {}
            for family, tag, us, work, nb in prof.launches():
                a = agg.setdefault((family, tag or ""), [0, 0.0, 0.0, 0.0])
                a[0] += 1
                a[1] += us
                a[2] += work
                a[3] += nb
            rows = [{"family": k[0], "tag": k[1], "launches_per_step": round(v[0] / args.steps, 2), "us_per_launch": round(v[1] / v[0], 2),
                     "us_per_step": round(v[1] / args.steps, 2), "work_per_launch": v[2] / v[0], "alg_bytes_per_launch": v[3] / v[0]}
                    for k, v in agg.items()]
            rows.sort(key=lambda r: -r["us_per_step"])
            with open(args.dump_launches, "w") as f:
                json.dump({"library_md5": roof["library_md5"], "steps": args.steps, "rows": rows}, f, indent=0)
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": "images/sec (train step) DenseNet161-BTS 352x1216" if (args.encoder, args.height, args.width) == ("densenet161_bts", 352, 1216) else "images/sec (train step) %s %dx%d" % (args.encoder, args.height, args.width),
            "value": round(args.batch * world * args.steps / elapsed, 3),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s train step (fwd+silog+bwd+AdamW), %dx%d, %d img/GPU, %s" %
                       (args.encoder, args.height, args.width, args.batch,
                        "kitti focal scaling" if args.dataset == "kitti" else "nyu (no focal scaling)"),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world, "grad_exchange": ("none" if not multi else ("torch DDP over %s" % args.backend if reducer is None else "bts GradAllReducer over %s" % args.backend)),
                       "encoder": "stock PyTorch-ROCm (%s autocast, %s, PYTORCH_MIOPEN_SUGGEST_NHWC=%s (BatchNorm: %s), cudnn.benchmark=%d, MIOPEN_FIND_MODE=%s, %s)" % (
                           args.dtype, "channels_last" if args.channels_last else "contiguous NCHW", os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC", "unset"), os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC_BATCHNORM", "unset"),
                           args.cudnn_benchmark, os.environ.get("MIOPEN_FIND_MODE", "unset"), miopen_db_note(args)), "decoder": "HIP kernels via libbts_amd.so", "launch": graph_note, "optimizer": "bts_adamw_step (fused HIP)" if own_opt else "torch.optim.AdamW(fused)",
                       "final_loss": round(final_loss, 5)},
        }
        if roof is not None:
            out.update(roof)
        if parity is not None:
            out["parity"] = parity
        if eager is not None:
            out["eager"] = eager
        if exchange is not None:
            out["exchange"] = exchange
        if multi:
            out["config"]["launch_note"] = ("N>1 runs the EAGER step (hook-driven exchange overlapped with backward): compare this `value` / n_gpus "
                                            "with the N=1 line's `eager.value`, not with its hipGraph-replayed `value`")
        if (args.f32_line and world == 1 and args.dtype == "bf16" and not multi
                and (args.encoder, args.height, args.width, args.batch) == ("densenet161_bts", 352, 1216, 8)):
            out["f32"] = f32_line_subprocess(args, parity)
            arb = (parity or {}).get("f64_arbiter") or {}
            out["config"]["f32_images_per_s"] = out["f32"].get("value")            # the configuration that meets the 1e-4 bound
            out["config"]["f32_ms_per_step"] = out["f32"].get("ms_per_step")
            out["config"]["f32_parity_1e-4_met"] = arb.get("met")
            out["config"]["f32_parity_elem_max_vs_f64"] = arb.get("product_elem_max")
            out["config"]["bf16_parity_elem_max_vs_f64"] = (arb.get("bf16_attribution") or {}).get("product_bf16_elem_max")
            out["config"]["bf16_torch_autocast_elem_max_vs_f64"] = (arb.get("bf16_attribution") or {}).get("torch_bf16_elem_max")
        if args.lpg_op and world == 1:
            try:
                out["roofline_lpg_op"] = lpg_op_roofline(args.batch, args.height, args.width)
                if (args.batch, args.height, args.width) == (8, 352, 1216):
                    # the same operator at BASELINE.json configs[4]'s shape (inference, 32 x 704 x 1216): 8x the bytes per launch, so
                    # the ~2 us a dependent launch costs under graph replay no longer dominates a 14-27 MB kernel
                    out["roofline_lpg_op_c5"] = lpg_op_roofline(32, 704, 1216, replays=3)
            except Exception as e:   # noqa: BLE001
                out["roofline_lpg_op"] = {"error": str(e)[:160]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_subprocess(args)
        emit(out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
