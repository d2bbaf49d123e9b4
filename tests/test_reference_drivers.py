"""The reference's own drivers, UNMODIFIED, on the drop-in: tools/run_reference.py runs
pytorch/bts_main.py (train, 3 steps + online eval + checkpoints), resumes it from the checkpoint it wrote, and runs
pytorch/bts_test.py on that checkpoint, over a tiny synthetic NYU-style dataset.

Runs only where the reference tree exists (the build container).  That box has no GPU, so the three kernel-launching
forward() bodies are served by the CPU checker (tests/ref_cpu_executor.py); everything else -- `from bts import *`, the
model file copied to <log_dir>/<model_name>/<model_name>.py and re-imported by name, weights_init_xavier / set_misc on
the module tree, the two AdamW parameter groups, DataParallel-prefixed checkpoints with the numpy `best_eval_steps`
entry, optimizer-state resume -- is the product's real code under the reference's real call sequence.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("BTS_REFERENCE_ROOT", "/root/reference")
REF_PY = os.path.join(REF, "pytorch")

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF_PY, "bts_main.py")),
                                reason="reference tree not present (GPU box)")


def _dataset(root, n):
    """n NYU-style samples: 640x480 RGB jpg + 16-bit depth png (millimetres), and the filenames file
    (`<rgb> <depth> <focal>`, train_test_inputs/nyudepthv2_train_files_with_gt.txt)."""
    from PIL import Image
    rng = np.random.RandomState(0)
    os.makedirs(os.path.join(root, "scene"), exist_ok=True)
    lines = []
    for i in range(n):
        rgb = (rng.rand(480, 640, 3) * 255).astype(np.uint8)
        depth = (rng.uniform(500, 9500, size=(480, 640))).astype(np.uint16)
        depth[rng.rand(480, 640) < 0.2] = 0
        Image.fromarray(rgb).save(os.path.join(root, "scene", "rgb_%05d.jpg" % i))
        Image.fromarray(depth).save(os.path.join(root, "scene", "sync_depth_%05d.png" % i))
        lines.append("scene/rgb_%05d.jpg scene/sync_depth_%05d.png 518.8579" % (i, i))
    fn = os.path.join(root, "files.txt")
    with open(fn, "w") as f:
        f.write("\n".join(lines) + "\n")
    return fn


def _args_file(path, **kw):
    with open(path, "w") as f:
        for k, v in kw.items():
            f.write("--%s\n" % k if v is True else "--%s %s\n" % (k, v))
    return path


def _run(script, argfile, workdir, timeout=900):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import run_reference
    finally:
        sys.path.pop(0)
    r = run_reference.run(REF_PY, script, [os.path.basename(argfile)], workdir=workdir, allow_cpu=True,
                          extra_env={"BTS_REF_POSTIMPORT": "tests.ref_cpu_executor", "OMP_NUM_THREADS": "8"},
                          timeout=timeout, capture=True)
    assert r.returncode == 0, "%s failed\nSTDOUT:\n%s\nSTDERR:\n%s" % (script, r.stdout[-3000:], r.stderr[-3000:])
    return r


def test_reference_train_resume_and_test_drivers(tmp_path):
    data = str(tmp_path / "data")
    files = _dataset(data, 2)       # 2 samples x 2 epochs: the checkpoint at global_step 2 falls on an epoch boundary, so the
    # resumed run's poly-LR base (1 - step/total) stays positive (bts_main.py:433, 457 restart the epoch at step 0)
    work = str(tmp_path / "work")
    os.makedirs(work)
    log = str(tmp_path / "log")
    os.makedirs(log)
    common = dict(encoder="densenet121_bts", dataset="nyu", data_path=data, gt_path=data, filenames_file=files,
                  input_height=64, input_width=96, max_depth=10, bts_size=128, log_directory=log, batch_size=1,
                  num_epochs=2, num_threads=0, log_freq=1, eval_freq=2, do_online_eval=True, data_path_eval=data,
                  gt_path_eval=data, filenames_file_eval=files, min_depth_eval=1e-3, max_depth_eval=10, eigen_crop=True,
                  adam_eps=1e-3)
    # ---- 1. train from scratch: `from bts import *`, cp bts.py -> <log>/m0/m0.py, 4 steps, online eval at step 2 ----
    _args_file(os.path.join(work, "train0.txt"), mode="train", model_name="m0", **common)
    r = _run("bts_main.py", os.path.join(work, "train0.txt"), work)
    assert "Fixing first conv layer" in r.stdout and "Total number of learning parameters" in r.stdout
    assert r.stdout.count("[epoch][s/s_per_e/gs]") == 4
    assert "Initial variables' sum" in r.stdout                       # the np.sum(list of tensors) hazard (bts_main.py:425-429)
    mdir = os.path.join(log, "m0")
    assert os.path.isfile(os.path.join(mdir, "m0.py"))                # the drop-in, copied by the driver
    assert open(os.path.join(mdir, "m0.py")).read() == open(os.path.join(ROOT, "dropin", "bts.py")).read()
    cks = sorted(f for f in os.listdir(mdir) if f.startswith("model-2-best_"))
    assert len(cks) == 9, os.listdir(mdir)                            # one per eval metric (bts_main.py:510-539)
    events = [json.loads(line) for line in open(os.path.join(mdir, "summaries", "events.jsonl"))]
    assert any(e["tag"] == "silog_loss" for e in events) and any(e["tag"].startswith("lpg8x8/image") for e in events)
    ck_path = os.path.join(mdir, cks[0])
    ck = torch.load(ck_path, weights_only=False)
    assert ck["global_step"] == 2 and isinstance(ck["best_eval_steps"], np.ndarray)
    assert all(k.startswith("module.") for k in ck["model"])
    # ---- 2. the checkpoint loads strictly into the product's modules, and the reference's (live) module tree agrees ----
    from types import SimpleNamespace as NS
    from bts_amd.model import BtsModel
    params = NS(encoder="densenet121_bts", max_depth=10.0, dataset="nyu", bts_size=128)
    m = torch.nn.DataParallel(BtsModel(params))
    m.load_state_dict(ck["model"])                                    # strict
    from oracle import ref_loader
    ref = ref_loader.load_reference()
    mref = torch.nn.DataParallel(ref.BtsModel(params))
    mref.load_state_dict(ck["model"])                                 # the same file loads in the REFERENCE's BtsModel
    assert len(ck["optimizer"]["state"]) == sum(1 for p in m.parameters() if True) - sum(
        1 for n, p in m.module.encoder.named_parameters() if ("conv0" in n or "norm" in n))
    # ---- 3. resume from it (model file imported BY NAME from the checkpoint's directory; optimizer state loaded) ----
    _args_file(os.path.join(work, "train1.txt"), mode="train", model_name="m1", checkpoint_path=ck_path, **common)
    r = _run("bts_main.py", os.path.join(work, "train1.txt"), work)
    assert "Loaded checkpoint" in r.stdout and "(global_step 2)" in r.stdout
    assert os.path.isfile(os.path.join(log, "m1", "m1.py"))
    assert "[epoch][s/s_per_e/gs]: [1][0/2/2]" in r.stdout            # continues at global_step 2, epoch 1
    # ---- 4. a checkpoint written by the product side loads in the driver: re-save through our modules, then test ----
    ck2 = os.path.join(mdir, "model-roundtrip")
    torch.save({"global_step": 2, "model": m.state_dict()}, ck2)
    _args_file(os.path.join(work, "test0.txt"), model_name="m0", encoder="densenet121_bts", data_path=data, dataset="nyu",
               filenames_file=files, max_depth=10, bts_size=128, checkpoint_path=ck2, save_lpg=True)
    r = _run("bts_test.py", os.path.join(work, "test0.txt"), work)
    assert "now testing 2 files" in r.stdout and "Done." in r.stdout
    raw = os.path.join(work, "result_m0", "raw")
    pngs = sorted(os.listdir(raw))
    assert len(pngs) == 2
    from PIL import Image
    a = np.array(Image.open(os.path.join(raw, pngs[0])))
    assert a.shape == (480, 640) and a.dtype in (np.uint16, np.int32) and a.max() <= 10000 and a.max() > 0
    assert len(os.listdir(os.path.join(work, "result_m0", "cmap"))) >= 2 * 4      # depth + 8x8/4x4/2x2 colour maps
