#!/usr/bin/env python
"""Run the UNMODIFIED reference drivers (cleinc/bts pytorch/bts_main.py, bts_test.py, bts_eval.py) on the bts_amd drop-in.

    python tools/run_reference.py --reference /path/to/bts/pytorch [--workdir DIR] bts_main.py arguments_train.txt
    python tools/run_reference.py --reference /path/to/bts/pytorch bts_test.py arguments_test.txt

What it does ("stack hazards"), without touching a byte of the reference tree:

  * builds a scratch working directory whose `bts.py` is dropin/bts.py (the file bts_main.py:569-585 copies into
    <log_dir>/<model_name>/<model_name>.py and later re-imports by name, bts_main.py:122-133, bts_test.py:68-74) and runs
    the driver with that directory as cwd and first on sys.path, the reference directory after it (bts_dataloader.py,
    distributed_sampler_no_evenly_divisible.py come from there);
  * stands in for packages the image lacks -- torchvision, tensorboardX, cv2 -- with the small modules in tools/ref_shims/
    (only those that are really missing; an installed package is never shadowed);
  * installs tools/ref_shims/site/sitecustomize.py through PYTHONPATH so that the process AND its mp.spawn children
    (bts_main.py:600-602) get the np.sum(list-of-tensors) fix and TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD=1.

The driver runs in a child interpreter as `runpy.run_path(<reference script>, run_name="__main__")`, i.e. exactly the
reference's code path including `if __name__ == '__main__': main()`.

With a GPU the drop-in executes its HIP kernels.  `--allow-cpu` (used by tests/test_reference_drivers.py on the
GPU-less build container) additionally makes `.cuda()` an identity; the model then needs an executor supplied by the
caller through BTS_REF_POSTIMPORT -- the product has none for CPU and raises.
"""
import argparse
import importlib.util
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIMS = os.path.join(ROOT, "tools", "ref_shims")
STANDINS = ("torchvision", "tensorboardX", "cv2")


def _installed(pkg):
    try:
        return importlib.util.find_spec(pkg) is not None
    except ValueError:          # a spec-less stub someone put into sys.modules (oracle/ref_loader.py does): not a real package
        return False


def prepare(reference, workdir):
    """Scratch cwd: bts.py = the drop-in; symlinks to the driver sources (bts_main.py `cp`s some of them next to the
    checkpoints, bts_main.py:581-585); shims/ = stand-ins for the packages that are missing."""
    os.makedirs(workdir, exist_ok=True)
    dst = os.path.join(workdir, "bts.py")
    if os.path.lexists(dst):
        os.remove(dst)
    with open(os.path.join(ROOT, "dropin", "bts.py")) as f:
        src = f.read()
    with open(dst, "w") as f:
        f.write(src)
    for name in os.listdir(reference):
        if name.endswith(".py") and name != "bts.py":
            link = os.path.join(workdir, name)
            if os.path.lexists(link):
                os.remove(link)
            os.symlink(os.path.join(reference, name), link)
    shim_dir = os.path.join(workdir, "_shims")
    os.makedirs(shim_dir, exist_ok=True)
    used = []
    for pkg in STANDINS:
        link = os.path.join(shim_dir, pkg)
        if os.path.lexists(link):
            os.remove(link)
        if not _installed(pkg):
            os.symlink(os.path.join(SHIMS, pkg), link)
            used.append(pkg)
    return shim_dir, used


def environment(reference, workdir, shim_dir, allow_cpu, extra_env=None):
    env = dict(os.environ)
    path = [os.path.join(SHIMS, "site"), shim_dir, workdir, reference, ROOT]
    if env.get("PYTHONPATH"):
        path.append(env["PYTHONPATH"])
    env["PYTHONPATH"] = os.pathsep.join(path)
    env["BTS_REF_SHIMS"] = "1"
    env["BTS_AMD_HOME"] = ROOT
    env["TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD"] = "1"
    env.setdefault("MPLBACKEND", "Agg")
    if allow_cpu:
        env["BTS_REF_ALLOW_CPU"] = "1"
    env.update(extra_env or {})
    return env


def run(reference, script, script_args, workdir=None, allow_cpu=False, extra_env=None, timeout=None, capture=False):
    reference = os.path.abspath(reference)
    script_path = os.path.join(reference, script)
    if not os.path.isfile(script_path):
        raise FileNotFoundError(script_path)
    workdir = os.path.abspath(workdir or tempfile.mkdtemp(prefix="bts_ref_"))
    shim_dir, used = prepare(reference, workdir)
    env = environment(reference, workdir, shim_dir, allow_cpu, extra_env)
    # the driver's sys.argv[0] stays its own file name: bts_main.py copies sys.argv[1] (the arguments file) around
    boot = ("import runpy, sys; sys.argv = [%r] + sys.argv[1:]; sys.path.insert(0, %r); "
            "runpy.run_path(%r, run_name='__main__')" % (script, workdir, script_path))
    cmd = [sys.executable, "-c", boot] + list(script_args)
    print("[run_reference] cwd=%s stand-ins=%s\n[run_reference] %s %s" % (workdir, used or "none", script, " ".join(script_args)),
          flush=True)
    return subprocess.run(cmd, cwd=workdir, env=env, timeout=timeout, capture_output=capture, text=capture)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", required=True, help="the reference's pytorch/ directory (read-only is fine)")
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--allow-cpu", action="store_true")
    ap.add_argument("script", help="bts_main.py | bts_test.py | bts_eval.py")
    ap.add_argument("script_args", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    sys.exit(run(a.reference, a.script, a.script_args, a.workdir, a.allow_cpu).returncode)


if __name__ == "__main__":
    main()
