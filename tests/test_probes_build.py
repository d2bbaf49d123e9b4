"""The stand-alone GPU probes under tools/probes/ compile the PRODUCT source (they #include bts_amd/csrc/*.hip with its
launch-shape constants turned into variables), so they rot the moment a kernel signature changes.  Cross-compiling them
for gfx950 needs no GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    return None


@pytest.mark.parametrize("probe", ["ew_probe.hip", "tr_probe.hip", "wgrad_tile_probe.hip"])
def test_probe_compiles_against_product_source(tmp_path, probe):
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip("hipcc not available")
    out = tmp_path / probe.replace(".hip", "")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value",
           os.path.join(ROOT, "tools", "probes", probe), "-o", str(out)]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.exists() and out.stat().st_size > 0


LEGACY = sorted(f for f in os.listdir(os.path.join(ROOT, "tools", "probes", "legacy")) if f.endswith(".hip"))


@pytest.mark.parametrize("src", LEGACY)       # whatever the tree holds (conv_legacy.hip only ever existed in a build container: .gitignore ate it)
def test_archived_kernel_variants_still_compile(tmp_path, src):
    """tools/probes/legacy/: convolution schedules that were measured and lost (register staging, whole-chunk prefetch, the
    three-stage ring, two staggered wave groups, the 2-byte-scatter weight gradient) are out of libbts_amd.so but keep compiling
    against csrc/conv_common.h, so an A/B can be repeated."""
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip("hipcc not available")
    out = tmp_path / src.replace(".hip", ".o")
    cmd = [hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-c",
           os.path.join(ROOT, "tools", "probes", "legacy", src), "-o", str(out)]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.exists() and out.stat().st_size > 0
