"""The C-ABI library loads and exports every symbol include/bts_amd.h declares, with the
prototypes the ctypes binding assumes (no compute calls: runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bts_amd.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int|long)\s+(bts_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        kinds = []
        for a in [x.strip() for x in args.split(",")]:
            if a in ("void", ""):
                continue
            if "*" in a or "bts_stream_t" in a:
                kinds.append("p")
            elif re.match(r"(const\s+)?float\b", a):
                kinds.append("f")
            elif re.match(r"(const\s+)?long\b", a):
                kinds.append("l")
            else:
                kinds.append("i")
        decls[name] = (ret, kinds)
    return decls


def _kind(ct):
    if ct is ctypes.c_float:
        return "f"
    if ct is ctypes.c_long:
        return "l"
    if ct is ctypes.c_int:
        return "i"
    return "p"


def test_header_matches_binding():
    from bts_amd import _lib
    decls = _declared()
    assert len(decls) >= 25
    assert set(decls) == set(_lib.SIGNATURES), set(decls) ^ set(_lib.SIGNATURES)
    for name, (ret, kinds) in decls.items():
        got = [_kind(a) for a in _lib.SIGNATURES[name]]
        assert got == kinds, (name, got, kinds)
        assert (name in _lib._LONG_RET) == (ret == "long"), name


def test_library_exports_all_symbols():
    from bts_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.bts_abi_version() == 4
    assert _lib.call("bts_silog_workspace_bytes", 1 << 20) > 0


def test_conv_desc_layout():
    """ctypes mirror of bts_conv_desc_t has the C struct's size (LP64, natural alignment)."""
    from bts_amd import _lib
    # 4 ints, nseg, 6*(8+4+4), 3 ints, 2 ints, 4*16 int16, pad, ptr, int, pad, ptr, 7 ints, float, pad, ptr, int, pad
    assert ctypes.sizeof(_lib.Seg) == 16
    assert ctypes.sizeof(_lib.ConvDesc) % 8 == 0


def test_missing_gpu_fails_loudly():
    import torch
    from bts_amd import _lib
    with pytest.raises(_lib.BtsAmdError):
        _lib.require_gpu(torch.zeros(1))


def test_header_is_plain_c99_and_struct_layouts_match(tmp_path):
    """include/bts_amd.h must be consumable by a C compiler (the reference-side binding is C / C++ / FFI), and the
    ctypes mirrors of its structs must have the C layout (sizes and a few offsets printed by a tiny C program)."""
    import shutil
    import subprocess
    from bts_amd import _lib
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "bts_amd.h"\n'
                   'int main(void) {\n'
                   '  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(bts_conv_desc_t), sizeof(bts_pack_job_t), sizeof(bts_unpack_job_t),\n'
                   '         sizeof(bts_aug_t), offsetof(bts_pack_job_t, first_block), offsetof(bts_unpack_job_t, first_block),\n'
                   '         offsetof(bts_aug_t, color), sizeof(bts_bn_desc_t), sizeof(bts_bn_seg_t), offsetof(bts_bn_desc_t, gamma),\n'
                   '         offsetof(bts_bn_desc_t, y2), offsetof(bts_conv_desc_t, fold_elu_y));\n  return 0;\n}\n')
    exe = tmp_path / "layout"
    subprocess.check_call([cc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(_lib.ConvDesc), ctypes.sizeof(_lib.PackJob), ctypes.sizeof(_lib.UnpackJob), ctypes.sizeof(_lib.AugParams),
            _lib.PackJob.first_block.offset, _lib.UnpackJob.first_block.offset, _lib.AugParams.color.offset,
            ctypes.sizeof(_lib.BnDesc), ctypes.sizeof(_lib.BnSeg), _lib.BnDesc.gamma.offset, _lib.BnDesc.y2.offset,
            _lib.ConvDesc.fold_elu_y.offset]
    assert got == want, (got, want)
