"""Where does a chunk of conv_igemm_dma go?  Runs a few decoder layers on the diagnostic build (tools/build_trace_lib.sh: s_memtime stamps
at the top of a chunk / after its vmcnt wait / after the barrier / after its last MFMA, lane 0 of waves 0 and 3 of 16 workgroups) and
prints, per layer, the mean cycles of the three sections over the steady-state chunks.  GPU box only; measurement tooling."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bts_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.environ.get("BTS_TRACE_LIB") or os.path.join(ROOT, "bts_amd", "lib", "libbts_amd_trace.so")
from bts_amd._lib import ACT_ELU  # noqa: E402
from bts_amd.conv import ConvLayer  # noqa: E402

os.environ["BTS_RES"] = "0"
DEV = "cuda"
CASES = [("daspp_dil12", 128, [256], 9, 12, False, (8, 44, 152)), ("daspp_1x1_960", 256, [960], 1, 1, False, (8, 44, 152)),
         ("upconv5", 512, [2208], 9, 1, True, (8, 11, 38)), ("conv5", 512, [512, 384], 9, 1, False, (8, 22, 76)),
         ("upconv4", 256, [512], 9, 1, True, (8, 22, 76))]


def main():
    lib = _lib.load()
    lib.bts_trace_dump.argtypes = [C.c_void_p]
    out = []
    for name, cout, segc, kk, dil, up, (N, H, W) in CASES:
        L = ConvLayer(name, cout, segc, kk, dil, up)
        dt = torch.bfloat16
        segs = [torch.randn(N, H, W, c, device=DEV).to(dt) for c in segc]
        k = 3 if kk == 9 else 1
        w = torch.randn(cout, sum(segc), k, k, device=DEV) * 0.05
        Ho, Wo = (2 * H, 2 * W) if up else (H, W)
        o = torch.empty(N, Ho, Wo, cout, dtype=dt, device=DEV)
        wp = L.pack_fwd(w, dt)
        for _ in range(3):
            L.forward(segs, wp, o, ACT_ELU)
        torch.cuda.synchronize()
        lib.bts_trace_clear()
        L.forward(segs, wp, o, ACT_ELU)          # (not timed: the stamped workgroups carry the stamps' overhead; the cycles are the result)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * (16 * 2 * 64 * 4))()
        lib.bts_trace_dump(buf)
        t = torch.tensor(list(buf), dtype=torch.float64).view(16, 2, 64, 4)
        rows = []
        for wg in range(16):
            for wv in range(2):
                st = t[wg, wv]
                n = int((st[:, 3] > 0).sum().item())
                if n < 8:
                    continue
                a = st[2:n - 1]                      # steady state: skip the first two chunks and the last
                nxt = st[3:n, 0]
                rows.append([(a[:, 1] - a[:, 0]).mean().item(), (a[:, 2] - a[:, 1]).mean().item(), (a[:, 3] - a[:, 2]).mean().item(),
                             (nxt - a[:, 3]).mean().item(), (nxt - a[:, 0]).mean().item(), n])
        if rows:
            r = torch.tensor(rows, dtype=torch.float64)
            m = r.mean(0)
            out.append(dict(lib=os.path.basename(_lib.LIB_PATH), case=name, waves=len(rows), chunks=int(m[5].item()),
                            prep_and_vmcnt_wait=round(m[0].item()), barrier_wait=round(m[1].item()), reads_mfma_section=round(m[2].item()),
                            loop_tail=round(m[3].item()), chunk_total=round(m[4].item()),
                            mfma_issue_floor=512, note="shader-clock cycles (s_memtime), means over the steady-state chunks of the stamped waves"))
        else:
            out.append(dict(case=name, error="no stamped workgroup (grid too small?)"))
        print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
