"""HIP-event timing of the C-ABI launches (used by bench.py for the roofline object).

When enabled, every tracked entry point is bracketed by two events recorded on the stream the
kernel is launched on (the current PyTorch stream); callers attach the algorithmic work of the
launch (FLOPs for the MFMA convs, bytes for the streaming kernels) with `note()`.
"""
import torch

ACTIVE = None
_pending = None


class Profiler:
    def __init__(self):
        self.records = []   # (family, kind, work, start_evt, end_evt)

    def add(self, family, kind, work, s, e):
        self.records.append((family, kind, work, s, e))

    def table(self):
        torch.cuda.synchronize()
        fam = {}
        for family, kind, work, s, e in self.records:
            f = fam.setdefault(family, {"kind": kind, "ms": 0.0, "work": 0.0, "launches": 0})
            f["ms"] += s.elapsed_time(e)
            f["work"] += work
            f["launches"] += 1
        return fam

    def summary(self, mfma_peak_tflops, hbm_peak_gbs, steps):
        fam = self.table()
        if not fam:
            return {}
        out = {}
        dom = max(fam.items(), key=lambda kv: kv[1]["ms"])
        out["roofline"] = self._roof(dom, mfma_peak_tflops, hbm_peak_gbs)
        lpg = {k: v for k, v in fam.items() if k.startswith("lpg_head")}
        if lpg:
            tot = {"kind": "hbm", "ms": sum(v["ms"] for v in lpg.values()), "work": sum(v["work"] for v in lpg.values()),
                   "launches": sum(v["launches"] for v in lpg.values())}
            # every LPG-head kernel of the step (fused reduction-chain + plane + LPG forward / recompute-backward,
            # or the separate plane-head + LPG kernels where a chain runs layer-wise), against algorithmic HBM bytes
            out["roofline_lpg"] = self._roof(("lpg_head* (reduction chain + plane + LPG, fwd+bwd, k=8,4,2,1)", tot), mfma_peak_tflops, hbm_peak_gbs)
        total_ms = sum(v["ms"] for v in fam.values())
        out["kernel_time_ms_per_step"] = {k: round(v["ms"] / steps, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])[:12]}
        out["hip_kernels_ms_per_step"] = round(total_ms / steps, 3)
        return out

    @staticmethod
    def _roof(item, mfma_peak, hbm_peak):
        name, f = item
        sec = f["ms"] * 1e-3
        if f["kind"] == "mfma":
            ach = f["work"] / sec / 1e12
            return {"kernel": name, "bound": "mfma", "achieved": round(ach, 2), "peak": mfma_peak, "unit": "TFLOP/s",
                    "frac": round(ach / mfma_peak, 4), "traffic": None, "launches": f["launches"],
                    "avg_launch_us": round(f["ms"] * 1e3 / f["launches"], 2)}
        ach = f["work"] / sec / 1e9
        return {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": hbm_peak, "unit": "GB/s",
                "frac": round(ach / hbm_peak, 4), "traffic": None, "launches": f["launches"],
                "avg_launch_us": round(f["ms"] * 1e3 / f["launches"], 2)}


def enable():
    global ACTIVE
    ACTIVE = Profiler()
    return ACTIVE


def disable():
    global ACTIVE
    ACTIVE = None


def note(family, kind, work):
    """Describe the next tracked launch (family name, 'mfma'|'hbm', FLOPs or bytes)."""
    global _pending
    if ACTIVE is not None:
        _pending = (family, kind, float(work))


def take():
    global _pending
    p, _pending = _pending, None
    return p
