"""HIP-event timing of the C-ABI launches (used by bench.py for the roofline object).

When enabled, EVERY entry point of libbts_amd.so that enqueues work is bracketed by two events recorded on the stream the
kernel is launched on (the current PyTorch stream).  Callers attach the algorithmic work of the launch (FLOPs for the MFMA
convolutions, bytes for the streaming kernels) with `note()`; a launch nobody described is still timed, under its entry-point
name with zero work, so `hip_kernels_ms_per_step` is the whole library's device time and not a subset of it.
"""
import torch

ACTIVE = None
_pending = None


class Profiler:
    def __init__(self):
        self.records = []   # (family, kind, work, start_evt, end_evt, tag, algorithmic bytes)

    def add(self, family, kind, work, s, e, tag=None, nbytes=0.0):
        self.records.append((family, kind, work, s, e, tag, nbytes))

    def table(self):
        torch.cuda.synchronize()
        fam = {}
        for family, kind, work, s, e, _tag, _nb in self.records:
            f = fam.setdefault(family, {"kind": kind, "ms": 0.0, "work": 0.0, "launches": 0})
            f["ms"] += s.elapsed_time(e)
            f["work"] += work
            f["launches"] += 1
        return fam

    def launches(self):
        """Per-launch list (family, tag, microseconds, work, algorithmic bytes) in issue order."""
        torch.cuda.synchronize()
        return [(family, tag, s.elapsed_time(e) * 1e3, work, nb) for family, kind, work, s, e, tag, nb in self.records]

    def summary(self, mfma_peak_tflops, hbm_peak_gbs, steps):
        fam = self.table()
        if not fam:
            return {}
        out = {}
        described = {k: v for k, v in fam.items() if v["work"] > 0}
        dom = max((described or fam).items(), key=lambda kv: kv[1]["ms"])
        out["roofline"] = self._roof(dom, mfma_peak_tflops, hbm_peak_gbs)
        lpg = {k: v for k, v in fam.items() if k.startswith("lpg_head")}
        if lpg:
            tot = {"kind": "hbm", "ms": sum(v["ms"] for v in lpg.values()), "work": sum(v["work"] for v in lpg.values()),
                   "launches": sum(v["launches"] for v in lpg.values())}
            # every LPG-head kernel of the step (fused reduction-chain + plane + LPG forward / recompute-backward,
            # or the separate plane-head + LPG kernels where a chain runs layer-wise), against algorithmic HBM bytes
            out["roofline_lpg"] = self._roof(("lpg_head* (reduction chain + plane + LPG, fwd+bwd, k=8,4,2,1)", tot), mfma_peak_tflops, hbm_peak_gbs)
        ew = {k: v for k, v in fam.items() if v["kind"] == "hbm" and v["work"] > 0 and not k.startswith("lpg_head")}
        if ew:
            tot = {"kind": "hbm", "ms": sum(v["ms"] for v in ew.values()), "work": sum(v["work"] for v in ew.values()),
                   "launches": sum(v["launches"] for v in ew.values())}
            out["roofline_elementwise"] = self._roof(("BatchNorm / activation / layout / pack / loss / AdamW kernels", tot), mfma_peak_tflops, hbm_peak_gbs)
        total_ms = sum(v["ms"] for v in fam.values())
        out["kernel_time_ms_per_step"] = {k: round(v["ms"] / steps, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])[:16]}
        out["launches_per_step"] = round(sum(v["launches"] for v in fam.values()) / steps, 1)
        out["hip_kernels_ms_per_step"] = round(total_ms / steps, 3)
        mf = [v for v in fam.values() if v["kind"] == "mfma" and v["work"] > 0]
        if mf:   # every MFMA convolution of the step together: the chip-level matrix-core utilisation of the decoder
            sec = sum(v["ms"] for v in mf) * 1e-3
            out["mfma_all_convs"] = {"achieved": round(sum(v["work"] for v in mf) / sec / 1e12, 1), "unit": "TFLOP/s",
                                     "frac": round(sum(v["work"] for v in mf) / sec / 1e12 / mfma_peak_tflops, 4),
                                     "ms_per_step": round(sec * 1e3 / steps, 3)}
        return out

    @staticmethod
    def _roof(item, mfma_peak, hbm_peak):
        name, f = item
        sec = f["ms"] * 1e-3
        per = f["work"] / max(f["launches"], 1)
        if f["kind"] == "mfma":
            ach = f["work"] / sec / 1e12
            return {"kernel": name, "bound": "mfma", "achieved": round(ach, 2), "peak": mfma_peak, "unit": "TFLOP/s",
                    "frac": round(ach / mfma_peak, 4), "traffic": None, "alg_flops_per_launch": round(per), "launches": f["launches"],
                    "avg_launch_us": round(f["ms"] * 1e3 / f["launches"], 2)}
        ach = f["work"] / sec / 1e9
        return {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": hbm_peak, "unit": "GB/s",
                "frac": round(ach / hbm_peak, 4), "traffic": None, "alg_bytes_per_launch": round(per), "launches": f["launches"],
                "avg_launch_us": round(f["ms"] * 1e3 / f["launches"], 2)}


def enable():
    global ACTIVE
    ACTIVE = Profiler()
    return ACTIVE


def disable():
    global ACTIVE
    ACTIVE = None


def note(family, kind, work, tag=None, nbytes=None):
    """Describe the next tracked launch: family name, 'mfma'|'hbm', FLOPs or bytes, optional per-launch tag, and the launch's
    ALGORITHMIC bytes (for 'hbm' launches that is `work` itself; MFMA launches pass it beside their FLOPs so that the counter
    traffic of profiles/pmc_traffic.json can be read as a ratio)."""
    global _pending
    if ACTIVE is not None:
        _pending = (family, kind, float(work), tag, float(work if (nbytes is None and kind == "hbm") else (nbytes or 0.0)))


def take():
    global _pending
    p, _pending = _pending, None
    return p
