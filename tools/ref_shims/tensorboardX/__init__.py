"""Stand-in for tensorboardX.SummaryWriter (bts_main.py:31, 408-415, 476-495): records every call as one JSON line in
<logdir>/events.jsonl so a run can be inspected without TensorBoard.  Installed by tools/run_reference.py only when
tensorboardX is absent."""
import json
import os


class SummaryWriter:
    def __init__(self, logdir=None, flush_secs=30, **kw):
        self.logdir = logdir or "runs"
        os.makedirs(self.logdir, exist_ok=True)
        self._f = open(os.path.join(self.logdir, "events.jsonl"), "a")

    def _put(self, kind, tag, step, **extra):
        self._f.write(json.dumps(dict(kind=kind, tag=tag, step=int(step) if step is not None else None, **extra)) + "\n")

    def add_scalar(self, tag, value, global_step=None, **kw):
        self._put("scalar", tag, global_step, value=float(value))

    def add_image(self, tag, img, global_step=None, **kw):
        shape = list(getattr(img, "shape", []))
        self._put("image", tag, global_step, shape=shape)

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()
