"""Test instrumentation: feed recorded discrete decisions (sampled index sets, dropout masks, proposal lists)
into a model so that a GPU run can be compared stage-by-stage with the CPU oracle.  Not a compute path."""
import torch


class Replay(object):
    def __init__(self, taps):
        self.d = {k: (list(v) if isinstance(v, list) else v) for k, v in taps.items()}

    def has(self, tag):
        return tag in self.d and self.d[tag] is not None

    def take_all(self, tag):
        return self.d.pop(tag, None)

    def take_next(self, tag):
        v = self.d.get(tag)
        if not v:
            return None
        return v.pop(0)
