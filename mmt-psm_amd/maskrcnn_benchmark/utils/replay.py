"""Test instrumentation: feed recorded RANDOM decisions (sampled index sets, dropout masks) into a model so that a GPU
run can be compared stage-by-stage with the CPU oracle.  Not a compute path.

Proposal and detection lists are NOT random decisions: by default they are computed by the product and compared with the
oracle's (tests/test_model_gpu.py, tests/test_proposals_gpu.py).  `substitute_lists=True` additionally swaps the product's
proposal / detection lists for the recorded ones; it exists for the one tolerance test that runs the network in bf16
arithmetic against the fp32 oracle (tests/test_irnet_gpu.py::test_irnet_bf16_products_mode), where discrete selections
legitimately differ and the recorded sampler indices would not refer to the same boxes."""
import torch


class Replay(object):
    def __init__(self, taps, substitute_lists=False):
        self.substitute_lists = substitute_lists
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

    def align(self, tag, boxlists, tol=1e-3, window=8):
        """The reference leaves the order of candidates with EQUAL scores unspecified, and two scores that
        differ by less than the fp32 noise between the device's and the host's convolutions are equal for that purpose:
        such a pair may come out swapped.  The recorded sampler indices are POSITIONS in these lists, so the product's own
        list is brought into the recorded order where -- and only where -- a row's box is found a few positions away
        (|shift| <= window) in the record.  Values are never taken from the record.  -> (lists, moved) where moved =
        [(image, product position, recorded position)] for the test to check that only near-tied rows moved."""
        rec = self.d.get(tag)
        moved = []
        if rec is None:
            return boxlists, moved
        out = []
        for n, (b, r) in enumerate(zip(boxlists, rec)):
            rb = r[0].to(b.bbox.device)
            if len(b) != rb.shape[0] or len(b) == 0:
                out.append(b)
                continue
            bad = ((b.bbox - rb).abs().amax(1) > tol).nonzero().squeeze(1).tolist()
            if not bad:
                out.append(b)
                continue
            perm = torch.arange(len(b), device=b.bbox.device)
            for j in bad:  # recorded position j: which nearby product row is it?
                lo, hi = max(0, j - window), min(len(b), j + window + 1)
                d = (b.bbox[lo:hi] - rb[j]).abs().amax(1)
                i = int(d.argmin()) + lo
                if float(d.min()) <= tol:
                    perm[j] = i
                    moved.append((n, i, j))
            if perm.sort()[0].equal(torch.arange(len(b), device=perm.device)):
                out.append(b[perm])
            else:
                out.append(b)
        return out, moved
