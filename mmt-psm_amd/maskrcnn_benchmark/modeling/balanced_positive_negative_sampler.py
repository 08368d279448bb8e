"""BalancedPositiveNegativeSampler (reference: modeling/balanced_positive_negative_sampler.py:5-72).

Device formulation: instead of nonzero + two randperm per image (4 host syncs per image) every candidate draws a uniform
key and the `num_pos` / `num_neg` smallest keys among positives / negatives are kept -- the same uniform-without-replacement
distribution, no data-dependent shapes -- for ALL images of the call in one launch (`mmt_sample_fg_bg`, csrc/select.hip).
`replay` (tests) substitutes recorded index sets so a run can be compared decision-for-decision with the CPU oracle."""
import torch

from maskrcnn_benchmark import _hip as H
from maskrcnn_benchmark.utils.miscellaneous import dev_const


KEY_ROWS = 4096   # lists up to this length draw their keys from a (images, KEY_ROWS) table (see __call__)


class BalancedPositiveNegativeSampler(object):
    def __init__(self, batch_size_per_image, positive_fraction):
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.replay = None  # callable(tag) -> list of (pos_idx, neg_idx) or None
        self.generator = None  # torch.Generator of the owning model (engine: one per model, so that the teacher's helper
        #                        thread and the student do not interleave draws from the global generator); None = global

    def __call__(self, matched_idxs, tag=None):
        rec = self.replay(tag) if (self.replay is not None and tag is not None) else None
        if rec is not None:
            pos_out, neg_out = [], []
            for m, (pi, ni) in zip(matched_idxs, rec):
                pm = torch.zeros_like(m, dtype=torch.bool)
                nm = torch.zeros_like(m, dtype=torch.bool)
                pm[pi.to(m.device)] = True
                nm[ni.to(m.device)] = True
                pos_out.append(pm)
                neg_out.append(nm)
            return pos_out, neg_out
        lens = [int(m.numel()) for m in matched_idxs]
        labels = torch.cat(matched_idxs, 0) if len(matched_idxs) > 1 else matched_idxs[0]
        off = [0]
        for n in lens:
            off.append(off[-1] + n)
        if lens and max(lens) <= KEY_ROWS:
            # short lists (the box head's proposals): the keys of image i are the first len_i entries of row i of a table of fixed
            # width, so a candidate's key depends on its position only -- a list kept at fixed capacity (rows behind its count are
            # never sampled) and the same list sliced to its count draw the same keys and give the same sampled set (SURVEY f-2)
            table = torch.rand((len(lens), KEY_ROWS), device=labels.device, generator=self.generator)
            keys = torch.cat([table[i, :n] for i, n in enumerate(lens)], 0) if len(lens) > 1 else table[0, :lens[0]]
        else:
            keys = torch.rand(labels.shape, device=labels.device, generator=self.generator)
        num_pos = int(self.batch_size_per_image * self.positive_fraction)
        pm, nm, _ = H.sample_fg_bg(labels, keys, dev_const(off, torch.int32, labels.device), self.batch_size_per_image, num_pos)
        return list(pm.split(lens, 0)), list(nm.split(lens, 0))
