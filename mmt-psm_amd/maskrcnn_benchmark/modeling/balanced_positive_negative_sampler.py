"""BalancedPositiveNegativeSampler (reference: modeling/balanced_positive_negative_sampler.py:5-72).

Device formulation: instead of nonzero + two randperm per image (4 host syncs per image) each candidate draws a
uniform key and the `num_pos` / `num_neg` smallest keys among positives / negatives are kept -- the same
uniform-without-replacement distribution, no data-dependent shapes.  `replay` (tests) substitutes recorded
index sets so a run can be compared decision-for-decision with the CPU oracle."""
import torch


class BalancedPositiveNegativeSampler(object):
    def __init__(self, batch_size_per_image, positive_fraction):
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.replay = None  # callable(tag) -> list of (pos_idx, neg_idx) or None
        self.generator = None  # torch.Generator of the owning model (engine: one per model, so that the teacher's helper
        #                        thread and the student do not interleave draws from the global generator); None = global

    def __call__(self, matched_idxs, tag=None):
        pos_out, neg_out = [], []
        rec = self.replay(tag) if (self.replay is not None and tag is not None) else None
        for i, m in enumerate(matched_idxs):
            if rec is not None:
                pi, ni = rec[i]
                pm = torch.zeros_like(m, dtype=torch.bool)
                nm = torch.zeros_like(m, dtype=torch.bool)
                pm[pi.to(m.device)] = True
                nm[ni.to(m.device)] = True
            else:
                pos, neg = m >= 1, m == 0
                num_pos = int(self.batch_size_per_image * self.positive_fraction)
                n_pos_avail = pos.sum()
                num_pos_t = torch.clamp(n_pos_avail, max=num_pos)
                num_neg_t = torch.minimum(neg.sum(), self.batch_size_per_image - num_pos_t)
                key = torch.rand(m.shape, device=m.device, generator=self.generator)
                pm = self._take(key, pos, num_pos_t, num_pos)
                nm = self._take(key, neg, num_neg_t, self.batch_size_per_image)
            pos_out.append(pm)
            neg_out.append(nm)
        return pos_out, neg_out

    @staticmethod
    def _take(key, member, count, kmax):
        """the `count` (device scalar, <= kmax) members with the smallest keys: one top-k, no sort, no sync"""
        k = torch.where(member, key, torch.full_like(key, 2.0))
        kk = min(int(kmax), k.numel())
        vals, idx = torch.topk(k, kk, largest=False, sorted=True)
        ok = (vals < 1.5) & (torch.arange(kk, device=k.device) < count)
        out = torch.zeros_like(member)
        out[idx] = ok
        return out
