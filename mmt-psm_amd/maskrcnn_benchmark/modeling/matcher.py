"""Matcher (reference: modeling/matcher.py:6-139): IoU-threshold assignment of predictions to ground truth."""
import torch


class Matcher(object):
    BELOW_LOW_THRESHOLD = -1
    BETWEEN_THRESHOLDS = -2

    def __init__(self, high_threshold, low_threshold, allow_low_quality_matches=False, top_k=1):
        assert low_threshold <= high_threshold
        self.high_threshold = high_threshold
        self.low_threshold = low_threshold
        self.allow_low_quality_matches = allow_low_quality_matches

    def __call__(self, match_quality_matrix):
        q = match_quality_matrix
        if q.numel() == 0:
            raise ValueError("No ground-truth boxes available for one of the images during training"
                             if q.shape[0] == 0 else
                             "No proposal boxes available for one of the images during training")
        vals, matches = q.max(dim=0)
        best = matches.clone() if self.allow_low_quality_matches else None
        # branch-free: no boolean-index writes (each would be a device->host sync)
        matches = torch.where(vals < self.low_threshold, torch.full_like(matches, Matcher.BELOW_LOW_THRESHOLD), matches)
        matches = torch.where((vals >= self.low_threshold) & (vals < self.high_threshold),
                              torch.full_like(matches, Matcher.BETWEEN_THRESHOLDS), matches)
        if self.allow_low_quality_matches:
            # predictions that are the (tied) best match of some gt keep their argmax (matcher.py:118-139)
            top, _ = q.max(dim=1)
            is_best = (q == top[:, None]).any(dim=0)
            matches = torch.where(is_best, best, matches)
        return matches
