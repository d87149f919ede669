"""`Matcher` with the reference's exact interface (/root/reference/modeling/matcher.py:20-120): constructor
(thresholds, labels, allow_low_quality_matches), `__call__(match_quality_matrix) -> (matches int64, match_labels int8,
matched_vals float32)`. Executed by the HIP kernel `unit_match_matrix`; the fused boxes->labels path the training step
uses is `unit_iou_match` (same decisions, one pass)."""

from .. import ops


class Matcher(object):
    def __init__(self, thresholds, labels, allow_low_quality_matches=False):
        thresholds = thresholds[:]
        assert thresholds[0] > 0
        thresholds.insert(0, -float("inf"))
        thresholds.append(float("inf"))
        assert all([low <= high for (low, high) in zip(thresholds[:-1], thresholds[1:])])
        assert all([l in [-1, 0, 1] for l in labels])
        assert len(labels) == len(thresholds) - 1
        self.thresholds = thresholds
        self.labels = labels
        self.allow_low_quality_matches = allow_low_quality_matches

    def __call__(self, match_quality_matrix):
        assert match_quality_matrix.dim() == 2
        return ops.match_matrix(match_quality_matrix.float(), self.thresholds[1:-1], self.labels, self.allow_low_quality_matches)
