"""`Matcher` with the reference's exact interface (/root/reference/modeling/matcher.py:20-120): constructor
(thresholds, labels, allow_low_quality_matches), `__call__(match_quality_matrix) -> (matches int64, match_labels int8,
matched_vals float32)`. Executed by the HIP kernel `unit_match_matrix`; the fused boxes->labels path the training step
uses is `unit_iou_match` (same decisions, one pass)."""

from .. import ops


class Matcher(object):
    def __init__(self, thresholds, labels, allow_low_quality_matches=False):
        """thresholds: ascending IoU cut points, the first one positive; labels: one of {-1, 0, 1} per interval they delimit (one more than
        cut points). Kept as the reference keeps them -- `thresholds` with the two infinite sentinels around the cut points -- because callers
        (and checkpoints of pickled configs) read these attributes."""
        cuts = [float(t) for t in thresholds]
        if not cuts or cuts[0] <= 0:
            raise AssertionError("Matcher: the lowest threshold must be positive")
        if any(b < a for a, b in zip(cuts, cuts[1:])):
            raise AssertionError("Matcher: thresholds must ascend")
        if len(labels) != len(cuts) + 1 or any(l not in (-1, 0, 1) for l in labels):
            raise AssertionError("Matcher: one label in {-1, 0, 1} per interval (len(thresholds) + 1 of them)")
        self.thresholds = [-float("inf")] + cuts + [float("inf")]
        self.labels = labels
        self.allow_low_quality_matches = allow_low_quality_matches

    def __call__(self, match_quality_matrix):
        assert match_quality_matrix.dim() == 2
        return ops.match_matrix(match_quality_matrix.float(), self.thresholds[1:-1], self.labels, self.allow_low_quality_matches)
