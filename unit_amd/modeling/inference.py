"""Inference path (filled in below the training step; see DESIGN.md 'next rows')."""


def inference(model, batched_inputs, do_postprocess=True):
    raise NotImplementedError("eval path is the next row after the training step (SURVEY section 8f rank 2)")
