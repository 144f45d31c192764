"""`CaptionModel`: the mode-dispatching base class of the reference API.

Mirrors reference `models/CaptionModel.py:21-26`: `model(*args, mode='sample', **kw)` calls
`self._sample(*args, **kw)`, the default mode is `'forward'` (-> `self._forward`).  The diverse
beam search of `CaptionModel.py:28-175` lives in `subgc/beam.py` (all sub-graphs and beams of an image in
one decode batch) and is reached through `mode='sample'` with `opt['beam_size'] > 1`, as in the reference; the
reference-signature `beam_search(init_state, init_logprobs, *args, opt=...)` is provided by AttModel (models/stepapi.py).
"""
from __future__ import annotations

import torch.nn as nn


class CaptionModel(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, *args, **kwargs):
        mode = kwargs.pop("mode", "forward")
        return getattr(self, "_" + mode)(*args, **kwargs)
